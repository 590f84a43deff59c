"""ORACLE -- TEST INFRASTRUCTURE.  CPU (torch fp32) restatement of SAM's ImageEncoderViT.

`segment_anything` (git+https://github.com/facebookresearch/segment-anything.git, un-pinned: README.md:235,
sam2image.py:55-61, editany_lora.py:36-57) is NOT under /root/reference; this follows its published
modeling/image_encoder.py (ImageEncoderViT / Block / Attention / window_partition /
add_decomposed_rel_pos / LayerNorm2d) and the reference's call sites sam2image.py:67-71,118.
Pinned against transformers.models.sam.modeling_sam.SamVisionEncoder (oracle/make_golden.py).
State-dict keys are upstream's (`image_encoder.` prefix stripped).
"""
import torch
import torch.nn.functional as F

PIXEL_MEAN = (123.675, 116.28, 103.53)
PIXEL_STD = (58.395, 57.12, 57.375)


def window_partition(x, ws):
    """[B,H,W,C] -> [B*nW, ws, ws, C] with zero padding to a multiple of ws (pad tokens are NOT masked)."""
    B, H, W, C = x.shape
    pad_h = (ws - H % ws) % ws
    pad_w = (ws - W % ws) % ws
    if pad_h or pad_w:
        x = F.pad(x, (0, 0, 0, pad_w, 0, pad_h))
    Hp, Wp = H + pad_h, W + pad_w
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(windows, ws, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = windows.shape[0] // (Hp * Wp // ws // ws)
    x = windows.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous()


def get_rel_pos(q_size, k_size, rel_pos):
    max_rel_dist = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel_dist:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel_dist, mode="linear")
        rel_pos = r.reshape(-1, max_rel_dist).permute(1, 0)
    q_coords = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k_coords = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (q_coords - k_coords) + (k_size - 1) * max(q_size / k_size, 1.0)
    return rel_pos[rel.long()]


def attention(sd, p, x, heads):
    """Attention.forward with decomposed rel-pos computed from the UNSCALED q."""
    B, H, W, C = x.shape
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(B, H * W, 3, heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * heads, H * W, -1).unbind(0)
    d = q.shape[-1]
    attn = (q * d ** -0.5) @ k.transpose(-2, -1)
    Rh = get_rel_pos(H, H, sd[p + "rel_pos_h"])
    Rw = get_rel_pos(W, W, sd[p + "rel_pos_w"])
    r_q = q.reshape(B * heads, H, W, d)
    rel_h = torch.einsum("bhwc,hkc->bhwk", r_q, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", r_q, Rw)
    attn = (attn.view(B * heads, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(B * heads, H * W, H * W)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).view(B, heads, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)
    return F.linear(x, sd[p + "proj.weight"], sd[p + "proj.bias"])


def block(sd, p, x, heads, window_size):
    ln = lambda t, n: F.layer_norm(t, (t.shape[-1],), sd[p + n + ".weight"], sd[p + n + ".bias"], 1e-6)
    shortcut = x
    x = ln(x, "norm1")
    if window_size > 0:
        H, W = x.shape[1], x.shape[2]
        x, pad_hw = window_partition(x, window_size)
    x = attention(sd, p + "attn.", x, heads)
    if window_size > 0:
        x = window_unpartition(x, window_size, pad_hw, (H, W))
    x = shortcut + x
    h = F.gelu(F.linear(ln(x, "norm2"), sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"]))
    return x + F.linear(h, sd[p + "mlp.lin2.weight"], sd[p + "mlp.lin2.bias"])


def layer_norm_2d(x, w, b, eps=1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def image_encoder(sd, cfg, x):
    """ImageEncoderViT.forward: x [B,3,S,S] normalised -> [B,out_chans,S/16,S/16]."""
    x = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=cfg["patch_size"])
    x = x.permute(0, 2, 3, 1) + sd["pos_embed"]
    for i in range(cfg["depth"]):
        ws = 0 if i in cfg["global_attn_indexes"] else cfg["window_size"]
        x = block(sd, f"blocks.{i}.", x, cfg["num_heads"], ws)
    x = x.permute(0, 3, 1, 2)
    x = F.conv2d(x, sd["neck.0.weight"])
    x = layer_norm_2d(x, sd["neck.1.weight"], sd["neck.1.bias"])
    x = F.conv2d(x, sd["neck.2.weight"], padding=1)
    return layer_norm_2d(x, sd["neck.3.weight"], sd["neck.3.bias"])


def preprocess(image_u8_hwc, img_size=1024):
    """Sam.preprocess after ResizeLongestSide: here the caller passes an image whose long side already equals
    img_size (BASELINE configs use square inputs); normalise, zero-pad bottom/right to img_size."""
    x = torch.as_tensor(image_u8_hwc).permute(2, 0, 1).float()
    mean = torch.tensor(PIXEL_MEAN).view(3, 1, 1)
    std = torch.tensor(PIXEL_STD).view(3, 1, 1)
    x = (x - mean) / std
    h, w = x.shape[-2:]
    return F.pad(x, (0, img_size - w, 0, img_size - h))[None]
