"""ORACLE -- TEST INFRASTRUCTURE.  CPU (torch fp32) restatement of SAM's prompt encoder, mask decoder and
`SamAutomaticMaskGenerator.generate`.

`segment_anything` (git+https://github.com/facebookresearch/segment-anything.git, un-pinned: README.md:235,
sam2image.py:55-61) is NOT under /root/reference.  This follows its published modeling/prompt_encoder.py,
modeling/mask_decoder.py, modeling/transformer.py, modeling/sam.py (postprocess_masks), predictor.py,
utils/transforms.py, utils/amg.py and automatic_mask_generator.py, with the generator defaults the reference relies on
(`SamAutomaticMaskGenerator(sam)` with no arguments, sam2image.py:71; call site `.generate(image)` sam2image.py:118):
points_per_side 32, points_per_batch 64, pred_iou_thresh 0.88, stability_score_thresh 0.95,
stability_score_offset 1.0, box_nms_thresh 0.7, crop_n_layers 0, min_mask_region_area 0, output "binary_mask".

Pinning: prompt encoder + mask decoder are checked against transformers.models.sam.modeling_sam
(SamPromptEncoder / SamMaskDecoder, an independent port) in tests/test_oracle.py via `to_hf_state_dict`;
the AMG post-processing (filters, stability score, boxes, NMS, output records) has no vectors anywhere in the
reference or in that port: "parity unpinned" for that part -- it is restated from the published algorithm only.
State-dict keys are upstream's (`prompt_encoder.*`, `mask_decoder.*`).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

AMG_DEFAULTS = dict(points_per_side=32, points_per_batch=64, pred_iou_thresh=0.88, stability_score_thresh=0.95,
                    stability_score_offset=1.0, box_nms_thresh=0.7, mask_threshold=0.0)


# ------------------------------------------------------------------ prompt encoder (modeling/prompt_encoder.py)
def pe_encoding(gauss, coords01):
    """PositionEmbeddingRandom._pe_encoding: coords in [0,1]^2 (x, y) -> [..., 2*num_pos_feats]."""
    c = 2.0 * coords01 - 1.0
    c = c @ gauss
    c = 2.0 * math.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(sd, size):
    """PromptEncoder.get_dense_pe: [1, C, h, w] positional encoding of the embedding grid."""
    h, w = size
    gauss = sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    y = (torch.arange(h, dtype=torch.float32) + 0.5) / h
    x = (torch.arange(w, dtype=torch.float32) + 0.5) / w
    grid = torch.stack([x[None, :].expand(h, w), y[:, None].expand(h, w)], dim=-1)
    return pe_encoding(gauss, grid).permute(2, 0, 1)[None]


def embed_points(sd, points, labels, input_size=1024):
    """PromptEncoder._embed_points with pad=True (no boxes): points [B, N, 2] in input-frame pixels (x, y),
    labels [B, N] in {1 foreground, 0 background} -> sparse embeddings [B, N + 1, C]."""
    gauss = sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    B = points.shape[0]
    pts = torch.cat([points + 0.5, torch.zeros(B, 1, 2)], dim=1)
    lab = torch.cat([labels.float(), -torch.ones(B, 1)], dim=1)
    emb = pe_encoding(gauss, pts / float(input_size))
    emb = torch.where(lab[..., None] == -1, torch.zeros_like(emb), emb)
    emb = emb + (lab[..., None] == -1) * sd["prompt_encoder.not_a_point_embed.weight"]
    emb = emb + (lab[..., None] == 0) * sd["prompt_encoder.point_embeddings.0.weight"]
    emb = emb + (lab[..., None] == 1) * sd["prompt_encoder.point_embeddings.1.weight"]
    return emb


def embed_boxes(sd, boxes, input_size=1024):
    """PromptEncoder._embed_boxes: boxes [B, 4] XYXY in input-frame pixels -> sparse embeddings [B, 2, C] (the two
    corners, + point_embeddings[2] / [3]).  With boxes and no points the prompt is exactly these two tokens (the
    padding point is only appended when there are NO boxes: `pad = boxes is None`).  Reference call site:
    sam2groundingdino_edit.py:176-183 (`predict_torch(point_coords=None, point_labels=None, boxes=...)`)."""
    gauss = sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    c = (boxes.float() + 0.5).reshape(-1, 2, 2)
    emb = pe_encoding(gauss, c / float(input_size))
    emb[:, 0] = emb[:, 0] + sd["prompt_encoder.point_embeddings.2.weight"][0]
    emb[:, 1] = emb[:, 1] + sd["prompt_encoder.point_embeddings.3.weight"][0]
    return emb


def apply_boxes(boxes, orig_hw, long_side=1024):
    """ResizeLongestSide.apply_boxes_torch: XYXY boxes of the original image -> the resized input frame."""
    h, w = orig_hw
    nh, nw = preprocess_shape(h, w, long_side)
    b = boxes.float().reshape(-1, 2, 2).clone()
    b[..., 0] = b[..., 0] * (nw / w)
    b[..., 1] = b[..., 1] * (nh / h)
    return b.reshape(-1, 4)


def remove_small_regions(mask, area_thresh, mode):
    """utils/amg.py remove_small_regions (cv2.connectedComponentsWithStats, 8-connectivity; here scipy.ndimage.label
    with a 3x3 structure -- the same components): mode "holes" fills holes smaller than area_thresh, "islands" removes
    islands smaller than it.  Returns (mask, changed).  Reference call: sam2groundingdino_edit.py:186-188."""
    from scipy import ndimage
    assert mode in ("holes", "islands")
    correct_holes = mode == "holes"
    working = (correct_holes ^ np.asarray(mask).astype(bool)).astype(np.uint8)
    regions, n = ndimage.label(working, structure=np.ones((3, 3), int))
    sizes = np.bincount(regions.ravel(), minlength=n + 1)[1:]
    small = [i + 1 for i, sz in enumerate(sizes) if sz < area_thresh]
    if len(small) == 0:
        return np.asarray(mask).astype(bool), False
    fill = [0] + small
    if not correct_holes:
        fill = [i for i in range(n + 1) if i not in fill]
        if len(fill) == 0:                    # every region is below the threshold: keep the largest
            fill = [int(np.argmax(sizes)) + 1]
    return np.isin(regions, fill), True


# ------------------------------------------------------------------ two-way transformer (modeling/transformer.py)
def attention(sd, p, q, k, v, heads):
    q = F.linear(q, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"])
    k = F.linear(k, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"])
    v = F.linear(v, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
    B, Nq, Ci = q.shape
    d = Ci // heads
    sp = lambda t: t.reshape(B, t.shape[1], heads, d).transpose(1, 2)
    a = (sp(q) @ sp(k).transpose(-2, -1)) / math.sqrt(d)
    o = torch.softmax(a, dim=-1) @ sp(v)
    o = o.transpose(1, 2).reshape(B, Nq, Ci)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def ln(sd, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], eps)


def two_way_transformer(sd, image_embedding, image_pe, point_embedding, heads=8, depth=2):
    """TwoWayTransformer.forward: image_embedding/image_pe [B, C, h, w], point_embedding [B, Np, C]."""
    p = "mask_decoder.transformer."
    keys = image_embedding.flatten(2).permute(0, 2, 1)
    key_pe = image_pe.flatten(2).permute(0, 2, 1)
    queries = point_embedding
    for i in range(depth):
        lp = f"{p}layers.{i}."
        if i == 0:
            queries = attention(sd, lp + "self_attn.", queries, queries, queries, heads)
        else:
            q = queries + point_embedding
            queries = queries + attention(sd, lp + "self_attn.", q, q, queries, heads)
        queries = ln(sd, lp + "norm1.", queries)
        q = queries + point_embedding
        k = keys + key_pe
        queries = ln(sd, lp + "norm2.", queries + attention(sd, lp + "cross_attn_token_to_image.", q, k, keys, heads))
        m = F.linear(F.relu(F.linear(queries, sd[lp + "mlp.lin1.weight"], sd[lp + "mlp.lin1.bias"])),
                     sd[lp + "mlp.lin2.weight"], sd[lp + "mlp.lin2.bias"])
        queries = ln(sd, lp + "norm3.", queries + m)
        q = queries + point_embedding
        k = keys + key_pe
        keys = ln(sd, lp + "norm4.", keys + attention(sd, lp + "cross_attn_image_to_token.", k, q, queries, heads))
    q = queries + point_embedding
    k = keys + key_pe
    queries = ln(sd, p + "norm_final_attn.", queries + attention(sd, p + "final_attn_token_to_image.", q, k, keys, heads))
    return queries, keys


def mlp3(sd, p, x):
    x = F.relu(F.linear(x, sd[p + "layers.0.weight"], sd[p + "layers.0.bias"]))
    x = F.relu(F.linear(x, sd[p + "layers.1.weight"], sd[p + "layers.1.bias"]))
    return F.linear(x, sd[p + "layers.2.weight"], sd[p + "layers.2.bias"])


def layernorm2d(x, w, b, eps=1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[None, :, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[None, :, None, None]


def mask_decoder(sd, image_embedding, image_pe, sparse, multimask_output=True, heads=8):
    """MaskDecoder.forward for a point batch: image_embedding [1, C, h, w], sparse [B, Np, C], no mask prompt
    (dense = no_mask_embed).  -> low-res mask logits [B, 3 or 1, 4h, 4w], iou predictions [B, 3 or 1]."""
    B = sparse.shape[0]
    C = image_embedding.shape[1]
    tokens = torch.cat([sd["mask_decoder.iou_token.weight"], sd["mask_decoder.mask_tokens.weight"]], 0)
    tokens = torch.cat([tokens[None].expand(B, -1, -1), sparse], dim=1)
    dense = sd["prompt_encoder.no_mask_embed.weight"].reshape(1, C, 1, 1)
    src = (image_embedding + dense).expand(B, -1, -1, -1)
    pos = image_pe.expand(B, -1, -1, -1)
    h, w = src.shape[-2:]
    hs, src = two_way_transformer(sd, src, pos, tokens, heads)
    iou_tok, mask_toks = hs[:, 0], hs[:, 1:5]
    src = src.transpose(1, 2).reshape(B, C, h, w)
    up = F.conv_transpose2d(src, sd["mask_decoder.output_upscaling.0.weight"], sd["mask_decoder.output_upscaling.0.bias"], stride=2)
    up = F.gelu(layernorm2d(up, sd["mask_decoder.output_upscaling.1.weight"], sd["mask_decoder.output_upscaling.1.bias"]))
    up = F.gelu(F.conv_transpose2d(up, sd["mask_decoder.output_upscaling.3.weight"], sd["mask_decoder.output_upscaling.3.bias"], stride=2))
    hyper = torch.stack([mlp3(sd, f"mask_decoder.output_hypernetworks_mlps.{i}.", mask_toks[:, i]) for i in range(4)], 1)
    b, c, hh, ww = up.shape
    masks = (hyper @ up.view(b, c, hh * ww)).view(b, 4, hh, ww)
    iou = mlp3(sd, "mask_decoder.iou_prediction_head.", iou_tok)
    sl = slice(1, None) if multimask_output else slice(0, 1)
    return masks[:, sl], iou[:, sl]


def postprocess_masks(masks, input_size, original_size, img_size=1024):
    """Sam.postprocess_masks: low-res logits -> original image resolution (two bilinear resizes around a crop)."""
    masks = F.interpolate(masks, (img_size, img_size), mode="bilinear", align_corners=False)
    masks = masks[..., :input_size[0], :input_size[1]]
    return F.interpolate(masks, original_size, mode="bilinear", align_corners=False)


# ------------------------------------------------------------------ AMG (automatic_mask_generator.py, utils/amg.py)
def build_point_grid(n):
    off = 1.0 / (2 * n)
    pts = np.linspace(off, 1 - off, n)
    return np.stack([np.tile(pts[None, :], (n, 1)), np.tile(pts[:, None], (1, n))], axis=-1).reshape(-1, 2)


def preprocess_shape(oldh, oldw, long_side=1024):
    """ResizeLongestSide.get_preprocess_shape."""
    scale = long_side * 1.0 / max(oldh, oldw)
    return int(oldh * scale + 0.5), int(oldw * scale + 0.5)


def stability_score(masks, thr, off):
    inter = (masks > (thr + off)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
    union = (masks > (thr - off)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
    return inter / union


def batched_mask_to_box(masks):
    """utils/amg.py batched_mask_to_box: [N, H, W] bool -> XYXY int boxes ([0,0,0,0] for empty masks)."""
    if masks.numel() == 0:
        return torch.zeros(*masks.shape[:-2], 4)
    h, w = masks.shape[-2:]
    in_h, _ = torch.max(masks, dim=-1)
    hc = in_h * torch.arange(h)[None, :]
    bottom, _ = torch.max(hc, dim=-1)
    hc = hc + h * (~in_h)
    top, _ = torch.min(hc, dim=-1)
    in_w, _ = torch.max(masks, dim=-2)
    wc = in_w * torch.arange(w)[None, :]
    right, _ = torch.max(wc, dim=-1)
    wc = wc + w * (~in_w)
    left, _ = torch.min(wc, dim=-1)
    empty = (right < left) | (bottom < top)
    out = torch.stack([left, top, right, bottom], dim=-1)
    return out * (~empty).unsqueeze(-1)


def is_box_near_crop_edge(boxes, crop_box, orig_box, atol=20.0):
    cb = torch.as_tensor(crop_box, dtype=torch.float)
    ob = torch.as_tensor(orig_box, dtype=torch.float)
    b = boxes.float()
    near_crop = torch.isclose(b, cb[None, :], atol=atol, rtol=0)
    near_img = torch.isclose(b, ob[None, :], atol=atol, rtol=0)
    return torch.any(near_crop & ~near_img, dim=1)


def box_iou(a, b):
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def nms(boxes, scores, thr):
    """torchvision.ops.nms semantics: greedy, descending score (stable), suppress IoU > thr; returns kept indices in
    descending-score order."""
    order = torch.argsort(scores, descending=True, stable=True)
    iou = box_iou(boxes[order], boxes[order])
    n = len(order)
    alive = torch.ones(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(int(order[i]))
        alive &= ~(iou[i] > thr)
        alive[i] = False
    return torch.as_tensor(keep, dtype=torch.long)


def generate(sd, image_embedding, orig_hw, cfg=None, img_size=1024):
    """SamAutomaticMaskGenerator.generate for one image whose encoder output `image_embedding` [1, C, 64, 64] is
    given (the encoder is oracle/sam_oracle.py).  Returns the list of records in NMS keep order
    (descending predicted_iou): segmentation bool [H, W], area, bbox XYWH, predicted_iou, point_coords,
    stability_score, crop_box."""
    c = dict(AMG_DEFAULTS)
    c.update(cfg or {})
    H, W = orig_hw
    in_h, in_w = preprocess_shape(H, W, img_size)
    pts = build_point_grid(c["points_per_side"]) * np.array([[W, H]])
    emb_hw = image_embedding.shape[-2:]
    pe = dense_pe(sd, emb_hw)
    crop_box = [0, 0, W, H]
    rec = dict(masks=[], iou=[], pts=[], stab=[], boxes=[])
    for s in range(0, len(pts), c["points_per_batch"]):
        p = torch.as_tensor(pts[s:s + c["points_per_batch"]], dtype=torch.float32)
        tp = p * torch.tensor([in_w / W, in_h / H])                 # ResizeLongestSide.apply_coords
        sparse = embed_points(sd, tp[:, None, :], torch.ones(len(p), 1), img_size)
        low, iou = mask_decoder(sd, image_embedding, pe, sparse, True)
        masks = postprocess_masks(low, (in_h, in_w), (H, W), img_size).flatten(0, 1)
        iou = iou.flatten(0, 1)
        pp = p.repeat_interleave(3, dim=0)
        k = iou > c["pred_iou_thresh"]
        masks, iou, pp = masks[k], iou[k], pp[k]
        st = stability_score(masks, c["mask_threshold"], c["stability_score_offset"])
        k = st >= c["stability_score_thresh"]
        masks, iou, pp, st = masks[k], iou[k], pp[k], st[k]
        mb = masks > c["mask_threshold"]
        boxes = batched_mask_to_box(mb)
        k = ~is_box_near_crop_edge(boxes, crop_box, [0, 0, W, H])
        rec["masks"].append(mb[k]); rec["iou"].append(iou[k]); rec["pts"].append(pp[k])
        rec["stab"].append(st[k]); rec["boxes"].append(boxes[k])
    masks = torch.cat(rec["masks"]); iou = torch.cat(rec["iou"]); ppts = torch.cat(rec["pts"])
    stab = torch.cat(rec["stab"]); boxes = torch.cat(rec["boxes"])
    keep = nms(boxes.float(), iou, c["box_nms_thresh"]) if len(iou) else torch.zeros(0, dtype=torch.long)
    out = []
    for i in keep.tolist():
        b = boxes[i].tolist()
        out.append(dict(segmentation=masks[i].numpy(), area=int(masks[i].sum()), bbox=[b[0], b[1], b[2] - b[0], b[3] - b[1]],
                        predicted_iou=float(iou[i]), point_coords=[ppts[i].tolist()], stability_score=float(stab[i]),
                        crop_box=[0, 0, W, H]))
    return out


# ------------------------------------------------------------------ name map to the independent port (pinning)
def to_hf_state_dict(sd):
    """upstream keys -> transformers SamPromptEncoder / SamMaskDecoder keys (prefixes `prompt_encoder.` / `mask_decoder.`)."""
    out = {}
    for k, v in sd.items():
        if k == "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix":
            out["prompt_encoder.shared_embedding.positional_embedding"] = v
        elif k.startswith("prompt_encoder.point_embeddings."):
            out[k.replace("point_embeddings", "point_embed")] = v
        elif k.startswith("prompt_encoder."):
            out[k] = v
        elif k.startswith("mask_decoder."):
            n = k
            for a, b in ((".norm1.", ".layer_norm1."), (".norm2.", ".layer_norm2."), (".norm3.", ".layer_norm3."),
                         (".norm4.", ".layer_norm4."), ("norm_final_attn", "layer_norm_final_attn"),
                         ("output_upscaling.0.", "upscale_conv1."), ("output_upscaling.1.", "upscale_layer_norm."),
                         ("output_upscaling.3.", "upscale_conv2.")):
                n = n.replace(a, b)
            if "output_hypernetworks_mlps" in n or "iou_prediction_head" in n:
                n = n.replace(".layers.0.", ".proj_in.").replace(".layers.2.", ".proj_out.").replace(".layers.1.", ".layers.0.")
            out[n] = v
    return out
