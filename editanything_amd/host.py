"""Host-side pre/post-processing on the `process()` call surface (cv2-free restatements).

  HWC3, resize_image         annotator/util.py:9-38
  show_anns                  sam2image.py:92-115 / editany_lora.py:426-449
  make_control / seeds       sam2image.py:154-167
  image / mask preparation   utils/stable_diffusion_controlnet_inpaint.py:142-388
"""
import random

import numpy as np
import torch
from PIL import Image


def HWC3(x):
    assert x.dtype == np.uint8
    if x.ndim == 2:
        x = x[:, :, None]
    assert x.ndim == 3
    H, W, C = x.shape
    assert C == 1 or C == 3 or C == 4
    if C == 3:
        return x
    if C == 1:
        return np.concatenate([x, x, x], axis=2)
    color = x[:, :, 0:3].astype(np.float32)
    alpha = x[:, :, 3:4].astype(np.float32) / 255.0
    y = color * alpha + 255.0 * (1.0 - alpha)
    return y.clip(0, 255).astype(np.uint8)


def resize_image(input_image, resolution):
    """Short side -> resolution, both sides rounded to x64.  When the size is unchanged this is a copy (the case of
    every BASELINE config); otherwise PIL LANCZOS (up) / BOX (down) stands in for cv2 LANCZOS4 / AREA -- not
    bit-identical to OpenCV (that resampling is unpinned)."""
    H, W, C = input_image.shape
    k = float(resolution) / min(float(H), float(W))
    Hn = int(np.round(H * k / 64.0)) * 64
    Wn = int(np.round(W * k / 64.0)) * 64
    if (Hn, Wn) == (H, W):
        return input_image.copy()
    im = Image.fromarray(input_image).resize((Wn, Hn), Image.LANCZOS if k > 1 else Image.BOX)
    return np.asarray(im)


def resize_longest_side(image_u8_hwc, long_side=1024):
    """segment_anything ResizeLongestSide.apply_image: `resize(to_pil_image(image), (newh, neww))` -- PIL bilinear."""
    h, w = image_u8_hwc.shape[:2]
    scale = long_side * 1.0 / max(h, w)
    newh, neww = int(h * scale + 0.5), int(w * scale + 0.5)
    return np.asarray(Image.fromarray(np.ascontiguousarray(image_u8_hwc)).resize((neww, newh), Image.BILINEAR))


def show_anns(anns, rng=None):
    """-> (PIL preview, float64 [H,W,3] id-map with ch0 = id % 256, ch1 = id // 256).  Keeps the reference quirk:
    ids follow the (unsorted) list order, later masks overwrite earlier ones."""
    if len(anns) == 0:
        return None
    m0 = np.asarray(anns[0]["segmentation"])
    idmap = np.zeros(m0.shape, dtype=np.uint16)
    full = np.zeros(m0.shape + (3,))
    rng = rng if rng is not None else np.random
    for i in range(len(anns)):
        m = np.asarray(anns[i]["segmentation"]) != 0
        idmap[m] = i + 1
        full[m] = rng.random((1, 3)).tolist()[0]
    res = np.zeros(m0.shape + (3,))
    res[:, :, 0] = idmap % 256
    res[:, :, 1] = idmap // 256
    return Image.fromarray(np.uint8(full * 255)), res


def show_anns_from_id_map(idmap, n, rng=None):
    """show_anns' return value from the id map itself (amg.generate_id_map): the preview colours are drawn once per
    record in list order -- the same `n` draws from the same stream as show_anns -- and looked up through the map (the
    last painter of a pixel is the largest record number, which is what the map holds)."""
    if n == 0:
        return None
    idmap = np.asarray(idmap).astype(np.uint16)
    rng = rng if rng is not None else np.random
    colours = np.zeros((n + 1, 3))
    for i in range(n):
        colours[i + 1] = rng.random((1, 3)).tolist()[0]
    full = colours[idmap]
    res = np.zeros(idmap.shape + (3,))
    res[:, :, 0] = idmap % 256
    res[:, :, 1] = idmap // 256
    return Image.fromarray(np.uint8(full * 255)), res


def make_control(detected_map, H, W, num_samples, device):
    """sam2image.py:154-161: uint8 truncation, HWC3, (bilinear) resize to (W, H), float 0..255, b c h w."""
    det = HWC3(detected_map.astype(np.uint8))
    if det.shape[:2] != (H, W):
        det = np.asarray(Image.fromarray(det).resize((W, H), Image.BILINEAR))
    control = torch.from_numpy(det.copy()).float().to(device)
    control = torch.stack([control for _ in range(num_samples)], dim=0)
    return control.permute(0, 3, 1, 2).contiguous()


def resolve_seed(seed):
    """seed == -1 -> random.randint(0, 65535); seed_everything + CPU generator (sam2image.py:163-167)."""
    if seed == -1:
        seed = random.randint(0, 65535)
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    generator = torch.manual_seed(seed)
    return seed, generator


def prepare_image(image):
    """…inpaint.py:142-163: tensor [3,H,W] / [B,3,H,W] (cast to float32), or PIL / uint8 HWC array / a list of either
    -> float32 [B,3,H,W] in [-1, 1]."""
    if isinstance(image, torch.Tensor):
        if image.ndim == 3:
            image = image.unsqueeze(0)
        return image.to(dtype=torch.float32)
    if isinstance(image, np.ndarray) and image.ndim == 4:      # [B,H,W,3]: a batch in one array (the batched tile
        image = list(image)                                     # refinement passes one; the reference takes HWC only)
    if isinstance(image, (Image.Image, np.ndarray)):
        image = [image]
    if isinstance(image[0], Image.Image):
        image = np.concatenate([np.array(i.convert("RGB"))[None, :] for i in image], axis=0)
    else:
        image = np.concatenate([np.asarray(i)[None, :] for i in image], axis=0)
    return torch.from_numpy(image.transpose(0, 3, 1, 2).copy()).to(dtype=torch.float32) / 127.5 - 1.0


def prepare_mask_image(mask):
    """…inpaint.py:290-326 -> [B,1,H,W] binarised at 0.5.  Same cases as the reference: a 2-D tensor is one mask, a
    3-D tensor is [1,H,W] (one mask) or [B,H,W] (a batch); PIL masks are converted to "L" and divided by 255, ndarray
    masks are taken AS THEY ARE ([H,W] each, already in [0, 1]); a list is stacked along the batch axis.  The one
    difference: the reference binarises a tensor argument in place, here the caller's tensor is left alone."""
    if isinstance(mask, torch.Tensor):
        m = mask
        if m.ndim == 2:
            m = m.unsqueeze(0).unsqueeze(0)
        elif m.ndim == 3 and m.shape[0] == 1:
            m = m.unsqueeze(0)
        elif m.ndim == 3:
            m = m.unsqueeze(1)
        m = m.clone()
    else:
        if isinstance(mask, (Image.Image, np.ndarray)):
            mask = [mask]
        if isinstance(mask[0], Image.Image):
            m = np.concatenate([np.array(x.convert("L"))[None, None, :] for x in mask], axis=0).astype(np.float32) / 255.0
        else:
            m = np.concatenate([np.asarray(x)[None, None, :] for x in mask], axis=0).copy()
        m = torch.from_numpy(m)
    m[m < 0.5] = 0
    m[m >= 0.5] = 1
    return m


def numpy_to_pil(images):
    if images.ndim == 3:
        images = images[None]
    images = (images * 255).round().astype("uint8")
    return [Image.fromarray(im) for im in images]


# ---------------------------------------------------------------------------------------------- editany_lora.py helpers
def resize_linear_u8(img, width, height):
    """`cv2.resize(img, (width, height), interpolation=cv2.INTER_LINEAR)` for uint8 HWC / HW arrays
    (editany_lora.py:742-752, 801-805, 892-897: SAM id map, inpaint mask and scale map onto the working resolution).
    OpenCV's 8-bit path restated: half-pixel centres, NO antialiasing when shrinking, 11-bit fixed-point tap weights,
    horizontal pass in int32 then `((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2`.  cv2 is not installed here, so the
    fixed-point rounding is not pinned against OpenCV itself; an unchanged size is an exact copy, which
    is the case of every BASELINE config."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    sh, sw = img.shape[:2]
    if (sh, sw) == (height, width):
        return img.copy()
    squeeze = img.ndim == 2
    src = img[:, :, None] if squeeze else img

    def taps(dst_n, src_n):
        scale = src_n / dst_n
        f = (np.arange(dst_n, dtype=np.float64) + 0.5) * scale - 0.5
        s = np.floor(f).astype(np.int64)
        f = f - s
        lo = s < 0
        s[lo], f[lo] = 0, 0.0
        hi = s >= src_n - 1
        s[hi], f[hi] = src_n - 1, 0.0
        s1 = np.minimum(s + 1, src_n - 1)
        w1 = np.clip(np.rint(f * 2048.0), -32768, 32767).astype(np.int64)
        w0 = np.clip(np.rint((1.0 - f) * 2048.0), -32768, 32767).astype(np.int64)
        return s, s1, w0, w1

    x0, x1, a0, a1 = taps(width, sw)
    y0, y1, b0, b1 = taps(height, sh)
    s64 = src.astype(np.int64)
    hor = s64[:, x0, :] * a0[None, :, None] + s64[:, x1, :] * a1[None, :, None]         # [sh, width, C] int
    r0, r1 = hor[y0], hor[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def resize_points(clicked_points, original_shape, resolution):
    """annotator/util.py:40-55."""
    oh, ow = float(original_shape[0]), float(original_shape[1])
    k = float(resolution) / min(oh, ow)
    return [(int(round(x * k)), int(round(y * k)), lab) for x, y, lab in clicked_points]


def get_bounding_box(mask):
    """annotator/util.py:57-74 -> [xmin, ymin, xmax, ymax] of the non-zero region of channel 0."""
    m = np.array(mask).astype(np.uint8)[:, :, 0]
    xs = np.where(np.any(m, axis=0))[0]
    ys = np.where(np.any(m, axis=1))[0]
    return [xs[0], ys[0], xs[-1], ys[-1]]


def make_inpaint_condition(image, image_mask):
    """editany_lora.py:330-338: image / 255 with the masked pixels set to -1 -> float64 [1, 3, H, W] (the inpaint
    ControlNet's conditioning image)."""
    image = np.asarray(image) / 255.0
    assert image.shape[0:1] == image_mask.shape[0:1], "image and image_mask must have the same image size"
    image[np.asarray(image_mask) > 128] = -1.0
    return torch.from_numpy(np.expand_dims(image, 0).transpose(0, 3, 1, 2).copy())


def draw_click_overlay(input_image, mask_image, clicked_points, radius=20, opacity_mask=0.75, opacity_edited=1.0):
    """editany_lora.py:587-607: filled circles at the clicks (red = foreground, blue = background), then
    `addWeighted(edited, 1.0, mask * green, 0.75, 0)` with uint8 saturation."""
    from PIL import ImageDraw
    edited = Image.fromarray(np.ascontiguousarray(input_image))
    d = ImageDraw.Draw(edited)
    for x, y, lab in clicked_points:
        color = (255, 0, 0) if lab == 1 else (0, 0, 255)
        d.ellipse([x - radius, y - radius, x + radius, y + radius], fill=color)
    edited = np.asarray(edited).astype(np.float64)
    green = (mask_image * np.array([0.0, 1.0, 0.0])).astype(np.uint8).astype(np.float64)
    out = np.rint(edited * opacity_edited + green * opacity_mask)
    return np.clip(out, 0, 255).astype(np.uint8)
