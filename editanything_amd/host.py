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
    """…inpaint.py:142-163: PIL / uint8 HWC / tensor -> float32 [B,3,H,W] in [-1, 1]."""
    if isinstance(image, torch.Tensor):
        return image if image.ndim == 4 else image[None]
    if isinstance(image, Image.Image):
        image = np.asarray(image.convert("RGB"))
    image = np.asarray(image)
    if image.ndim == 3:
        image = image[None]
    return torch.from_numpy(image.transpose(0, 3, 1, 2).copy()).float() / 127.5 - 1.0


def prepare_mask_image(mask):
    """…inpaint.py:166-187: -> float32 [B,1,H,W] binarised at 0.5."""
    if isinstance(mask, torch.Tensor):
        m = mask.float()
        if m.ndim == 2:
            m = m[None, None]
        elif m.ndim == 3:
            m = m[:, None] if m.shape[0] != 1 else m[None]
    else:
        if isinstance(mask, Image.Image):
            mask = np.asarray(mask.convert("L"))
        m = np.asarray(mask).astype(np.float32) / 255.0
        if m.ndim == 3:
            m = m[..., 0]
        m = torch.from_numpy(m)[None, None]
    m = m.clone()
    m[m < 0.5] = 0
    m[m >= 0.5] = 1
    return m


def numpy_to_pil(images):
    if images.ndim == 3:
        images = images[None]
    images = (images * 255).round().astype("uint8")
    return [Image.fromarray(im) for im in images]
