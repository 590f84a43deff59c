"""ORACLE -- TEST INFRASTRUCTURE.  Pure-Python/numpy restatement of the reference's host-side integer logic.

  show_anns      sam2image.py:92-115 / editany_lora.py:426-449  (id map -> 2-byte RGB encoding; bit-exact target)
  HWC3           annotator/util.py:9-25
  resize_shape   annotator/util.py:28-37 (the size arithmetic; the cv2 resampling itself is identity at 512/512)
  control_tensor sam2image.py:154-161
"""
import numpy as np


def show_anns_idmap(anns):
    """The reference sorts by area but then indexes the UNSORTED list (ann = anns[i]), so ids follow list order and
    later masks overwrite earlier ones.  Returns the float64 [H,W,3] `res` array (ch0 = id % 256, ch1 = id // 256)."""
    if len(anns) == 0:
        return None
    m0 = anns[0]["segmentation"]
    h, w = m0.shape
    idmap = [[0] * w for _ in range(h)]
    for i in range(len(anns)):
        m = anns[i]["segmentation"]
        for y in range(h):
            row = m[y]
            for x in range(w):
                if row[x] != 0:
                    idmap[y][x] = i + 1
    res = np.zeros((h, w, 3))
    for y in range(h):
        for x in range(w):
            v = idmap[y][x] & 0xFFFF
            res[y, x, 0] = v % 256
            res[y, x, 1] = v // 256
    return res


def hwc3(x):
    assert x.dtype == np.uint8
    if x.ndim == 2:
        x = x[:, :, None]
    h, w, c = x.shape
    assert c in (1, 3, 4)
    if c == 3:
        return x
    if c == 1:
        return np.concatenate([x, x, x], axis=2)
    color = x[:, :, 0:3].astype(np.float32)
    alpha = x[:, :, 3:4].astype(np.float32) / 255.0
    y = color * alpha + 255.0 * (1.0 - alpha)
    return y.clip(0, 255).astype(np.uint8)


def resize_shape(h, w, resolution):
    k = float(resolution) / min(float(h), float(w))
    return int(np.round(h * k / 64.0)) * 64, int(np.round(w * k / 64.0)) * 64


def control_tensor(res, num_samples):
    """detected_map.astype(uint8) -> HWC3 -> float [n,3,H,W], values 0..255 NOT divided by 255."""
    det = hwc3(res.astype(np.uint8))
    c = det.astype(np.float32)
    return np.stack([c] * num_samples, 0).transpose(0, 3, 1, 2).copy()
