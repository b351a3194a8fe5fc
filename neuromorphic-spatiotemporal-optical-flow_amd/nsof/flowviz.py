"""Middlebury colour coding of a flow field: host-side mirror of the reference's ``flow_viz`` module
(/root/reference/flow_viz.py:20-135 -- ``make_colorwheel``, ``flow_uv_to_colors``, ``flow_to_image``), which the
scripts use to write their flow PNGs (``viz``, optical_flow_seg.py:106-113).  SURVEY.md section 8f row 4.

The colour wheel is the published one of Baker et al. (ICCV 2007): 55 hues in six segments RY/YG/GC/CB/BM/MR of
15/6/4/11/13/6 steps.  Hue = direction, saturation = magnitude / max magnitude; vectors longer than the
normaliser are dimmed to 75 %.  Plain NumPy on the host (a presentation step, not on the hot path); float64
like the reference, which feeds it the float64 canvas of ``opticalFlow3D``.
"""
import numpy as np

_SEGMENTS = (("RY", 15), ("YG", 6), ("GC", 4), ("CB", 11), ("BM", 13), ("MR", 6))


def make_colorwheel():
    """[55, 3] float64 RGB wheel.  Each segment ramps one channel up or down while another sits at 255."""
    n = sum(k for _, k in _SEGMENTS)
    wheel = np.zeros((n, 3))
    # (channel held at 255, channel ramped, ramp direction) per segment
    plan = ((0, 1, +1), (1, 0, -1), (1, 2, +1), (2, 1, -1), (2, 0, +1), (0, 2, -1))
    at = 0
    for (_, k), (hold, ramp, sign) in zip(_SEGMENTS, plan):
        step = np.floor(255 * np.arange(k) / k)
        wheel[at:at + k, hold] = 255
        wheel[at:at + k, ramp] = step if sign > 0 else 255 - step
        at += k
    return wheel


def flow_uv_to_colors(u, v, convert_to_bgr=False):
    """u, v already divided by the normaliser.  Returns uint8 [H, W, 3]."""
    u = np.asarray(u)
    v = np.asarray(v)
    wheel = make_colorwheel()
    n = wheel.shape[0]
    rad = np.sqrt(np.square(u) + np.square(v))
    pos = (np.arctan2(-v, -u) / np.pi + 1) / 2 * (n - 1)
    lo = np.floor(pos).astype(np.int32)
    hi = lo + 1
    hi[hi == n] = 0
    frac = pos - lo
    small = rad <= 1
    img = np.zeros(u.shape + (3,), np.uint8)
    for c in range(3):
        col = (1 - frac) * (wheel[lo, c] / 255.0) + frac * (wheel[hi, c] / 255.0)
        col = np.where(small, 1 - rad * (1 - col), col * 0.75)
        img[..., 2 - c if convert_to_bgr else c] = np.floor(255 * col)
    return img


def flow_to_image(flow_uv, clip_flow=None, convert_to_bgr=False, max_flow=None):
    """flow_uv [H, W, 2] -> uint8 [H, W, 3].  Normalised by the largest magnitude (+1e-5) unless max_flow is given."""
    flow_uv = np.asarray(flow_uv)
    if flow_uv.ndim != 3 or flow_uv.shape[2] != 2:
        raise ValueError("input flow must have shape [H,W,2]")
    if clip_flow is not None:
        flow_uv = np.clip(flow_uv, 0, clip_flow)
    u, v = flow_uv[..., 0], flow_uv[..., 1]
    top = np.max(np.sqrt(np.square(u) + np.square(v))) if max_flow is None else max_flow
    return flow_uv_to_colors(u / (top + 1e-5), v / (top + 1e-5), convert_to_bgr)


def viz(flo, imgname):
    """``viz`` of the scripts (optical_flow_seg.py:11-19): colour-code the flow and save it with the channels in
    B,G,R order, as the reference does (it hands ``flo[:, :, [2, 1, 0]]`` to PIL).  Needs Pillow."""
    from PIL import Image
    img = flow_to_image(flo)[:, :, [2, 1, 0]]
    Image.fromarray(np.ascontiguousarray(img)).save(imgname)
