"""Frame preparation of the frame-driven accumulator: host-side mirror of ``crop_image`` / ``compress_image`` /
``process_images`` in /root/reference/simulation/simulationcode_v4_transistor_uav.m:104-143.

``compress_image`` is ``imresize(im2double(image), [floor(H/n), floor(W/m)], 'lanczos3')``.  MATLAB's imresize is not
in the reference; this follows its published algorithm (the ``contributions`` routine of imresize.m): for a
shrinking axis the Lanczos-3 kernel is stretched by 1/scale (antialiasing), every output sample takes
``ceil(6/scale) + 2`` source taps starting at ``floor(u - 3/scale)`` with ``u = x/scale + 0.5 (1 - 1/scale)``, the
weights are normalised to sum 1, indices outside the image are mirrored, and the axis with the smaller scale is
resized first (rows first on a tie).  PARITY UNPINNED (no MATLAB/Octave here, no stored compressed frames in the
reference).  The outputs are 4x4 grids: this is a cold path on the host (NumPy, float64); the per-pixel ODE of the
grid runs on the GPU (``nsof.simulate_frames``).
"""
import numpy as np

from .errors import NsofValueError


def _lanczos3(x):
    x = np.asarray(x, np.float64)
    eps = np.finfo(np.float64).eps
    f = (np.sin(np.pi * x) * np.sin(np.pi * x / 3) + eps) / ((np.pi ** 2 * x ** 2 / 3) + eps)
    return f * (np.abs(x) < 3)


def _contributions(in_len, out_len, scale):
    """(weights [out_len][P], indices [out_len][P]) of one axis, 0-based indices."""
    kernel_width = 6.0
    if scale < 1:
        h = lambda t: scale * _lanczos3(scale * t)  # noqa: E731
        kernel_width /= scale
    else:
        h = _lanczos3
    x = np.arange(1, out_len + 1, dtype=np.float64)
    u = x / scale + 0.5 * (1 - 1 / scale)
    left = np.floor(u - kernel_width / 2)
    p = int(np.ceil(kernel_width)) + 2
    ind = left[:, None] + np.arange(p)[None, :]                   # 1-based
    wts = h(u[:, None] - ind)
    wts = wts / wts.sum(axis=1, keepdims=True)
    aux = np.concatenate([np.arange(1, in_len + 1), np.arange(in_len, 0, -1)])
    ind = aux[np.mod(ind.astype(np.int64) - 1, aux.size)] - 1     # mirrored, 0-based
    keep = np.any(wts != 0, axis=0)
    return wts[:, keep], ind[:, keep]


def imresize_lanczos3(image, out_h, out_w):
    """``imresize(image, [out_h out_w], 'lanczos3')`` for float64 images [H][W] or [H][W][C] (no clamping, as MATLAB
    does for double input)."""
    img = np.asarray(image, np.float64)
    if img.ndim not in (2, 3) or out_h < 1 or out_w < 1:
        raise NsofValueError("imresize_lanczos3: [H][W] or [H][W][C] image and a positive size expected")
    scales = (out_h / img.shape[0], out_w / img.shape[1])
    order = (0, 1) if scales[0] <= scales[1] else (1, 0)
    for ax in order:
        wts, ind = _contributions(img.shape[ax], (out_h, out_w)[ax], scales[ax])
        moved = np.moveaxis(img, ax, 0)                            # [in][...]
        out = np.zeros((wts.shape[0],) + moved.shape[1:], np.float64)
        for k in range(wts.shape[1]):                               # taps in MATLAB's order (left to right)
            out += wts[:, k].reshape((-1,) + (1,) * (moved.ndim - 1)) * moved[ind[:, k]]
        img = np.moveaxis(out, 0, ax)
    return img


def im2double(image):
    a = np.asarray(image)
    if a.dtype == np.uint8:
        return a.astype(np.float64) / 255.0
    if a.dtype == np.uint16:
        return a.astype(np.float64) / 65535.0
    return a.astype(np.float64)


def crop_image(image, region_ul, region_lr):
    """``image(ul(1):lr(1), ul(2):lr(2), :)`` with MATLAB's 1-based inclusive corners ([y, x])."""
    return np.asarray(image)[region_ul[0] - 1:region_lr[0], region_ul[1] - 1:region_lr[1]]


def compress_image(image, m, n):
    """:111-121 -- Lanczos-3 resize of the double image to ``[floor(H/n), floor(W/m)]``."""
    h, w = np.asarray(image).shape[:2]
    return imresize_lanczos3(im2double(image), h // n, w // m)


def process_images(images, m, n, region_ul=None, region_lr=None):
    """:128-143 without the JPEG side effects: crop (optional) and compress every frame -> float64 [n][h][w]
    (gray frames) ready for ``nsof.simulate_frames``."""
    out = []
    for im in images:
        if region_ul is not None:
            im = crop_image(im, region_ul, region_lr)
        out.append(compress_image(im, m, n))
    return np.stack(out)
