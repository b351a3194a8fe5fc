"""Frame prediction by flow warp + SSIM: host-side mirror of the prediction task of
/root/reference/optical_flow_prediction.py (``task_results`` :255-353, baseline warp :581-591,
``calculateIntegralError`` :113-115) and of the cv2 / scikit-image calls inside it.  SURVEY.md section 8f row 2.

    flow_map = (grid + flow).astype(np.float32)
    new_frame[y0:y1, x0:x1, c] = cv2.remap(next_frame[:, :, c], flow_map[..., 0], flow_map[..., 1],
                                           cv2.INTER_LINEAR, borderMode=cv2.BORDER_REPLICATE)
    ssim = structural_similarity(true[:, :, 2], new_frame[:, :, 2], data_range=255.0)

All per-pixel work runs in libnsof.so (csrc/warp_kernels.hip): the map is formed inside the warp kernel from the flow,
the three channels are warped together, the SSIM statistics are exact integer window sums reduced in a fixed order.
"""
import time

import numpy as np

from . import _lib
from .context import default_context, dev_ptr
from .errors import NsofValueError

INTER_LINEAR = 1                         # cv2.INTER_LINEAR
BORDER_CONSTANT, BORDER_REPLICATE = 0, 1  # cv2.BORDER_*


def gray_u8_dev(d_frame, d_gray, code="RGB2GRAY", *, ctx=None):
    """``cv2.cvtColor(frame, COLOR_RGB2GRAY | COLOR_BGR2GRAY)`` on DEVICE memory: ``d_frame`` uint8 [H][W][3] torch CUDA
    tensor (rows may be strided), ``d_gray`` uint8 [H][W].  Same bytes as ``nsof.gating.frame_to_gray``; asynchronous on
    the context's stream -- decoded frames need not visit the host before the flow call."""
    ctx = ctx or default_context()
    if code not in ("RGB2GRAY", "BGR2GRAY"):
        raise NsofValueError("gray_u8_dev: code must be RGB2GRAY or BGR2GRAY", _lib.NSOF_EINVAL)
    h, w = int(d_frame.shape[0]), int(d_frame.shape[1])
    if tuple(d_frame.shape) != (h, w, 3) or tuple(d_gray.shape) != (h, w) or d_frame.stride(2) != 1 or d_frame.stride(1) != 3 \
            or d_gray.stride(1) != 1:
        raise NsofValueError("gray_u8_dev: uint8 [H][W][3] interleaved frame and [H][W] output expected", _lib.NSOF_ESHAPE)
    rc = ctx._lib.nsof_gray_u8_dev(ctx.ptr, dev_ptr(d_frame), int(d_frame.stride(0)), w, h, 1 if code == "BGR2GRAY" else 0,
                                   dev_ptr(d_gray), int(d_gray.stride(0)))
    ctx.check(rc, "gray_u8_dev")
    return d_gray


def remap(src, map1, map2, interpolation=INTER_LINEAR, dst=None, borderMode=BORDER_CONSTANT, borderValue=0,  # noqa: N803
          *, ctx=None):
    """``cv2.remap`` for uint8 images (1 or 3 channels), two float32 maps and INTER_LINEAR."""
    import torch
    ctx = ctx or default_context()
    if interpolation != INTER_LINEAR:
        raise NsofValueError("remap: only INTER_LINEAR is provided", _lib.NSOF_EUNSUPPORTED)
    src = np.asarray(src)
    if src.dtype != np.uint8 or src.ndim not in (2, 3):
        raise NsofValueError("remap: uint8 image expected", _lib.NSOF_EINVAL)
    map1 = np.ascontiguousarray(map1, np.float32)
    map2 = np.ascontiguousarray(map2, np.float32)
    if map1.ndim != 2 or map1.shape != map2.shape:
        raise NsofValueError("remap: map1 and map2 must be 2-D float32 arrays of one shape", _lib.NSOF_ESHAPE)
    if isinstance(borderValue, (tuple, list)):
        borderValue = borderValue[0]  # noqa: N806
    cn = 1 if src.ndim == 2 else src.shape[2]
    sh, sw = src.shape[:2]
    dh, dw = map1.shape
    dev = torch.device("cuda", ctx.device)
    d_src = torch.from_numpy(np.ascontiguousarray(src)).to(dev)
    d_mx, d_my = torch.from_numpy(map1).to(dev), torch.from_numpy(map2).to(dev)
    d_dst = torch.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    rc = ctx._lib.nsof_remap_linear_u8_dev(ctx.ptr, dev_ptr(d_src), sw * cn, sw, sh, cn, dev_ptr(d_mx), dw,
                                           dev_ptr(d_my), dw, dw, dh, int(borderMode), int(borderValue),
                                           dev_ptr(d_dst), dw * cn)
    ctx.check(rc, "remap")
    ctx.synchronize()
    out = d_dst.cpu().numpy()
    if dst is not None:
        dst[...] = out
        return dst
    return out


def predict_region(next_frame, flow, rect, out=None, sign=-1, borderMode=BORDER_REPLICATE, *, ctx=None):  # noqa: N803
    """Warp ``next_frame`` inside ``rect = (x0, y0, x1, y1)`` with ``sign * flow`` (``flow`` = the frame-sized flow
    canvas, un-negated Farneback output for ``sign=-1``); the rest of ``out`` (default: a copy of the frame,
    prediction.py:263) is left as it is."""
    ctx = ctx or default_context()
    frame = np.asarray(next_frame)
    if frame.dtype != np.uint8 or frame.ndim not in (2, 3) or (frame.ndim == 3 and frame.strides[2] != 1):
        raise NsofValueError("predict_region: uint8 frame expected", _lib.NSOF_EINVAL)
    if frame.strides[1] != (1 if frame.ndim == 2 else frame.shape[2]):
        frame = np.ascontiguousarray(frame)
    h, w = frame.shape[:2]
    cn = 1 if frame.ndim == 2 else frame.shape[2]
    x0, y0, x1, y1 = (int(v) for v in rect)
    if out is None:
        out = frame.copy()
    if x1 <= x0 or y1 <= y0:
        return out
    crop = np.asarray(flow)[y0:y1, x0:x1]
    if crop.dtype != np.float32 or crop.strides[2] != 4 or crop.strides[1] != 8:
        crop = np.ascontiguousarray(crop, np.float32)     # float64 canvases hold float32 values
    rc = ctx._lib.nsof_predict_warp_u8(ctx.ptr, frame.ctypes.data, frame.strides[0], w, h, cn, crop.ctypes.data,
                                       crop.strides[0], int(sign), x0, y0, x1, y1, int(borderMode), out.ctypes.data,
                                       out.strides[0])
    ctx.check(rc, "predict_region")
    return out


def predict_region_dev(d_frame, d_flow, d_out, height, width, rect, channels=3, sign=-1, borderMode=BORDER_REPLICATE,  # noqa: N803
                       *, ctx=None):
    """Device-resident variant (torch tensors or raw addresses, dense layouts); asynchronous."""
    ctx = ctx or default_context()
    x0, y0, x1, y1 = (int(v) for v in rect)
    rc = ctx._lib.nsof_predict_warp_u8_dev(ctx.ptr, dev_ptr(d_frame), width * channels, width, height, channels,
                                           dev_ptr(d_flow), 2 * width, int(sign), x0, y0, x1, y1, int(borderMode),
                                           dev_ptr(d_out), width * channels)
    ctx.check(rc, "predict_region_dev")


def task_results(prev_frame, next_frame, flow, num_labels, regions_info, EST_FLAG=2, MERGE_FLAG=False, padding=20,  # noqa: N803
                 sign=1, times=None, *, ctx=None):
    """optical_flow_prediction.py:255-353.  ``flow`` is the canvas the caller already negated (:545), hence
    ``sign=1`` here; pass the raw ``opticalFlow3D`` canvas with ``sign=-1`` to skip that host pass."""
    t0 = time.time()
    h, w = prev_frame.shape[:2]
    new_frame = np.array(next_frame, copy=True)
    if num_labels > 1:
        if EST_FLAG == 1 and MERGE_FLAG:
            boxes = [(max(0, min(r[0] for r in regions_info) - padding), max(0, min(r[1] for r in regions_info) - padding),
                      min(w, max(r[2] for r in regions_info) + padding), min(h, max(r[3] for r in regions_info) + padding))]
        elif EST_FLAG == 1:
            boxes = list(regions_info)
        else:
            boxes = [tuple(regions_info)]
        for box in boxes:   # later boxes see the original frame, like the reference (it always warps next_frame)
            predict_region(next_frame, flow, box, out=new_frame, sign=sign, ctx=ctx)
    if times is not None:
        times.append(time.time() - t0)
    return new_frame


def structural_similarity(im1, im2, data_range=255.0, *, ctx=None):
    """``skimage.metrics.structural_similarity(im1, im2, data_range=...)`` with its defaults, for 2-D uint8 views."""
    ctx = ctx or default_context()
    a, b = np.asarray(im1), np.asarray(im2)
    if a.dtype != np.uint8 or b.dtype != np.uint8 or a.ndim != 2 or a.shape != b.shape:
        raise NsofValueError("structural_similarity: two uint8 2-D images of one shape expected", _lib.NSOF_ESHAPE)
    if min(a.shape) < 7:
        raise NsofValueError("win_size exceeds image extent", _lib.NSOF_ESHAPE)
    if a.strides[1] < 1 or b.strides[1] < 1:
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    import ctypes as C
    out = C.c_double()
    rc = ctx._lib.nsof_ssim_u8(ctx.ptr, a.ctypes.data, a.strides[0], a.strides[1], b.ctypes.data, b.strides[0],
                               b.strides[1], a.shape[1], a.shape[0], float(data_range), C.byref(out))
    ctx.check(rc, "structural_similarity")
    return out.value


def calculateIntegralError(prediction, true, *, ctx=None):  # noqa: N802
    """optical_flow_prediction.py:113-115: SSIM of channel 2 (red of a BGR frame)."""
    return structural_similarity(np.asarray(true)[:, :, 2], np.asarray(prediction)[:, :, 2], data_range=255.0, ctx=ctx)
