"""Motion-segmentation head: host-side mirror of the reference's ``task_results`` / ``process_flow_region``
(/root/reference/optical_flow_seg.py:253-357; baseline copy :503-537) and of the cv2 calls inside them.
SURVEY.md section 8f row 3.

    mag, ang = cv2.cartToPolar(fx, fy); mask[mag > SEG_TH] = 255
    5 x { cv2.dilate(mask, ellipse 10x10); cv2.erode(mask, ellipse 10x10) }; threshold(1) -> 0/255

Everything per pixel runs in libnsof.so (bit-packed masks, one fused launch for the whole dilate/erode chain;
csrc/segment_kernels.hip).  The hue/value image the reference also builds (:327-343) never reaches its result and
is not computed.
"""
import time

import numpy as np

from . import _lib
from .context import default_context, dev_ptr
from .errors import NsofValueError

MORPH_RECT, MORPH_CROSS, MORPH_ELLIPSE = 0, 1, 2     # cv2.MORPH_*


def getStructuringElement(shape, ksize):  # noqa: N802
    """``cv2.getStructuringElement(shape, (width, height))`` -> uint8 [height][width] of 0/1."""
    kw, kh = int(ksize[0]), int(ksize[1])
    out = np.zeros((max(kh, 0), max(kw, 0)), np.uint8)
    rc = _lib.load().nsof_structuring_element(int(shape), kw, kh, out.ctypes.data)
    if rc:
        raise NsofValueError(f"getStructuringElement({shape}, {ksize}): bad argument", rc)
    return out


def _morph(op, src, kernel, anchor, iterations, ctx):
    import torch
    ctx = ctx or default_context()
    src = np.asarray(src)
    if src.ndim == 3:   # the baseline path pushes a 3-channel copy of the mask through (seg.py:526-533)
        return np.stack([_morph(op, src[..., c], kernel, anchor, iterations, ctx) for c in range(src.shape[2])], -1)
    if src.ndim != 2 or src.dtype != np.uint8:
        raise NsofValueError("dilate/erode: uint8 single-channel mask expected", _lib.NSOF_EINVAL)
    kernel = np.ascontiguousarray(kernel, np.uint8)
    h, w = src.shape
    dev = torch.device("cuda", ctx.device)
    d_src = torch.from_numpy(np.ascontiguousarray(src)).to(dev)
    d_dst = torch.empty_like(d_src)
    torch.cuda.synchronize(dev)
    rc = ctx._lib.nsof_morph_binary_u8_dev(ctx.ptr, op, dev_ptr(d_src), w, w, h, kernel.ctypes.data, kernel.shape[1],
                                           kernel.shape[0], int(anchor[0]), int(anchor[1]), int(iterations),
                                           dev_ptr(d_dst), w)
    ctx.check(rc, "dilate" if op else "erode")
    ctx.synchronize()
    return d_dst.cpu().numpy()


def dilate(src, kernel, anchor=(-1, -1), iterations=1, *, ctx=None):
    """``cv2.dilate`` for two-valued uint8 masks (non-zero = set; result 0/255, which is what cv2 returns for the
    0/255 masks the reference feeds it)."""
    return _morph(1, src, kernel, anchor, iterations, ctx)


def erode(src, kernel, anchor=(-1, -1), iterations=1, *, ctx=None):
    """``cv2.erode`` for two-valued uint8 masks."""
    return _morph(0, src, kernel, anchor, iterations, ctx)


def motion_mask(flow, seg_th=1, ksize=10, iterations=5, out=None, *, ctx=None):
    """The whole head on a flow field or a strided crop of one: uint8 [H][W] of 0/255.  float64 canvases (what
    ``opticalFlow3D`` returns) hold float32 values and are narrowed without loss; the sign of the flow is
    irrelevant."""
    ctx = ctx or default_context()
    flow = np.asarray(flow)
    if flow.ndim != 3 or flow.shape[2] != 2:
        raise NsofValueError("flow must have shape [H,W,2]", _lib.NSOF_ESHAPE)
    if flow.dtype != np.float32 or flow.strides[2] != 4 or flow.strides[1] != 8:
        flow = np.ascontiguousarray(flow, np.float32)
    h, w = flow.shape[:2]
    if out is None:
        out = np.empty((h, w), np.uint8)
    if h == 0 or w == 0:
        return out
    rc = ctx._lib.nsof_motion_mask(ctx.ptr, flow.ctypes.data, flow.strides[0], w, h, float(seg_th), int(ksize),
                                   int(iterations), out.ctypes.data, out.strides[0])
    ctx.check(rc, "motion_mask")
    return out


def motion_mask_dev(d_flow, d_mask, height, width, seg_th=1, ksize=10, iterations=5, *, flow_stride=None,
                    mask_stride=None, ctx=None):
    """Device-resident variant (torch tensors or raw addresses); asynchronous on the context's stream."""
    ctx = ctx or default_context()
    rc = ctx._lib.nsof_motion_mask_dev(ctx.ptr, dev_ptr(d_flow), 2 * width if flow_stride is None else flow_stride,
                                       width, height, float(seg_th), int(ksize), int(iterations), dev_ptr(d_mask),
                                       width if mask_stride is None else mask_stride)
    ctx.check(rc, "motion_mask_dev")


def process_flow_region(mag, ang=None, seg_th=1, *, ctx=None):
    """optical_flow_seg.py:322-357 with the same arguments (``ang`` only feeds the unused hue image)."""
    mask = np.where(np.asarray(mag) > seg_th, np.uint8(255), np.uint8(0))
    k = getStructuringElement(MORPH_ELLIPSE, (10, 10))
    for _ in range(5):
        mask = erode(dilate(mask, k, ctx=ctx), k, ctx=ctx)
    return np.where(mask > 1, np.uint8(255), np.uint8(0))


def task_results(prev_frame, next_frame, flow, num_labels, regions_info, EST_FLAG=2, MERGE_FLAG=False,  # noqa: N803
                 padding=20, seg_th=1, times=None, *, ctx=None):
    """optical_flow_seg.py:253-320: the head on the gated region(s) only, pasted into an all-zero mask.
    ``regions_info`` is what ``opticalFlow3D`` returned (a list of boxes for FLAG 1, one box for FLAG 2)."""
    t0 = time.time()
    h, w = prev_frame.shape[:2]
    motion_binary = np.zeros((h, w), np.uint8)
    if num_labels > 1:
        if EST_FLAG == 1 and MERGE_FLAG:
            boxes = [(max(0, min(r[0] for r in regions_info) - padding), max(0, min(r[1] for r in regions_info) - padding),
                      min(w, max(r[2] for r in regions_info) + padding), min(h, max(r[3] for r in regions_info) + padding))]
        elif EST_FLAG == 1:
            boxes = list(regions_info)
        else:
            boxes = [tuple(regions_info)]
        for x_min, y_min, x_max, y_max in boxes:
            if x_max > x_min and y_max > y_min:
                motion_mask(flow[y_min:y_max, x_min:x_max], seg_th, out=motion_binary[y_min:y_max, x_min:x_max], ctx=ctx)
    if times is not None:
        times.append(time.time() - t0)
    return motion_binary
