"""Dense Farneback flow: drop-in for ``cv2.calcOpticalFlowFarneback``.

Reference call sites: /root/reference/optical_flow_seg.py:158,203,494 (and the _ob/_prediction/
_yolo twins); always ``(prev_region, next_region, None, **farneback_params)`` with the keys
``pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags`` (:73-81); inputs may be
strided ROI views ``gray[y0:y1, x0:x1]`` (:186-187).
"""
import ctypes as C
from dataclasses import asdict, dataclass

import numpy as np

from . import _lib
from .context import Context, default_context, dev_ptr
from .errors import NsofValueError


@dataclass(frozen=True)
class FarnebackParams:
    """Keyword set of the reference's ``farneback_params`` dict (optical_flow_seg.py:73-81)."""
    pyr_scale: float = 0.5
    levels: int = 3
    winsize: int = 15
    iterations: int = 3
    poly_n: int = 5
    poly_sigma: float = 1.2
    flags: int = 0

    def as_kwargs(self):
        return asdict(self)


# parameter sets the reference ships (data/*/Parameters.txt)
PARAMS_A = FarnebackParams(0.5, 3, 15, 3, 5, 1.2, 0)    # grasp, uavnew2
PARAMS_B = FarnebackParams(0.6, 3, 3, 3, 10, 1.05, 0)   # autodriving, uav
PARAMS_C = FarnebackParams(0.6, 3, 4, 2, 1, 1.05, 0)    # tabletennis


def _as_gray_u8(a, name):
    if not isinstance(a, np.ndarray):
        raise NsofValueError(f"{name} is not a numpy array (got {type(a).__name__})")
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    if a.ndim != 2:
        raise NsofValueError(f"{name} must be single-channel (shape {a.shape}); cv2 asserts channels() == 1")
    if a.dtype != np.uint8:
        raise NsofValueError(f"{name} must be uint8 (got {a.dtype}); the reference only passes 8-bit gray frames")
    if a.size and a.strides[1] != 1:  # pixel stride must be 1; row stride is free (ROI views)
        a = np.ascontiguousarray(a)
    return a


_DEFAULT_EXACT = None   # install(exact=...) sets it: None = the context's setting (NSOF_EXACT_ROWSUMS), True / False = forced
_DEFAULT_LOW_LATENCY = None   # install(low_latency=...): None = the context's setting (NSOF_ROW_BANDS)


def calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags,  # noqa: A002,N802
                             *, ctx=None, exact=None, low_latency=None):
    """Same signature and result as ``cv2.calcOpticalFlowFarneback``: float32 (H, W, 2), (u, v) interleaved,
    such that ``next(x+u, y+v) ~ prev(x, y)``.  ``flow=None`` allocates; a matching float32 array is reused.
    ``exact`` (keyword only): True = box-filter row sums in the library's own order for this call
    (``NSOF_OPT_EXACT_ROWSUMS``, the context's default: bit-identical to the CPU restatement on any input),
    False = the fast mode (each pixel's window summed directly: a few per cent faster, up to ~8e-4 off where 2x2 systems
    are rank deficient), None = whatever the context / ``install(exact=...)`` says.
    ``low_latency`` (keyword only): True = the fast mode with row bands in the iteration kernel for this call
    (``NSOF_OPT_ROW_BANDS``: the fast mode's lone 1080p call drops from 3.7 to 1.1 ms -- the exact default is at that speed
    on its own since the small-batch form of round 3, so this is only of interest with ``exact=False``; column sums restart
    per band, so the flow moves in its 5th decimal, more where the 2x2 system is rank deficient -- the automatic mode
    therefore applies from winsize 9 up); ignored together with an explicit ``exact=True``.
    Raises ``nsof.error`` (a ``cv2.error`` where cv2 imports) on every failure, device-side ones included: a launch whose
    workgroup hand-over timed out never returns a flow field (``NSOF_EDEVICE``).
    The two keywords set context options around the call; the context's lock makes that safe for concurrent callers of
    one context."""
    prev = _as_gray_u8(prev, "prev")
    next = _as_gray_u8(next, "next")  # noqa: A001
    if prev.shape != next.shape:
        raise NsofValueError(f"prev {prev.shape} and next {next.shape} sizes differ", _lib.NSOF_ESHAPE)
    h, w = prev.shape
    if h == 0 or w == 0:
        raise NsofValueError("empty input image", _lib.NSOF_ESHAPE)
    if (isinstance(flow, np.ndarray) and flow.dtype == np.float32 and flow.shape == (h, w, 2)
            and flow.strides[2] == 4 and flow.strides[1] == 8 and flow.flags.writeable):
        out = flow
    else:
        out = np.empty((h, w, 2), np.float32)
    ctx = ctx or default_context()
    exact = _DEFAULT_EXACT if exact is None else exact
    low_latency = _DEFAULT_LOW_LATENCY if low_latency is None else low_latency
    if low_latency and exact is None:
        exact = False   # row bands belong to the fast row-sum mode
    with ctx.lock:
        saved = saved_bands = None
        if exact is not None:
            saved = ctx.get_option(_lib.OPT_EXACT_ROWSUMS)
            ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 1 if exact else 0)
        if low_latency is not None:
            saved_bands = ctx.get_option(_lib.OPT_ROW_BANDS)
            ctx.set_option(_lib.OPT_ROW_BANDS, 1 if low_latency else 0)
        try:
            rc = ctx._lib.nsof_farneback_u8(ctx.ptr, prev.ctypes.data, prev.strides[0], next.ctypes.data, next.strides[0],
                                            w, h, out.ctypes.data, out.strides[0], float(pyr_scale), int(levels),
                                            int(winsize), int(iterations), int(poly_n), float(poly_sigma), int(flags))
        finally:
            if saved is not None:
                ctx.set_option(_lib.OPT_EXACT_ROWSUMS, saved)
            if saved_bands is not None:
                ctx.set_option(_lib.OPT_ROW_BANDS, saved_bands)
        ctx.check(rc, "calcOpticalFlowFarneback")
    return out


def farneback_batch(d_prev, d_next, d_flow, n_pairs, height, width, params, *, row_stride=None, pair_stride=None,
                    ctx=None):
    """Device-resident batch: ``d_prev/d_next`` uint8 [n][H][row_stride], ``d_flow`` float32 [n][H][W][2]
    (torch tensors or raw device addresses).  Asynchronous on the context's stream."""
    ctx = ctx or default_context()
    row_stride = width if row_stride is None else row_stride
    pair_stride = row_stride * height if pair_stride is None else pair_stride
    p = params
    rc = ctx._lib.nsof_farneback_u8_batch_dev(ctx.ptr, n_pairs, dev_ptr(d_prev), dev_ptr(d_next), row_stride,
                                              pair_stride, width, height, dev_ptr(d_flow), p.pyr_scale, p.levels,
                                              p.winsize, p.iterations, p.poly_n, p.poly_sigma, p.flags)
    ctx.check(rc, "farneback_batch")


def farneback_sequence(d_frames, d_flow, n_frames, height, width, params, *, row_stride=None, frame_stride=None,
                       ctx=None):
    """Device-resident sequence: ``d_frames`` uint8 [n_frames][H][row_stride]; ``d_flow`` float32
    [n_frames-1][H][W][2] with flow i = frame i -> frame i+1 (the consecutive-pair walk of the reference's scripts).
    Per-frame work (pyramid, polynomial expansion) is shared between neighbouring pairs."""
    ctx = ctx or default_context()
    row_stride = width if row_stride is None else row_stride
    frame_stride = row_stride * height if frame_stride is None else frame_stride
    p = params
    rc = ctx._lib.nsof_farneback_u8_sequence_dev(ctx.ptr, n_frames, dev_ptr(d_frames), row_stride, frame_stride,
                                                 width, height, dev_ptr(d_flow), p.pyr_scale, p.levels, p.winsize,
                                                 p.iterations, p.poly_n, p.poly_sigma, p.flags)
    ctx.check(rc, "farneback_sequence")


class _PinnedOwner:
    """Keeps a page-locked allocation alive for as long as numpy views of it exist."""

    def __init__(self, nbytes):
        self._lib = _lib.load()
        self.ptr = self._lib.nsof_host_alloc(max(int(nbytes), 1))
        if not self.ptr:
            raise MemoryError(f"nsof_host_alloc({nbytes}) failed")

    def __del__(self):
        if getattr(self, "ptr", None):
            self._lib.nsof_host_free(self.ptr)
            self.ptr = None


def pinned_empty(shape, dtype=np.float32):
    """``np.empty`` in page-locked host memory: frames / flow fields in such arrays are copied to and from the GPU
    directly by ``farneback_pairs`` (no staging copy on the host)."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    owner = _PinnedOwner(n)
    buf = (C.c_char * max(n, 1)).from_address(owner.ptr)
    arr = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
    buf._nsof_owner = owner          # numpy keeps `buf` alive as the array's base; `buf` keeps the allocation
    return arr


def _desc_array(pairs, flows, host):
    """ctypes array of nsof_pair_desc for (prev, next) pairs and their flow fields (numpy arrays or, with
    host=False, objects exposing data_ptr()/shape/stride() like torch CUDA tensors)."""
    descs = (_lib.PairDesc * len(pairs))()
    keep = []
    for i, ((prev, nxt), flow) in enumerate(zip(pairs, flows)):
        d = descs[i]
        if host:
            prev, nxt = _as_gray_u8(prev, "prev"), _as_gray_u8(nxt, "next")
            if prev.shape != nxt.shape:
                raise NsofValueError(f"pair {i}: prev {prev.shape} and next {nxt.shape} sizes differ", _lib.NSOF_ESHAPE)
            keep += [prev, nxt]
            h, w = prev.shape
            d.prev, d.prev_stride, d.next, d.next_stride = prev.ctypes.data, prev.strides[0], nxt.ctypes.data, nxt.strides[0]
            d.flow, d.flow_stride = flow.ctypes.data, flow.strides[0]
        else:
            h, w = int(prev.shape[0]), int(prev.shape[1])
            if tuple(nxt.shape[:2]) != (h, w):
                raise NsofValueError(f"pair {i}: prev and next sizes differ", _lib.NSOF_ESHAPE)
            if prev.stride(1) != 1 or nxt.stride(1) != 1 or flow.stride(2) != 1 or flow.stride(1) != 2:
                raise NsofValueError(f"pair {i}: pixel strides must be 1 (row strides are free)")
            d.prev, d.prev_stride, d.next, d.next_stride = prev.data_ptr(), prev.stride(0), nxt.data_ptr(), nxt.stride(0)
            d.flow, d.flow_stride = flow.data_ptr(), flow.stride(0) * 4
        if h == 0 or w == 0:
            raise NsofValueError(f"pair {i}: empty input image", _lib.NSOF_ESHAPE)
        d.width, d.height = w, h
    return descs, keep


def farneback_pairs(pairs, params, flows=None, *, pinned=False, ctx=None):
    """Flow of MANY independent (prev, next) pairs of ANY shapes with one parameter set -- the gated path's ROI
    calls (optical_flow_seg.py:129-164, :186-203) and full-frame calls (:492-496) of a whole sequence in one go
    (``nsof_farneback_u8_batch``): the pairs share every kernel launch and upload / compute / download overlap.

    ``pairs``: [(prev, next), ...] uint8 2-D numpy arrays (strided ROI views allowed).  ``flows``: optional list of
    float32 (h, w, 2) arrays to write into (views ``canvas[y0:y1, x0:x1]`` of a frame-sized canvas are written in
    place -- the paste of :162/:204); by default fresh arrays are returned, page-locked when ``pinned`` (then the
    result is copied straight from the GPU into the array).  Each result equals ``calcOpticalFlowFarneback`` of
    that pair bit for bit."""
    ctx = ctx or default_context()
    kw = params.as_kwargs() if hasattr(params, "as_kwargs") else dict(params)
    pairs = list(pairs)
    if flows is None:
        alloc = pinned_empty if pinned else np.empty
        flows = [alloc((p.shape[0], p.shape[1], 2), np.float32) for p, _ in pairs]
    else:
        flows = list(flows)
        for i, ((p, _), f) in enumerate(zip(pairs, flows)):
            if not (isinstance(f, np.ndarray) and f.dtype == np.float32 and f.shape == (p.shape[0], p.shape[1], 2)
                    and f.strides[2] == 4 and f.strides[1] == 8 and f.flags.writeable):
                raise NsofValueError(f"flows[{i}] must be a writeable float32 ({p.shape[0]}, {p.shape[1]}, 2) array "
                                     "with contiguous pixels")
    if len(flows) != len(pairs):
        raise NsofValueError("flows and pairs differ in length")
    if not pairs:
        return []
    descs, keep = _desc_array(pairs, flows, host=True)
    rc = ctx._lib.nsof_farneback_u8_batch(ctx.ptr, len(pairs), descs, float(kw["pyr_scale"]), int(kw["levels"]),
                                          int(kw["winsize"]), int(kw["iterations"]), int(kw["poly_n"]),
                                          float(kw["poly_sigma"]), int(kw["flags"]))
    ctx.check(rc, "farneback_pairs")
    del keep
    return flows


def farneback_pairs_dev(pairs, flows, params, *, ctx=None):
    """Device-resident twin (``nsof_farneback_u8_batch_desc_dev``): ``pairs`` = [(prev, next), ...] of uint8 CUDA
    tensors (any row stride: crops ``frame[y0:y1, x0:x1]`` of frames in HBM), ``flows`` = float32 (h, w, 2) CUDA
    tensors or crops of a frame-sized canvas, written in place.  Asynchronous on the context's stream."""
    ctx = ctx or default_context()
    kw = params.as_kwargs() if hasattr(params, "as_kwargs") else dict(params)
    pairs, flows = list(pairs), list(flows)
    if len(flows) != len(pairs):
        raise NsofValueError("flows and pairs differ in length")
    if not pairs:
        return
    descs, _ = _desc_array(pairs, flows, host=False)
    rc = ctx._lib.nsof_farneback_u8_batch_desc_dev(ctx.ptr, len(pairs), descs, float(kw["pyr_scale"]), int(kw["levels"]),
                                                   int(kw["winsize"]), int(kw["iterations"]), int(kw["poly_n"]),
                                                   float(kw["poly_sigma"]), int(kw["flags"]))
    ctx.check(rc, "farneback_pairs_dev")


def farneback_roi_sequence_dev(frames, counts, rects, flows, params, *, gate_frame=0, ctx=None):
    """The gated path of a frame sequence on the device (``nsof_farneback_u8_roi_sequence_dev``; opticalFlow3D's crop ->
    flow -> paste loop, optical_flow_seg.py:129-164, 186-204): ``frames`` uint8 CUDA tensor [n][H][W] (row stride free),
    ``counts`` / ``rects`` the device ROI table of ``gating.roi_from_surface_dev``, ``flows`` float32 CUDA tensor
    [n-1][H][W][2] (contiguous; zero-filled by the call).  Pair k is gated by the rectangles of frame ``k + gate_frame``:
    0 (default) = the map of the pair's first frame, as the shipped scripts gate (``memimg2 := memimg1``,
    optical_flow_seg.py:435; ``GatingConfig.bug_compatible``), 1 = the map of its second frame (what ``opticalFlow3D``
    is written to use).  All crops of all pairs form one work list; overlapping crops of a pair are pasted in label order.
    -> (n_crops, crop_pixels)."""
    ctx = ctx or default_context()
    kw = params.as_kwargs() if hasattr(params, "as_kwargs") else dict(params)
    n, h, w = (int(v) for v in frames.shape)
    if tuple(flows.shape) != (n - 1, h, w, 2) or not flows.is_contiguous() or frames.stride(2) != 1:
        raise NsofValueError("flows must be a contiguous (n-1, H, W, 2) tensor and the frames' pixel stride 1")
    if tuple(rects.shape[:1]) != (n,) or rects.shape[2] != 4 or not rects.is_contiguous() or not counts.is_contiguous():
        raise NsofValueError("rects must be a contiguous (n, max_rects, 4) int32 tensor")
    calls, pixels = C.c_longlong(), C.c_longlong()
    rc = ctx._lib.nsof_farneback_u8_roi_sequence_dev(
        ctx.ptr, n, dev_ptr(frames), int(frames.stride(1)), int(frames.stride(0)), w, h, dev_ptr(counts), dev_ptr(rects),
        int(rects.shape[1]), dev_ptr(flows), float(kw["pyr_scale"]), int(kw["levels"]), int(kw["winsize"]),
        int(kw["iterations"]), int(kw["poly_n"]), float(kw["poly_sigma"]), int(kw["flags"]), int(gate_frame), C.byref(calls),
        C.byref(pixels))
    ctx.check(rc, "farneback_roi_sequence_dev")
    return calls.value, pixels.value


def effective_levels(width, height, pyr_scale, levels):
    return _lib.load().nsof_farneback_effective_levels(width, height, pyr_scale, levels)


def level_size(width, height, pyr_scale, level):
    """-> (level_width, level_height, blur_ksize, blur_sigma)"""
    lw, lh, ks, sg = C.c_int(), C.c_int(), C.c_int(), C.c_double()
    rc = _lib.load().nsof_farneback_level_size(width, height, pyr_scale, level, C.byref(lw), C.byref(lh),
                                               C.byref(ks), C.byref(sg))
    if rc:
        raise NsofValueError(f"bad level geometry ({width}x{height}, pyr_scale={pyr_scale}, level={level})", rc)
    return lw.value, lh.value, ks.value, sg.value


class StreamPool:
    """K contexts (one HIP stream and workspace each) with one worker thread per context.  Small images -- the ROI
    crops of the gated path -- leave most of the 256 CUs idle; independent calls issued from several streams overlap
    on the GPU (measured: 64 ROI pairs of 520x200 take 0.95 ms each on one stream, 0.36 ms on eight).  ctypes
    releases the GIL inside the C call, so plain threads are enough."""

    def __init__(self, n_streams=8, device=None):
        from concurrent.futures import ThreadPoolExecutor
        self._ctxs = [Context(device) for _ in range(int(n_streams))]
        self._free = list(self._ctxs)
        import threading
        self._lock = threading.Lock()
        self._pool = ThreadPoolExecutor(max_workers=len(self._ctxs))

    def _call(self, prev, nxt, kw):
        with self._lock:
            ctx = self._free.pop()
        try:
            return calcOpticalFlowFarneback(prev, nxt, None, **kw, ctx=ctx)
        finally:
            with self._lock:
                self._free.append(ctx)

    def map(self, pairs, params):
        """``[(prev, next), ...]`` -> list of flows, in order; ``params``: FarnebackParams or a kwargs dict."""
        kw = params.as_kwargs() if hasattr(params, "as_kwargs") else dict(params)
        return list(self._pool.map(lambda pq: self._call(pq[0], pq[1], kw), pairs))

    def close(self):
        self._pool.shutdown(wait=True)
        for c in self._ctxs:
            c.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def farneback_many(pairs, params, n_streams=8, pool=None):
    """Flow of many independent (prev, next) pairs of any shapes, overlapped over ``n_streams`` HIP streams."""
    if pool is not None:
        return pool.map(pairs, params)
    with StreamPool(min(n_streams, max(len(pairs), 1))) as sp:
        return sp.map(pairs, params)


_saved_cv2_fn = None


def install(cv2_module=None, exact=None, low_latency=None):
    """Assign ``calcOpticalFlowFarneback`` onto ``cv2`` (the reference looks the attribute up at call
    time, so its scripts then run on the GPU unmodified).  Returns the patched module.  ``exact=False`` makes every
    call through the drop-in use the fast row-sum mode instead of the library's own order (the default, bit-faithful on
    any footage); ``low_latency=True`` makes every call use that mode with row bands (one call
    per camera frame: 1.1 instead of 3.7 ms at 1080p)."""
    global _saved_cv2_fn, _DEFAULT_EXACT, _DEFAULT_LOW_LATENCY
    _DEFAULT_EXACT = exact
    _DEFAULT_LOW_LATENCY = low_latency
    if cv2_module is None:
        import cv2 as cv2_module  # raises ImportError where cv2 is absent: nothing to patch
    if _saved_cv2_fn is None:
        _saved_cv2_fn = getattr(cv2_module, "calcOpticalFlowFarneback", None)
    cv2_module.calcOpticalFlowFarneback = calcOpticalFlowFarneback
    return cv2_module


def uninstall(cv2_module=None):
    global _saved_cv2_fn, _DEFAULT_EXACT, _DEFAULT_LOW_LATENCY
    _DEFAULT_EXACT = None
    _DEFAULT_LOW_LATENCY = None
    if cv2_module is None:
        import cv2 as cv2_module
    if _saved_cv2_fn is not None:
        cv2_module.calcOpticalFlowFarneback = _saved_cv2_fn
        _saved_cv2_fn = None
    return cv2_module
