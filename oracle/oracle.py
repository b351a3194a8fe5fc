"""ctypes wrapper around the CPU oracle (oracle/_build/libnsof_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.  See the headers of
oracle/farneback_ref.c and oracle/accum_ref.c for what each function restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libnsof_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.nsof_ref_farneback_u8.restype = C.c_int
        _lib.nsof_ref_farneback_u8.argtypes = [
            C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_void_p, C.c_ssize_t,
            C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
        _lib.nsof_ref_pyr_level.restype = C.c_int
        _lib.nsof_ref_pyr_level.argtypes = [C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
        _lib.nsof_ref_level_geometry.restype = None
        _lib.nsof_ref_level_geometry.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int] + [C.c_void_p] * 4
        _lib.nsof_ref_effective_levels.restype = C.c_int
        _lib.nsof_ref_effective_levels.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int]
        _lib.nsof_ref_polyexp.restype = C.c_int
        _lib.nsof_ref_polyexp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
        _lib.nsof_ref_update_matrices.restype = C.c_int
        _lib.nsof_ref_update_matrices.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4
        _lib.nsof_ref_update_flow_blur.restype = C.c_int
        _lib.nsof_ref_update_flow_blur.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4
        _lib.nsof_ref_resize_linear.restype = C.c_int
        _lib.nsof_ref_resize_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        _lib.nsof_ref_gaussian_kernel.restype = C.c_int
        _lib.nsof_ref_gaussian_kernel.argtypes = [C.c_int, C.c_double, C.c_void_p]
        _lib.nsof_ref_poly_prepare.restype = C.c_int
        _lib.nsof_ref_poly_prepare.argtypes = [C.c_int, C.c_double] + [C.c_void_p] * 4
        if hasattr(_lib, "nsof_ref_accum_update_state"):
            _bind_accum(_lib)
    return _lib


def _bind_accum(l):
    l.nsof_ref_accum_update_state.restype = None
    l.nsof_ref_accum_update_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    l.nsof_ref_accum_resistance.restype = None
    l.nsof_ref_accum_resistance.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    l.nsof_ref_accum_slice_bounds.restype = C.c_int64
    l.nsof_ref_accum_slice_bounds.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64]
    l.nsof_ref_accum_simulate.restype = C.c_int
    l.nsof_ref_accum_simulate.argtypes = [
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
        C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]


def accum_frames(imgs, dt=5e-4, n_sub=1000, th1=0.7, th2=1.5):
    """Frame-driven variant (simulation/simulationcode_v4_transistor_uav.m). imgs: float64 [n][H][W] in [0,1]."""
    imgs = np.ascontiguousarray(imgs, np.float64)
    n, H, W = imgs.shape
    w = np.empty((H, W), np.float64)
    res = np.empty((n, H, W), np.float64)
    l = lib()
    l.nsof_ref_accum_frames.restype = C.c_int
    l.nsof_ref_accum_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double,
                                        C.c_double, C.c_void_p, C.c_void_p]
    _chk(l.nsof_ref_accum_frames(imgs.ctypes.data, n, H, W, dt, n_sub, th1, th2, w.ctypes.data, res.ctypes.data),
         "accum_frames")
    return w, res


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed: rc={rc}")


def _u8(a):
    a = np.asarray(a)
    assert a.dtype == np.uint8 and a.ndim == 2
    if a.strides[1] != 1:
        a = np.ascontiguousarray(a)
    return a


# ---------------------------------------------------------------- CPU-baseline build (bench.py cpu_baseline leg)
def host_cpu():
    """(model name, logical cores, physical cores) of this machine, from /proc/cpuinfo."""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name":
                    model = v
                elif k == "physical id":
                    pid = v
                elif k == "core id":
                    cid = v
                elif not k and pid is not None:
                    phys.add((pid, cid))
                    pid = cid = None
            if pid is not None:
                phys.add((pid, cid))
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    # a container's CPU share (cgroup quota) is what "all cores" can really use: more threads than that only thrash
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda s: s.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda s: [s.strip(), None])):
        try:
            with open(path) as f:
                q, per = parse(f.read())
            if per is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    per = f.read().strip()
            if q not in ("max", "-1") and int(per) > 0:
                usable = max(1, min(usable, -(-int(q) // int(per))))
            break
        except (OSError, ValueError):
            continue
    if os.environ.get("NSOF_CPU_THREADS"):
        usable = max(1, int(os.environ["NSOF_CPU_THREADS"]))
    return model, logical, (len(phys) or logical), usable


_perf = None


def perf_lib():
    """The -O3 -march=native -fopenmp build of the same sources, compiled ON the machine that times it (the file
    name carries a tag of the host CPU so that a copy built elsewhere is never picked up).  Same arithmetic as the
    checker build: -ffp-contract=off, no reassociation."""
    global _perf
    if _perf is None:
        import hashlib
        tag = hashlib.sha1(host_cpu()[0].encode()).hexdigest()[:10]
        so = os.path.join(_HERE, "_build", f"libnsof_oracle_perf_{tag}.so")
        srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "perf", f"PERF_OUT=_build/libnsof_oracle_perf_{tag}.so"])
        l = C.CDLL(so)
        l.nsof_ref_farneback_u8_many.restype = C.c_int
        l.nsof_ref_farneback_u8_many.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_ssize_t, C.c_int,
                                                 C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_double, C.c_int, C.c_int]
        l.nsof_ref_has_openmp.restype = C.c_int
        _perf = l
        l.nsof_ref_set_pyr_fma.argtypes = [C.c_int]
        l.nsof_ref_set_pyr_fma.restype = None
        l.nsof_ref_set_pyr_fma(1 if _pyr_fma else 0)
    return _perf


def accum_slices_per_s(x, y, t, H, W, slice_us=1000, active_v=-6.0, silent_v=0.0, n_slices=20, n_threads=1):
    """CPU rate of the reference's scheme-1 slice (V fill + event scatter + dense update_state over H x W) through the
    perf build: the first ``n_slices`` slices of the stream, ``n_threads`` OpenMP threads.  Returns (slices/s, w)."""
    import time
    l = perf_lib()
    l.nsof_ref_set_threads.argtypes = [C.c_int]
    l.nsof_ref_accum_slice_v1.restype = None
    l.nsof_ref_accum_slice_v1.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                          C.c_float, C.c_float]
    x = np.ascontiguousarray(x, np.int16)
    y = np.ascontiguousarray(y, np.int16)
    idx = accum_slice_bounds(np.ascontiguousarray(t, np.int64), slice_us)
    n_slices = min(n_slices, len(idx) - 1)
    w = np.full((H, W), 0.5, np.float32)
    V = np.empty((H, W), np.float32)
    l.nsof_ref_set_threads(int(n_threads))
    t0 = time.perf_counter()
    for s in range(n_slices):
        lo, hi = int(idx[s]), int(idx[s + 1])
        l.nsof_ref_accum_slice_v1(w.ctypes.data, V.ctypes.data, w.size, x[lo:].ctypes.data, y[lo:].ctypes.data, hi - lo, W,
                                  active_v, silent_v)
    dt = time.perf_counter() - t0
    l.nsof_ref_set_threads(1)
    return n_slices / dt, w


def set_pyr_fma(on):
    """Arithmetic variant of the pyramid stages for subsequent calls (see farneback_ref.c): False = every product and sum
    rounded (default), True = one fused multiply-add per tap / blend.  Applies to the checker and the perf build."""
    global _pyr_fma
    _pyr_fma = bool(on)
    for l in (lib(), _perf):
        if l is not None:
            l.nsof_ref_set_pyr_fma.argtypes = [C.c_int]
            l.nsof_ref_set_pyr_fma.restype = None
            l.nsof_ref_set_pyr_fma(1 if on else 0)


_pyr_fma = False


def farneback_many(prevs, nexts, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags=0, n_threads=1):
    """[n][H][W] uint8 x2 -> [n][H][W][2] float32 through the perf build, ``n_threads`` pairs at a time."""
    prevs, nexts = np.ascontiguousarray(prevs, np.uint8), np.ascontiguousarray(nexts, np.uint8)
    n, h, w = prevs.shape
    flow = np.empty((n, h, w, 2), np.float32)
    rc = perf_lib().nsof_ref_farneback_u8_many(n, prevs.ctypes.data, nexts.ctypes.data, w, w * h, w, h, flow.ctypes.data,
                                               pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags,
                                               int(n_threads))
    _chk(rc, "farneback_many")
    return flow


# ---------------------------------------------------------------- Farneback
def farneback(prev, nxt, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags=0):
    prev, nxt = _u8(prev), _u8(nxt)
    h, w = prev.shape
    flow = np.empty((h, w, 2), np.float32)
    rc = lib().nsof_ref_farneback_u8(prev.ctypes.data, prev.strides[0], nxt.ctypes.data, nxt.strides[0], w, h,
                                     flow.ctypes.data, flow.strides[0], pyr_scale, levels, winsize, iterations,
                                     poly_n, poly_sigma, flags)
    _chk(rc, "farneback")
    return flow


def effective_levels(w, h, pyr_scale, levels):
    return lib().nsof_ref_effective_levels(w, h, pyr_scale, levels)


def level_geometry(w, h, pyr_scale, k):
    wk, hk, ks = C.c_int(), C.c_int(), C.c_int()
    sg = C.c_double()
    lib().nsof_ref_level_geometry(w, h, pyr_scale, k, C.byref(wk), C.byref(hk), C.byref(ks), C.byref(sg))
    return wk.value, hk.value, ks.value, sg.value


def pyr_level(img, pyr_scale, k):
    img = _u8(img)
    h, w = img.shape
    wk, hk, _, _ = level_geometry(w, h, pyr_scale, k)
    out = np.empty((hk, wk), np.float32)
    _chk(lib().nsof_ref_pyr_level(img.ctypes.data, img.strides[0], w, h, pyr_scale, k, out.ctypes.data), "pyr_level")
    return out


def polyexp(img_f32, n, sigma):
    a = np.ascontiguousarray(img_f32, np.float32)
    h, w = a.shape
    out = np.empty((h, w, 5), np.float32)
    _chk(lib().nsof_ref_polyexp(a.ctypes.data, w, h, n, sigma, out.ctypes.data), "polyexp")
    return out


def update_matrices(R0, R1, flow, M=None, y0=0, y1=None):
    R0 = np.ascontiguousarray(R0, np.float32)
    R1 = np.ascontiguousarray(R1, np.float32)
    flow = np.ascontiguousarray(flow, np.float32)
    h, w = flow.shape[:2]
    if M is None:
        M = np.zeros((h, w, 5), np.float32)
    _chk(lib().nsof_ref_update_matrices(R0.ctypes.data, R1.ctypes.data, flow.ctypes.data, M.ctypes.data, w, h, y0,
                                        h if y1 is None else y1), "update_matrices")
    return M


def update_flow_blur(R0, R1, flow, M, winsize, update):
    """Returns (new_flow, new_M); inputs are not modified."""
    R0 = np.ascontiguousarray(R0, np.float32)
    R1 = np.ascontiguousarray(R1, np.float32)
    flow = np.array(flow, np.float32, order="C")
    M = np.array(M, np.float32, order="C")
    h, w = flow.shape[:2]
    _chk(lib().nsof_ref_update_flow_blur(R0.ctypes.data, R1.ctypes.data, flow.ctypes.data, M.ctypes.data, w, h,
                                         winsize, int(bool(update))), "update_flow_blur")
    return flow, M


def resize_linear(src, dw, dh):
    a = np.ascontiguousarray(src, np.float32)
    sh, sw = a.shape[:2]
    cn = 1 if a.ndim == 2 else a.shape[2]
    out = np.empty((dh, dw) if a.ndim == 2 else (dh, dw, cn), np.float32)
    _chk(lib().nsof_ref_resize_linear(a.ctypes.data, sw, sh, cn, out.ctypes.data, dw, dh), "resize")
    return out


def gaussian_kernel(n, sigma):
    out = np.empty(n, np.float32)
    _chk(lib().nsof_ref_gaussian_kernel(n, sigma, out.ctypes.data), "gaussian_kernel")
    return out


def poly_prepare(n, sigma):
    g = np.empty(2 * n + 1, np.float32)
    xg = np.empty_like(g)
    xxg = np.empty_like(g)
    ig = np.empty(4, np.float64)
    _chk(lib().nsof_ref_poly_prepare(n, sigma, g.ctypes.data, xg.ctypes.data, xxg.ctypes.data, ig.ctypes.data),
         "poly_prepare")
    return g, xg, xxg, ig


# ---------------------------------------------------------------- accumulator
def accum_update_state(w, V):
    w = np.ascontiguousarray(w, np.float32)
    V = np.ascontiguousarray(V, np.float32)
    out = np.empty_like(w)
    lib().nsof_ref_accum_update_state(w.ctypes.data, V.ctypes.data, out.ctypes.data, w.size)
    return out


def accum_resistance(w):
    w = np.ascontiguousarray(w, np.float32)
    out = np.empty_like(w)
    lib().nsof_ref_accum_resistance(w.ctypes.data, out.ctypes.data, w.size)
    return out


def accum_slice_bounds(t, slice_us):
    t = np.ascontiguousarray(t, np.int64)
    n = lib().nsof_ref_accum_slice_bounds(t.ctypes.data, t.size, slice_us, None, 0)
    idx = np.empty(n, np.int64)
    lib().nsof_ref_accum_slice_bounds(t.ctypes.data, t.size, slice_us, idx.ctypes.data, n)
    return idx


def accum_simulate(x, y, p, t, H, W, version, polarity, slice_us, active_v, silent_v):
    """polarity: 'split' | 'magnitude'.  Returns dict(w_final, resistances[, w_final_b, resistances_b])."""
    x = np.ascontiguousarray(x, np.int16)
    y = np.ascontiguousarray(y, np.int16)
    p = np.ascontiguousarray(p, np.int8)
    t = np.ascontiguousarray(t, np.int64)
    idx = accum_slice_bounds(t, slice_us)
    nslices = max(len(idx) - 1, 0)
    every = max(1, nslices // 100)
    nsnap = (nslices + every - 1) // every
    split = 1 if (version == 2 and polarity == "split") else 0
    wa = np.empty((H, W), np.float32)
    ra = np.empty((nsnap, H, W), np.float32)
    wb = np.empty((H, W), np.float32) if split else None
    rb = np.empty((nsnap, H, W), np.float32) if split else None
    rc = lib().nsof_ref_accum_simulate(
        x.ctypes.data, y.ctypes.data, p.ctypes.data, t.ctypes.data, t.size, H, W, version, split, slice_us,
        active_v, silent_v, wa.ctypes.data, ra.ctypes.data, wb.ctypes.data if split else None,
        rb.ctypes.data if split else None, nsnap)
    _chk(rc, "accum_simulate")
    out = dict(w_final=wa, resistances=ra)
    if split:
        out.update(w_final_b=wb, resistances_b=rb)
    return out


# ---------------------------------------------------------------- segmentation head
def structuring_element(shape, kw, kh):
    """shape 0 rect / 1 cross / 2 ellipse, as cv2.getStructuringElement(shape, (kw, kh))."""
    out = np.empty((kh, kw), np.uint8)
    l = lib()
    l.nsof_ref_structuring_element.restype = C.c_int
    l.nsof_ref_structuring_element.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    _chk(l.nsof_ref_structuring_element(shape, kw, kh, out.ctypes.data), "structuring_element")
    return out


def morph(op, src, elem, anchor=(-1, -1)):
    """op 0 erode / 1 dilate on a uint8 image (cv2 semantics: un-reflected element, outside ignored)."""
    src = np.ascontiguousarray(src, np.uint8)
    elem = np.ascontiguousarray(elem, np.uint8)
    h, w = src.shape
    out = np.empty_like(src)
    l = lib()
    l.nsof_ref_morph_u8.restype = C.c_int
    l.nsof_ref_morph_u8.argtypes = [C.c_int, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p, C.c_ssize_t]
    _chk(l.nsof_ref_morph_u8(op, src.ctypes.data, w, w, h, elem.ctypes.data, elem.shape[1], elem.shape[0],
                             anchor[0], anchor[1], out.ctypes.data, w), "morph")
    return out


def motion_mask(flow, thresh=1.0, ksize=10, iters=5):
    flow = np.ascontiguousarray(flow, np.float32)
    h, w = flow.shape[:2]
    out = np.empty((h, w), np.uint8)
    l = lib()
    l.nsof_ref_motion_mask.restype = C.c_int
    l.nsof_ref_motion_mask.argtypes = [C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                       C.c_void_p, C.c_ssize_t]
    _chk(l.nsof_ref_motion_mask(flow.ctypes.data, 2 * w, w, h, thresh, ksize, iters, out.ctypes.data, w),
         "motion_mask")
    return out


# ---------------------------------------------------------------- prediction warp + SSIM
def remap_linear(src, mapx, mapy, border=0, cval=0):
    """cv2.remap(src, mapx, mapy, INTER_LINEAR, borderMode=CONSTANT(0)|REPLICATE(1)) for uint8 [H][W] or [H][W][C]."""
    src = np.ascontiguousarray(src, np.uint8)
    mapx = np.ascontiguousarray(mapx, np.float32)
    mapy = np.ascontiguousarray(mapy, np.float32)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dh, dw = mapx.shape
    out = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.uint8)
    l = lib()
    l.nsof_ref_remap_linear_u8.restype = C.c_int
    l.nsof_ref_remap_linear_u8.argtypes = [C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_ssize_t]
    _chk(l.nsof_ref_remap_linear_u8(src.ctypes.data, sw * cn, sw, sh, cn, mapx.ctypes.data, mapy.ctypes.data, dw, dw,
                                    dh, border, cval, out.ctypes.data, dw * cn), "remap")
    return out


def flow_map(flow, rect, sign=-1):
    """(mapx, mapy) float32 for the crop rect=(x0, y0, x1, y1) of a float32 flow canvas, as prediction.py builds it."""
    flow = np.ascontiguousarray(flow, np.float32)
    x0, y0, x1, y1 = rect
    mx = np.empty((y1 - y0, x1 - x0), np.float32)
    my = np.empty_like(mx)
    l = lib()
    l.nsof_ref_flow_map.restype = C.c_int
    l.nsof_ref_flow_map.argtypes = [C.c_void_p, C.c_ssize_t] + [C.c_int] * 5 + [C.c_void_p] * 2
    _chk(l.nsof_ref_flow_map(flow.ctypes.data, 2 * flow.shape[1], x0, y0, x1, y1, sign, mx.ctypes.data,
                             my.ctypes.data), "flow_map")
    return mx, my


def ssim_u8(a, b, data_range=255.0):
    """structural_similarity(a, b, data_range=...) for 2-D uint8 views (any strides)."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.dtype == np.uint8 and b.dtype == np.uint8 and a.shape == b.shape and a.ndim == 2
    h, w = a.shape
    out = C.c_double()
    l = lib()
    l.nsof_ref_ssim_u8.restype = C.c_int
    l.nsof_ref_ssim_u8.argtypes = [C.c_void_p, C.c_ssize_t, C.c_int, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int,
                                   C.c_int, C.c_double, C.POINTER(C.c_double)]
    _chk(l.nsof_ref_ssim_u8(a.ctypes.data, a.strides[0], a.strides[1], b.ctypes.data, b.strides[0], b.strides[1], w, h,
                            data_range, C.byref(out)), "ssim")
    return out.value


def imresize_lanczos3(img, oh, ow):
    """imresize(double image, [oh ow], 'lanczos3') -- see the header of the function in accum_ref.c."""
    img = np.ascontiguousarray(img, np.float64)
    h, w = img.shape
    out = np.empty((oh, ow), np.float64)
    l = lib()
    l.nsof_ref_imresize_lanczos3.restype = C.c_int
    l.nsof_ref_imresize_lanczos3.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    _chk(l.nsof_ref_imresize_lanczos3(img.ctypes.data, h, w, oh, ow, out.ctypes.data), "imresize")
    return out
