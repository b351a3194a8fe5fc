"""Synaptic accumulator: host-side mirror of /root/reference/eventsim/event_mem_sim.py.

Same names and argument meaning as the reference (``PARAMS``, ``DT``, ``update_state(w, V)``,
``resistance_exp(w)``, ``slice_indices(t, slice_us)``, ``load_events(h5_path)``, ``simulate(...)``);
the arithmetic runs in libnsof.so on the GPU.
"""
import ctypes as C
import gzip
import json
from pathlib import Path

import numpy as np

from . import _lib
from .context import default_context, dev_ptr
from .errors import NsofValueError

# event_mem_sim.py:20-34 (informational mirror; the device constants are compiled into libnsof.so)
PARAMS = dict(alphaoff=1, alphaon=1, voff=-0.2, von=0.1, koff=51.03, kon=-2.91, son=0.2, soff=0.8,
              bon=-5.12, boff=3.10, Ron=163_305, Roff=2_104_377, won=1, woff=0, wini=0.5)
DT = 5e-4
THETA_EVENTS = 1
REFRACTORY_US = 800


def _torch():
    import torch
    return torch


def update_state(w, V, *, ctx=None):  # noqa: N803
    """``update_state(w, V)`` of event_mem_sim.py:40-57 for float32 arrays (numpy in -> numpy out)."""
    ctx = ctx or default_context()
    w = np.ascontiguousarray(w, np.float32)
    V = np.ascontiguousarray(V, np.float32)  # noqa: N806
    if w.shape != V.shape:
        raise NsofValueError(f"w {w.shape} and V {V.shape} shapes differ", _lib.NSOF_ESHAPE)
    torch = _torch()
    dev = torch.device("cuda", ctx.device)
    dw, dv = torch.from_numpy(w).to(dev), torch.from_numpy(V).to(dev)
    out = torch.empty_like(dw)
    torch.cuda.synchronize(dev)
    ctx.check(ctx._lib.nsof_accum_update_state_dev(ctx.ptr, dev_ptr(dw), dev_ptr(dv), dev_ptr(out), w.size),
              "update_state")
    ctx.synchronize()
    return out.cpu().numpy()


def resistance_exp(w, *, ctx=None):
    """``resistance_exp(w)`` of event_mem_sim.py:60-63, returned as float32 (as the reference stores it, :292)."""
    ctx = ctx or default_context()
    w = np.ascontiguousarray(w, np.float32)
    torch = _torch()
    dev = torch.device("cuda", ctx.device)
    dw = torch.from_numpy(w).to(dev)
    out = torch.empty_like(dw)
    torch.cuda.synchronize(dev)
    ctx.check(ctx._lib.nsof_accum_resistance_dev(ctx.ptr, dev_ptr(dw), dev_ptr(out), w.size), "resistance_exp")
    ctx.synchronize()
    return out.cpu().numpy()


def slice_index_array(t, slice_us):
    """searchsorted(t, arange(t[0], t[-1]+slice_us, slice_us)) of event_mem_sim.py:78-83 as one int64 array."""
    t = np.ascontiguousarray(t, np.int64)
    lib = _lib.load()
    n = lib.nsof_accum_slice_bounds(t.ctypes.data, t.size, int(slice_us), None, 0)
    idx = np.empty(n, np.int64)
    lib.nsof_accum_slice_bounds(t.ctypes.data, t.size, int(slice_us), idx.ctypes.data, n)
    return idx


def slice_indices(t, slice_us):
    """Generator of ``slice(start, stop)`` objects, as the reference's ``slice_indices``."""
    idx = slice_index_array(t, slice_us)
    for i in range(len(idx) - 1):
        yield slice(int(idx[i]), int(idx[i + 1]))


def bincount_2d(x, y, H, W, *, ctx=None):  # noqa: N803
    """``bincount_2d`` of event_mem_sim.py:100-104: events per pixel, int32 [H][W] (atomic adds on the GPU)."""
    ctx = ctx or default_context()
    x, y = np.asarray(x), np.asarray(y)
    if x.shape != y.shape or x.ndim != 1:
        raise NsofValueError("bincount_2d: x and y must be 1-D arrays of one length")
    if x.size and (x.min() < -32768 or x.max() > 32767 or y.min() < -32768 or y.max() > 32767):
        raise NsofValueError("bincount_2d: coordinates outside the int16 range of /CD/events")
    x16, y16 = np.ascontiguousarray(x, np.int16), np.ascontiguousarray(y, np.int16)
    out = np.empty((int(H), int(W)), np.int32)
    ctx.check(ctx._lib.nsof_accum_bincount_2d(ctx.ptr, x16.ctypes.data, y16.ctypes.data, x16.size, int(H), int(W),
                                              out.ctypes.data), "bincount_2d")
    return out


def generate_synthetic_events(H=240, W=320, box_h=50, box_w=50, speed_pps=300, duration_s=1.5):  # noqa: N803
    """``generate_synthetic_events`` of event_mem_sim.py:109-158: a white box moving left to right; ON (+1) events
    on the pixels it newly covers, OFF (-1) on the ones it leaves, one frame every ``DT`` seconds.  Same event order
    as the reference (per step: ON pixels in row-major order, then OFF pixels).  Host generator (test input)."""
    t_step_us = int(DT * 1_000_000)
    duration_us = int(duration_s * 1_000_000)
    y0 = (H - box_h) // 2
    rows = np.arange(y0, min(y0 + box_h, H))
    rows = rows[rows >= 0]
    xs, ys, ps, ts = [], [], [], []
    prev = (0, 0)                                   # covered column range [a, b) of the previous frame
    for t_us in range(0, duration_us, t_step_us):
        start = int((t_us / 1_000_000) * speed_pps)
        end = start + box_w
        cur = (max(0, start), min(W, end)) if (start < W and end > 0) else (0, 0)
        cols_cur = np.arange(cur[0], cur[1])
        cols_prev = np.arange(prev[0], prev[1])
        for cols, pol in ((np.setdiff1d(cols_cur, cols_prev), 1), (np.setdiff1d(cols_prev, cols_cur), -1)):
            if cols.size and rows.size:
                yy, xx = np.meshgrid(rows, cols, indexing="ij")   # row-major, like np.where
                xs.append(xx.ravel()); ys.append(yy.ravel())
                ps.append(np.full(xx.size, pol)); ts.append(np.full(xx.size, t_us))
        prev = cur
    if not xs:
        e = np.array([], dtype=int)
        return e, e.copy(), e.copy(), e.copy()
    return (np.concatenate(xs).astype(int), np.concatenate(ys).astype(int), np.concatenate(ps).astype(int),
            np.concatenate(ts).astype(int))


def load_events(h5_path):
    """``load_events`` of event_mem_sim.py:69-75 (needs h5py; sensor size inferred from the data)."""
    import h5py
    with h5py.File(h5_path, "r") as f:
        evs = f["/CD/events"]
        x, y, p, t = evs["x"][:], evs["y"][:], evs["p"][:].astype(int), evs["t"][:]
    H, W = int(y.max()) + 1, int(x.max()) + 1  # noqa: N806
    return x, y, p, t, H, W


class Accumulator:
    """Device-resident array state (w, refractory maps) that can be advanced chunk by chunk."""

    def __init__(self, height, width, version=1, polarity="split", active_v=-8.0, silent_v=0.0, *, ctx=None,
                 dense=None, frames_path=None):
        if version not in (1, 2):
            raise NsofValueError("version must be 1 or 2")
        if polarity not in ("split", "magnitude"):
            raise NsofValueError("polarity must be 'split' or 'magnitude'")
        self.ctx = ctx or default_context()
        self.H, self.W, self.version = int(height), int(width), version
        self.split = version == 2 and polarity == "split"
        p = C.c_void_p()
        self.ctx.check(self.ctx._lib.nsof_accum_create(self.ctx.ptr, self.H, self.W, version, int(self.split),
                                                       float(active_v), float(silent_v), C.byref(p)), "accum_create")
        self._p = p
        if dense is not None:   # None: automatic; True: the every-pixel pass; False: the event-pixel update where it is exact
            self.ctx.check(self.ctx._lib.nsof_accum_set_dense(self._p, 1 if dense else -1), "accum_set_dense")
        if frames_path is not None:   # run_frames: "tiles" (default where it applies) or "copy_patch"
            self.ctx.check(self.ctx._lib.nsof_accum_set_frames_path(self._p, {"tiles": 0, "copy_patch": 1}[frames_path]),
                           "accum_set_frames_path")

    def reset(self):
        self.ctx.check(self.ctx._lib.nsof_accum_reset(self._p), "accum_reset")

    def step(self, x, y, p, t, bounds, snap_every=0):
        """Advance over slices ``[bounds[i], bounds[i+1])`` of the host event arrays."""
        x = np.ascontiguousarray(x, np.int16)
        y = np.ascontiguousarray(y, np.int16)
        p = np.ascontiguousarray(p, np.int8)
        t = np.ascontiguousarray(t, np.int64)
        bounds = np.ascontiguousarray(bounds, np.int64)
        if bounds.size < 2:
            return
        if bounds[-1] > x.size or not (x.size == y.size == p.size == t.size):
            raise NsofValueError("event arrays shorter than the slice bounds")
        self.ctx.check(self.ctx._lib.nsof_accum_step_events(self._p, x.ctypes.data, y.ctypes.data, p.ctypes.data,
                                                            t.ctypes.data, bounds.ctypes.data, bounds.size - 1,
                                                            int(snap_every)), "accum_step_events")

    def set_events(self, x, y, p, t, bounds):
        """Upload the events of all slices once (``nsof_accum_set_events``); ``run`` then advances over sub-ranges."""
        x = np.ascontiguousarray(x, np.int16)
        y = np.ascontiguousarray(y, np.int16)
        p = np.ascontiguousarray(p, np.int8)
        t = np.ascontiguousarray(t, np.int64)
        bounds = np.ascontiguousarray(bounds, np.int64)
        if bounds.size < 1 or bounds[-1] > x.size or not (x.size == y.size == p.size == t.size):
            raise NsofValueError("event arrays shorter than the slice bounds")
        self.n_staged = bounds.size - 1
        self.ctx.check(self.ctx._lib.nsof_accum_set_events(self._p, x.ctypes.data, y.ctypes.data, p.ctypes.data,
                                                           t.ctypes.data, bounds.ctypes.data, bounds.size - 1),
                       "accum_set_events")

    def set_slice_times(self, t_first, t_last):
        """Scheme 2 on a row band: the first / last event time of every slice of the WHOLE stream
        (``nsof_accum_set_slice_times``; ``nsof.dist.global_slice_times``) in place of those of the staged band events."""
        t_first = np.ascontiguousarray(t_first, np.int64)
        t_last = np.ascontiguousarray(t_last, np.int64)
        if t_first.shape != t_last.shape or t_first.ndim != 1:
            raise NsofValueError("t_first / t_last: one entry per staged slice")
        self.ctx.check(self.ctx._lib.nsof_accum_set_slice_times(self._p, t_first.ctypes.data, t_last.ctypes.data, t_first.size),
                       "accum_set_slice_times")

    def run(self, first_slice, n_slices, snap_every=0):
        self.ctx.check(self.ctx._lib.nsof_accum_run(self._p, int(first_slice), int(n_slices), int(snap_every)),
                       "accum_run")

    def run_surface(self, first_slice, n_slices, d_out, which=0, row_stride=None, mode="state"):
        """``run`` + ``surface_u8`` of the state after the last slice as one call (``nsof_accum_run_surface``): the dense
        scheme-1 update's last pass writes the frame itself -- same bytes as the two calls, one pass over the array less."""
        self.ctx.check(self.ctx._lib.nsof_accum_run_surface(self._p, int(first_slice), int(n_slices), which,
                                                            {"current": 0, "state": 1}[mode], dev_ptr(d_out),
                                                            self.W if row_stride is None else int(row_stride)),
                       "accum_run_surface")

    def run_frames(self, first_slice, n_frames, every, frames, which=0, mode="state"):
        """``n_frames`` intervals of ``every`` slices, the surface after each into ``frames[k]`` (uint8 CUDA tensor [n][H][W]):
        one call (``nsof_accum_run_frames``).  Scheme 1 with ``silent_v`` in the dead zone (and ``dense`` not forced): frame
        k is frame k-1 copied and patched at the event pixels -- byte-identical to ``run_surface`` per interval."""
        if int(frames.stride(2)) != 1:
            raise NsofValueError("frames: pixel stride must be 1")
        self.ctx.check(self.ctx._lib.nsof_accum_run_frames(self._p, int(first_slice), int(n_frames), int(every), which,
                                                           {"current": 0, "state": 1}[mode], dev_ptr(frames),
                                                           int(frames.stride(1)), int(frames.stride(0)) if n_frames > 1 else 0),
                       "accum_run_frames")

    def surface_u8(self, d_out, which=0, row_stride=None, mode="state"):
        """The current surface as an 8-bit frame written to DEVICE memory (torch uint8 tensor / address).
        ``mode``: "current" = the reference's current -> gray map (optical_flow_seg.py:426-431; saturates at 255 for
        w >= 0.42), "state" = uint8(255 * w)."""
        self.ctx.check(self.ctx._lib.nsof_accum_surface_u8_dev(self._p, which, {"current": 0, "state": 1}[mode],
                                                               dev_ptr(d_out),
                                                               self.W if row_stride is None else int(row_stride)),
                       "accum_surface_u8")

    def block_current(self, memsize, which=0, snapshot=-1, v_ds=1.0):
        """Block maximum of the device current ``v_ds / R`` over ``memsize x memsize`` pixel blocks, float64
        [H // memsize][W // memsize], reduced on the GPU (``nsof_accum_block_current``): the input of the gating image
        (``current_to_gray``) without downloading the surface.  ``snapshot`` >= 0 takes a stored snapshot (not
        consumed), -1 the current state."""
        out = np.empty((self.H // int(memsize), self.W // int(memsize)), np.float64)
        self.ctx.check(self.ctx._lib.nsof_accum_block_current(self._p, int(which), int(snapshot), int(memsize),
                                                              float(v_ds), out.ctypes.data), "accum_block_current")
        return out

    def block_current_dev(self, memsize, d_out, which=0, snapshot=-1, v_ds=1.0):
        """``block_current`` written to DEVICE memory (``d_out``: torch float64 tensor / address of
        (H // memsize) * (W // memsize) doubles), asynchronous: the input of ``gating.roi_from_surface_dev``."""
        self.ctx.check(self.ctx._lib.nsof_accum_block_current_dev(self._p, int(which), int(snapshot), int(memsize),
                                                                  float(v_ds), dev_ptr(d_out)), "accum_block_current_dev")

    def state(self, which=0):
        """-> dict(w float32 [H][W], next_ok int64 [H][W], slice_counter) -- everything a resume needs."""
        w = np.empty((self.H, self.W), np.float32)
        nok = np.empty((self.H, self.W), np.int64)
        cnt = C.c_int64()
        self.ctx.check(self.ctx._lib.nsof_accum_read_state(self._p, which, w.ctypes.data, nok.ctypes.data,
                                                           C.byref(cnt)), "accum_read_state")
        return dict(w=w, next_ok=nok, slice_counter=cnt.value)

    def load_state(self, state, which=0):
        """Inverse of ``state()`` (e.g. ``w_final`` of a stored run: ``load_state(dict(w=w_final, slice_counter=n))``)."""
        w = np.ascontiguousarray(state["w"], np.float32)
        if w.shape != (self.H, self.W):
            raise NsofValueError(f"state is {w.shape}, the array is {(self.H, self.W)}")
        nok = state.get("next_ok")
        nok = None if nok is None else np.ascontiguousarray(nok, np.int64)
        self.ctx.check(self.ctx._lib.nsof_accum_write_state(self._p, which, w.ctypes.data,
                                                            None if nok is None else nok.ctypes.data,
                                                            int(state.get("slice_counter", 0))), "accum_write_state")

    def w(self, which=0):
        out = np.empty((self.H, self.W), np.float32)
        self.ctx.check(self.ctx._lib.nsof_accum_read_w(self._p, which, out.ctypes.data), "accum_read_w")
        return out

    def resistance(self, which=0):
        out = np.empty((self.H, self.W), np.float32)
        self.ctx.check(self.ctx._lib.nsof_accum_read_resistance(self._p, which, out.ctypes.data), "accum_read_R")
        return out

    def snapshot_count(self):
        return int(self.ctx._lib.nsof_accum_snapshot_count(self._p))

    def snapshots(self):
        """-> list (one per array) of float32 [count][H][W]; clears the device ring."""
        n = self.ctx._lib.nsof_accum_snapshot_count(self._p)
        outs = []
        for which in range(2 if self.split else 1):
            out = np.empty((n, self.H, self.W), np.float32)
            self.ctx.check(self.ctx._lib.nsof_accum_read_snapshots(self._p, which, out.ctypes.data, n),
                           "accum_read_snapshots")
            outs.append(out)
        return outs

    def close(self):
        if getattr(self, "_p", None) is not None:
            self.ctx._lib.nsof_accum_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def simulate(events, version=1, slice_us=1_000, active_v=-8.0, silent_v=0.0, save_video=False, polarity="split",
             *, sensor_size=None, out_prefix=None, ctx=None, dense=None):
    """``simulate`` of event_mem_sim.py:164-286.

    ``events``: an HDF5 path with a ``/CD/events`` group (as the reference) or a tuple ``(x, y, p, t)`` of arrays.
    Returns ``dict(w_final, resistances[, w_final_b, resistances_b])`` with the reference's snapshot cadence
    (every ``max(1, nslices // 100)`` slices).  With ``out_prefix`` (or an HDF5 path) the same
    ``.V{version}.npz`` / ``.V2_b.npz`` / ``.json.gz`` files as the reference (:289-322) are written.
    ``save_video`` is accepted for signature compatibility; MP4 previews are not produced.
    """
    if version not in (1, 2):
        raise NsofValueError("version must be 1 or 2")
    if polarity not in ("split", "magnitude"):
        raise NsofValueError("polarity must be 'split' or 'magnitude'")
    h5_path = None
    if isinstance(events, (str, Path)):
        h5_path = Path(events)
        x, y, p, t, H, W = load_events(h5_path)  # noqa: N806
    else:
        x, y, p, t = events
        x, y, t = np.asarray(x), np.asarray(y), np.asarray(t)
        if x.size == 0:
            raise NsofValueError("empty event stream")
        if sensor_size is None:
            H, W = int(y.max()) + 1, int(x.max()) + 1  # noqa: N806  (event_mem_sim.py:74)
        else:
            H, W = sensor_size  # noqa: N806
    idx = slice_index_array(t, slice_us)
    nslices = max(len(idx) - 1, 0)
    every = max(1, nslices // 100)
    acc = Accumulator(H, W, version, polarity, active_v, silent_v, ctx=ctx, dense=dense)
    try:
        acc.step(x, y, p, t, idx, snap_every=every)
        snaps = acc.snapshots()
        out = dict(w_final=acc.w(0), resistances=snaps[0])
        if acc.split:
            out.update(w_final_b=acc.w(1), resistances_b=snaps[1])
    finally:
        acc.close()
    prefix = Path(out_prefix) if out_prefix is not None else h5_path
    if prefix is not None:
        _save_outputs(prefix, out, version, slice_us, polarity, h5_path)
    return out


def simulate_frames(compressed_images, dt=0.0005, n_sub_steps=1000, th1=0.7, th2=1.5, *, ctx=None):
    """Frame-driven accumulator of /root/reference/simulation/simulationcode_v4_transistor_uav.m
    (``simulate_memristor_array``, :187-227): ``compressed_images`` float64 [n][H][W] in [0,1] (the output of the
    script's Lanczos ``compress_image``).  Returns ``(w_array, resistances_over_time)`` with the initial snapshot
    first, as the script stores them.  uav: th1=0.7, th2=1.5; vehicle: th1=2."""
    ctx = ctx or default_context()
    imgs = np.ascontiguousarray(compressed_images, np.float64)
    if imgs.ndim != 3:
        raise NsofValueError("compressed_images must be [n][H][W]")
    n, H, W = imgs.shape  # noqa: N806
    w = np.empty((H, W), np.float64)
    res = np.empty((n, H, W), np.float64)
    ctx.check(ctx._lib.nsof_accum_frames_f64(ctx.ptr, imgs.ctypes.data, n, H, W, float(dt), int(n_sub_steps),
                                             float(th1), float(th2), w.ctypes.data, res.ctypes.data),
              "simulate_frames")
    return w, res


def _save_outputs(prefix, out, version, slice_us, polarity, h5_path):
    """File set of event_mem_sim.py:289-322 (npz keys ``w_final`` / ``resistances``; json.gz metadata)."""
    np.savez_compressed(prefix.with_suffix(f".V{version}.npz"), w_final=out["w_final"],
                        resistances=out["resistances"].astype(np.float32))
    if version == 2:
        if "w_final_b" in out:
            np.savez_compressed(prefix.with_suffix(".V2_b.npz"), w_final=out["w_final_b"],
                                resistances=out["resistances_b"].astype(np.float32))
        else:
            np.savez_compressed(prefix.with_suffix(".V2_b.npz"), w_final=np.array([]), resistances=np.array([]))
    meta = dict(version=version, slice_us=slice_us, fps=1_000_000 / slice_us, params=PARAMS, dt=DT,
                scheme="boxcar" if version == 1 else "dc_bias_overlay", polarity=polarity if version == 2 else None,
                theta_events=THETA_EVENTS if version == 1 else None,
                refractory_us=REFRACTORY_US if version == 2 else None, event_file=str(h5_path or prefix))
    with gzip.open(prefix.with_suffix(f".V{version}.json.gz"), "wt") as fp:
        json.dump(meta, fp, indent=2)
