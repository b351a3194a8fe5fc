"""GPU parity: HIP accumulator (through the C ABI) vs the reference-generated golden vectors and the oracle.

Tolerances: the state update uses float32 pow; the reference's NumPy (SVML pow), glibc powf (oracle) and the
device (double-precision exp/log, rounded once) differ by <= 1 float ulp per step, so w is compared at
5e-7 absolute (about 8 ulp at w ~ 0.5..1) after 50-400 slices and resistances at 2e-6 relative."""
import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu

W_ATOL = 5e-7
R_RTOL = 2e-6
CASES = ["v1", "v2_split", "v2_magnitude", "v1_leak", "v2_split_bias", "v1_exact", "v2_split_pm1"]


def test_update_state_golden(nsof_lib, ctx):
    g = np.load(golden_path("accum_update_state.npz"))
    out = nsof_lib.update_state(g["w_grid"], g["V_grid"], ctx=ctx)
    assert out.dtype == np.float32 and out.shape == g["w_grid"].shape
    assert np.abs(out - g["out_grid"]).max() <= 1.2e-7
    out = nsof_lib.update_state(g["w_rand"], g["V_rand"], ctx=ctx)
    assert np.abs(out - g["out_rand"]).max() <= 1.2e-7
    # known answers recorded in SURVEY.md section 8c (w0 = 0.5)
    V = np.array([-8, -6, -1, -0.21, -0.2, 0, 0.1, 0.11, 1, 3], np.float32)
    want = np.array([0.7042345, 0.6518667, 0.5209471, 0.50026184, 0.5, 0.5, 0.5, 0.49975047, 0.47754133, 0.4276332],
                    np.float32)
    got = nsof_lib.update_state(np.full(V.shape, 0.5, np.float32), V, ctx=ctx)
    assert np.abs(got - want).max() <= 6e-8


def test_resistance_golden(nsof_lib, ctx):
    g = np.load(golden_path("accum_update_state.npz"))
    r = nsof_lib.resistance_exp(g["w_rand"], ctx=ctx)
    assert (np.abs(r - g["res_rand"]) / g["res_rand"]).max() <= R_RTOL
    r = nsof_lib.resistance_exp(np.array([0.5, 0.6518667, 0.50026184], np.float32), ctx=ctx)
    assert np.allclose(r, [586221.1990, 397622.2887, 585828.9265], rtol=R_RTOL)


@pytest.mark.parametrize("name", CASES)
def test_simulate_vs_reference_golden(nsof_lib, ctx, oracle, name):
    d = np.load(golden_path(f"accum_sim_{name}.npz"))
    H, W = d["w_final"].shape
    out = nsof_lib.simulate((d["x"], d["y"], d["p"], d["t"]), version=int(d["version"]),
                            slice_us=int(d["slice_us"]), active_v=float(d["active_v"]),
                            silent_v=float(d["silent_v"]), polarity=str(d["polarity"]), sensor_size=(H, W), ctx=ctx)
    assert out["resistances"].shape[0] == int(d["n_snapshots"])
    assert np.abs(out["w_final"] - d["w_final"]).max() <= W_ATOL
    idx = d["snap_idx"]
    rr = out["resistances"][idx]
    assert (np.abs(rr - d["resistances"]) / d["resistances"]).max() <= R_RTOL
    if "w_final_b" in d:
        assert np.abs(out["w_final_b"] - d["w_final_b"]).max() <= W_ATOL
        assert (np.abs(out["resistances_b"][idx] - d["resistances_b"]) / d["resistances_b"]).max() <= R_RTOL
    else:
        assert "w_final_b" not in out
    # and against the C oracle on every snapshot
    ref = oracle.accum_simulate(d["x"], d["y"], d["p"], d["t"], H, W, int(d["version"]), str(d["polarity"]),
                                int(d["slice_us"]), float(d["active_v"]), float(d["silent_v"]))
    assert np.abs(out["w_final"] - ref["w_final"]).max() <= W_ATOL
    assert (np.abs(out["resistances"] - ref["resistances"]) / ref["resistances"]).max() <= R_RTOL


@pytest.mark.parametrize("name", ["v1", "v2_split", "v2_magnitude"])
def test_dense_path_equals_sparse_path(nsof_lib, ctx, name):
    """With silent_v in the dead zone the event-pixel (sparse) path must equal the every-pixel path bit for bit."""
    d = np.load(golden_path(f"accum_sim_{name}.npz"))
    H, W = d["w_final"].shape
    kw = dict(version=int(d["version"]), slice_us=int(d["slice_us"]), active_v=float(d["active_v"]),
              silent_v=float(d["silent_v"]), polarity=str(d["polarity"]), sensor_size=(H, W), ctx=ctx)
    ev = (d["x"], d["y"], d["p"], d["t"])
    a = nsof_lib.simulate(ev, dense=False, **kw)      # the event-pixel update
    b = nsof_lib.simulate(ev, dense=True, **kw)       # the every-pixel pass
    c = nsof_lib.simulate(ev, **kw)                   # automatic choice
    for k in a:
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k]), k


def test_chunked_stepping_equals_one_shot(nsof_lib, ctx):
    d = np.load(golden_path("accum_sim_v2_split.npz"))
    H, W = d["w_final"].shape
    from nsof.accumulator import Accumulator, slice_index_array
    idx = slice_index_array(d["t"], 1000)
    every = max(1, (len(idx) - 1) // 100)
    acc = Accumulator(H, W, 2, "split", -6.0, 0.0, ctx=ctx)
    for lo in range(0, len(idx) - 1, 37):
        acc.step(d["x"], d["y"], d["p"], d["t"], idx[lo:lo + 38], snap_every=every)
    wa, wb = acc.w(0), acc.w(1)
    snaps = acc.snapshots()
    acc.close()
    one = nsof_lib.simulate((d["x"], d["y"], d["p"], d["t"]), version=2, slice_us=1000, active_v=-6.0, silent_v=0.0,
                            polarity="split", sensor_size=(H, W), ctx=ctx)
    assert np.array_equal(wa, one["w_final"]) and np.array_equal(wb, one["w_final_b"])
    assert np.array_equal(snaps[0], one["resistances"]) and np.array_equal(snaps[1], one["resistances_b"])


def test_large_sensor_many_slices_vs_oracle(nsof_lib, ctx, oracle):
    """1280x720 synthetic stream, 300 slices, scheme 1 and 2."""
    from nsof import synth
    x, y, p, t = synth.make_events(2024, 1280, 720, 60_000, 300_000)
    for version, pol in [(1, "split"), (2, "split"), (2, "magnitude")]:
        out = nsof_lib.simulate((x, y, p, t), version=version, slice_us=1000, active_v=-6.0, silent_v=0.0,
                                polarity=pol, sensor_size=(720, 1280), ctx=ctx)
        ref = oracle.accum_simulate(x, y, p, t, 720, 1280, version, pol, 1000, -6.0, 0.0)
        assert np.abs(out["w_final"] - ref["w_final"]).max() <= W_ATOL
        assert out["resistances"].shape == ref["resistances"].shape
        assert (np.abs(out["resistances"][::17] - ref["resistances"][::17]) / ref["resistances"][::17]).max() <= R_RTOL
        if "w_final_b" in ref:
            assert np.abs(out["w_final_b"] - ref["w_final_b"]).max() <= W_ATOL


def test_event_outside_sensor_is_rejected(nsof_lib, ctx):
    x = np.array([5, 700], np.int16)
    y = np.array([5, 5], np.int16)
    p = np.array([1, 1], np.int8)
    t = np.array([0, 10], np.int64)
    with pytest.raises(nsof_lib.error):
        nsof_lib.simulate((x, y, p, t), version=1, sensor_size=(64, 64), ctx=ctx)


def test_frame_driven_variant_vs_oracle(nsof_lib, ctx, oracle):
    """float64, 1000 Euler sub-steps per frame pair (simulation/simulationcode_v4_transistor_uav.m): the device's
    double pow differs from libm's in the last bits, hence 1e-9 on w after 4000 sub-steps."""
    rng = np.random.default_rng(3)
    for shape in [(4, 4), (9, 31)]:
        imgs = rng.random((5,) + shape)
        imgs[3] = imgs[2]
        w, res = nsof_lib.simulate_frames(imgs, ctx=ctx)
        wr, rr = oracle.accum_frames(imgs)
        assert w.shape == shape and res.shape == (5,) + shape
        assert np.abs(w - wr).max() < 1e-9
        assert (np.abs(res - rr) / rr).max() < 1e-9
    w2, _ = nsof_lib.simulate_frames(imgs, th1=2.0, n_sub_steps=10, ctx=ctx)   # vehicle variant, fast simulation
    assert np.abs(w2 - oracle.accum_frames(imgs, n_sub=10, th1=2.0)[0]).max() < 1e-12


def test_output_files_match_reference_format(nsof_lib, ctx, tmp_path):
    """simulate(out_prefix=...) writes the reference's file set (event_mem_sim.py:289-322): .V{v}.npz with
    w_final/resistances (float32), .V2_b.npz (empty arrays in magnitude mode), .V{v}.json.gz metadata."""
    import gzip
    import json
    d = np.load(golden_path("accum_sim_v2_split.npz"))
    H, W = d["w_final"].shape
    ev = (d["x"], d["y"], d["p"], d["t"])
    prefix = tmp_path / "stream.hdf5"
    out = nsof_lib.simulate(ev, version=2, slice_us=1000, active_v=-6.0, silent_v=0.0, polarity="split",
                            sensor_size=(H, W), out_prefix=prefix, ctx=ctx)
    a = np.load(tmp_path / "stream.V2.npz")
    b = np.load(tmp_path / "stream.V2_b.npz")
    assert sorted(a.files) == ["resistances", "w_final"] and a["resistances"].dtype == np.float32
    assert np.array_equal(a["w_final"], out["w_final"]) and np.array_equal(b["w_final"], out["w_final_b"])
    with gzip.open(tmp_path / "stream.V2.json.gz", "rt") as f:
        meta = json.load(f)
    assert set(meta) == {"version", "slice_us", "fps", "params", "dt", "scheme", "polarity", "theta_events",
                         "refractory_us", "event_file"}
    assert meta["scheme"] == "dc_bias_overlay" and meta["refractory_us"] == 800 and meta["theta_events"] is None
    assert meta["params"]["koff"] == 51.03 and meta["dt"] == 5e-4 and meta["fps"] == 1000.0
    nsof_lib.simulate(ev, version=2, polarity="magnitude", active_v=-6.0, sensor_size=(H, W),
                      out_prefix=tmp_path / "m.hdf5", ctx=ctx)
    assert np.load(tmp_path / "m.V2_b.npz")["w_final"].size == 0
    nsof_lib.simulate(ev, version=1, active_v=-6.0, sensor_size=(H, W), out_prefix=tmp_path / "one.hdf5", ctx=ctx)
    with gzip.open(tmp_path / "one.V1.json.gz", "rt") as f:
        m1 = json.load(f)
    assert m1["scheme"] == "boxcar" and m1["polarity"] is None and m1["theta_events"] == 1
    assert not (tmp_path / "one.V2_b.npz").exists()


@pytest.mark.gpu
def test_bincount_2d_matches_reference_and_numpy(nsof_lib):
    """bincount_2d against the histograms the reference produced and against np.bincount."""
    import numpy as np
    from conftest import golden_path
    g = np.load(golden_path("synth_events.npz"))
    x, y = g["default_x"], g["default_y"]
    got = nsof_lib.bincount_2d(x, y, 240, 320)
    assert got.dtype == np.int32 and np.array_equal(got, g["hist_all"])
    assert np.array_equal(nsof_lib.bincount_2d(x[:2000], y[:2000], 145, 320), g["hist_first_2000"])
    rng = np.random.default_rng(1)
    xs, ys = rng.integers(0, 333, 200_000), rng.integers(0, 77, 200_000)
    xs[:5000] = 7                                    # a hot pixel: many atomics on one address
    ys[:5000] = 9
    want = np.bincount(ys.astype(np.int64) * 333 + xs, minlength=77 * 333).reshape(77, 333).astype(np.int32)
    assert np.array_equal(nsof_lib.bincount_2d(xs, ys, 77, 333), want)
    assert nsof_lib.bincount_2d(np.array([], int), np.array([], int), 4, 5).sum() == 0
    with pytest.raises(nsof_lib.NsofError):
        nsof_lib.bincount_2d(np.array([5]), np.array([0]), 4, 5)


@pytest.mark.gpu
def test_full_size_sensor_properties(nsof_lib, ctx, oracle):
    """BASELINE config 5 size (3840x2160): sparse path == forced-dense path bit for bit; untouched pixels keep the
    initial state exactly; a leaking silent voltage (outside the dead zone: every pixel changes every slice) matches
    the oracle on the whole array; chunked == one-shot."""
    from nsof import synth
    from nsof.accumulator import Accumulator, slice_index_array
    H, W = 2160, 3840
    x, y, p, t = synth.make_events(5, W, H, 40_000, 40_000, box=(400, 300))
    idx = slice_index_array(t, 1000)
    res = {}
    for dense in (False, True):
        acc = Accumulator(H, W, 1, "split", -6.0, 0.0, ctx=ctx, dense=dense)
        acc.step(x, y, p, t, idx, snap_every=0)
        res[dense] = acc.w(0)
        acc.close()
    assert np.array_equal(res[False], res[True])
    touched = np.zeros((H, W), bool)
    n_in = int(idx[-1])
    touched[y[:n_in], x[:n_in]] = True
    assert np.all(res[False][~touched] == np.float32(0.5)) and np.all(res[False][touched] > 0.5)
    # leak: silent_v = 0.4 > von -> the whole array decays every slice (dense math on all 8.3 Mpx)
    acc = Accumulator(H, W, 1, "split", -6.0, 0.4, ctx=ctx)
    k = 8                                                    # slices, one shot vs two chunks
    acc.step(x, y, p, t, idx[:k + 1], snap_every=0)
    one = acc.w(0)
    acc.reset()
    acc.step(x, y, p, t, idx[:4], snap_every=0)
    acc.step(x, y, p, t, idx[3:k + 1], snap_every=0)
    two = acc.w(0)
    acc.close()
    assert np.array_equal(one, two)
    w = np.full((H, W), 0.5, np.float32)
    for s in range(k):
        V = np.full((H, W), 0.4, np.float32)
        V[y[idx[s]:idx[s + 1]], x[idx[s]:idx[s + 1]]] = -6.0
        w = oracle.accum_update_state(w, V)
    assert np.abs(one - w).max() <= W_ATOL


@pytest.mark.parametrize("version,polarity", [(1, "split"), (2, "split"), (2, "magnitude")])
def test_staged_events_and_resume(nsof_lib, ctx, version, polarity):
    """nsof_accum_set_events + nsof_accum_run over sub-ranges == one nsof_accum_step_events call, and a run that is
    checkpointed (w, refractory map, slice counter), torn down and resumed in a NEW accumulator ends in the same state
    with the same snapshots (the reference persists only w_final, event_mem_sim.py:289-303)."""
    from nsof import synth
    W, H = 96, 64
    x, y, p, t = synth.make_events(5, W, H, 20000, 300_000, box=(20, 12))
    idx = nsof_lib.accumulator.slice_index_array(t, 1000)
    n = len(idx) - 1
    one = nsof_lib.Accumulator(H, W, version, polarity, -6.0, 0.0, ctx=ctx)
    one.step(x, y, p, t, idx, snap_every=7)
    want_w = [one.w(k) for k in range(2 if one.split else 1)]
    want_snaps = one.snapshots()
    one.close()

    a = nsof_lib.Accumulator(H, W, version, polarity, -6.0, 0.0, ctx=ctx)
    a.set_events(x, y, p, t, idx)
    cut = n // 3 + 5
    a.run(0, cut, snap_every=7)
    states = [a.state(k) for k in range(2 if a.split else 1)]
    first_snaps = a.snapshots()
    a.close()
    assert states[0]["slice_counter"] == cut
    b = nsof_lib.Accumulator(H, W, version, polarity, -6.0, 0.0, ctx=ctx)
    for k, st in enumerate(states):
        b.load_state(st, k)
    b.set_events(x, y, p, t, idx)
    b.run(cut, n - cut, snap_every=7)
    for k in range(len(want_w)):
        assert np.array_equal(b.w(k), want_w[k])
    second = b.snapshots()
    for k in range(len(want_w)):
        assert np.array_equal(np.concatenate([first_snaps[k], second[k]], 0), want_snaps[k])
    b.close()


def test_surface_u8_matches_host_map(nsof_lib, ctx, torch_dev):
    """nsof_accum_surface_u8_dev: mode "current" == the reference's current -> gray map (optical_flow_seg.py:426-431)
    applied on the host to the resistances the accumulator reports (identical but for the last ulp of log10 at integer
    boundaries); mode "state" == uint8(255 * w) exactly.  A leaking run (silent_v above von) brings w below the 0.42
    where the reference's map leaves saturation."""
    import torch
    from nsof import gating, synth
    W, H = 160, 120
    x, y, p, t = synth.make_events(9, W, H, 30000, 200_000, box=(30, 20))
    idx = nsof_lib.accumulator.slice_index_array(t, 1000)
    acc = nsof_lib.Accumulator(H, W, 1, "split", -8.0, 3.0, ctx=ctx)
    acc.step(x, y, p, t, idx)
    out = torch.zeros((2, H, W + 16), dtype=torch.uint8, device=torch_dev)
    torch.cuda.synchronize()
    acc.surface_u8(out[0], row_stride=W + 16, mode="current")
    acc.surface_u8(out[1], row_stride=W + 16, mode="state")
    ctx.synchronize()
    got = out.cpu().numpy()
    w = acc.w()
    want = gating.current_to_gray(1.0 / acc.resistance().astype(np.float64))
    acc.close()
    assert not got[:, :, W:].any()
    d = np.abs(got[0, :, :W].astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    assert want.max() > want.min() and want.min() < 255
    assert np.array_equal(got[1, :, :W], (w * np.float32(255.0)).astype(np.uint8))


def test_row_bands_with_real_accumulator(nsof_lib, ctx):
    """nsof.dist.simulate_banded with the GPU accumulator as the band callback on an RCCL process group (world size
    1 on this box): bands on the global slice grid, empty-band contract, gather on the device."""
    import torch.distributed as dist
    from nsof import dist as nd
    from nsof import synth
    import os
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(29950 + os.getpid() % 40)
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("LOCAL_RANK", "0")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl")
    try:
        W, H = 64, 47
        x, y, p, t = synth.make_events(11, W, H, 3000, 40_000, box=(12, 9))

        def band(xb, yb, pb, tb, idx_b, hw, slice_times):
            if hw[0] == 0:
                return np.zeros(hw, np.float32)
            acc = nsof_lib.Accumulator(hw[0], hw[1], 1, "split", -6.0, 0.0, ctx=ctx)
            try:
                acc.step(xb, yb, pb, tb, idx_b)
                return acc.w()
            finally:
                acc.close()

        out = nd.simulate_banded(x, y, p, t, (H, W), 1000, band)
        full = nsof_lib.simulate((x, y, p, t), version=1, slice_us=1000, active_v=-6.0, silent_v=0.0,
                                 sensor_size=(H, W), ctx=ctx)["w_final"]
        assert np.array_equal(out.numpy(), full)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("polarity", ["split", "magnitude"])
def test_scheme2_row_bands_stitch_to_the_unsharded_run(nsof_lib, ctx, polarity):
    """Scheme 2 over row bands (the refractory rule reads the first / last event time of the WHOLE
    stream's slice, /root/reference/eventsim/event_mem_sim.py:243-267): three bands, each an accumulator staged with
    its own events + nsof_accum_set_slice_times(global table), stitched == the unsharded simulate, both arrays; and
    WITHOUT the global table a band differs (that is what the call is for)."""
    from nsof import dist as nd
    from nsof import synth
    from test_dist_cpu import _refractory_stream
    W, H = 72, 50   # noqa: N806
    x, y, p, t = _refractory_stream(H, W, seed=9, duration_us=60_000)   # OFF = 0, as scheme 2 / split expects
    full = nsof_lib.simulate((x, y, p, t), version=2, slice_us=1000, active_v=-0.3, silent_v=0.0, polarity=polarity,
                             sensor_size=(H, W), ctx=ctx)
    tf, tl = nd.global_slice_times(t, 1000)
    rows_a, rows_b, plain_differs = [], [], False
    for (y0, y1) in nd.band_bounds(H, 3):
        xb, yb, pb, tb, _ = nd.events_in_band(x, y, p, t, y0, y1)
        idx = nd.band_slice_bounds(t, tb, 1000)
        for use_table in (True, False):
            acc = nsof_lib.Accumulator(y1 - y0, W, 2, polarity, -0.3, 0.0, ctx=ctx)
            try:
                acc.set_events(xb, yb, pb, tb, idx)
                if use_table:
                    acc.set_slice_times(tf, tl)
                acc.run(0, len(idx) - 1)
                wa = acc.w(0)
                wb = acc.w(1) if polarity == "split" else None
            finally:
                acc.close()
            if use_table:
                rows_a.append(wa)
                rows_b.append(wb)
            else:
                plain_differs |= not np.array_equal(wa, full["w_final"][y0:y1])
    assert np.array_equal(np.concatenate(rows_a, 0), full["w_final"])
    if polarity == "split":
        assert np.array_equal(np.concatenate(rows_b, 0), full["w_final_b"])
    assert plain_differs, "the band's own first / last event times gave the global result: the stream does not exercise the rule"
    with pytest.raises(nsof_lib.NsofError):
        acc = nsof_lib.Accumulator(8, 8, 2, polarity, -0.3, 0.0, ctx=ctx)
        try:
            acc.set_events(x[:0], y[:0], p[:0], t[:0], np.zeros(3, np.int64))
            acc.set_slice_times(tf[:5], tl[:5])   # 2 slices staged, 5 given
        finally:
            acc.close()


def test_sharded_sequence_pipeline_equals_single_gpu(nsof_lib, ctx):
    """events -> accumulator row bands -> all-gather of the surface frames -> sharded frame pairs
    (nsof.pipeline.events_to_flow_sequence_sharded) on an RCCL process group of one rank: same
    frames and flows as the single-GPU pipeline; and with the stream cut into two bands by hand (what two ranks would
    compute) the stitched frames are the same bytes."""
    import os

    import torch
    import torch.distributed as dist
    from nsof import dist as nd
    from nsof import pipeline, synth
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(29910 + os.getpid() % 40)
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("LOCAL_RANK", "0")
    W, H, every = 192, 135, 4   # noqa: N806
    x, y, p, t = synth.make_events(23, W, H, 20000, 24_000, box=(40, 30))
    frames1, flows1 = pipeline.events_to_flow_sequence(x, y, p, t, (H, W), snapshot_every=every, active_v=-6.0,
                                                       silent_v=0.5, ctx=ctx)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl")
    try:
        (lo, hi), frames, flows = pipeline.events_to_flow_sequence_sharded(x, y, p, t, (H, W), snapshot_every=every,
                                                                            active_v=-6.0, silent_v=0.5, ctx=ctx)
    finally:
        if created:
            dist.destroy_process_group()
    assert (lo, hi) == (0, frames1.shape[0] - 1) and frames1.shape[0] >= 4
    assert torch.equal(frames, frames1) and torch.equal(flows, flows1)
    assert int(frames1.max()) > int(frames1.min())
    # two bands by hand, each through the GPU accumulator on the global slice grid
    n_frames = frames1.shape[0]
    parts = []
    for (y0, y1) in nd.band_bounds(H, 2):
        xb, yb, pb, tb, _ = nd.events_in_band(x, y, p, t, y0, y1)
        idx = nd.band_slice_bounds(t, tb, 1000)
        acc = nsof_lib.Accumulator(y1 - y0, W, 1, "split", -6.0, 0.5, ctx=ctx, dense=True)
        out = torch.empty((n_frames, y1 - y0, W), dtype=torch.uint8, device=frames1.device)
        try:
            acc.set_events(xb, yb, pb, tb, idx)
            for k in range(n_frames):
                acc.run(k * every, every)
                acc.surface_u8(out[k])
            ctx.synchronize()
        finally:
            acc.close()
        parts.append(out)
    assert torch.equal(torch.cat(parts, 1), frames1)


def test_hdf5_event_file_round_trip(nsof_lib, tmp_path):
    """load_events + simulate(h5_path) on a /CD/events file (event_mem_sim.py:69-75, 359-364) and the file set the
    reference writes next to it (:289-322).  h5py lives in the image's conda python only, so the run is a child
    process under that interpreter (ctypes + numpy are all nsof.simulate needs); the golden stream and expected
    state come from the reference itself (tests/golden/accum_sim_v2_split.npz)."""
    import gzip
    import json
    import os
    import subprocess
    conda = "/opt/conda/bin/python3.9"
    if not os.path.exists(conda) or subprocess.run([conda, "-c", "import h5py"], capture_output=True).returncode:
        pytest.skip("no interpreter with h5py on this machine")
    from conftest import PKG, golden_path
    g = np.load(golden_path("accum_sim_v2_split.npz"))
    h5 = tmp_path / "stream.hdf5"
    script = f"""
import sys, numpy as np, h5py
sys.path.insert(0, {PKG!r})
g = np.load({golden_path("accum_sim_v2_split.npz")!r})
with h5py.File({str(h5)!r}, "w") as f:
    ev = f.create_group("CD").create_group("events")
    for k, dt in (("x", np.int16), ("y", np.int16), ("p", np.int8), ("t", np.int64)):
        ev.create_dataset(k, data=g[k].astype(dt))
from nsof.accumulator import load_events, simulate
x, y, p, t, H, W = load_events({str(h5)!r})
assert (H, W) == (int(g["y"].max()) + 1, int(g["x"].max()) + 1) and np.array_equal(t, g["t"]) and p.dtype.kind == "i"
out = simulate({str(h5)!r}, version=2, slice_us=int(g["slice_us"]), active_v=float(g["active_v"]),
               silent_v=float(g["silent_v"]), polarity="split")
print("ok", out["w_final"].shape)
"""
    r = subprocess.run([conda, "-c", script], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, NSOF_HIP_RUNTIME="system"))
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    a = np.load(tmp_path / "stream.V2.npz")
    b = np.load(tmp_path / "stream.V2_b.npz")
    H, W = a["w_final"].shape
    assert g["w_final"].shape == (H, W)
    assert np.abs(a["w_final"] - g["w_final"]).max() <= W_ATOL and np.abs(b["w_final"] - g["w_final_b"]).max() <= W_ATOL
    assert a["resistances"].dtype == np.float32 and a["resistances"].shape[0] == int(g["n_snapshots"])
    with gzip.open(tmp_path / "stream.V2.json.gz", "rt") as fp:
        meta = json.load(fp)
    assert meta["version"] == 2 and meta["polarity"] == "split" and meta["refractory_us"] == 800
    assert meta["scheme"] == "dc_bias_overlay" and meta["slice_us"] == int(g["slice_us"])


def test_block_current_on_device_equals_host_reduction(nsof_lib, ctx):
    """nsof_accum_block_current (block maximum of v_ds / R on the GPU, stored snapshots and the current state) == the
    host reduction over the downloaded float32 resistance maps, bit for bit; block sizes that do not divide the sensor
    leave the remainder out, as the host version does."""
    from nsof import pipeline, synth
    from nsof.accumulator import slice_index_array
    W, H = 333, 217   # noqa: N806
    x, y, p, t = synth.make_events(5, W, H, 30000, 60_000, box=(50, 40))
    idx = slice_index_array(t, 1000)
    acc = nsof_lib.Accumulator(H, W, 1, "split", -6.0, 0.5, ctx=ctx)
    try:
        acc.step(x, y, p, t, idx, snap_every=20)
        n = acc.snapshot_count()
        assert n >= 3
        dev = {ms: [acc.block_current(ms, snapshot=k) for k in range(n)] for ms in (1, 7, 20, 64)}
        cur = {ms: acc.block_current(ms) for ms in (7, 20)}
        r_now = acc.resistance()
        snaps = acc.snapshots()[0]
    finally:
        acc.close()
    for ms, got in dev.items():
        for k in range(n):
            want = pipeline.surface_to_block_current(snaps[k], ms)
            assert got[k].shape == want.shape == (H // ms, W // ms) and np.array_equal(got[k], want), (ms, k)
    for ms, got in cur.items():
        assert np.array_equal(got, pipeline.surface_to_block_current(r_now, ms))
    assert dev[20][-1].max() > dev[20][-1].min()


def test_config5_pipeline_flow_vs_oracle_chain(nsof_lib, ctx, oracle):
    """BASELINE config 5 joined (events -> leaky-integrate surface -> 8-bit frames -> Farneback between consecutive
    surface frames, nsof.pipeline.events_to_flow_sequence) at a reduced sensor against the oracle CHAIN:
    oracle/accum_ref.c for the slices -> uint8(255 w) -> oracle/farneback_ref.c.  The surface frames -- a flat field plus
    sparse dots -- are the low-texture class where rank-deficient 2x2 systems amplify the last bits of the box sums, so
    this is the leg that needs the library's row-sum order: frames byte-identical, flows bit-identical in the default
    mode; the opt-in fast row sums stay finite and close but are not held to 1e-4 here."""
    import torch
    from nsof import _lib, pipeline, synth
    from nsof.farneback import PARAMS_A
    W, H, every = 640, 360, 33   # noqa: N806
    x, y, p, t = synth.make_events(31, W, H, 60_000, 140_000, box=(60, 40))
    frames, flows = pipeline.events_to_flow_sequence(x, y, p, t, (H, W), snapshot_every=every, ctx=ctx)
    n_frames = frames.shape[0]
    assert n_frames >= 3
    gf, gfl = frames.cpu().numpy(), flows.cpu().numpy()
    pa = [getattr(PARAMS_A, k) for k in ("pyr_scale", "levels", "winsize", "iterations", "poly_n", "poly_sigma", "flags")]
    ref_frames = []
    for k in range(3):
        _, w = oracle.accum_slices_per_s(x, y, t, H, W, 1000, -6.0, 0.0, n_slices=(k + 1) * every, n_threads=4)
        ref_frames.append((np.float32(255.0) * w).astype(np.uint8))
        assert np.array_equal(gf[k], ref_frames[k]), k
    assert int(gf[2].max()) > int(gf[2].min())                     # events did move the surface
    for k in range(2):
        ref = oracle.farneback(ref_frames[k], ref_frames[k + 1], *pa)
        assert np.array_equal(gfl[k], ref), (k, float(np.abs(gfl[k] - ref).max()))
    fast = nsof_lib.calcOpticalFlowFarneback(gf[0], gf[1], None, *pa, ctx=ctx, exact=False)
    assert np.isfinite(fast).all()
    assert ctx.get_option(_lib.OPT_EXACT_ROWSUMS) == 1


@pytest.mark.gpu
def test_run_surface_equals_run_then_surface(nsof_lib, ctx):
    """nsof_accum_run_surface (the dense update's last pass writes the 8-bit frame itself) == nsof_accum_run followed by
    nsof_accum_surface_u8_dev, byte for byte: both modes, dense and sparse updates (the sparse one falls back to the separate
    surface launch), widths that are not a multiple of 4, strided frames, a silent voltage outside the dead zone."""
    import torch
    from nsof import synth
    from nsof.accumulator import Accumulator, slice_index_array
    dev = torch.device("cuda", ctx.device)
    for (H, W, silent, dense) in [(120, 160, 0.0, True), (77, 131, 0.0, True), (90, 202, 0.5, True), (120, 160, 0.0, False)]:
        x, y, p, t = synth.make_events(9, W, H, 6000, 80_000, box=(20, 16))
        idx = slice_index_array(t, 1000)
        for mode in ("state", "current"):
            frames = {}
            for fused in (False, True):
                acc = Accumulator(H, W, 1, "split", -6.0, silent, ctx=ctx, dense=dense)
                try:
                    acc.set_events(x, y, p, t, idx)
                    buf = torch.zeros((2, H, W + 12), dtype=torch.uint8, device=dev)   # strided rows
                    torch.cuda.synchronize()
                    for k in range(2):
                        if fused:
                            acc.run_surface(k * 33, 33, buf[k], row_stride=W + 12, mode=mode)
                        else:
                            acc.run(k * 33, 33)
                            acc.surface_u8(buf[k], row_stride=W + 12, mode=mode)
                    ctx.synchronize()
                    frames[fused] = buf.cpu().numpy()
                finally:
                    acc.close()
            assert np.array_equal(frames[True], frames[False]), (H, W, silent, dense, mode)
            assert frames[True][:, :, :W].any() and not frames[True][:, :, W:].any()


@pytest.mark.gpu
def test_run_frames_copy_patch_equals_dense_frames(nsof_lib, ctx, oracle):
    """nsof_accum_run_frames: with the silent voltage in the dead zone the frames come from an event-driven form -- the tile
    walk (a wave owns 1024 pixels for the whole run, frames write-only: two launches per call) or, where the shape does
    not allow it / on request, copy + patch per interval -- frames and final state byte-identical to the every-pixel pass
    per interval (dense=True takes n x run_surface), both surface modes, intervals of 33 / 7 / 64 / 20 slices, dense and
    strided frame tensors, widths that are / are not multiples of 16, tiles that end inside the image, a second call that
    continues the stream; the state against the CPU oracle; a silent voltage OUTSIDE the dead zone takes the generic path
    through the same entry."""
    import torch
    from nsof import synth
    from nsof.accumulator import Accumulator, slice_index_array
    dev = torch.device("cuda", ctx.device)
    for (H, W, every, pad, silent) in [(120, 160, 33, 0, 0.0), (77, 131, 7, 5, 0.0), (96, 128, 64, 0, 0.05), (90, 202, 33, 0, 0.5),
                                       (50, 208, 33, 16, 0.0), (129, 1040, 20, 0, 0.0)]:
        x, y, p, t = synth.make_events(9, W, H, 9000, 200_000, box=(20, 16))
        idx = slice_index_array(t, 1000)
        n_fr = (len(idx) - 1) // every
        assert n_fr >= 3
        for mode in ("state", "current"):
            got = {}
            # the every-pixel pass per interval; copy + patch per interval; the tile walk (default where the shape allows)
            for dense, fpath in ((True, None), ("cp", "copy_patch"), (None, None)):
                acc = Accumulator(H, W, 1, "split", -6.0, silent, ctx=ctx, dense=True if dense is True else None, frames_path=fpath)
                try:
                    acc.set_events(x, y, p, t, idx)
                    buf = torch.zeros((n_fr, H, W + pad), dtype=torch.uint8, device=dev)
                    torch.cuda.synchronize()
                    acc.run_frames(0, 2, every, buf[:2, :, :W], mode=mode)                  # two calls: the second continues
                    acc.run_frames(2 * every, n_fr - 2, every, buf[2:, :, :W], mode=mode)
                    ctx.synchronize()
                    got[dense] = (buf.cpu().numpy(), acc.w())
                finally:
                    acc.close()
            assert np.array_equal(got[None][0], got[True][0]), (H, W, every, silent, mode)
            assert np.array_equal(got[None][1], got[True][1])
            assert np.array_equal(got["cp"][0], got[True][0]) and np.array_equal(got["cp"][1], got[True][1]), (H, W, every, silent, mode)
            assert got[None][0][:, :, :W].any() and not got[None][0][:, :, W:].any()
            if mode == "state":   # (the current -> gray map saturates at 255 for w >= 0.42: those frames are constant)
                assert len({got[None][0][k].tobytes() for k in range(n_fr)}) > 1          # the frames do change
        _, w_ref = oracle.accum_slices_per_s(x, y, t, H, W, 1000, -6.0, silent, n_slices=n_fr * every, n_threads=2)
        assert np.abs(got[None][1] - w_ref).max() <= 5e-7


@pytest.mark.gpu
def test_dense_groups_of_64_slices(nsof_lib, ctx, oracle):
    """The dense scheme-1 update fuses up to 64 slices per pass (two mask words per pixel): runs whose groups exceed 32
    slices equal the event-pixel path bit for bit (silent voltage in the dead zone) and the oracle within the usual
    tolerance (silent voltage outside it: every pixel integrates in every slice, in slice order)."""
    from nsof import synth
    from nsof.accumulator import Accumulator, slice_index_array
    H, W = 90, 131
    x, y, p, t = synth.make_events(17, W, H, 9000, 150_000, box=(20, 16))
    idx = slice_index_array(t, 1000)
    n = len(idx) - 1
    assert n >= 140
    ws = {}
    for dense in (False, True):
        acc = Accumulator(H, W, 1, "split", -6.0, 0.0, ctx=ctx, dense=dense)
        try:
            acc.set_events(x, y, p, t, idx)
            acc.run(0, 70)          # dense: one group of 64 + one of 6
            acc.run(70, 33)         # one group of 33
            acc.run(103, n - 103)
            ws[dense] = acc.w()
        finally:
            acc.close()
    assert np.array_equal(ws[True], ws[False])
    ref = oracle.accum_simulate(x, y, p, t, H, W, 1, "split", 1000, -6.0, 0.0)
    assert np.abs(ws[True] - ref["w_final"]).max() <= W_ATOL
    acc = Accumulator(H, W, 1, "split", -6.0, 0.5, ctx=ctx)       # leaks in every silent slice: the every-pixel pass
    try:
        acc.set_events(x, y, p, t, idx)
        acc.run(0, n)
        w_leak = acc.w()
    finally:
        acc.close()
    ref = oracle.accum_simulate(x, y, p, t, H, W, 1, "split", 1000, -6.0, 0.5)
    assert np.abs(w_leak - ref["w_final"]).max() <= 2e-6, float(np.abs(w_leak - ref["w_final"]).max())
