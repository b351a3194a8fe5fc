"""CPU: the accumulator oracle (oracle/accum_ref.c) against golden vectors produced by RUNNING the reference
(/root/reference/eventsim/event_mem_sim.py) in the build container -- tests/golden/gen_accum_golden.py.

Tolerance: NumPy 1.26 evaluates float32 pow/exp with SIMD kernels that are 1-3 ulp off glibc's powf/expf,
so update_state is compared at 1.2e-7 absolute (w <= 1, i.e. <= 2 ulp at w ~ 1) and resistances at 1e-6 relative."""
import numpy as np
import pytest

from conftest import golden_path

CASES = ["v1", "v2_split", "v2_magnitude", "v1_leak", "v2_split_bias", "v1_exact", "v2_split_pm1"]


def test_update_state_grid_and_random(oracle):
    g = np.load(golden_path("accum_update_state.npz"))
    assert np.abs(oracle.accum_update_state(g["w_grid"], g["V_grid"]) - g["out_grid"]).max() <= 1.2e-7
    assert np.abs(oracle.accum_update_state(g["w_rand"], g["V_rand"]) - g["out_rand"]).max() <= 1.2e-7


def test_update_state_dead_zone_and_clip(oracle):
    w = np.array([0.0, 0.3, 1.0], np.float32)
    for v in (-0.2, 0.0, 0.1):  # thresholds are exclusive: V < voff, V > von (event_mem_sim.py:44-45)
        assert np.array_equal(oracle.accum_update_state(w, np.full(3, v, np.float32)), w)
    assert oracle.accum_update_state(np.float32([0.999999]), np.float32([-8]))[0] == 1.0
    assert oracle.accum_update_state(np.float32([1e-6]), np.float32([6]))[0] == 0.0


def test_known_answers_from_survey(oracle):
    V = np.array([-8, -6, -1, -0.21, -0.2, 0, 0.1, 0.11, 1, 3], np.float32)
    want = np.array([0.7042345, 0.6518667, 0.5209471, 0.50026184, 0.5, 0.5, 0.5, 0.49975047, 0.47754133, 0.4276332],
                    np.float32)
    assert np.abs(oracle.accum_update_state(np.full(V.shape, 0.5, np.float32), V) - want).max() <= 6e-8
    r = oracle.accum_resistance(np.array([0.5, 0.6518667, 0.50026184], np.float32))
    assert np.allclose(r, [586221.1990, 397622.2887, 585828.9265], rtol=1e-6)


def test_resistance(oracle):
    g = np.load(golden_path("accum_update_state.npz"))
    r = oracle.accum_resistance(g["w_rand"])
    assert (np.abs(r - g["res_rand"]) / g["res_rand"]).max() <= 1e-6


def test_slice_bounds_match_numpy(oracle):
    rng = np.random.default_rng(0)
    for span, n, sl in [(200_000, 6000, 1000), (50_000, 2500, 1000), (999, 10, 1000), (10_000, 500, 333)]:
        t = np.sort(rng.integers(0, span, n)).astype(np.int64)
        bounds = np.arange(t[0], t[-1] + sl, sl, dtype=t.dtype)
        assert np.array_equal(oracle.accum_slice_bounds(t, sl), np.searchsorted(t, bounds))
    # span an exact multiple of slice_us: the events at t[-1] fall outside the last slice (reference quirk)
    t = np.array([0, 10, 1000, 2000, 2000], np.int64)
    idx = oracle.accum_slice_bounds(t, 1000)
    assert list(idx) == [0, 2, 3] and idx[-1] < len(t)


@pytest.mark.parametrize("name", CASES)
def test_simulate_vs_reference(oracle, name):
    d = np.load(golden_path(f"accum_sim_{name}.npz"))
    H, W = d["w_final"].shape
    out = oracle.accum_simulate(d["x"], d["y"], d["p"], d["t"], H, W, int(d["version"]), str(d["polarity"]),
                                int(d["slice_us"]), float(d["active_v"]), float(d["silent_v"]))
    assert out["resistances"].shape[0] == int(d["n_snapshots"])
    assert np.abs(out["w_final"] - d["w_final"]).max() <= 3e-7
    idx = d["snap_idx"]
    assert (np.abs(out["resistances"][idx] - d["resistances"]) / d["resistances"]).max() <= 1e-6
    if "w_final_b" in d:
        assert np.abs(out["w_final_b"] - d["w_final_b"]).max() <= 3e-7
        assert (np.abs(out["resistances_b"][idx] - d["resistances_b"]) / d["resistances_b"]).max() <= 1e-6


def test_frame_driven_variant_properties(oracle):
    """simulation/*.m restatement (parity unpinned: no MATLAB): checked against the source's own structure."""
    rng = np.random.default_rng(1)
    imgs = rng.random((5, 4, 4))
    imgs[2] = imgs[1]                        # an unchanged frame pair: every pixel leaks
    w, res = oracle.accum_frames(imgs, n_sub=1000)
    assert res.shape == (5, 4, 4) and np.allclose(res[0], np.sqrt(163305.0 * 2104377.0))   # w = 0.5 initially
    assert np.all((w >= 0) & (w <= 1))
    lam = np.log(2104377 / 163305)
    assert np.allclose(res[-1], 163305 / np.exp(-lam * (1 - w)), rtol=1e-12)
    # unchanged pixels get d = 0 -> V = -3.3 -> v_mod = +12.9 > von: w decreases (leak)
    w1, r1 = oracle.accum_frames(imgs[1:3], n_sub=1000)
    assert np.all(w1 < 0.5)
    # strongly changed pixels (d > th2) get v_mod < voff: w increases
    a = np.zeros((2, 2, 2)); a[1] = 0.5
    w2, _ = oracle.accum_frames(a, n_sub=1000)
    assert np.all(w2 > 0.5)
    # sub-stepping converges: 1000 and 2000 sub-steps agree to 1e-4, 1 and 1000 differ visibly
    wa, _ = oracle.accum_frames(imgs, n_sub=2000)
    assert np.abs(wa - w).max() < 1e-4
    # the float64 ODE with ONE sub-step of dt equals the event simulator's float32 update_state
    g = np.load(golden_path("accum_update_state.npz"))
    from oracle import oracle as O
    l = O.lib()
    import ctypes as C
    l.nsof_ref_frame_drive.restype = C.c_double
    l.nsof_ref_frame_drive.argtypes = [C.c_double] * 4
    assert abs(l.nsof_ref_frame_drive(0.0, 0.0, 0.7, 1.5) - 12.9) < 1e-12      # -(3*(-3.3) - 3)
    assert abs(l.nsof_ref_frame_drive(0.0, 0.5, 0.7, 1.5) - (-0.3 * 132 * 0.75)) < 1e-12
