#!/opt/conda/bin/python3.9
"""Generate golden vectors for the accumulator by RUNNING THE REFERENCE ITSELF.

Run in the build container only (the reference never travels):
    /opt/conda/bin/python3.9 tests/golden/gen_accum_golden.py
Imports /root/reference/eventsim/event_mem_sim.py (numpy 1.26.4, h5py 3.3.0) with a stub
`cv2` module (cv2 is only used by write_video(), which is never called: save_video=False).
Writes tests/golden/accum_update_state.npz and tests/golden/accum_sim_*.npz: inputs and the
reference's outputs only (data, no reference source).
"""
import os
import sys
import tempfile
import types
from pathlib import Path

sys.modules["cv2"] = types.ModuleType("cv2")
sys.path.insert(0, "/root/reference/eventsim")
import h5py  # noqa: E402
import numpy as np  # noqa: E402
import event_mem_sim as ems  # noqa: E402

OUT = Path(__file__).resolve().parent


def gen_update_state():
    V = np.array([-8, -6, -3, -1, -0.5, -0.21, -0.2000001, -0.2, -0.19, 0, 0.05, 0.1, 0.1000001, 0.11, 0.5, 1, 3, 6],
                 np.float32)
    w = np.array([0, 1e-6, 0.01, 0.1, 0.25, 0.5, 0.50026184, 0.6518667, 0.75, 0.9, 0.99, 0.999999, 1], np.float32)
    Vg, wg = np.meshgrid(V, w, indexing="ij")
    Vg = np.ascontiguousarray(Vg, np.float32)
    wg = np.ascontiguousarray(wg, np.float32)
    out = ems.update_state(wg, Vg)
    rng = np.random.default_rng(7)
    wr = rng.random(4096, dtype=np.float32)
    Vr = (rng.random(4096, dtype=np.float32) * 16 - 8).astype(np.float32)
    outr = ems.update_state(wr, Vr)
    res = np.asarray(ems.resistance_exp(wr), dtype=np.float32)
    res_grid = np.asarray(ems.resistance_exp(w), dtype=np.float32)
    np.savez_compressed(OUT / "accum_update_state.npz", V_grid=Vg, w_grid=wg, out_grid=out.astype(np.float32),
                        w_rand=wr, V_rand=Vr, out_rand=outr.astype(np.float32), res_rand=res, w_res=w,
                        res_grid=res_grid)
    print("update_state grid", out.dtype, "resistance dtype", ems.resistance_exp(wr).dtype)


def make_stream(seed, W, H, n_events, t_span, p_values, exact_multiple=None):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, W, n_events).astype(np.int16)
    y = rng.integers(0, H, n_events).astype(np.int16)
    # cluster a third of the events on a small patch: duplicates within a slice + refractory hits
    k = n_events // 3
    x[:k] = rng.integers(10, 16, k)
    y[:k] = rng.integers(20, 24, k)
    x[-1], y[-1] = W - 1, H - 1  # the loader infers the sensor size from the max coordinate
    p = rng.choice(np.array(p_values, np.int8), n_events).astype(np.int8)
    t = np.sort(rng.integers(0, t_span, n_events)).astype(np.int64)
    if exact_multiple is not None:
        t[0] = 0
        t[-3:] = exact_multiple  # several events exactly on the last boundary -> dropped by slicing
    perm = rng.permutation(n_events)
    x, y, p = x[perm], y[perm], p[perm]  # t stays sorted, pixels shuffled
    x[-1], y[-1] = W - 1, H - 1
    return x, y, p, t


def run_case(name, stream, version, polarity, slice_us, active_v, silent_v, keep=(0, 1, -1)):
    x, y, p, t = stream
    with tempfile.TemporaryDirectory() as d:
        h5 = Path(d) / "s.hdf5"
        with h5py.File(h5, "w") as f:
            g = f.create_group("/CD/events")
            g.create_dataset("x", data=x, dtype=np.int16)
            g.create_dataset("y", data=y, dtype=np.int16)
            g.create_dataset("p", data=p, dtype=np.int8)
            g.create_dataset("t", data=t, dtype=np.int64)
        ems.simulate(h5, version=version, slice_us=slice_us, active_v=active_v, silent_v=silent_v,
                     save_video=False, polarity=polarity)
        a = np.load(h5.with_suffix(f".V{version}.npz"))
        out = dict(x=x, y=y, p=p, t=t, version=version, polarity=polarity, slice_us=slice_us,
                   active_v=np.float32(active_v), silent_v=np.float32(silent_v), w_final=a["w_final"],
                   n_snapshots=a["resistances"].shape[0], snap_idx=np.array(keep),
                   resistances=a["resistances"][list(keep)])
        if version == 2:
            b = np.load(h5.with_suffix(".V2_b.npz"))
            if polarity == "split":
                out.update(w_final_b=b["w_final"], resistances_b=b["resistances"][list(keep)])
            else:
                assert b["w_final"].size == 0
    np.savez_compressed(OUT / f"accum_sim_{name}.npz", **out)
    print(name, "slices->snapshots", out["n_snapshots"], "w range", out["w_final"].min(), out["w_final"].max())


def main():
    gen_update_state()
    s01 = make_stream(11, 64, 48, 6000, 200_000, (0, 1))
    run_case("v1", s01, 1, "split", 1000, -6.0, 0.0)
    run_case("v2_split", s01, 2, "split", 1000, -6.0, 0.0)
    run_case("v2_magnitude", s01, 2, "magnitude", 1000, -6.0, 0.0)
    # silent voltage outside the dead zone: every pixel leaks each slice (dense path), larger pulse
    run_case("v1_leak", s01, 1, "split", 1000, -8.0, 0.5)
    run_case("v2_split_bias", s01, 2, "split", 500, -3.0, -0.5)
    # span an exact multiple of slice_us: the events at t == t[-1] are dropped by slice_indices
    s_exact = make_stream(12, 40, 30, 2500, 50_000, (-1, 1), exact_multiple=50_000)
    run_case("v1_exact", s_exact, 1, "split", 1000, -6.0, 0.0)
    run_case("v2_split_pm1", s_exact, 2, "split", 1000, -6.0, 0.0)  # OFF = -1 never matches p == 0


if __name__ == "__main__":
    main()
