#!/opt/conda/bin/python3.9
"""Golden data for the reference's synthetic event generator and 2-D histogram, produced by RUNNING THE REFERENCE
(eventsim/event_mem_sim.py: generate_synthetic_events :109-158, bincount_2d :100-104) in the build container:
    /opt/conda/bin/python3.9 tests/golden/gen_synth_events_golden.py
Stub cv2 as in gen_accum_golden.py (cv2 is only used by write_video).  Output: tests/golden/synth_events.npz with the
event arrays of two generator settings (int32/int8/int64, a few hundred KB compressed) and histograms of two slices."""
import sys
import types
from pathlib import Path

sys.modules["cv2"] = types.ModuleType("cv2")
sys.path.insert(0, "/root/reference/eventsim")
import numpy as np  # noqa: E402
import event_mem_sim as ems  # noqa: E402

OUT = Path(__file__).resolve().parent / "synth_events.npz"


def main():
    d = {}
    for name, kw in {"default": {}, "small": dict(H=60, W=90, box_h=20, box_w=11, speed_pps=700, duration_s=0.2)}.items():
        x, y, p, t = ems.generate_synthetic_events(**kw)
        d[f"{name}_x"] = np.asarray(x, np.int16)
        d[f"{name}_y"] = np.asarray(y, np.int16)
        d[f"{name}_p"] = np.asarray(p, np.int8)
        d[f"{name}_t"] = np.asarray(t, np.int64)
        print(name, len(x), int(t.min()), int(t.max()), int(y.max()) + 1, int(x.max()) + 1, sorted(set(p.tolist())))
    x, y = d["default_x"], d["default_y"]
    d["hist_all"] = ems.bincount_2d(x, y, 240, 320)
    d["hist_first_2000"] = ems.bincount_2d(x[:2000], y[:2000], 145, 320)
    np.savez_compressed(OUT, **d)
    print(OUT, OUT.stat().st_size)


if __name__ == "__main__":
    main()
