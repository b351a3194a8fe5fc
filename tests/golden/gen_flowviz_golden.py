#!/usr/bin/env python3
"""Golden vectors for the Middlebury flow colour coding: outputs of the reference's own
/root/reference/flow_viz.py (numpy only; vendored tomrunia code, flow_to_image :109-135, make_colorwheel :20-69) on
seeded flow fields.  Run in the build container:  python tests/golden/gen_flowviz_golden.py
Writes tests/golden/flowviz_golden.npz (inputs + expected uint8 images).  The reference file is imported from where
it lies; nothing of it is copied."""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
import flow_viz as ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(42)
    out = {"colorwheel": ref.make_colorwheel()}
    cases = {
        "smooth": np.stack(np.meshgrid(np.linspace(-6, 6, 40), np.linspace(-3, 9, 31)), -1).astype(np.float32),
        "noise": (rng.standard_normal((23, 37, 2)) * 4).astype(np.float32),
        "tiny": (rng.standard_normal((5, 7, 2)) * 1e-7).astype(np.float32),
        "zero": np.zeros((4, 6, 2), np.float32),
        "f64": rng.standard_normal((9, 11, 2)) * 20,
    }
    for name, flow in cases.items():
        out[f"{name}_flow"] = flow
        out[f"{name}_rgb"] = ref.flow_to_image(flow)
        out[f"{name}_bgr"] = ref.flow_to_image(flow, convert_to_bgr=True)
        out[f"{name}_clip"] = ref.flow_to_image(flow, clip_flow=2.5)
    np.savez_compressed(os.path.join(HERE, "flowviz_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
