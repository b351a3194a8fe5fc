#!/usr/bin/env python3
"""Extract a few slices of the reference's gating data (data/*/constructed_3D_matrix.mat, key
constructed3DMatrix: device currents in ampere) into tests/golden/gating_maps.json.  Data only; the .mat files are
parsed with scipy.io.loadmat (a binary parser, nothing is executed).  Run in the build container:
    python tests/golden/gen_gating_fixtures.py"""
import json
import os

import scipy.io as sio

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gating_maps.json")
SLICES = {"grasp": [0, 1, 2, 3, 50], "autodriving": [15, 16, 40, 60, 90], "uav": [15, 16, 30, 60, 100],
          "uavnew2": [0, 1, 2, 20], "tabletennis": [0, 1, 2, 10]}
SIZES = {"grasp": (1920, 1080), "autodriving": (801, 801), "uav": (161, 161), "uavnew2": (600, 600),
         "tabletennis": (160, 160)}   # frame (h, w) of data/*/RGB


def main():
    out = {}
    for name, sl in SLICES.items():
        m = sio.loadmat(f"/root/reference/data/{name}/constructed_3D_matrix.mat")["constructed3DMatrix"]
        out[name] = {"frame_hw": SIZES[name], "stack_shape": list(m.shape),
                     "slices": {str(k): [[repr(float(v)) for v in row] for row in m[:, :, k]] for k in sl}}
    with open(OUT, "w") as f:
        json.dump(out, f)
    print(OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
