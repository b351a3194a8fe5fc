#!/usr/bin/env python3
"""Data fixtures for the multi-dataset (BASELINE config 4) harness, extracted from the reference's data/ directory:

  gating_stacks.npz      the complete ``constructed3DMatrix`` stack of every dataset (device currents in ampere,
                         float64 rows x cols x slices) from data/*/constructed_3D_matrix.mat, parsed with
                         scipy.io.loadmat (a binary parser; nothing in the file is executed)
  frames/<dataset>/*.jpg the first three RGB frames of autodriving, uav, uavnew2 and tabletennis, copied byte for byte
                         (grasp's first two frames are already under demo/)

Data only -- no reference source text.  Run in the build container:  python tests/golden/gen_dataset_fixtures.py"""
import os
import re
import shutil

import numpy as np
import scipy.io as sio

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/data"
DATASETS = ["grasp", "autodriving", "uav", "uavnew2", "tabletennis"]


def numeric_sorted(names):
    return sorted(names, key=lambda s: int(re.match(r"(\d+)", s).group(1)))


def main():
    stacks = {}
    for name in DATASETS:
        stacks[name] = np.asarray(sio.loadmat(f"{REF}/{name}/constructed_3D_matrix.mat")["constructed3DMatrix"], np.float64)
    np.savez_compressed(os.path.join(HERE, "gating_stacks.npz"), **stacks)
    for name in DATASETS[1:]:
        src = f"{REF}/{name}/RGB"
        dst = os.path.join(HERE, "frames", name)
        os.makedirs(dst, exist_ok=True)
        for f in numeric_sorted(os.listdir(src))[:3]:
            shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
            os.chmod(os.path.join(dst, f), 0o644)
    print({k: v.shape for k, v in stacks.items()}, os.path.getsize(os.path.join(HERE, "gating_stacks.npz")))


if __name__ == "__main__":
    main()
