#!/opt/conda/bin/python3.9
"""Golden values for the SSIM score of the prediction task (optical_flow_prediction.py:113-115):
``structural_similarity(true[:,:,2], prediction[:,:,2], data_range=255.0)`` from scikit-image, the library the
reference imports (requirements.txt pins 0.23.2; the build container holds 0.18.3 under /opt/conda -- same defaults:
7x7 uniform window, sample covariance, K1=0.01, K2=0.03).  Inputs and scores go to tests/golden/ssim_golden.npz.
Run in the build container:  /opt/conda/bin/python3.9 tests/golden/gen_ssim_golden.py"""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from skimage.metrics import structural_similarity  # noqa: E402
import skimage  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ssim_golden.npz")


def main():
    rng = np.random.default_rng(11)
    cases = {}
    for name, (h, w) in {"small": (7, 7), "odd": (37, 53), "wide": (24, 200), "frame": (120, 160)}.items():
        yy, xx = np.mgrid[0:h, 0:w]
        base = (127 + 90 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + rng.normal(0, 12, (h, w))).clip(0, 255)
        a = np.zeros((h, w, 3), np.uint8)
        b = np.zeros((h, w, 3), np.uint8)
        a[..., 2] = base.astype(np.uint8)
        b[..., 2] = (np.roll(base, 1, axis=1) + rng.normal(0, 6, (h, w))).clip(0, 255).astype(np.uint8)
        a[..., :2] = rng.integers(0, 256, (h, w, 2))       # other channels must not influence the score
        b[..., :2] = rng.integers(0, 256, (h, w, 2))
        cases[name + "_a"] = a
        cases[name + "_b"] = b
        cases[name + "_ssim"] = np.float64(structural_similarity(a[:, :, 2], b[:, :, 2], data_range=255.0))
    flat = np.full((20, 30, 3), 77, np.uint8)
    cases["flat_a"] = flat
    cases["flat_b"] = flat.copy()
    cases["flat_ssim"] = np.float64(structural_similarity(flat[:, :, 2], flat[:, :, 2], data_range=255.0))
    cases["skimage_version"] = np.array(skimage.__version__)
    np.savez_compressed(OUT, **cases)
    print(OUT, os.path.getsize(OUT), {k: float(v) for k, v in cases.items() if k.endswith("_ssim")})


if __name__ == "__main__":
    main()
