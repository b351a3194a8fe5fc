#!/usr/bin/env python3
"""The reference's demo.ipynb (grasp pair 1.jpg -> 2.jpg) on the MI355X path, end to end:

    gating map -> ROI -> Farneback on the ROI crop (GPU) -> negate -> motion mask (GPU) ; full-frame flow (GPU)

and a side-by-side of the panels the authors recorded with cv2 (tests/golden/demo/panel_*.png, extracted from the
notebook) against the same panels rendered from this build's results.

    python examples/demo_grasp.py [--out gpurun_out/demo]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "demo"))
    a = ap.parse_args()
    import numpy as np
    from PIL import Image
    import nsof

    demo = os.path.join(ROOT, "tests", "golden", "demo")
    bgr = [np.ascontiguousarray(np.asarray(Image.open(os.path.join(demo, f"grasp_{k}.jpg")).convert("RGB"))[..., ::-1])
           for k in (1, 2)]                                            # what cv2.imread returns
    g1, g2 = (nsof.frame_to_gray(f) for f in bgr)                      # cv2.cvtColor(frame, cv2.COLOR_RGB2GRAY)
    stack = json.load(open(os.path.join(ROOT, "tests", "golden", "gating_maps.json")))["grasp"]["slices"]
    mem = [nsof.current_to_gray(np.array([[float(v) for v in row] for row in stack[k]])) for k in ("0", "1")]
    cfg = nsof.dataset_config("grasp")                                 # data/grasp/Parameters.txt

    nsof.calcOpticalFlowFarneback(g1, g2, None, **cfg.farneback_params.as_kwargs())   # warm-up (context, workspace)
    t0 = time.perf_counter()
    out = nsof.opticalFlow3D(mem[0], mem[1], g1, g2, cfg.MEMSIZE, cfg.MEMSIZE, cfg)
    flow, rect = out[0], out[-1]
    flow = -flow                                                       # "Invert for Farneback"
    t1 = time.perf_counter()
    mask = nsof.task_results(bgr[0], bgr[1], flow, 2, rect)
    t2 = time.perf_counter()
    full = -nsof.calcOpticalFlowFarneback(g1, g2, None, **cfg.farneback_params.as_kwargs())
    t3 = time.perf_counter()

    def small(img):
        return np.asarray(Image.fromarray(img).resize((247, 438), Image.BILINEAR))

    ours = {"full_flow": small(nsof.flow_to_image(full)), "roi_flow": small(nsof.flow_to_image(flow)),
            "seg_mask": np.repeat(small(mask)[..., None], 3, 2)}
    ref = {k: np.asarray(Image.open(os.path.join(demo, f"panel_{k}.png")).convert("RGB")) for k in ours}
    os.makedirs(a.out, exist_ok=True)
    rows = [np.hstack([ref[k] for k in ours]), np.hstack([ours[k] for k in ours])]
    Image.fromarray(np.vstack(rows)).save(os.path.join(a.out, "panels_reference_top_ours_bottom.png"), optimize=True)
    report = {"roi_rect_x0y0x1y1": [int(v) for v in rect],
              "ms": {"gated_roi_flow": round((t1 - t0) * 1e3, 2), "motion_mask": round((t2 - t1) * 1e3, 2),
                     "full_frame_flow": round((t3 - t2) * 1e3, 2)},
              "mean_abs_colour_diff": {k: round(float(np.abs(ours[k].astype(float) - ref[k]).mean()), 3) for k in ours},
              "mask_iou": round(float(((ours["seg_mask"] > 127) & (ref["seg_mask"] > 127)).sum() /
                                      ((ours["seg_mask"] > 127) | (ref["seg_mask"] > 127)).sum()), 4)}
    with open(os.path.join(a.out, "report.json"), "w") as f:
        json.dump(report, f)
    print(json.dumps(report))


if __name__ == "__main__":
    main()
