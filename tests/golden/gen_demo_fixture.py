#!/usr/bin/env python3
"""Extract the only recorded OUTPUT of the real cv2 pipeline that the reference holds: the figure stored in the last
cell of demo.ipynb (grasp pair 1.jpg -> 2.jpg, params A, run by the authors with opencv-python).  Its panels are

    bottom-left   flow_to_image(-cv2.calcOpticalFlowFarneback(gray1, gray2, None, **params_A))   "full_flow"
    bottom-middle flow_to_image(-ROI-gated flow)                                                  "roi_flow"
    bottom-right  motion segmentation mask                                                       "seg_mask"
    top-right     memimg2 (imshow, cmap=hot)                                                     "mem_map"

each drawn by matplotlib at 247x438 px (the frames are 1080x1920).  This script crops the panels out of the PNG (data:
an output image, nothing executable) and copies the two input JPEGs the figure was computed from, so that the tests can
compare this build's flow, rendered the same way, with what cv2 produced.  Run in the build container:
    python tests/golden/gen_demo_fixture.py"""
import base64
import io
import json
import os
import shutil

import numpy as np
from PIL import Image

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "demo")
REF = "/root/reference"
# panel boxes (row0, row1, col0, col1) found from the non-white extents of the top-row photographs
PANELS = {"full_flow": (537, 975, 30, 277), "roi_flow": (537, 975, 605, 852), "seg_mask": (537, 975, 1181, 1428),
          "mem_map": (62, 500, 1185, 1423), "prev_frame": (62, 500, 30, 277)}


def main():
    os.makedirs(HERE, exist_ok=True)
    nb = json.load(open(os.path.join(REF, "demo.ipynb")))
    png = None
    for cell in nb["cells"]:
        for out in cell.get("outputs", []):
            if "image/png" in out.get("data", {}):
                png = base64.b64decode(out["data"]["image/png"])
    fig = np.asarray(Image.open(io.BytesIO(png)).convert("RGB"))
    assert fig.shape == (985, 1463, 3), fig.shape
    for name, (r0, r1, c0, c1) in PANELS.items():
        Image.fromarray(fig[r0:r1, c0:c1]).save(os.path.join(HERE, f"panel_{name}.png"), optimize=True)
    for k in (1, 2):
        shutil.copyfile(os.path.join(REF, "data/grasp/RGB", f"{k}.jpg"), os.path.join(HERE, f"grasp_{k}.jpg"))
        os.chmod(os.path.join(HERE, f"grasp_{k}.jpg"), 0o644)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
