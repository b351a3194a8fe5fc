"""ROI gating harness: host logic on CPU with the reference's own gating data
(tests/golden/gating_maps.json = slices of data/*/constructed_3D_matrix.mat), flow calls on the GPU."""
import json
import os

import numpy as np
import pytest

from conftest import golden_path

# union boxes (x0, y0, x1, y1) of the first frame pairs, as recorded independently in SURVEY.md section 8f
SURVEY_MERGED = {("grasp", 0): (460, 0, 1060, 1920), ("grasp", 1): (300, 0, 1060, 1140),
                 ("grasp", 2): (540, 940, 1060, 1140), ("uavnew2", 0): (0, 60, 600, 600),
                 ("uavnew2", 1): (180, 100, 420, 340), ("tabletennis", 0): (0, 0, 160, 160),
                 ("tabletennis", 1): (70, 10, 160, 160)}


@pytest.fixture(scope="module")
def maps():
    with open(golden_path("gating_maps.json")) as f:
        d = json.load(f)
    return {name: (tuple(ent["frame_hw"]),
                   {int(k): np.array([[float(v) for v in row] for row in rows]) for k, rows in ent["slices"].items()})
            for name, ent in d.items()}


def test_current_to_gray(nsof_lib):
    g = nsof_lib.current_to_gray(np.array([[1e-6, 4.752e-7, 3.8e-6, 6e-11, 0.0, 1.0]]))
    # -3366/log10(1e-6) - 306 = 255 exactly; 1/Roff maps to ~226; anything >= 1 uA saturates; tiny currents -> 0
    assert g.dtype == np.uint8 and g[0, 0] == 255 and g[0, 2] == 255 and g[0, 3] == 23 and g[0, 1] == 226
    assert g[0, 4] == 0  # log10(0) = -inf -> -306 -> clipped to 0 (the reference silences the warning the same way)


def test_connected_components_match_scipy(nsof_lib):
    """Partition, statistics AND label order against scipy.ndimage.label (raster order of each component's first
    pixel).  PARITY UNPINNED vs cv2 for the label ORDER: cv2.connectedComponentsWithStats numbers its components in
    the order its two-pass algorithm (SAUF / Spaghetti, depending on the build) resolves them, which for 4-connectivity
    is the raster order of the first pixel in every case checked by hand but is not documented.  The order only matters
    for FLAG 1 datasets when component boxes overlap (the later paste wins, optical_flow_seg.py:162); the union box
    of FLAG 2 and every non-overlapping case do not depend on it.  cv2 is not installable here; the live cross-check
    lives in tests/test_cv2_crosscheck.py."""
    from scipy import ndimage
    rng = np.random.default_rng(0)
    for shape in [(4, 4), (24, 13), (15, 15), (16, 16), (7, 31)]:
        for p in (0.2, 0.5, 0.8):
            img = (rng.random(shape) < p).astype(np.uint8) * 255
            n, labels, stats, cents = nsof_lib.connectedComponentsWithStats(img, connectivity=4)
            ref, nref = ndimage.label(img, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
            assert n == nref + 1
            # same partition (label numbering: raster order of the first pixel in both)
            assert np.array_equal(labels, ref)
            for lab in range(1, n):
                ys, xs = np.nonzero(ref == lab)
                assert tuple(stats[lab]) == (xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1,
                                             ys.size)
            n8, *_ = nsof_lib.connectedComponentsWithStats(img, connectivity=8)
            assert n8 - 1 == ndimage.label(img, structure=np.ones((3, 3)))[1]


def test_roi_rectangles_from_reference_data(nsof_lib, maps):
    from nsof import gating
    calls = []

    def fake_flow(prev, nxt, flow, **kw):
        calls.append((prev.shape, kw))
        return np.full(prev.shape + (2,), 1.5, np.float32)

    for (name, k), want in SURVEY_MERGED.items():
        (h, w), slices = maps[name]
        cfg = gating.dataset_config(name, FLAG=2)
        g = gating.current_to_gray(slices[k])
        img = np.zeros((h, w), np.uint8)
        calls.clear()
        flow, cal, vel, regions, rect = gating.opticalFlow3D(g, g, img, img, cfg.MEMSIZE, cfg.MEMSIZE, cfg, fake_flow)
        assert rect == want, (name, k, rect)
        x0, y0, x1, y1 = rect
        assert flow.dtype == np.float64 and flow.shape == (h, w, 2)
        assert np.all(flow[y0:y1, x0:x1] == 1.5) and flow.sum() == 1.5 * 2 * (y1 - y0) * (x1 - x0)
        assert len(calls) == 1 and calls[0][0] == (y1 - y0, x1 - x0)
        assert calls[0][1] == cfg.farneback_params.as_kwargs()
        assert abs(regions[0] - (y1 - y0) * (x1 - x0) / (h * w) * 100) < 1e-9


def test_separate_regions_and_empty_map(nsof_lib, maps):
    from nsof import gating
    (h, w), slices = maps["uavnew2"]
    cfg = gating.dataset_config("uavnew2")   # FLAG 1
    g = gating.current_to_gray(slices[2])
    img = np.zeros((h, w), np.uint8)
    seen = []
    out = gating.opticalFlow3D(g, g, img, img, cfg.MEMSIZE, cfg.MEMSIZE, cfg,
                               lambda p, n, f, **kw: (seen.append(p.shape), np.zeros(p.shape + (2,), np.float32))[1])
    flow, cal, vel, regions, num_labels, info = out
    assert num_labels == 4 and info == [(220, 100, 380, 260), (260, 140, 420, 300), (220, 180, 380, 340)]
    assert seen == [(160, 160)] * 3 and len(regions) == 3
    # no pixel above threshold -> zero flow, regions (0,0,0,0) / []
    (h2, w2), sl2 = maps["autodriving"]
    cfg2 = gating.dataset_config("autodriving")
    g2 = gating.current_to_gray(sl2[16])
    img2 = np.zeros((h2, w2), np.uint8)
    flow2, _, _, regions2, n2, info2 = gating.opticalFlow3D(g2, g2, img2, img2, cfg2.MEMSIZE, cfg2.MEMSIZE, cfg2,
                                                            lambda *a, **k: 1 / 0)
    assert n2 == 1 and regions2 == [] and info2 == [] and not flow2.any()
    cfg3 = gating.dataset_config("autodriving", FLAG=2)
    out3 = gating.opticalFlow3D(g2, g2, img2, img2, cfg3.MEMSIZE, cfg3.MEMSIZE, cfg3, lambda *a, **k: 1 / 0)
    assert out3[4] == (0, 0, 0, 0)


def test_gating_maps_bug_compatibility(nsof_lib, maps):
    from nsof import gating
    (h, w), slices = maps["grasp"]
    stack = np.stack([slices[0], slices[1], slices[2]], -1)
    cfg = gating.dataset_config("grasp")
    m1, m2 = gating.gating_maps(stack, 0, cfg)
    assert np.array_equal(m1, m2)                                   # shipped scripts: memimg2 := memimg1
    cfg.bug_compatible = False
    m1, m2 = gating.gating_maps(stack, 0, cfg)
    assert np.array_equal(m2, gating.current_to_gray(slices[1])) and not np.array_equal(m1, m2)


@pytest.mark.gpu
def test_roi_flow_equals_standalone_crop(nsof_lib, ctx, maps):
    """End to end on the GPU: the flow pasted into the canvas is the flow of the cropped pair."""
    from nsof import gating, synth
    (h, w), slices = maps["uavnew2"]
    prev, nxt = synth.make_pair(77, h, w)
    cfg = gating.dataset_config("uavnew2", FLAG=2)
    g = gating.current_to_gray(slices[1])
    flow_fn = lambda p, n, f, **kw: nsof_lib.calcOpticalFlowFarneback(p, n, f, **kw, ctx=ctx)  # noqa: E731
    flow, _, _, _, (x0, y0, x1, y1) = gating.opticalFlow3D(g, g, prev, nxt, cfg.MEMSIZE, cfg.MEMSIZE, cfg, flow_fn)
    assert (x0, y0, x1, y1) == (180, 100, 420, 340)
    crop = nsof_lib.calcOpticalFlowFarneback(prev[y0:y1, x0:x1].copy(), nxt[y0:y1, x0:x1].copy(), None,
                                             **cfg.farneback_params.as_kwargs(), ctx=ctx)
    assert np.array_equal(flow[y0:y1, x0:x1], crop.astype(np.float64))
    outside = flow.copy()
    outside[y0:y1, x0:x1] = 0
    assert not outside.any()


@pytest.mark.gpu
def test_events_to_roi_to_flow_pipeline(nsof_lib, ctx):
    """Config-3 style chain on a small sensor: a drifting box emits events, the accumulator's surface gates an ROI
    around it, and the flow is computed only there."""
    from nsof import gating, pipeline, synth
    H, W = 240, 320
    x, y, p, t = synth.make_events(7, W, H, n_background=0, duration_us=120_000, box=(40, 30), speed_pps=500.0)
    cfg = gating.GatingConfig(MEMSIZE=20, EXTEND_HEIGHT_UPPER=10, EXTEND_HEIGHT_LOWER=10, EXTEND_WIDTH_LEFT=10,
                              EXTEND_WIDTH_RIGHT=10, THRES=240, FLAG=2, farneback_params=nsof_lib.farneback.PARAMS_A)
    # silent_v = 0.5 V > von: idle devices leak towards Roff (gray 226 < THRES) while event pixels are driven up
    rois = pipeline.events_to_rois(x, y, p, t, (H, W), cfg, slice_us=1000, silent_v=0.5, snapshot_every=40, ctx=ctx)
    assert len(rois) == 3
    g, rects = rois[-1]
    assert g.shape == (H // 20, W // 20) and len(rects) >= 1
    # the box travelled from x=40 to x=100 in rows 105..135: every ROI lies on that band
    for (x0, y0, x1, y1) in rects:
        assert y0 <= 105 and y1 >= 135 and x0 < 110 and x1 > 30 and (x1 - x0) < W
    prev, nxt = synth.make_pair(3, H, W)
    fl = lambda a, b, f, **kw: nsof_lib.calcOpticalFlowFarneback(a, b, f, **kw, ctx=ctx)  # noqa: E731
    flow, _, _, _, rect = pipeline.gated_flow(g, prev, nxt, cfg, flow_fn=fl)
    x0, y0, x1, y1 = rect
    assert flow[y0:y1, x0:x1].any() and not flow[:max(y0 - 1, 0)].any()


def test_segmentation_harness_rows_and_csv(nsof_lib, oracle, tmp_path):
    """run_segmentation = the main loop of optical_flow_seg.py: CSV schema and formatting, accuracies, gating per
    pair -- here with the CPU oracle injected as flow / mask backend (the GPU backends are the defaults)."""
    import csv
    import json
    from conftest import golden_path
    from nsof import pipeline, synth
    g = json.load(open(golden_path("gating_maps.json")))["grasp"]
    stack = np.stack([np.array([[float(v) for v in row] for row in g["slices"][k]]) for k in ("0", "1", "2")], -1)
    hm, wm = stack.shape[:2]
    cfg = nsof_lib.dataset_config("grasp", MEMSIZE=16, EXTEND_HEIGHT_UPPER=4, EXTEND_HEIGHT_LOWER=4,
                                  EXTEND_WIDTH_LEFT=4, EXTEND_WIDTH_RIGHT=4)
    h, w = hm * 16, wm * 16
    frames = []
    for k in range(3):
        a, _ = synth.make_pair(40 + k, h, w)
        frames.append(np.repeat(a[..., None], 3, 2))
    gts = [np.zeros((h, w, 3), np.uint8) for _ in range(3)]
    gts[1][10:40, 20:60] = 255
    far = lambda a, b, _f, **kw: oracle.farneback(a, b, **kw)  # noqa: E731
    rows, acc_mem, acc_orig = pipeline.run_segmentation(frames, gts, stack, cfg, csv_path=str(tmp_path / "r.csv"),
                                                        flow_fn=far, mask_fn=lambda f: oracle.motion_mask(f, 1.0, 10, 5))
    assert len(rows) == 1 and len(rows[0]) == len(pipeline.SEG_CSV_COLUMNS) == 13
    assert rows[0][0] == "2.jpg-1.jpg" and 0 <= acc_mem <= 100 and 0 <= acc_orig <= 100
    with open(tmp_path / "r.csv") as fh:
        got = list(csv.reader(fh))
    assert got[0] == pipeline.SEG_CSV_COLUMNS and got[1][0] == "2.jpg-1.jpg" and len(got) == 2
    for col in (1, 2, 3, 5, 6, 7, 8, 9):
        assert len(got[1][col].split(".")[1]) == 4          # '%.4f'
    assert pipeline.calculate_pixel_accuracy(np.array([[1, 2]]), np.array([[1, 3]])) == 50.0


@pytest.mark.gpu
def test_segmentation_harness_gpu_equals_oracle_backends(nsof_lib, oracle):
    import json
    from conftest import golden_path
    from nsof import pipeline, synth
    g = json.load(open(golden_path("gating_maps.json")))["grasp"]
    stack = np.stack([np.array([[float(v) for v in row] for row in g["slices"][k]]) for k in ("0", "1", "2", "3")], -1)
    hm, wm = stack.shape[:2]
    h, w = hm * 16, wm * 16
    frames = [np.repeat(synth.make_pair(70 + k, h, w)[0][..., None], 3, 2) for k in range(4)]
    gts = [np.zeros((h, w, 3), np.uint8) for _ in range(4)]
    gts[2][30:80, 40:100] = 255
    res = []
    for backend in ("gpu", "oracle"):
        cfg = nsof_lib.dataset_config("grasp", MEMSIZE=16, EXTEND_HEIGHT_UPPER=4, EXTEND_HEIGHT_LOWER=4,
                                      EXTEND_WIDTH_LEFT=4, EXTEND_WIDTH_RIGHT=4)
        kw = {} if backend == "gpu" else dict(flow_fn=lambda a, b, _f, **k: oracle.farneback(a, b, **k),
                                              mask_fn=lambda f: oracle.motion_mask(f, 1.0, 10, 5))
        res.append(pipeline.run_segmentation(frames, gts, stack, cfg, **kw))
    assert len(res[0][0]) == 2
    assert res[0][1:] == res[1][1:]                                   # mean accuracies identical
    assert [r[8:11] for r in res[0][0]] == [r[8:11] for r in res[1][0]]   # per-pair accuracies and region shares


def test_float32_canvas_option(nsof_lib, maps, oracle):
    """cfg.canvas_dtype=float32: same values as the reference's float64 canvas."""
    from nsof import synth
    _, slices = maps["grasp"]
    cur = slices[1]
    hm, wm = cur.shape
    h, w = hm * 8, wm * 8
    a, b = synth.make_pair(5, h, w)
    mem = nsof_lib.current_to_gray(cur)
    far = lambda p, q, _f, **kw: oracle.farneback(p, q, **kw)  # noqa: E731
    out = []
    for dt in (np.float64, np.float32):
        cfg = nsof_lib.dataset_config("grasp", MEMSIZE=8, EXTEND_HEIGHT_UPPER=2, EXTEND_HEIGHT_LOWER=2,
                                      EXTEND_WIDTH_LEFT=2, EXTEND_WIDTH_RIGHT=2, canvas_dtype=dt)
        out.append(nsof_lib.opticalFlow3D(mem, mem, a, b, 8, 8, cfg, flow_fn=far)[0])
    assert out[0].dtype == np.float64 and out[1].dtype == np.float32 and np.array_equal(out[0], out[1])
    assert np.abs(out[0]).max() > 0


@pytest.mark.gpu
def test_device_gating_equals_reference_gating_on_all_stacks(nsof_lib, ctx):
    """nsof_roi_from_surface_dev (one wavefront per gating map: gray map, threshold, connected components by bit-parallel
    flood fill in raster order, scaled / extended rectangles) against the Python mirror of opticalFlow3D's gating
    (optical_flow_seg.py:115-121, 211-252) on EVERY slice of the reference's five constructed_3D_matrix.mat stacks, both
    FLAG settings, connectivity 4 and 8: rectangles equal as integers, in the same order; gray maps byte-identical."""
    import torch
    from nsof import gating
    from nsof import workload as wl
    stacks = np.load(os.path.join(os.path.dirname(__file__), "golden", "gating_stacks.npz"))
    dev = torch.device("cuda", ctx.device)
    n = 0
    for name, (h, w, _) in wl.DATASET_FRAMES.items():
        st = np.ascontiguousarray(np.moveaxis(stacks[name], 2, 0))           # [slices][rows][cols]
        d = torch.from_numpy(st).to(dev)
        torch.cuda.synchronize()
        for flag in (1, 2):
            for conn in (4, 8):
                cfg = gating.dataset_config(name, FLAG=flag, CONNECT=conn)
                counts, rects, gray = gating.roi_from_surface_dev(d, st.shape[0], st.shape[1:], (h, w), cfg, ctx=ctx,
                                                                  want_gray=True)
                got = gating.rects_to_host(counts, rects, ctx=ctx)
                g = gray.cpu().numpy()
                for k in range(st.shape[0]):
                    assert np.array_equal(g[k], gating.current_to_gray(st[k])), (name, k)
                    want = gating.roi_from_surface(st[k], (h, w), cfg)      # host C mirror (itself == Python, test_workload)
                    assert got[k] == want, (name, flag, conn, k, got[k], want)
                    n += len(want)
    assert n > 1000


@pytest.mark.gpu
def test_device_gating_random_maps_up_to_64x64(nsof_lib, ctx):
    """Random maps of every size class up to 64 x 64 cells (snakes, rings, checkerboards included), 4- and 8-connectivity,
    more components than max_rects -> counts report the true number."""
    import torch
    from nsof import gating
    rng = np.random.default_rng(3)
    dev = torch.device("cuda", ctx.device)
    for (rows, cols, dens) in [(1, 1, 1.0), (4, 4, 0.5), (13, 24, 0.3), (32, 32, 0.55), (64, 64, 0.45), (64, 17, 0.6), (5, 64, 0.5)]:
        ms = 7
        h, w = rows * ms + 3, cols * ms + 5
        on = rng.random((6, rows, cols)) < dens
        if rows >= 8 and cols >= 8:          # a snake that needs many flood iterations, and a ring
            on[0] = False
            on[0, ::2, :] = True
            for r in range(1, rows, 2):
                on[0, r, (cols - 1) if (r // 2) % 2 == 0 else 0] = True
            on[1] = False
            on[1, 1:-1, 1] = on[1, 1:-1, -2] = on[1, 1, 1:-1] = on[1, -2, 1:-1] = True
            on[2] = (np.add.outer(np.arange(rows), np.arange(cols)) % 2) == 0
        cur = np.where(on, 1e-5, 1e-9)      # gray 255 vs 68
        d = torch.from_numpy(cur).to(dev)
        torch.cuda.synchronize()
        for flag in (1, 2):
            for conn in (4, 8):
                cfg = gating.GatingConfig(MEMSIZE=ms, THRES=200, FLAG=flag, CONNECT=conn, EXTEND_HEIGHT_UPPER=2,
                                          EXTEND_HEIGHT_LOWER=3, EXTEND_WIDTH_LEFT=1, EXTEND_WIDTH_RIGHT=4)
                want = [gating.roi_from_surface(cur[k], (h, w), cfg) for k in range(6)]
                cap = max(1, max(len(v) for v in want))
                counts, rects = gating.roi_from_surface_dev(d, 6, (rows, cols), (h, w), cfg, max_rects=cap, ctx=ctx)
                assert gating.rects_to_host(counts, rects, ctx=ctx) == want, (rows, cols, flag, conn)
                if cap > 1:                                      # too little room: the count is still the true one
                    c2, r2 = gating.roi_from_surface_dev(d, 6, (rows, cols), (h, w), cfg, max_rects=1, ctx=ctx)
                    ctx.synchronize()
                    assert c2.cpu().numpy().tolist() == [len(v) for v in want]
                    first = r2.cpu().numpy()[:, 0]
                    assert all(tuple(int(q) for q in first[k]) == v[0] for k, v in enumerate(want) if v)


@pytest.mark.gpu
def test_events_to_rois_device_path_equals_host_path(nsof_lib, ctx):
    """pipeline.events_to_rois (accumulator -> block currents -> gating kernel, all in HBM, one small copy at the end)
    returns what the host-side mirror of the reference's gating returns on the downloaded block currents."""
    from nsof import gating, pipeline, synth
    H, W = 240, 320   # noqa: N806
    x, y, p, t = synth.make_events(7, W, H, n_background=3000, duration_us=160_000, box=(40, 30), speed_pps=500.0)
    for flag in (1, 2):
        cfg = gating.GatingConfig(MEMSIZE=20, EXTEND_HEIGHT_UPPER=10, EXTEND_HEIGHT_LOWER=10, EXTEND_WIDTH_LEFT=10,
                                  EXTEND_WIDTH_RIGHT=10, THRES=240, FLAG=flag, farneback_params=nsof_lib.farneback.PARAMS_A)
        a = pipeline.events_to_rois(x, y, p, t, (H, W), cfg, slice_us=1000, silent_v=0.5, snapshot_every=40, ctx=ctx)
        b = pipeline.events_to_rois_host(x, y, p, t, (H, W), cfg, slice_us=1000, silent_v=0.5, snapshot_every=40, ctx=ctx)
        assert len(a) == len(b) == 4
        for (ga, ra), (gb, rb) in zip(a, b):
            assert np.array_equal(ga, gb)
            want = rb if flag == 1 else ([(min(r[0] for r in rb), min(r[1] for r in rb), max(r[2] for r in rb), max(r[3] for r in rb))] if rb else [])
            assert ra == want
        assert any(len(r) for _, r in a)
        # a table too small for a map's components is grown by one more (tiny) gating launch, not an error
        a1 = pipeline.events_to_rois(x, y, p, t, (H, W), cfg, slice_us=1000, silent_v=0.5, snapshot_every=40, ctx=ctx, max_rects=1)
        assert [r for _, r in a1] == [r for _, r in a]
    # a gating grid beyond the kernel's 64 x 64 cells (MEMSIZE 4 -> 60 x 80) takes the host mirror, same contract
    for flag in (1, 2):
        cfg = gating.GatingConfig(MEMSIZE=4, EXTEND_HEIGHT_UPPER=2, EXTEND_HEIGHT_LOWER=2, EXTEND_WIDTH_LEFT=2,
                                  EXTEND_WIDTH_RIGHT=2, THRES=240, FLAG=flag, farneback_params=nsof_lib.farneback.PARAMS_A)
        a = pipeline.events_to_rois(x, y, p, t, (H, W), cfg, slice_us=1000, silent_v=0.5, snapshot_every=40, ctx=ctx)
        b = pipeline.events_to_rois_host(x, y, p, t, (H, W), cfg, slice_us=1000, silent_v=0.5, snapshot_every=40, ctx=ctx)
        assert len(a) == len(b) == 4 and a[0][0].shape == (60, 80)
        for (ga, ra), (gb, rb) in zip(a, b):
            assert np.array_equal(ga, gb)
            assert ra == (rb if flag == 1 else ([(min(r[0] for r in rb), min(r[1] for r in rb), max(r[2] for r in rb), max(r[3] for r in rb))] if rb else []))


@pytest.mark.gpu
def test_config3_roi_flow_pipeline_vs_oracle_chain(nsof_lib, ctx, oracle):
    """BASELINE config 3 joined on the device (pipeline.events_to_roi_flows: events -> surface frames + gating maps ->
    device ROI rectangles -> flow of every ROI crop as one work list, pasted into zero canvases) against the oracle
    CHAIN: oracle/accum_ref.c for the slices -> uint8(255 w) frames and V_ds / resistance block maxima -> the host mirror
    of the reference's gating (current -> gray, threshold, components, boxes) -> oracle/farneback_ref.c on each crop,
    pasted in label order.  Frames byte-identical, rectangles identical, flow canvases bit-identical; both FLAG modes."""
    from nsof import gating, pipeline, synth
    from nsof.farneback import PARAMS_B
    H, W, every = 240, 320, 40   # noqa: N806
    x, y, p, t = synth.make_events(7, W, H, n_background=3000, duration_us=160_000, box=(40, 30), speed_pps=500.0)
    pb = [getattr(PARAMS_B, k) for k in ("pyr_scale", "levels", "winsize", "iterations", "poly_n", "poly_sigma", "flags")]
    ref_frames, ref_cur = [], []
    for k in range(4):
        _, w = oracle.accum_slices_per_s(x, y, t, H, W, 1000, -6.0, 0.5, n_slices=(k + 1) * every, n_threads=4)
        ref_frames.append((np.float32(255.0) * w).astype(np.uint8))
        ref_cur.append(pipeline.surface_to_block_current(oracle.accum_resistance(w), 20))
    some_overlap = False
    for flag, compat in ((1, True), (2, True), (1, False), (2, False)):
        # which map gates pair (k, k+1): the scripts' (memimg2 := memimg1 -> frame k) or opticalFlow3D's as written (frame k+1)
        cfg = gating.GatingConfig(MEMSIZE=20, EXTEND_HEIGHT_UPPER=10, EXTEND_HEIGHT_LOWER=10, EXTEND_WIDTH_LEFT=10,
                                  EXTEND_WIDTH_RIGHT=10, THRES=240, FLAG=flag, farneback_params=PARAMS_B, bug_compatible=compat)
        gi = 0 if compat else 1
        tm = {}
        frames, rects, flows = pipeline.events_to_roi_flows(x, y, p, t, (H, W), cfg, slice_us=1000, silent_v=0.5,
                                                            snapshot_every=every, ctx=ctx, timings=tm)
        gf, gfl = frames.cpu().numpy(), flows.cpu().numpy()
        assert gf.shape == (4, H, W) and gfl.shape == (3, H, W, 2) and tm["roi_calls"] >= 1
        for k in range(4):
            assert np.array_equal(gf[k], ref_frames[k]), k
            g = gating.current_to_gray(ref_cur[k])
            tp = gating.update_transition_pic(g, np.zeros_like(g, dtype=np.float64), cfg.THRES).astype(np.uint8)
            n, _, stats, _ = gating.connectedComponentsWithStats(tp, cfg.CONNECT)
            want = [gating._roi(*[int(v) for v in stats[i, :4]], W, H, 20, 20, cfg) for i in range(1, n)]
            if flag == 2 and want:
                want = [(min(r[0] for r in want), min(r[1] for r in want), max(r[2] for r in want), max(r[3] for r in want))]
            assert rects[k] == want, (flag, k, rects[k], want)
        for k in range(3):
            canvas = np.zeros((H, W, 2), np.float32)
            done = []
            for (x0, y0, x1, y1) in rects[k + gi]:
                canvas[y0:y1, x0:x1] = oracle.farneback(np.ascontiguousarray(ref_frames[k][y0:y1, x0:x1]),
                                                        np.ascontiguousarray(ref_frames[k + 1][y0:y1, x0:x1]), *pb)
                some_overlap |= any(x0 < b[2] and b[0] < x1 and y0 < b[3] and b[1] < y1 for b in done)
                done.append((x0, y0, x1, y1))
            assert np.array_equal(gfl[k], canvas), (flag, compat, k, float(np.abs(gfl[k] - canvas).max()))
        assert any(len(r) for r in rects[1:])


@pytest.mark.gpu
def test_roi_sequence_call_empty_tables_and_errors(nsof_lib, ctx):
    """nsof_farneback_u8_roi_sequence_dev: no rectangles -> zeroed canvases and (0, 0); a count beyond the table or a
    rectangle that leaves the frame -> NSOF_EINVAL with a message, nothing launched."""
    import torch
    from nsof import synth
    dev = torch.device("cuda", ctx.device)
    a, b = synth.make_pair(5, 200, 300)
    frames = torch.from_numpy(np.stack([a, b, a])).to(dev)
    counts = torch.zeros(3, dtype=torch.int32, device=dev)
    rects = torch.zeros((3, 4, 4), dtype=torch.int32, device=dev)
    flows = torch.full((2, 200, 300, 2), 7.0, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    assert nsof_lib.farneback_roi_sequence_dev(frames, counts, rects, flows, nsof_lib.farneback.PARAMS_A, gate_frame=1, ctx=ctx) == (0, 0)
    ctx.synchronize()
    assert float(flows.abs().max().item()) == 0.0
    counts[1] = 9
    with pytest.raises(nsof_lib.NsofError, match="rectangles"):
        nsof_lib.farneback_roi_sequence_dev(frames, counts, rects, flows, nsof_lib.farneback.PARAMS_A, gate_frame=1, ctx=ctx)
    counts[1] = 1
    rects[1, 0] = torch.tensor([10, 10, 400, 100], dtype=torch.int32)
    with pytest.raises(nsof_lib.NsofError, match="leaves"):
        nsof_lib.farneback_roi_sequence_dev(frames, counts, rects, flows, nsof_lib.farneback.PARAMS_A, gate_frame=1, ctx=ctx)
    # an overlapping pair of rectangles is pasted in order: the second one wins where they overlap
    counts[1] = 2
    rects[1, 0] = torch.tensor([20, 30, 180, 150], dtype=torch.int32)
    rects[1, 1] = torch.tensor([100, 60, 280, 190], dtype=torch.int32)
    n, px = nsof_lib.farneback_roi_sequence_dev(frames, counts, rects, flows, nsof_lib.farneback.PARAMS_A, gate_frame=1, ctx=ctx)
    ctx.synchronize()
    assert (n, px) == (2, 160 * 120 + 180 * 130)
    got = flows[0].cpu().numpy()
    second = nsof_lib.calcOpticalFlowFarneback(np.ascontiguousarray(a[60:190, 100:280]), np.ascontiguousarray(b[60:190, 100:280]),
                                               None, **nsof_lib.farneback.PARAMS_A.as_kwargs(), ctx=ctx)
    first = nsof_lib.calcOpticalFlowFarneback(np.ascontiguousarray(a[30:150, 20:180]), np.ascontiguousarray(b[30:150, 20:180]),
                                              None, **nsof_lib.farneback.PARAMS_A.as_kwargs(), ctx=ctx)
    assert np.array_equal(got[60:190, 100:280], second) and np.array_equal(got[30:60, 20:180], first[:30])
    assert not got[:30].any() and not got[190:].any() and float(np.abs(flows[1].cpu().numpy()).max()) == 0.0
