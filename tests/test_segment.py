"""Motion-segmentation head: oracle vs the reference's recorded cv2 mask, HIP path vs oracle.

Pin: the "Final Motion Segmentation" panel of the reference's demo.ipynb (tests/golden/demo/panel_seg_mask.png) is
the cv2 result of gating -> ROI Farneback -> negate -> mag > 1 -> 5 x (dilate, erode) with the 10x10 ellipse.  The
panel is the 1080x1920 mask drawn at 247x438 px, so the comparison is an overlap measure (IoU of the >127 sets after
the same down-sampling), not bits: required > 0.99, measured 0.997; the textbook closing (element reflected in the
dilation), which cv2 does not do, gives 0.911 and must fail.
HIP vs oracle is bit-exact (two-valued masks).
"""
import numpy as np
import pytest

import test_demo_pin as demo

ELLIPSE_10 = ["0000010000", "0011111110", "0111111111", "1111111111", "1111111111", "1111111111", "1111111111",
              "1111111111", "0111111111", "0011111110"]


def test_structuring_elements(oracle):
    assert ["".join(map(str, r)) for r in oracle.structuring_element(2, 10, 10)] == ELLIPSE_10
    # well-known cv2 results
    assert oracle.structuring_element(2, 5, 5).tolist() == [[0, 0, 1, 0, 0]] + [[1] * 5] * 3 + [[0, 0, 1, 0, 0]]
    assert oracle.structuring_element(2, 3, 3).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    assert oracle.structuring_element(1, 3, 3).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    assert oracle.structuring_element(0, 4, 2).tolist() == [[1] * 4] * 2


def test_abi_structuring_element_matches_oracle(oracle, nsof_lib):
    for shape in (0, 1, 2):
        for kw, kh in ((1, 1), (3, 3), (5, 3), (10, 10), (7, 12), (31, 31), (32, 32)):
            assert np.array_equal(nsof_lib.getStructuringElement(shape, (kw, kh)),
                                  oracle.structuring_element(shape, kw, kh)), (shape, kw, kh)


def test_oracle_morph_basics(oracle):
    img = np.zeros((9, 11), np.uint8)
    img[4, 5] = 255
    k = oracle.structuring_element(0, 3, 2)           # even height: anchor row 1 -> offsets -1..0
    d = oracle.morph(1, img, k)
    ys, xs = np.nonzero(d)
    assert (ys.min(), ys.max(), xs.min(), xs.max()) == (4, 5, 4, 6)     # dst(y) = src(y + i - 1), i in {0, 1}
    assert np.array_equal(oracle.morph(0, np.full((5, 5), 255, np.uint8), k), np.full((5, 5), 255, np.uint8))


def _iou(mask, panel):
    small = np.asarray(demo.PIL.fromarray(mask).resize((247, 438), demo.PIL.BILINEAR)) > 127
    ref = panel > 127
    return (small & ref).sum() / (small | ref).sum()


def _demo_mask(flow_fn, mask_fn):
    flow, rect = demo._roi_flow(flow_fn)
    x0, y0, x1, y1 = rect
    full = np.zeros(flow.shape[:2], np.uint8)
    full[y0:y1, x0:x1] = mask_fn(-flow[y0:y1, x0:x1])
    return full


def test_oracle_mask_matches_cv2_panel(oracle):
    panel = np.asarray(demo.PIL.open(demo.os.path.join(demo.DEMO, "panel_seg_mask.png")).convert("L"))
    far = lambda a, b, _f, **kw: oracle.farneback(a, b, **kw)  # noqa: E731
    good = _demo_mask(far, lambda f: oracle.motion_mask(f, 1.0, 10, 5))
    assert _iou(good, panel) > 0.99

    def reflected(f):   # proper closing: dilate with the reflected element
        m = np.where(np.hypot(f[..., 0].astype(np.float64), f[..., 1]) > 1, 255, 0).astype(np.uint8)
        k = oracle.structuring_element(2, 10, 10)
        for _ in range(5):
            m = oracle.morph(0, oracle.morph(1, m, k[::-1, ::-1].copy(), anchor=(4, 4)), k)
        return m
    assert _iou(_demo_mask(far, reflected), panel) < 0.95


# ------------------------------------------------------------------------------------------------ GPU
def _rand_mask(rng, h, w, density):
    return np.where(rng.random((h, w)) < density, 255, 0).astype(np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,kw,kh", [(0, 3, 3), (1, 5, 5), (2, 10, 10), (2, 7, 3), (0, 1, 1), (2, 31, 31),
                                          (0, 32, 2), (0, 2, 32)])
def test_gpu_morph_matches_oracle(oracle, nsof_lib, shape, kw, kh):
    rng = np.random.default_rng(kw * 100 + kh)
    k = oracle.structuring_element(shape, kw, kh)
    for h, w in ((1, 1), (7, 70), (64, 256), (65, 257), (130, 333), (200, 31)):
        for density in (0.02, 0.5, 0.98):
            img = _rand_mask(rng, h, w, density)
            for op, fn in ((1, nsof_lib.dilate), (0, nsof_lib.erode)):
                assert np.array_equal(fn(img, k), oracle.morph(op, img, k)), (shape, kw, kh, h, w, density, op)


@pytest.mark.gpu
def test_gpu_morph_anchor_iterations_and_values(oracle, nsof_lib):
    rng = np.random.default_rng(5)
    img = _rand_mask(rng, 90, 150, 0.1)
    k = (rng.random((5, 9)) < 0.5).astype(np.uint8)
    k[2, 4] = 1
    for anchor in ((-1, -1), (0, 0), (8, 4), (3, 1)):
        for op, fn in ((1, nsof_lib.dilate), (0, nsof_lib.erode)):
            want = img
            for _ in range(3):
                want = oracle.morph(op, want, k, anchor=anchor)
            assert np.array_equal(fn(img, k, anchor=anchor, iterations=3), want), (anchor, op)
    # long chains are split into several launches when the halo would exceed the tile
    k3 = oracle.structuring_element(0, 15, 15)
    want = img
    for _ in range(12):
        want = oracle.morph(1, want, k3)
    assert np.array_equal(nsof_lib.dilate(img, k3, iterations=12), want)
    # non-zero values other than 255 count as set
    g = (img // 255 * 7).astype(np.uint8)
    assert np.array_equal(nsof_lib.dilate(g, k), oracle.morph(1, img, k))


@pytest.mark.gpu
def test_gpu_morph_rejects_bad_arguments(nsof_lib):
    img = np.zeros((8, 8), np.uint8)
    with pytest.raises(nsof_lib.NsofError):
        nsof_lib.dilate(img, np.ones((33, 3), np.uint8))
    with pytest.raises(nsof_lib.NsofError):
        nsof_lib.dilate(img, np.ones((3, 3), np.uint8), anchor=(3, 0))
    with pytest.raises(nsof_lib.NsofError):
        nsof_lib.dilate(img.astype(np.float32), np.ones((3, 3), np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", [(1, 1), (37, 53), (128, 256), (270, 481), (1080, 1920)])
def test_gpu_motion_mask_matches_oracle(oracle, nsof_lib, h, w):
    rng = np.random.default_rng(h + w)
    # smooth field crossing the threshold in blobs, plus speckle, plus exact ties at |flow| == 1
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    flow = np.stack([1.3 * np.sin(xx / 17.0) * np.cos(yy / 23.0), 0.9 * np.cos(xx / 11.0 + yy / 29.0)], -1)
    flow += (rng.random((h, w, 2)) < 0.002) * 3.0
    flow = flow.astype(np.float32)
    flow[0, 0] = (1.0, 0.0)
    flow[-1, -1] = (0.6, 0.8)
    for iters, ksize in ((5, 10), (0, 10), (1, 3), (2, 5)):
        got = nsof_lib.motion_mask(flow, 1, ksize, iters)
        assert np.array_equal(got, oracle.motion_mask(flow, 1.0, ksize, iters)), (iters, ksize)


@pytest.mark.gpu
def test_gpu_motion_mask_on_strided_crop_and_f64_canvas(oracle, nsof_lib):
    rng = np.random.default_rng(3)
    canvas = (rng.standard_normal((300, 400, 2)) * 0.8).astype(np.float32)
    crop = canvas[40:260, 33:350]
    want = oracle.motion_mask(np.ascontiguousarray(crop), 1.0, 10, 5)
    assert np.array_equal(nsof_lib.motion_mask(crop), want)
    assert np.array_equal(nsof_lib.motion_mask(-crop.astype(np.float64)), want)
    out = np.zeros((300, 400), np.uint8)
    nsof_lib.motion_mask(crop, out=out[40:260, 33:350])
    assert np.array_equal(out[40:260, 33:350], want) and out[:40].sum() == 0 and out[:, :33].sum() == 0
    # process_flow_region / task_results mirrors
    mag = np.hypot(crop[..., 0].astype(np.float64), crop[..., 1])
    assert np.array_equal(nsof_lib.process_flow_region(mag, None), want)
    tr = nsof_lib.task_results(np.zeros((300, 400, 3), np.uint8), None, canvas, 2, (33, 40, 350, 260))
    assert np.array_equal(tr[40:260, 33:350], want) and tr.sum() == want.sum()
    assert nsof_lib.task_results(np.zeros((300, 400, 3), np.uint8), None, canvas, 1, (0, 0, 0, 0)).sum() == 0


@pytest.mark.gpu
def test_gpu_mask_matches_cv2_panel(nsof_lib):
    panel = np.asarray(demo.PIL.open(demo.os.path.join(demo.DEMO, "panel_seg_mask.png")).convert("L"))
    got = _demo_mask(nsof_lib.calcOpticalFlowFarneback, nsof_lib.motion_mask)
    assert _iou(got, panel) > 0.99
