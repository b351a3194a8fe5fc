"""CPU: analytic pins of the Farneback oracle (oracle/farneback_ref.c).

PARITY UNPINNED against cv2: the reference holds no golden flow and cv2 is not installable here, so these
tests pin the restatement by properties that any faithful implementation of the algorithm must satisfy."""
import numpy as np
import pytest

A = (0.5, 3, 15, 3, 5, 1.2, 0)
B = (0.6, 3, 3, 3, 10, 1.05, 0)
Cc = (0.6, 3, 4, 2, 1, 1.05, 0)


def test_gaussian_kernel_values(oracle):
    assert np.array_equal(oracle.gaussian_kernel(3, 0.0), np.float32([0.25, 0.5, 0.25]))
    e = np.exp(-2.0)
    assert np.allclose(oracle.gaussian_kernel(3, 0.5), np.float32([e, 1, e]) / np.float32(1 + 2 * e), rtol=1e-7)
    for n, s in [(5, 0.888888), (9, 1.5), (19, 3.5)]:
        k = oracle.gaussian_kernel(n, s)
        x = np.arange(n) - (n - 1) / 2
        ref = np.exp(-x * x / (2 * s * s))
        assert np.allclose(k, ref / ref.sum(), rtol=2e-7) and abs(k.sum() - 1) < 1e-6 and np.array_equal(k, k[::-1])


def test_level_geometry(oracle):
    # 1080p, pyr_scale .5: 960x540, 480x270, 240x135; blur sizes 3,3,9,19 (cvRound is round-half-even: 2.5 -> 2)
    got = [oracle.level_geometry(1920, 1080, 0.5, k)[:3] for k in range(4)]
    assert got == [(1920, 1080, 3), (960, 540, 3), (480, 270, 9), (240, 135, 19)]
    got = [oracle.level_geometry(801, 801, 0.6, k)[:3] for k in range(4)]
    assert got == [(801, 801, 3), (481, 481, 3), (288, 288, 5), (173, 173, 9)]
    assert oracle.effective_levels(1920, 1080, 0.5, 3) == 3
    assert oracle.effective_levels(520, 200, 0.5, 3) == 2      # 200 * .125 = 25 < 32
    assert oracle.effective_levels(40, 33, 0.5, 3) == 0


def test_resize_linear_properties(oracle):
    rng = np.random.default_rng(1)
    a = rng.random((37, 53)).astype(np.float32)
    assert np.array_equal(oracle.resize_linear(a, 53, 37), a)            # same size: plain copy
    c = np.full((20, 30), 3.25, np.float32)
    assert np.allclose(oracle.resize_linear(c, 61, 41), 3.25, rtol=1e-6)  # constants survive
    # exact 2x decimation: each output is the mean of a 2x2 block
    d = oracle.resize_linear(a[:36, :52], 26, 18)
    ref = 0.25 * (a[0:36:2, 0:52:2] + a[1:36:2, 0:52:2] + a[0:36:2, 1:52:2] + a[1:36:2, 1:52:2])
    assert np.allclose(d, ref, rtol=1e-6)
    f = rng.random((9, 11, 2)).astype(np.float32)
    up = oracle.resize_linear(f, 22, 18)
    assert up.shape == (18, 22, 2) and up.min() >= f.min() - 1e-6 and up.max() <= f.max() + 1e-6


@pytest.mark.parametrize("n,sigma", [(5, 1.2), (10, 1.05), (1, 1.05), (7, 1.5)])
def test_polyexp_reproduces_quadratic(oracle, n, sigma):
    """A quadratic image is its own polynomial expansion: R = (f_y, f_x, f_yy/2, f_xx/2, f_xy) locally."""
    h, w = 48, 64
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    a0, a1, a2, a3, a4, a5 = 10.0, 0.7, -0.4, 0.013, -0.021, 0.009
    img = (a0 + a1 * xx + a2 * yy + a3 * xx * xx + a4 * yy * yy + a5 * xx * yy).astype(np.float32)
    R = oracle.polyexp(img, n, sigma)[n:-n, n:-n]
    yy, xx = yy[n:-n, n:-n], xx[n:-n, n:-n]
    want = [a2 + 2 * a4 * yy + a5 * xx, a1 + 2 * a3 * xx + a5 * yy, a4 + 0 * xx, a3 + 0 * xx, a5 + 0 * xx]
    for c in range(5):
        assert np.abs(R[..., c] - want[c]).max() < 2e-3, c


def test_update_matrices_zero_displacement_identity(oracle):
    """flow = 0 and R1 == R0 in the interior: h = 0 and G = (r4^2+r6^2, (r4+r5) r6, r5^2+r6^2) of R0 itself."""
    rng = np.random.default_rng(2)
    img = (rng.random((40, 50)) * 255).astype(np.float32)
    R = oracle.polyexp(img, 5, 1.2)
    M = oracle.update_matrices(R, R, np.zeros((40, 50, 2), np.float32))
    inner = M[5:-5, 5:-5]
    r4, r5, r6 = R[5:-5, 5:-5, 2], R[5:-5, 5:-5, 3], R[5:-5, 5:-5, 4] * np.float32(0.5)
    assert np.all(inner[..., 3] == 0) and np.all(inner[..., 4] == 0)
    assert np.allclose(inner[..., 0], r4 * r4 + r6 * r6, rtol=1e-6)
    assert np.allclose(inner[..., 1], (r4 + r5) * r6, rtol=1e-5, atol=1e-6)


def test_blur_solve_matches_float64_box_filter(oracle):
    rng = np.random.default_rng(3)
    h, w, ws = 31, 45, 7
    M = rng.random((h, w, 5)).astype(np.float32) + np.float32([2, 0, 2, 0, 0])
    R = np.zeros((h, w, 5), np.float32)
    flow, _ = oracle.update_flow_blur(R, R, np.zeros((h, w, 2), np.float32), M, ws, False)
    m = ws // 2
    pad = np.pad(M.astype(np.float64), ((m, m), (m, m), (0, 0)), mode="edge")
    box = sum(pad[i:i + h, j:j + w] for i in range(ws) for j in range(ws)) / (ws * ws)
    g11, g12, g22, h1, h2 = (box[..., i] for i in range(5))
    idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3)
    assert np.allclose(flow[..., 0], (g11 * h2 - g12 * h1) * idet, rtol=2e-5, atol=1e-6)
    assert np.allclose(flow[..., 1], (g22 * h1 - g12 * h2) * idet, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("params,tol", [(A, 0.04), (B, 0.2), (Cc, 0.15)], ids=["A", "B", "C"])
def test_recovers_known_motion(oracle, nsof_lib, params, tol):
    from nsof import synth
    prev, nxt = synth.make_pair(1234, 270, 480)
    flow = oracle.farneback(prev, nxt, *params)
    tf = synth.true_flow(270, 480)
    assert flow.shape == (270, 480, 2) and flow.dtype == np.float32
    err = np.abs(flow - tf)[40:-40, 40:-40]
    assert err.mean() < tol


def test_identical_frames_interior_is_exactly_zero(oracle, nsof_lib):
    """Away from the right/bottom border (where the out-of-image branch of the matrix update injects a
    non-zero h) identical frames give exactly zero flow."""
    from nsof import synth
    prev, _ = synth.make_pair(5, 200, 300)
    for params in (B, Cc):
        z = oracle.farneback(prev, prev, *params)
        assert np.abs(z[:100, :150]).max() == 0.0


def test_strided_view_equals_copy_and_inputs_untouched(oracle, nsof_lib):
    from nsof import synth
    prev, nxt = synth.make_pair(6, 160, 240)
    keep = prev.copy()
    pv, nv = prev[20:140, 30:200], nxt[20:140, 30:200]
    a = oracle.farneback(pv, nv, *A)
    b = oracle.farneback(pv.copy(), nv.copy(), *A)
    assert np.array_equal(a, b) and np.array_equal(prev, keep)


def test_levels_are_truncated_for_small_images(oracle, nsof_lib):
    from nsof import synth
    prev, nxt = synth.make_pair(8, 40, 50)
    a = oracle.farneback(prev, nxt, 0.5, 3, 15, 3, 5, 1.2, 0)
    b = oracle.farneback(prev, nxt, 0.5, 0, 15, 3, 5, 1.2, 0)   # 50 * .5 = 25 < 32 -> no coarser level
    assert np.array_equal(a, b)


def test_flags_are_rejected(oracle, nsof_lib):
    from nsof import synth
    prev, nxt = synth.make_pair(8, 40, 50)
    with pytest.raises(RuntimeError):
        oracle.farneback(prev, nxt, 0.5, 3, 15, 3, 5, 1.2, 256)
