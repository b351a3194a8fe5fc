"""GPU parity: HIP Farneback (through the C ABI) vs the CPU oracle on identical inputs.

Tolerances.  north_star asks for max-abs 1e-4 against cv2; the oracle restates cv2's arithmetic
(parity unpinned: cv2 is not available, see oracle/farneback_ref.c).  The HIP path follows the same
operation order, so the stage tests demand bit-exact results wherever the order is identical
(pyramid level, polynomial expansion, matrix update, flow resample) and <= 2 float ulps where only the
double-precision row-sum order differs (blur + solve).  Whole-pipeline tolerance: 1e-5 px.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import ulp_diff

pytestmark = pytest.mark.gpu

A = (0.5, 3, 15, 3, 5, 1.2, 0)
B = (0.6, 3, 3, 3, 10, 1.05, 0)
Cc = (0.6, 3, 4, 2, 1, 1.05, 0)
PIPE_TOL = 1e-5


def _dev(torch_dev, a):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).to(torch_dev)
    torch.cuda.synchronize()  # libnsof launches on its own stream
    return t


def _planar(aos):  # (h,w,5) -> (5,h,w)   (layout of M)
    return np.ascontiguousarray(np.moveaxis(aos, -1, 0))


def _rlayout(aos):
    """Oracle R (h,w,5) -> device layout of one image: [h][w][4] (channels 0..3) then [h][w] (channel 4), flat."""
    return np.concatenate([np.ascontiguousarray(aos[..., :4]).ravel(), np.ascontiguousarray(aos[..., 4]).ravel()])


@pytest.fixture(scope="module")
def frames(nsof_lib):
    from nsof import synth
    return {(h, w): synth.make_pair(100 + h + w, h, w) for (h, w) in [(135, 240), (200, 303), (97, 131), (270, 480)]}


@pytest.mark.parametrize("pyr_scale,level", [(0.5, 0), (0.5, 1), (0.5, 2), (0.5, 3), (0.6, 1), (0.6, 2), (0.6, 3),
                                             (0.75, 2)])
@pytest.mark.parametrize("shape", [(270, 480), (200, 303)])
def test_pyr_level_bit_exact(ctx, oracle, torch_dev, frames, pyr_scale, level, shape):
    import torch
    img = frames[shape][0]
    h, w = shape
    want = oracle.pyr_level(img, pyr_scale, level)
    hk, wk = want.shape
    # two images in one launch, the second with a padded row stride
    pitch = w + 13
    buf = np.zeros((2, h, pitch), np.uint8)
    buf[0, :, :w] = img
    buf[1, :, :w] = frames[shape][1]
    d = _dev(torch_dev, buf)
    out = torch.empty((2, hk, wk), dtype=torch.float32, device=torch_dev)
    ctx.check(ctx._lib.nsof_stage_pyr_level(ctx.ptr, 2, d.data_ptr(), pitch, h * pitch, w, h, pyr_scale, level,
                                            out.data_ptr()))
    ctx.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got[0], want)
    assert np.array_equal(got[1], oracle.pyr_level(frames[shape][1], pyr_scale, level))


@pytest.mark.parametrize("level", [1, 2, 3])
@pytest.mark.parametrize("shape", [(272, 480), (24, 16), (184, 1040), (1080, 1920), (1920, 1080), (200, 72),
                                   (96, 64)])
def test_pyr_level_exact_decimation_path(ctx, oracle, torch_dev, level, shape):
    """Dense, 16-byte aligned frames whose size divides by 2^level take the decimating walker kernel
    (k_prep_decim); same bits as the oracle, and as the generic kernels (NSOF_PREP_NODECIM)."""
    import torch
    h, w = shape
    if h % (1 << level) or w % (1 << level) or h <= (3, 9, 19)[level - 1]:
        pytest.skip("shape does not decimate exactly at this level")
    rng = np.random.default_rng(h * 7 + w + level)
    buf = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
    buf[1] = (np.add.outer(np.arange(h), np.arange(w)) % 256).astype(np.uint8)      # ramps: reflect errors show up
    want = [oracle.pyr_level(buf[i], 0.5, level) for i in range(3)]
    hk, wk = want[0].shape
    assert (hk, wk) == (h >> level, w >> level)
    d = _dev(torch_dev, buf)
    out = torch.empty((3, hk, wk), dtype=torch.float32, device=torch_dev)
    ctx.check(ctx._lib.nsof_stage_pyr_level(ctx.ptr, 3, d.data_ptr(), w, h * w, w, h, 0.5, level, out.data_ptr()))
    ctx.synchronize()
    got = out.cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], want[i]), i


@pytest.mark.parametrize("n,sigma", [(1, 1.05), (2, 0.9), (3, 1.1), (5, 1.2), (7, 1.5), (10, 1.05), (4, 0.0)])
@pytest.mark.parametrize("shape", [(135, 240), (97, 131), (33, 517)])
def test_polyexp_bit_exact(ctx, oracle, torch_dev, n, sigma, shape):
    import torch
    rng = np.random.default_rng(n * 1000 + shape[0])
    imgs = (rng.random((3,) + shape) * 255).astype(np.float32)
    d = _dev(torch_dev, imgs)
    out = torch.empty((3, 5 * shape[0] * shape[1]), dtype=torch.float32, device=torch_dev)
    ctx.check(ctx._lib.nsof_stage_polyexp(ctx.ptr, 3, d.data_ptr(), shape[1], shape[0], n, sigma, out.data_ptr()))
    ctx.synchronize()
    got = out.cpu().numpy()
    for i in range(3):
        want = _rlayout(oracle.polyexp(imgs[i], n, sigma))
        assert np.array_equal(got[i], want), f"image {i}: max ulp {ulp_diff(got[i], want).max()}"


def _level_state(oracle, prev, nxt, n, sigma, seed):
    I0 = oracle.pyr_level(prev, 0.5, 0)
    I1 = oracle.pyr_level(nxt, 0.5, 0)
    R0, R1 = oracle.polyexp(I0, n, sigma), oracle.polyexp(I1, n, sigma)
    rng = np.random.default_rng(seed)
    flow = (rng.standard_normal(I0.shape + (2,)) * 3).astype(np.float32)
    flow[:5, :7] += 40  # force out-of-image samples
    return R0, R1, flow


@pytest.mark.parametrize("shape", [(135, 240), (97, 131)])
def test_update_matrices_bit_exact(ctx, oracle, torch_dev, frames, shape):
    import torch
    h, w = shape
    prev, nxt = frames[shape]
    R0, R1, flow = _level_state(oracle, prev, nxt, 5, 1.2, 5)
    want = _planar(oracle.update_matrices(R0, R1, flow))
    Rp = np.stack([np.stack([_rlayout(R0), _rlayout(R1)])] * 2)  # 2 pairs
    flows = np.stack([flow, flow])
    dR, dF = _dev(torch_dev, Rp), _dev(torch_dev, flows)
    out = torch.empty((2, 5, h, w), dtype=torch.float32, device=torch_dev)
    ctx.check(ctx._lib.nsof_stage_update_matrices(ctx.ptr, 2, dR.data_ptr(), dF.data_ptr(), w, h, out.data_ptr()))
    ctx.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want)


@pytest.mark.parametrize("winsize", [15, 3, 4, 2, 31])
@pytest.mark.parametrize("shape", [(135, 240), (97, 531)])
def test_blur_solve(ctx, oracle, torch_dev, winsize, shape):
    import torch
    from nsof import synth
    h, w = shape
    prev, nxt = synth.make_pair(9, h, w)
    R0, R1, flow = _level_state(oracle, prev, nxt, 5, 1.2, 6)
    M = oracle.update_matrices(R0, R1, flow)
    want, _ = oracle.update_flow_blur(R0, R1, flow, M, winsize, False)
    dM = _dev(torch_dev, _planar(M)[None])
    out = torch.empty((1, h, w, 2), dtype=torch.float32, device=torch_dev)
    ctx.check(ctx._lib.nsof_stage_blur_solve(ctx.ptr, 1, dM.data_ptr(), w, h, winsize, out.data_ptr()))
    ctx.synchronize()
    got = out.cpu().numpy()[0]
    # column sums are order-identical; only the double row-sum order differs -> a few float ulps at most
    d = np.abs(got - want)
    assert d.max() <= 1e-6 * max(1.0, np.abs(want).max()), d.max()
    assert (got != want).mean() < 0.01


@pytest.mark.parametrize("winsize", [15, 3, 4, 2, 9])
@pytest.mark.parametrize("shape", [(135, 240), (97, 531), (20, 40)])
def test_fused_iteration(ctx, oracle, torch_dev, winsize, shape):
    """k_iterate == update_matrices followed by blur+solve (the unfused oracle stages)."""
    import torch
    from nsof import synth
    h, w = shape
    prev, nxt = synth.make_pair(9, h, w)
    R0, R1, flow = _level_state(oracle, prev, nxt, 5, 1.2, 6)
    M = oracle.update_matrices(R0, R1, flow)
    want, _ = oracle.update_flow_blur(R0, R1, flow, M, winsize, False)
    Rp = np.stack([np.stack([_rlayout(R0), _rlayout(R1)])] * 2)
    dR, dF = _dev(torch_dev, Rp), _dev(torch_dev, np.stack([flow, flow]))
    out = torch.zeros((2, h, w, 2), dtype=torch.float32, device=torch_dev)
    torch.cuda.synchronize()
    ctx.check(ctx._lib.nsof_stage_iterate(ctx.ptr, 2, dR.data_ptr(), dF.data_ptr(), w, h, winsize, out.data_ptr()))
    ctx.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got[0], got[1])
    d = np.abs(got[0] - want)
    assert d.max() <= 1e-6 * max(1.0, np.abs(want).max()), d.max()
    assert (got[0] != want).mean() < 0.01


@pytest.mark.parametrize("pyr_scale,src,dst", [(0.5, (68, 120), (135, 240)), (0.6, (58, 79), (97, 131)),
                                               (0.5, (135, 240), (270, 480)), (0.6, (97, 173), (161, 288)),
                                               (0.5, (100, 129), (200, 257)), (0.75, (120, 300), (160, 400)),
                                               (0.5, (33, 130), (67, 259)), (0.5, (540, 960), (1080, 1920))])
def test_flow_upsample_bit_exact(ctx, oracle, torch_dev, pyr_scale, src, dst):
    import torch
    rng = np.random.default_rng(3)
    f = (rng.standard_normal((2,) + src + (2,)) * 4).astype(np.float32)
    d = _dev(torch_dev, f)
    out = torch.empty((2,) + dst + (2,), dtype=torch.float32, device=torch_dev)
    ctx.check(ctx._lib.nsof_stage_flow_upsample(ctx.ptr, 2, d.data_ptr(), src[1], src[0], out.data_ptr(), dst[1],
                                                dst[0], pyr_scale))
    ctx.synchronize()
    got = out.cpu().numpy()
    for i in range(2):
        want = oracle.resize_linear(f[i], dst[1], dst[0]) * np.float32(1.0 / pyr_scale)
        assert np.array_equal(got[i], want)


@pytest.mark.parametrize("params", [A, B, Cc, (0.5, 1, 9, 1, 7, 1.5, 0), (0.8, 5, 7, 2, 3, 0.0, 0)],
                         ids=["A", "B", "C", "D", "E"])
@pytest.mark.parametrize("shape", [(135, 240), (200, 303), (97, 131)])
def test_pipeline_vs_oracle(nsof_lib, ctx, oracle, frames, params, shape):
    prev, nxt = frames[shape]
    got = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *params, ctx=ctx)
    want = oracle.farneback(prev, nxt, *params)
    assert got.shape == want.shape and got.dtype == np.float32
    err = np.abs(got - want).max()
    assert err <= PIPE_TOL, f"max-abs {err}"


def test_pipeline_1080p_vs_oracle(nsof_lib, ctx, oracle):
    """BASELINE config 2 shape: one seeded synthetic 1920x1080 pair, params A and B."""
    from nsof import synth
    prev, nxt = synth.make_pair(1234, 1080, 1920)
    for params in (A, B):
        got = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *params, ctx=ctx)
        want = oracle.farneback(prev, nxt, *params)
        err = np.abs(got - want).max()
        assert err <= PIPE_TOL, f"{params}: max-abs {err}"
    # known motion is recovered (interior, params A)
    got = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *A, ctx=ctx)
    tf = synth.true_flow(1080, 1920)
    assert np.abs(got - tf)[100:-100, 100:-100].mean() < 0.05


def test_roi_view_equals_standalone(nsof_lib, ctx, frames):
    """A strided ROI view (optical_flow_seg.py:186-187) gives the same flow as a contiguous copy."""
    prev, nxt = frames[(270, 480)]
    pv, nv = prev[40:240, 60:363], nxt[40:240, 60:363]
    assert not pv.flags.c_contiguous
    a = nsof_lib.calcOpticalFlowFarneback(pv, nv, None, *A, ctx=ctx)
    b = nsof_lib.calcOpticalFlowFarneback(pv.copy(), nv.copy(), None, *A, ctx=ctx)
    assert np.array_equal(a, b)


def test_flow_argument_reused_and_keywords(nsof_lib, ctx, frames):
    prev, nxt = frames[(135, 240)]
    kw = dict(pyr_scale=0.5, levels=3, winsize=15, iterations=3, poly_n=5, poly_sigma=1.2, flags=0)
    buf = np.zeros((135, 240, 2), np.float32)
    out = nsof_lib.calcOpticalFlowFarneback(prev, nxt, buf, **kw, ctx=ctx)
    assert out is buf
    ref = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, **kw, ctx=ctx)
    assert np.array_equal(out, ref)
    assert np.array_equal(prev, frames[(135, 240)][0])  # inputs untouched


def test_identical_frames_zero_interior(nsof_lib, ctx, frames):
    prev, _ = frames[(270, 480)]
    z = nsof_lib.calcOpticalFlowFarneback(prev, prev, None, *B, ctx=ctx)
    assert np.abs(z[:150, :250]).max() == 0.0


def test_batch_equals_single_calls(nsof_lib, ctx, torch_dev):
    import torch
    from nsof import synth
    h, w, n = 120, 200, 5
    pairs = [synth.make_pair(50 + i, h, w, shift=(1.0 + i, -0.5 * i)) for i in range(n)]
    dp = _dev(torch_dev, np.stack([p for p, _ in pairs]))
    dn = _dev(torch_dev, np.stack([q for _, q in pairs]))
    df = torch.empty((n, h, w, 2), dtype=torch.float32, device=torch_dev)
    P = nsof_lib.FarnebackParams(*A)
    nsof_lib.farneback_batch(dp, dn, df, n, h, w, P, ctx=ctx)
    ctx.synchronize()
    got = df.cpu().numpy()
    for i, (p, q) in enumerate(pairs):
        one = nsof_lib.calcOpticalFlowFarneback(p, q, None, *A, ctx=ctx)
        assert np.array_equal(got[i], one), i


def test_error_behaviour(nsof_lib, ctx, frames):
    prev, nxt = frames[(135, 240)]
    with pytest.raises(nsof_lib.error):
        nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, 1.0, 3, 15, 3, 5, 1.2, 0, ctx=ctx)  # pyr_scale < 1
    with pytest.raises(nsof_lib.error):
        nsof_lib.calcOpticalFlowFarneback(prev, nxt[:-1], None, *A, ctx=ctx)  # sizes differ
    with pytest.raises(nsof_lib.error) as e:
        nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, 0.5, 3, 15, 3, 5, 1.2, 4, ctx=ctx)
    assert e.value.status == -5
    with pytest.raises(nsof_lib.error):
        nsof_lib.calcOpticalFlowFarneback(prev.astype(np.float32), nxt, None, *A, ctx=ctx)
    # levels are truncated for small images (min_size 32), tiny images still work
    small = nsof_lib.calcOpticalFlowFarneback(prev[:33, :40], nxt[:33, :40], None, *A, ctx=ctx)
    assert small.shape == (33, 40, 2) and np.isfinite(small).all()


def test_pipeline_4k_vs_oracle(nsof_lib, ctx, oracle):
    """BASELINE config 5 frame size: one 3840x2160 pair, params A (the oracle needs ~4 s)."""
    from nsof import synth
    prev, nxt = synth.make_pair(5, 2160, 3840, shift=(3.0, 1.5))
    got = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *A, ctx=ctx)
    want = oracle.farneback(prev, nxt, *A)
    assert np.abs(got - want).max() <= PIPE_TOL


@pytest.mark.parametrize("shape", [(33, 40), (64, 64), (161, 161), (801, 801), (1080, 1920)])
def test_context_reuse_across_shapes(nsof_lib, ctx, oracle, shape):
    """One context serves calls of changing size (workspace grows and is reused), as the ROI dispatcher needs."""
    from nsof import synth
    prev, nxt = synth.make_pair(shape[0], *shape)
    for params in (Cc, A):
        got = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *params, ctx=ctx)
        assert np.abs(got - oracle.farneback(prev, nxt, *params)).max() <= PIPE_TOL


def test_large_window_takes_unfused_path(nsof_lib, ctx, oracle, frames):
    """winsize > 15 is outside the fused iteration kernels: the unfused pair must give the same result class."""
    prev, nxt = frames[(200, 303)]
    p = (0.5, 2, 25, 2, 5, 1.2, 0)
    got = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *p, ctx=ctx)
    assert np.abs(got - oracle.farneback(prev, nxt, *p)).max() <= PIPE_TOL


def test_fuzz_parameters_vs_oracle(nsof_lib, ctx, oracle):
    """Seeded sweep over shapes and parameter combinations (exercises every templated kernel instance: poly_n 1..10,
    window half-widths 1..8 and the unfused path beyond, blur sizes 3..31 incl. the runtime-size pyramid kernels)."""
    from nsof import synth
    rng = np.random.default_rng(2024)
    worst = 0.0
    for case in range(48):
        h, w = int(rng.integers(33, 260)), int(rng.integers(33, 420))
        p = (float(rng.choice([0.5, 0.6, 0.75, 0.8])), int(rng.integers(0, 6)), int(rng.integers(2, 20)),
             int(rng.integers(0, 4)), int(rng.integers(1, 11)), float(rng.choice([0.0, 0.8, 1.1, 1.5, 2.0])), 0)
        prev, nxt = synth.make_pair(1000 + case, h, w, shift=(float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4))),
                                    rot_deg=float(rng.uniform(-1, 1)))
        got = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *p, ctx=ctx)
        want = oracle.farneback(prev, nxt, *p)
        err = float(np.abs(got - want).max())
        worst = max(worst, err)
        assert err <= PIPE_TOL * max(1.0, float(np.abs(want).max()) / 10), (case, (h, w), p, err)
    assert worst <= 1e-4


@pytest.mark.parametrize("shape", [(256, 272), (264, 328), (1080, 1920)])
def test_three_level_pyramid_launch_batch_and_sequence(nsof_lib, ctx, oracle, torch_dev, shape):
    """pyr_scale 0.5 with three coarser levels on frames that decimate exactly by 8: the batch driver makes levels 1-3
    in ONE launch (k_prep_decim3; 16- and 8-column lanes) and level 0 inside the expansion kernel.  A batch (prev and
    next arrays apart: two launches) and a sequence (one array) through the fused-kernel path, against the oracle."""
    import torch
    from nsof import _lib, synth
    h, w = shape
    n = 2 if h > 1000 else 3
    assert nsof_lib.effective_levels(w, h, 0.5, 3) == 3
    base, _ = synth.make_pair(31, h + 16, w + 16)
    frames = np.stack([np.ascontiguousarray(base[2 * i:2 * i + h, 3 * i:3 * i + w]) for i in range(n + 1)])
    want = [oracle.farneback(frames[i], frames[i + 1], *A) for i in range(n)]
    P = nsof_lib.FarnebackParams(*A)
    ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, 0)
    try:
        dp, dn = _dev(torch_dev, frames[:-1].copy()), _dev(torch_dev, frames[1:].copy())
        df = torch.empty((n, h, w, 2), dtype=torch.float32, device=torch_dev)
        nsof_lib.farneback_batch(dp, dn, df, n, h, w, P, ctx=ctx)
        ctx.synchronize()
        got = df.cpu().numpy()
        ds = _dev(torch_dev, frames)
        out = torch.empty((n, h, w, 2), dtype=torch.float32, device=torch_dev)
        nsof_lib.farneback_sequence(ds, out, n + 1, h, w, P, ctx=ctx)
        ctx.synchronize()
        seq = out.cpu().numpy()
    finally:
        ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, 64)
    for i in range(n):
        assert np.array_equal(got[i], want[i]), ("batch", i, float(np.abs(got[i] - want[i]).max()))
        assert np.array_equal(seq[i], want[i]), ("sequence", i, float(np.abs(seq[i] - want[i]).max()))


def test_sequence_equals_consecutive_pairs(nsof_lib, ctx, torch_dev):
    """nsof_farneback_u8_sequence_dev == one call per consecutive pair, bit for bit (params A and B)."""
    import torch
    from nsof import synth
    h, w, n = 120, 200, 6
    base, _ = synth.make_pair(9, h + 40, w + 40)
    frames = np.stack([np.ascontiguousarray(base[10 + 2 * i:10 + 2 * i + h, 5 + 3 * i:5 + 3 * i + w]) for i in range(n)])
    d = _dev(torch_dev, frames)
    for params in (A, B):
        out = torch.empty((n - 1, h, w, 2), dtype=torch.float32, device=torch_dev)
        nsof_lib.farneback_sequence(d, out, n, h, w, nsof_lib.FarnebackParams(*params), ctx=ctx)
        ctx.synchronize()
        got = out.cpu().numpy()
        for i in range(n - 1):
            one = nsof_lib.calcOpticalFlowFarneback(frames[i], frames[i + 1], None, *params, ctx=ctx)
            assert np.array_equal(got[i], one), (params, i)


def test_farneback_many_streams_equal_sequential(nsof_lib):
    """Independent pairs of different shapes over a pool of HIP streams (threads + one context each) give the same
    flows as one call after the other; the FLAG-1 gating path uses the pool for the components of a pair."""
    from nsof import synth
    shapes = [(200, 520), (161, 161), (97, 131), (240, 320), (64, 64), (135, 240), (120, 333), (200, 520), (90, 90)]
    pairs = [synth.make_pair(300 + i, h, w) for i, (h, w) in enumerate(shapes)]
    want = [nsof_lib.calcOpticalFlowFarneback(a, b, None, *A) for a, b in pairs]
    with nsof_lib.StreamPool(4) as pool:
        for _ in range(2):
            got = pool.map(pairs, nsof_lib.FarnebackParams(*A))
            assert all(np.array_equal(g, w_) for g, w_ in zip(got, want))
        # gating, FLAG 1: components of a pair on the pool
        tp = np.zeros((6, 8), np.uint8)
        tp[1, 1] = tp[1, 2] = tp[4, 6] = 255
        tp[3, 3] = 255
        base, nxt = synth.make_pair(77, 6 * 40, 8 * 40)
        res = []
        for sp in (None, pool):
            cfg = nsof_lib.dataset_config("uav", FLAG=1, THRES=200, MEMSIZE=40, stream_pool=sp)
            res.append(nsof_lib.opticalFlow3D(tp, tp, base, nxt, 40, 40, cfg))
        assert np.array_equal(res[0][0], res[1][0]) and res[0][5] == res[1][5] and len(res[0][5]) == 3
    assert len(nsof_lib.farneback_many(pairs[:3], nsof_lib.FarnebackParams(*A))) == 3


def test_polyexp_float_mode_is_opt_in_and_close(ctx, nsof_lib, torch_dev):
    """NSOF_OPT_POLYEXP_F32: default off (exact path); switched on, the expansion coefficients stay within float
    rounding of the exact ones and the flow within the measured bound DESIGN.md states (not bit-identical)."""
    import torch
    from nsof import _lib, synth
    assert ctx.get_option(_lib.OPT_POLYEXP_F32) == 0
    h, w = 270, 480
    prev, nxt = synth.make_pair(77, h, w)
    img = torch.from_numpy(prev.astype(np.float32)).to(torch_dev)
    r_exact = torch.empty(5 * h * w, dtype=torch.float32, device=torch_dev)
    r_fast = torch.empty_like(r_exact)
    torch.cuda.synchronize()
    ctx.check(ctx._lib.nsof_stage_polyexp(ctx.ptr, 1, img.data_ptr(), w, h, 5, 1.2, r_exact.data_ptr()))
    ctx.set_option(_lib.OPT_POLYEXP_F32, 1)
    try:
        ctx.check(ctx._lib.nsof_stage_polyexp(ctx.ptr, 1, img.data_ptr(), w, h, 5, 1.2, r_fast.data_ptr()))
        f_fast = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *A, ctx=ctx)
    finally:
        ctx.set_option(_lib.OPT_POLYEXP_F32, 0)
    ctx.synchronize()
    f_exact = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *A, ctx=ctx)
    re, rf = r_exact.cpu().numpy(), r_fast.cpu().numpy()
    assert not np.array_equal(re, rf)                       # it IS a different arithmetic ...
    assert float(np.abs(re - rf).max()) <= 2e-5 * float(np.abs(re).max())   # ... within float rounding of the sums
    assert float(np.abs(f_fast - f_exact).max()) < 1e-3


@pytest.mark.gpu
def test_row_bands_mode_is_opt_in_and_close(ctx, nsof_lib, oracle):
    """NSOF_OPT_ROW_BANDS (low-latency mode for lone calls of the FAST row-sum mode, NSOF_OPT_EXACT_ROWSUMS = 0): off by
    default; on, every strip of that mode's iteration kernel is
    cut into row bands whose column sums start from a direct sum instead of the library's running sum from row 0 --
    same numbers to ~1e-16, hence a flow that stays inside the oracle tolerance on textured frames (params A, wide
    window) and moves in the 4th decimal where 3x3 windows are rank deficient (params B; same sensitivity as the
    row-sum order; which is why the automatic mode leaves windows below 9 alone: value 1 with
    params B is bit-identical to the default).  Ragged last bands, bands shorter than the window, and images shorter
    than one band (no split: bit-identical to the default) are covered."""
    from nsof import _lib, synth
    assert ctx.get_option(_lib.OPT_ROW_BANDS) == 0
    assert ctx.get_option(_lib.OPT_EXACT_ROWSUMS) == 1          # the library's row-sum order is the default
    for bad in (-1, 2, 3):
        assert ctx._lib.nsof_set_option(ctx.ptr, _lib.OPT_ROW_BANDS, bad) == _lib.NSOF_EINVAL
    assert ctx._lib.nsof_set_option(ctx.ptr, _lib.OPT_EXACT_ROWSUMS, 2) == _lib.NSOF_EINVAL
    ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 0)                   # row bands belong to the fast row-sum mode
    try:
        _row_bands_body(ctx, nsof_lib, oracle)
    finally:
        ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 1)


def _row_bands_body(ctx, nsof_lib, oracle):
    from nsof import _lib, synth
    ns = 1e-4   # the north-star tolerance (max-abs end-point error)
    cases = [((1080, 1920), A, 1, ns), ((203, 317), A, 32, ns), ((203, 317), A, 4, ns),
             ((135, 240), Cc, 8, 1e-3), ((801, 801), B, 32, 1e-3), ((801, 801), B, 1, 0.0), ((30, 200), A, 32, 0.0)]
    for (h, w), params, bands, tol in cases:
        prev, nxt = synth.make_pair(4242 + h, h, w)
        base = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *params, ctx=ctx)
        ctx.set_option(_lib.OPT_ROW_BANDS, bands)
        try:
            assert ctx.get_option(_lib.OPT_ROW_BANDS) == bands
            got = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *params, ctx=ctx)
        finally:
            ctx.set_option(_lib.OPT_ROW_BANDS, 0)
        if tol == 0.0:
            assert np.array_equal(got, base)
            continue
        want = oracle.farneback(prev, nxt, *params)
        err, dev = float(np.abs(got - want).max()), float(np.abs(got - base).max())
        assert err <= tol and dev <= tol, f"{(h, w)} bands={bands}: vs oracle {err}, vs default {dev}"
    # the keyword of the drop-in switches it per call and restores the context's setting
    prev, nxt = synth.make_pair(7, 270, 480)
    a = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *A, ctx=ctx, low_latency=True)
    assert ctx.get_option(_lib.OPT_ROW_BANDS) == 0
    ctx.set_option(_lib.OPT_ROW_BANDS, 1)
    try:
        b = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *A, ctx=ctx)
        c = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *A, ctx=ctx, exact=True)   # exact order ignores bands
    finally:
        ctx.set_option(_lib.OPT_ROW_BANDS, 0)
    assert np.array_equal(a, b)
    assert np.array_equal(c, oracle.farneback(prev, nxt, *A))


@pytest.mark.gpu
def test_recip_matches_ieee_division(ctx, torch_dev):
    """The 2x2 solves take 1/det from nsof_recip_normal (the compiler's own division sequence minus its operand
    scaling and special-case fix-ups).  Over the determinants' range and far beyond it -- 2^-900 .. 2^900, 64 M
    log-uniform values plus mantissa edge cases -- it returns the bits of IEEE division."""
    import torch
    g = torch.Generator(device=torch_dev).manual_seed(5)
    n = 1 << 24
    for rep in range(4):
        e = torch.rand(n, dtype=torch.float64, device=torch_dev, generator=g) * (1800.0 if rep else 110.0) - (900.0 if rep else 10.0)
        x = torch.exp2(torch.floor(e)) * (1.0 + torch.rand(n, dtype=torch.float64, device=torch_dev, generator=g))
        if rep == 3:   # mantissas next to a power of two and all-ones, exact powers of two
            k = torch.arange(n, device=torch_dev) % 4
            m = torch.where(k == 0, torch.tensor(1.0, dtype=torch.float64, device=torch_dev),
                            torch.where(k == 1, torch.tensor(2.0 - 2.0 ** -52, dtype=torch.float64, device=torch_dev),
                                        torch.where(k == 2, torch.tensor(1.0 + 2.0 ** -52, dtype=torch.float64, device=torch_dev),
                                                    torch.tensor(1.5, dtype=torch.float64, device=torch_dev))))
            x = torch.exp2(torch.floor(e)) * m
        fast, ieee = torch.empty_like(x), torch.empty_like(x)
        torch.cuda.synchronize()
        ctx.check(ctx._lib.nsof_stage_recip(ctx.ptr, n, x.data_ptr(), fast.data_ptr(), ieee.data_ptr()))
        ctx.synchronize()
        assert torch.equal(fast.view(torch.int64), ieee.view(torch.int64)), f"rep {rep}"
        assert torch.equal(ieee, 1.0 / x)


@pytest.mark.gpu
def test_batches_larger_than_memory_are_chunked(nsof_lib, ctx, torch_dev, monkeypatch):
    """A batch whose workspace would not fit the device is run in chunks of as many pairs as fit (here capped by hand
    with NSOF_MAX_PAIRS): same result as the unchunked call, pairs and consecutive-frame sequences, default and exact
    order."""
    import torch
    from nsof import _lib, synth
    h, w, n = 96, 160, 7
    base, _ = synth.make_pair(19, h + 40, w + 40)
    frames = np.stack([np.ascontiguousarray(base[5 + 2 * i:5 + 2 * i + h, 3 * i:3 * i + w]) for i in range(n + 1)])
    d = _dev(torch_dev, frames)
    P = nsof_lib.FarnebackParams(*A)
    saved = ctx.get_option(_lib.OPT_EXACT_ROWSUMS)
    for exact in (0, 1):
        ctx.set_option(_lib.OPT_EXACT_ROWSUMS, exact)
        try:
            res = {}
            for cap in (None, "3", "1"):
                if cap is None:
                    monkeypatch.delenv("NSOF_MAX_PAIRS", raising=False)
                else:
                    monkeypatch.setenv("NSOF_MAX_PAIRS", cap)
                seq = torch.empty((n, h, w, 2), dtype=torch.float32, device=torch_dev)
                par = torch.empty_like(seq)
                nsof_lib.farneback_sequence(d, seq, n + 1, h, w, P, ctx=ctx)
                nsof_lib.farneback_batch(d[:-1], d[1:], par, n, h, w, P, ctx=ctx)
                ctx.synchronize()
                res[cap] = (seq.cpu().numpy(), par.cpu().numpy())
            for cap in ("3", "1"):
                assert np.array_equal(res[cap][0], res[None][0]) and np.array_equal(res[cap][1], res[None][1]), (exact, cap)
            assert np.array_equal(res[None][0], res[None][1])
        finally:
            ctx.set_option(_lib.OPT_EXACT_ROWSUMS, saved)


@pytest.mark.gpu
def test_more_pairs_than_one_grid_holds(nsof_lib, ctx, torch_dev):
    """40 000 small pairs in ONE call (a launch addresses at most 32767 pairs through gridDim.z; the driver chunks):
    spot-checked against single calls."""
    import torch
    n, h, w = 40000, 24, 40
    g = torch.Generator(device=torch_dev).manual_seed(9)
    frames = torch.randint(0, 256, (n + 1, h, w), dtype=torch.uint8, device=torch_dev, generator=g)
    flow = torch.empty((n, h, w, 2), dtype=torch.float32, device=torch_dev)
    P = nsof_lib.FarnebackParams(0.5, 2, 5, 2, 5, 1.1, 0)
    nsof_lib.farneback_batch(frames[:-1], frames[1:], flow, n, h, w, P, ctx=ctx)
    ctx.synchronize()
    host = frames.cpu().numpy()
    for i in (0, 1, 32766, 32767, 32768, n - 1):
        one = nsof_lib.calcOpticalFlowFarneback(host[i], host[i + 1], None, *P.as_kwargs().values(), ctx=ctx)
        assert np.array_equal(flow[i].cpu().numpy(), one), i


@pytest.mark.gpu
def test_pyramid_fma_variant_twin(nsof_lib, ctx, oracle, torch_dev):
    """NSOF_OPT_PYR_FMA: the arithmetic-variant twin of the pyramid stages (float Gaussian blur + bilinear resamples with
    one fused multiply-add per tap / blend, as an AVX2+FMA3 build of the library's vector loops contracts them).  GPU and
    CPU oracle agree bit for bit IN EACH VARIANT -- pyramid levels of every kernel family (same-size, exact decimation,
    generic scales), the flow resample, and the whole call for the reference's three parameter sets -- and the two
    variants differ from each other (by what DESIGN.md section 2 reports)."""
    import torch
    from nsof import _lib, synth
    assert ctx.get_option(_lib.OPT_PYR_FMA) == 0
    shapes = [(270, 480), (200, 303), (256, 512)]
    try:
        res = {}
        for fma in (0, 1):
            ctx.set_option(_lib.OPT_PYR_FMA, fma)
            oracle.set_pyr_fma(bool(fma))
            for shape in shapes:
                prev, nxt = synth.make_pair(100 + shape[0], *shape)
                d = _dev(torch_dev, np.stack([prev, nxt]))
                for pyr_scale, level in [(0.5, 0), (0.5, 1), (0.5, 2), (0.5, 3), (0.6, 1), (0.6, 2), (0.6, 3), (0.75, 2)]:
                    wk, hk, _, _ = nsof_lib.level_size(shape[1], shape[0], pyr_scale, level)
                    out = torch.empty((2, hk, wk), dtype=torch.float32, device=torch_dev)
                    ctx.check(ctx._lib.nsof_stage_pyr_level(ctx.ptr, 2, d.data_ptr(), shape[1], shape[0] * shape[1], shape[1],
                                                            shape[0], pyr_scale, level, out.data_ptr()))
                    ctx.synchronize()
                    got = out.cpu().numpy()
                    for i, img in enumerate((prev, nxt)):
                        want = oracle.pyr_level(img, pyr_scale, level)
                        assert np.array_equal(got[i], want), (fma, shape, pyr_scale, level, float(np.abs(got[i] - want).max()))
            rng = np.random.default_rng(5)
            for (sh, sw), (dh, dw), ps in [((68, 120), (135, 240), 0.5), ((58, 79), (97, 131), 0.6), ((135, 400), (270, 800), 0.5)]:
                src = (rng.standard_normal((2, sh, sw, 2)) * 3).astype(np.float32)
                dsrc = _dev(torch_dev, src)
                out = torch.empty((2, dh, dw, 2), dtype=torch.float32, device=torch_dev)
                ctx.check(ctx._lib.nsof_stage_flow_upsample(ctx.ptr, 2, dsrc.data_ptr(), sw, sh, out.data_ptr(), dw, dh, ps))
                ctx.synchronize()
                want = oracle.resize_linear(src[0], dw, dh) * np.float32(1. / ps)
                assert np.array_equal(out.cpu().numpy()[0], want), (fma, sh, sw)
            for name, P in (("A", A), ("B", B), ("C", Cc)):
                prev, nxt = synth.make_pair(77, 270, 480)
                got = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *P, ctx=ctx)
                want = oracle.farneback(prev, nxt, *P)
                assert np.array_equal(got, want), (fma, name, float(np.abs(got - want).max()))
                res[(fma, name)] = got
    finally:
        ctx.set_option(_lib.OPT_PYR_FMA, 0)
        oracle.set_pyr_fma(False)
    for name in "ABC":
        d = float(np.abs(res[(0, name)] - res[(1, name)]).max())
        assert 0 < d < 5e-3, (name, d)


@pytest.mark.gpu
def test_small_batch_form_is_bit_identical_to_fused_kernel(nsof_lib, ctx, oracle):
    """NSOF_OPT_SMALL_BATCH_JOBS: calls with few (strip, image) jobs run the exact-order iteration as three wide kernels
    (matrices / column sums / row scan + solve, farneback_iterate_lat.hip) instead of the fused strip walker -- same
    arithmetic, same order: both forms equal the oracle bit for bit, lone calls and ROI work lists, every window
    the fused kernel covers, ragged and tiny sizes included."""
    from nsof import _lib, synth
    assert ctx.get_option(_lib.OPT_SMALL_BATCH_JOBS) == 64
    cases = [((135, 240), A), ((200, 303), B), ((97, 131), Cc), ((33, 17), (0.5, 2, 5, 2, 5, 1.1, 0)),
             ((70, 450), (0.6, 3, 15, 2, 7, 1.5, 0)), ((257, 64), (0.5, 1, 2, 3, 5, 1.2, 0)), ((16, 16), (0.5, 0, 9, 1, 5, 1.1, 0)),
             ((540, 960), A)]
    try:
        for k, (shape, params) in enumerate(cases):
            prev, nxt = synth.make_pair(60 + k, *shape)
            want = oracle.farneback(prev, nxt, *params)
            for jobs in (0, 1 << 30):
                ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, jobs)
                got = nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *params, ctx=ctx)
                assert np.array_equal(got, want), (shape, params, jobs, float(np.abs(got - want).max()))
        # a ragged ROI work list (the gated path's calls) in both forms
        big_p, big_n = synth.make_pair(77, 400, 600)
        rects = [(0, 0, 600, 400), (10, 20, 210, 140), (301, 7, 364, 390), (100, 100, 133, 121), (17, 250, 590, 399)]
        pairs = [(big_p[y0:y1, x0:x1], big_n[y0:y1, x0:x1]) for (x0, y0, x1, y1) in rects]
        ref = [oracle.farneback(np.ascontiguousarray(a), np.ascontiguousarray(b), *B) for a, b in pairs]
        for jobs in (0, 1 << 30):
            ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, jobs)
            flows = nsof_lib.farneback_pairs(pairs, nsof_lib.farneback.PARAMS_B, ctx=ctx)
            for f, r in zip(flows, ref):
                assert np.array_equal(f, r), jobs
        with pytest.raises(nsof_lib.NsofError):
            ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, -1)
    finally:
        ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, 64)


@pytest.mark.gpu
def test_small_batch_form_edge_shapes_batches_and_sequences(nsof_lib, ctx, oracle, torch_dev):
    """The three-kernel form on the shapes that stress its tiling (2x2, a 3-row strip, widths below one tile, odd sizes), on a
    uniform device batch of several pairs and on a frame sequence: equal to the oracle / to the fused kernel bit for bit."""
    import torch
    from nsof import _lib, synth
    try:
        for (h, w) in [(2, 2), (3, 70), (5, 5), (9, 33), (64, 31), (130, 257)]:
            big_a, big_b = synth.make_pair(11, max(h, 8), max(w, 8))
            a, b = np.ascontiguousarray(big_a[:h, :w]), np.ascontiguousarray(big_b[:h, :w])
            for params in [(0.5, 2, 3, 2, 5, 1.1, 0), (0.5, 1, 15, 1, 5, 1.2, 0), (0.7, 3, 6, 3, 7, 1.5, 0)]:
                want = oracle.farneback(a, b, *params)
                for jobs in (0, 1 << 30):
                    ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, jobs)
                    got = nsof_lib.calcOpticalFlowFarneback(a, b, None, *params, ctx=ctx)
                    assert np.array_equal(got, want), (h, w, params, jobs)
        a, b = synth.make_pair(5, 200, 300)
        pv = torch.from_numpy(np.stack([a, b, a])).to(torch_dev)
        nx = torch.from_numpy(np.stack([b, a, a])).to(torch_dev)
        fr = torch.from_numpy(np.stack([a, b, a, b])).to(torch_dev)
        res = {}
        for jobs in (0, 1 << 30):
            ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, jobs)
            f1 = torch.empty((3, 200, 300, 2), dtype=torch.float32, device=torch_dev)
            f2 = torch.empty((3, 200, 300, 2), dtype=torch.float32, device=torch_dev)
            torch.cuda.synchronize()
            nsof_lib.farneback_batch(pv, nx, f1, 3, 200, 300, nsof_lib.farneback.PARAMS_B, ctx=ctx)
            nsof_lib.farneback_sequence(fr, f2, 4, 200, 300, nsof_lib.farneback.PARAMS_A, ctx=ctx)
            ctx.synchronize()
            res[jobs] = (f1.cpu().numpy(), f2.cpu().numpy())
        assert np.array_equal(res[0][0], res[1 << 30][0]) and np.array_equal(res[0][1], res[1 << 30][1])
        assert np.array_equal(res[0][0][0], oracle.farneback(a, b, *B))
    finally:
        ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, 64)


@pytest.mark.parametrize("form", ["fused_kernel", "small_batch_form"])
def test_lost_handover_is_reported(nsof_lib, oracle, form):
    """cv2 raises where it fails (/root/reference/optical_flow_seg.py:203 would propagate cv2.error).  The one failure mode
    the exact-order kernels add -- a hand-over between workgroups (k_iterate_x: strip-to-strip carries) or waves
    (k_lat_colsum: turns) that never arrives -- must fail the CALL THE REFERENCE MAKES, in bounded time, and leave the
    context usable: NSOF_OPT_DEBUG_FAULT makes one strip withhold its carries / one wave its turn."""
    import time
    from nsof import _lib, synth
    from nsof.errors import NsofDeviceError
    prev, nxt = synth.make_pair(31, 300, 420)          # 3 strips of 192 columns; 10 turns of 32 rows
    c = nsof_lib.Context(0)
    try:
        c.set_option(_lib.OPT_SMALL_BATCH_JOBS, 0 if form == "fused_kernel" else 64)
        want = oracle.farneback(prev, nxt, *A)
        assert np.array_equal(nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *A, ctx=c), want)
        c.set_option(_lib.OPT_DEBUG_FAULT, 1 if form == "fused_kernel" else 2)
        t0 = time.perf_counter()
        with pytest.raises(NsofDeviceError) as ei:
            nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *A, ctx=c)
        dt = time.perf_counter() - t0
        assert ei.value.status == _lib.NSOF_EDEVICE and "hand-over" in str(ei.value)
        assert dt < 10.0, f"a failed launch must drain quickly, took {dt:.1f} s"
        # the pipelined list entry and the device-resident entry + synchronize report it as well
        with pytest.raises(NsofDeviceError):
            nsof_lib.farneback_pairs([(prev, nxt)], nsof_lib.farneback.PARAMS_A, ctx=c)
        c.set_option(_lib.OPT_DEBUG_FAULT, 0)
        # the error is consumed; the same context computes the oracle's bits again
        c.synchronize()
        assert np.array_equal(nsof_lib.calcOpticalFlowFarneback(prev, nxt, None, *A, ctx=c), want)
    finally:
        c.close()
