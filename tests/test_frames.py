"""Frame preparation of the frame-driven accumulator: crop + Lanczos-3 compress_image.
PARITY UNPINNED against MATLAB (no MATLAB/Octave, no stored compressed frames): the NumPy mirror is checked against
the independent scalar restatement in oracle/accum_ref.c (1e-13) and against properties of imresize."""
import numpy as np
import pytest


def test_imresize_matches_scalar_restatement(oracle, nsof_lib):
    from nsof import frames
    rng = np.random.default_rng(0)
    for h, w, oh, ow in ((161, 161, 4, 4), (801, 801, 4, 4), (50, 70, 25, 10), (20, 30, 40, 45), (33, 90, 8, 2)):
        img = rng.random((h, w))
        got = frames.imresize_lanczos3(img, oh, ow)
        assert got.shape == (oh, ow)
        assert np.abs(got - oracle.imresize_lanczos3(img, oh, ow)).max() < 1e-13, (h, w, oh, ow)


def test_imresize_properties(nsof_lib):
    from nsof import frames
    assert np.abs(frames.imresize_lanczos3(np.full((161, 161), 0.3), 4, 4) - 0.3).max() < 1e-14   # weights sum to 1
    rng = np.random.default_rng(1)
    img = rng.random((24, 24))
    assert np.abs(frames.imresize_lanczos3(img, 24, 24) - img).max() < 1e-12                       # scale 1: identity
    a = frames.imresize_lanczos3(img, 6, 6)
    assert np.abs(frames.imresize_lanczos3(img[::-1, ::-1], 6, 6) - a[::-1, ::-1]).max() < 1e-13   # symmetric kernel
    rgb = rng.random((40, 40, 3))
    out = frames.imresize_lanczos3(rgb, 5, 5)
    assert out.shape == (5, 5, 3)
    assert np.abs(out[..., 1] - frames.imresize_lanczos3(rgb[..., 1], 5, 5)).max() < 1e-15
    # a shrinking resize is an antialiased average: a fine checkerboard goes to its mean
    chk = (np.add.outer(np.arange(160), np.arange(160)) % 2).astype(np.float64)
    assert np.abs(frames.imresize_lanczos3(chk, 4, 4) - 0.5).max() < 1e-3


def test_crop_and_compress_follow_the_script(nsof_lib):
    from nsof import frames
    img = np.arange(300 * 400, dtype=np.uint32).reshape(300, 400) % 251
    img = img.astype(np.uint8)
    c = frames.crop_image(img, (276 - 200, 79), (236, 239))      # MATLAB corners are 1-based and inclusive
    assert c.shape == (161, 161) and c[0, 0] == img[75, 78] and c[-1, -1] == img[235, 238]
    small = frames.compress_image(c, 40, 40)
    assert small.shape == (4, 4) and small.dtype == np.float64 and 0 <= small.min() and small.max() <= 1.001
    assert frames.im2double(np.array([[255, 0]], np.uint8)).tolist() == [[1.0, 0.0]]
    stack = frames.process_images([img, img[::-1]], 40, 40, (76, 79), (236, 239))
    assert stack.shape == (2, 4, 4) and np.array_equal(stack[0], small)


@pytest.mark.gpu
def test_gpu_frames_to_accumulator(oracle, nsof_lib):
    """process_images -> simulate_frames, the whole frame-driven chain, against the oracle on the same frames."""
    from nsof import frames
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (161, 161), dtype=np.uint8)
    seq = [base.copy() for _ in range(4)]
    seq[2][40:120, 30:90] = 255 - seq[2][40:120, 30:90]          # a change in one quadrant
    comp = frames.process_images(seq, 40, 40)
    w, res = nsof_lib.simulate_frames(comp, 5e-4, 50, 0.7, 1.5)
    w_ref, res_ref = oracle.accum_frames(comp, 5e-4, 50, 0.7, 1.5)
    assert np.abs(w - w_ref).max() < 1e-9 and np.abs(res / res_ref - 1).max() < 1e-9
