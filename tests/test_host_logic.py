"""CPU: host-side mirror of the reference interfaces -- argument checking happens before any device work."""
import numpy as np
import pytest


def test_signature_and_keywords(nsof_lib):
    import inspect
    sig = inspect.signature(nsof_lib.calcOpticalFlowFarneback)
    names = list(sig.parameters)
    # cv2.calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags)
    assert names[:10] == ["prev", "next", "flow", "pyr_scale", "levels", "winsize", "iterations", "poly_n",
                          "poly_sigma", "flags"]
    # the reference's farneback_params dict (optical_flow_seg.py:73-81) unpacks into it
    p = nsof_lib.farneback.PARAMS_A
    assert p.as_kwargs() == dict(pyr_scale=0.5, levels=3, winsize=15, iterations=3, poly_n=5, poly_sigma=1.2, flags=0)
    sig.bind(np.zeros((4, 4), np.uint8), np.zeros((4, 4), np.uint8), None, **p.as_kwargs())


@pytest.mark.parametrize("bad", ["shape", "dtype", "channels", "notarray", "empty"])
def test_bad_inputs_raise_before_touching_the_device(nsof_lib, bad):
    a = np.zeros((32, 48), np.uint8)
    b = a.copy()
    if bad == "shape":
        b = np.zeros((32, 47), np.uint8)
    elif bad == "dtype":
        a = a.astype(np.float32)
    elif bad == "channels":
        a = np.zeros((32, 48, 3), np.uint8)
    elif bad == "notarray":
        a = [[0] * 48] * 32
    elif bad == "empty":
        a = b = np.zeros((0, 48), np.uint8)
    with pytest.raises(nsof_lib.error) as e:
        nsof_lib.calcOpticalFlowFarneback(a, b, None, 0.5, 3, 15, 3, 5, 1.2, 0)
    assert isinstance(e.value, ValueError)


def test_install_patches_a_cv2_like_module(nsof_lib):
    import types
    fake = types.ModuleType("cv2")
    fake.calcOpticalFlowFarneback = sentinel = object()
    nsof_lib.install(fake)
    assert fake.calcOpticalFlowFarneback is nsof_lib.calcOpticalFlowFarneback
    nsof_lib.uninstall(fake)
    assert fake.calcOpticalFlowFarneback is sentinel


def test_simulate_argument_checks(nsof_lib):
    ev = (np.int16([1]), np.int16([1]), np.int8([1]), np.int64([0]))
    with pytest.raises(nsof_lib.error):
        nsof_lib.simulate(ev, version=3)
    with pytest.raises(nsof_lib.error):
        nsof_lib.simulate(ev, version=2, polarity="both")
    with pytest.raises(nsof_lib.error):
        nsof_lib.simulate((np.int16([]), np.int16([]), np.int8([]), np.int64([])), version=1)
    assert nsof_lib.PARAMS["koff"] == 51.03 and nsof_lib.DT == 5e-4 and nsof_lib.REFRACTORY_US == 800


def test_synth_pair_is_deterministic_and_contiguous(nsof_lib):
    from nsof import synth
    a1, b1 = synth.make_pair(3, 64, 96)
    a2, b2 = synth.make_pair(3, 64, 96)
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2) and a1.flags.c_contiguous and a1.dtype == np.uint8
    x, y, p, t = synth.make_events(1, 160, 120, 1000, 20_000, box=(20, 16))
    assert x.dtype == np.int16 and t.dtype == np.int64 and np.all(np.diff(t) >= 0)
    assert x.max() < 160 and y.max() < 120 and set(np.unique(p)) <= {0, 1}


def test_shard_bounds(nsof_lib):
    from nsof.dist import shard_bounds
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(3, 8)[:4] == [(0, 1), (1, 2), (2, 3), (3, 3)]
    for n in (0, 1, 7, 64, 360):
        for w in (1, 2, 3, 8):
            b = shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b, b[1:]))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
