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


def test_flow_colour_coding_properties(tmp_path):
    """flow_viz mirror: wheel layout, white at rest, known hues on the axes, 75 % dimming beyond the normaliser, viz()."""
    import numpy as np
    from nsof import flowviz
    wheel = flowviz.make_colorwheel()
    assert wheel.shape == (55, 3) and wheel[0].tolist() == [255, 0, 0] and wheel[15].tolist() == [255, 255, 0]
    assert wheel[21].tolist() == [0, 255, 0] and wheel[25].tolist() == [0, 255, 255] and wheel[36].tolist() == [0, 0, 255]
    flow = np.zeros((4, 4, 2), np.float32)
    flow[0, 0] = (1, 0)
    flow[0, 1] = (0, 1)
    flow[0, 2] = (-1, 0)
    flow[0, 3] = (0.5, 0)
    img = flowviz.flow_to_image(flow)
    assert img.dtype == np.uint8 and img.shape == (4, 4, 3)
    assert img[1, 1].tolist() == [255, 255, 255]                         # no motion -> white
    assert img[0, 0].tolist() == [255, 0, 0] and img[0, 2].tolist() == [0, 209, 255]   # +x red, -x azure (wheel[27])
    assert img[0, 1].tolist() == [255, 229, 0]                           # +y: wheel[13.5]
    assert img[0, 3].tolist() == [255, 127, 127]                         # half magnitude -> half saturation
    assert np.array_equal(flowviz.flow_to_image(flow, convert_to_bgr=True), img[..., ::-1])
    dim = flowviz.flow_to_image(flow, max_flow=0.5)
    assert dim[0, 0].tolist() == [191, 0, 0]                             # beyond the normaliser: colour * 0.75
    from PIL import Image
    flowviz.viz(flow, str(tmp_path / "f.png"))
    assert np.array_equal(np.asarray(Image.open(tmp_path / "f.png")), img[:, :, [2, 1, 0]])


def test_load_gating_stack_and_gray(tmp_path):
    import numpy as np
    import scipy.io
    from nsof import gating
    stack = np.full((3, 2, 4), 1e-7)
    stack[1, 1, 2] = 1e-6
    scipy.io.savemat(tmp_path / "m.mat", {"constructed3DMatrix": stack})
    got = gating.load_gating_stack(str(tmp_path / "m.mat"))
    assert got.shape == (3, 2, 4) and got[1, 1, 2] == 1e-6
    assert gating.current_to_gray(got[:, :, 2])[1, 1] == 255
    f = np.zeros((2, 2, 3), np.uint8)
    f[0, 0] = (255, 0, 0)
    f[1, 1] = (10, 200, 30)
    assert gating.frame_to_gray(f, "RGB2GRAY")[0, 0] == (255 * 9798 + (1 << 14)) >> 15
    assert gating.frame_to_gray(f, "BGR2GRAY")[0, 0] == (255 * 3735 + (1 << 14)) >> 15
    assert gating.frame_to_gray(f)[1, 1] == (10 * 9798 + 200 * 19235 + 30 * 3735 + (1 << 14)) >> 15


def test_generate_synthetic_events_matches_reference_output():
    """event_mem_sim.generate_synthetic_events: same events, same order, as the arrays the reference
    itself produced (tests/golden/synth_events.npz, generated by gen_synth_events_golden.py)."""
    import numpy as np
    from conftest import golden_path
    from nsof.accumulator import generate_synthetic_events
    g = np.load(golden_path("synth_events.npz"))
    for name, kw in {"default": {}, "small": dict(H=60, W=90, box_h=20, box_w=11, speed_pps=700, duration_s=0.2)}.items():
        x, y, p, t = generate_synthetic_events(**kw)
        assert x.dtype.kind == "i" and len(x) == len(g[f"{name}_x"])
        for got, key in ((x, "x"), (y, "y"), (p, "p"), (t, "t")):
            assert np.array_equal(got, g[f"{name}_{key}"]), (name, key)
    x, y, p, t = generate_synthetic_events(duration_s=0.0)
    assert x.size == y.size == p.size == t.size == 0


def test_flowviz_matches_reference_goldens(nsof_lib):
    """Outputs of the reference's own flow_viz.py (tests/golden/gen_flowviz_golden.py ran it) reproduced byte for
    byte: colour wheel, RGB / BGR order, clip_flow, float32 and float64 inputs, all-zero and tiny flows."""
    from conftest import golden_path
    from nsof import flowviz
    with np.load(golden_path("flowviz_golden.npz")) as z:
        assert np.array_equal(flowviz.make_colorwheel(), z["colorwheel"])
        for name in ("smooth", "noise", "tiny", "zero", "f64"):
            flow = z[f"{name}_flow"]
            assert np.array_equal(flowviz.flow_to_image(flow), z[f"{name}_rgb"]), name
            assert np.array_equal(flowviz.flow_to_image(flow, convert_to_bgr=True), z[f"{name}_bgr"]), name
            assert np.array_equal(flowviz.flow_to_image(flow, clip_flow=2.5), z[f"{name}_clip"]), name


def test_global_slice_times_and_band_bounds(nsof_lib):
    """nsof.dist helpers of the row-band sharding (host arithmetic): the first / last event time of every slice of the whole
    stream (what scheme 2's refractory rule reads, /root/reference/eventsim/event_mem_sim.py:243-267), a band's events on
    the GLOBAL slice grid, and the bands themselves."""
    import numpy as np
    from nsof import dist as nd
    from nsof.accumulator import slice_index_array
    t = np.array([5, 7, 1003, 1004, 1999, 4100, 4100, 4990], np.int64)       # slices of 1000 us from t[0] = 5
    y = np.array([0, 9, 3, 8, 1, 9, 2, 7], np.int64)
    x = np.arange(8)
    p = np.ones(8, np.int64)
    idx = slice_index_array(t, 1000)
    tf, tl = nd.global_slice_times(t, 1000)
    assert len(tf) == len(idx) - 1 == 5
    assert tf.tolist() == [5, 1999, 0, 0, 4100] and tl.tolist() == [1004, 1999, 0, 0, 4990]   # [5,1005) holds 5, 7, 1003, 1004
    for s in range(len(idx) - 1):                                             # == the reference's t_us[sl.start] / t_us[sl.stop - 1]
        if idx[s + 1] > idx[s]:
            assert tf[s] == t[idx[s]] and tl[s] == t[idx[s + 1] - 1]
    assert nd.band_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]
    xb, yb, pb, tb, sel = nd.events_in_band(x, y, p, t, 4, 10)
    assert sel.tolist() == [1, 3, 5, 7] and yb.tolist() == [5, 4, 5, 3]
    ib = nd.band_slice_bounds(t, tb, 1000)
    assert len(ib) == len(idx) and ib.tolist() == [0, 2, 2, 2, 2, 4][:len(idx)]
    # a band's own first event of slice 0 (7) differs from the stream's (5): the table exists for this reason
    assert tb[ib[0]] == 7 != tf[0]

