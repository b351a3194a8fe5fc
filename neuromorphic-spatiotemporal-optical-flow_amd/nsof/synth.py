"""Deterministic synthetic inputs: textured frame pairs with a known rigid
motion, and event streams with the reference's ``/CD/events`` schema.  Pure numpy."""
import numpy as np


def _gauss_blur(img, sigma):
    r = int(np.ceil(3 * sigma))
    k = np.exp(-np.arange(-r, r + 1, dtype=np.float64) ** 2 / (2 * sigma * sigma))
    k /= k.sum()
    out = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 1, img)
    return np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 0, out)


def _bilinear(img, yy, xx):
    h, w = img.shape
    x0 = np.clip(np.floor(xx).astype(np.int64), 0, w - 2)
    y0 = np.clip(np.floor(yy).astype(np.int64), 0, h - 2)
    fx = np.clip(xx - x0, 0.0, 1.0)
    fy = np.clip(yy - y0, 0.0, 1.0)
    return ((1 - fy) * ((1 - fx) * img[y0, x0] + fx * img[y0, x0 + 1]) +
            fy * ((1 - fx) * img[y0 + 1, x0] + fx * img[y0 + 1, x0 + 1]))


def make_pair(seed, height=1080, width=1920, shift=(2.5, -1.25), rot_deg=0.2, sigma=3.0):
    """uint8 frames (prev, next) with next(x+u, y+v) = prev(x, y) for the rigid motion
    (u, v) = shift + rotation by ``rot_deg`` about the image centre."""
    pad = 24
    rng = np.random.default_rng(seed)
    base = _gauss_blur(rng.random((height + 2 * pad, width + 2 * pad)), sigma)
    base = (base - base.min()) / (base.max() - base.min()) * 255.0
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    prev = base[pad:pad + height, pad:pad + width]
    # inverse map: next(p) = prev(R^-1 (p - c - shift) + c)
    th = np.deg2rad(rot_deg)
    cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
    dx, dy = xx - cx - shift[0], yy - cy - shift[1]
    sx = np.cos(th) * dx + np.sin(th) * dy + cx
    sy = -np.sin(th) * dx + np.cos(th) * dy + cy
    nxt = _bilinear(base, sy + pad, sx + pad)
    to_u8 = lambda a: np.ascontiguousarray(np.clip(np.rint(a), 0, 255).astype(np.uint8))  # noqa: E731
    return to_u8(prev), to_u8(nxt)


def true_flow(height, width, shift=(2.5, -1.25), rot_deg=0.2):
    """The (u, v) field make_pair() applies, float64 (H, W, 2)."""
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    th = np.deg2rad(rot_deg)
    cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
    dx, dy = xx - cx, yy - cy
    u = np.cos(th) * dx - np.sin(th) * dy + cx + shift[0] - xx
    v = np.sin(th) * dx + np.cos(th) * dy + cy + shift[1] - yy
    return np.stack([u, v], -1)


def make_events(seed, width=1280, height=720, n_background=200_000, duration_us=1_000_000, box=(120, 80),
                speed_pps=400.0, p_values=(0, 1)):
    """Event stream (x int16, y int16, p int8, t int64 sorted): uniform background plus the leading /
    trailing edges of a box drifting left to right (schema of event_mem_sim.py:69-75, :359-364)."""
    rng = np.random.default_rng(seed)
    x = rng.integers(0, width, n_background)
    y = rng.integers(0, height, n_background)
    p = rng.choice(np.asarray(p_values), n_background)
    t = rng.integers(0, duration_us, n_background)
    bw, bh = box
    y0 = (height - bh) // 2
    step_us = 500
    ts = np.arange(0, duration_us, step_us)
    lead = (ts * 1e-6 * speed_pps).astype(np.int64) + bw
    keep = lead < width
    ts, lead = ts[keep], lead[keep]
    rows = np.arange(y0, y0 + bh)
    ex = np.concatenate([np.repeat(lead, bh), np.repeat(np.maximum(lead - bw, 0), bh)])
    ey = np.concatenate([np.tile(rows, lead.size), np.tile(rows, lead.size)])
    ep = np.concatenate([np.full(lead.size * bh, p_values[-1]), np.full(lead.size * bh, p_values[0])])
    et = np.concatenate([np.repeat(ts, bh), np.repeat(ts, bh)])
    x, y, p, t = np.concatenate([x, ex]), np.concatenate([y, ey]), np.concatenate([p, ep]), np.concatenate([t, et])
    order = np.argsort(t, kind="stable")
    return (x[order].astype(np.int16), y[order].astype(np.int16), p[order].astype(np.int8),
            t[order].astype(np.int64))


def make_event_stream_4k(seed=5, width=3840, height=2160, n_events=1_000_000, duration_us=1_000_000,
                         window=(400, 300), window_share=0.3, speed_pps=600.0):
    """The roofline-run stream of SURVEY.md section 8d config 5: ``n_events`` events over ``duration_us`` on a
    ``width`` x ``height`` sensor, sorted by time; x, y uniform except for ``window_share`` of the events, which fall
    into a ``window`` drifting left to right at ``speed_pps`` px/s; p ~ Bernoulli(0.5).  Schema of /CD/events."""
    rng = np.random.default_rng(seed)
    t = np.sort(rng.integers(0, duration_us, n_events)).astype(np.int64)
    x = rng.integers(0, width, n_events)
    y = rng.integers(0, height, n_events)
    inside = rng.random(n_events) < window_share
    ww, wh = window
    x0 = (t * 1e-6 * speed_pps).astype(np.int64) % max(width - ww, 1)
    y0 = (height - wh) // 2
    x = np.where(inside, x0 + rng.integers(0, ww, n_events), x)
    y = np.where(inside, y0 + rng.integers(0, wh, n_events), y)
    p = (rng.random(n_events) < 0.5).astype(np.int8)
    return x.astype(np.int16), y.astype(np.int16), p, t
