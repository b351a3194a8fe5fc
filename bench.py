#!/usr/bin/env python3
"""Headline benchmark: 1080p frame-pairs/s of the HIP Farneback path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one pass of the hot path over one batch of `--pairs` synthetic 1920x1080 frame pairs per
GPU (inputs already resident in HBM, flow left in HBM), Farneback parameter set A of the reference
(pyr_scale .5, levels 3, winsize 15, iterations 3, poly_n 5, poly_sigma 1.2, flags 0 --
/root/reference/optical_flow_seg.py:73-81 and data/grasp/Parameters.txt).  Independent pairs are
sharded over ranks with no data-path collective (weak scaling: fixed pairs per GPU); RCCL is used only
for the barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
  roofline     -- dominant kernel by summed device time, timed live with HIP events on the launch
                  stream inside the timed region; algorithmic bytes per DESIGN.md section 5
  roofline_polyexp -- the same for the polynomial-expansion kernel (the north-star kernel)
  cpu_baseline -- the CPU oracle (oracle/farneback_ref.c, 1 thread) on a bounded sample of the same pairs
  max_abs_epe_vs_oracle -- GPU flow vs oracle flow on that sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def synth_pairs_gpu(torch, dev, n, h, w, seed0):
    """n distinct textured u8 frame pairs generated on the device (same recipe as nsof.synth.make_pair:
    blurred uniform noise, next = prev moved by a per-pair translation + 0.2 deg rotation)."""
    import math
    import torch.nn.functional as F  # noqa: N812
    pad = 24
    g = torch.Generator(device=dev)
    prevs = torch.empty((n, h, w), dtype=torch.uint8, device=dev)
    nexts = torch.empty((n, h, w), dtype=torch.uint8, device=dev)
    r = 9
    k = torch.exp(-torch.arange(-r, r + 1, device=dev, dtype=torch.float32) ** 2 / (2 * 3.0 * 3.0))
    kt = (k / k.sum()).tolist()
    ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32),
                            torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
    for i in range(n):
        g.manual_seed(seed0 + i)
        base = torch.rand((1, 1, h + 2 * pad, w + 2 * pad), generator=g, device=dev)
        # separable Gaussian as explicit tap sums (plain elementwise kernels: no MIOpen find / user database, which
        # eight ranks starting at once would all hit, and the same bits on every rank and box)
        padded = F.pad(base, (r, r, 0, 0), mode="reflect")
        acc = torch.zeros_like(base)
        for j in range(2 * r + 1):
            acc.add_(padded[..., j:j + base.shape[-1]], alpha=float(kt[j]))
        padded = F.pad(acc, (0, 0, r, r), mode="reflect")
        base = torch.zeros_like(acc)
        for j in range(2 * r + 1):
            base.add_(padded[..., j:j + acc.shape[-2], :], alpha=float(kt[j]))
        base = (base - base.min()) / (base.max() - base.min()) * 255.0
        u, v = 2.5 + 0.25 * (i % 5), -1.25 - 0.2 * (i % 3)
        th = math.radians(0.2)
        cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
        dx, dy = xs - cx - u, ys - cy - v
        sx = math.cos(th) * dx + math.sin(th) * dy + cx + pad
        sy = -math.sin(th) * dx + math.cos(th) * dy + cy + pad
        hh, ww = h + 2 * pad, w + 2 * pad
        grid = torch.stack([(sx + 0.5) / ww * 2 - 1, (sy + 0.5) / hh * 2 - 1], -1)[None]
        nxt = F.grid_sample(base, grid, mode="bilinear", padding_mode="border", align_corners=False)
        prevs[i] = base[0, 0, pad:pad + h, pad:pad + w].round().clamp(0, 255).to(torch.uint8)
        nexts[i] = nxt[0, 0].round().clamp(0, 255).to(torch.uint8)
    return prevs, nexts


def level_sizes(nsof, w, h, p):
    L = nsof.effective_levels(w, h, p.pyr_scale, p.levels)  # noqa: N806
    return [nsof.level_size(w, h, p.pyr_scale, k)[:2] for k in range(L + 1)]


def algorithmic_bytes_per_pair(nsof, w, h, p, fused_level0=True):
    """Per-kernel compulsory HBM bytes for ONE frame pair.
    fused_level0: the default path forms the full-resolution level image inside the expansion kernel (no pyramid launch
    at level 0, 1 B instead of 4 B read per pixel there); the float-expansion mode keeps the two kernels."""
    from nsof import _lib
    sizes = level_sizes(nsof, w, h, p)
    n0 = w * h
    out = {k: 0 for k in range(_lib.K_COUNT)}
    for k, (wk, hk) in enumerate(sizes):
        nk = wk * hk
        if k == 0 and fused_level0:
            out[_lib.K_POLYEXP] += 2 * 21 * nk                 # 1 B read (the frame) + 5x4 B written per pixel, x2 frames
        else:
            out[_lib.K_PREP] += 2 * (n0 + 4 * nk)              # u8 frame in, f32 level image out, x2 frames
            out[_lib.K_POLYEXP] += 2 * 24 * nk                 # 4 B read + 5x4 B written per pixel, x2 frames
        out[_lib.K_UPDMAT] += p.iterations * 68 * nk           # R0 20 + R1 20 + flow 8 -> M 20
        out[_lib.K_BLUR] += p.iterations * 28 * nk             # M 20 -> flow 8
        out[_lib.K_ITERATE] += p.iterations * 56 * nk          # fused: R0 20 + R1 20 + flow 8 -> flow 8
        if k + 1 < len(sizes):
            nk1 = sizes[k + 1][0] * sizes[k + 1][1]
            out[_lib.K_UPSAMPLE] += 8 * nk + 8 * nk1           # coarse flow in, fine flow out
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=256, help="frame pairs per GPU per step")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--params", choices=["A", "B", "C"], default="A")
    ap.add_argument("--cpu-sample", type=int, default=8, help="pairs timed on the CPU oracle (rank 0, N=1 only)")
    ap.add_argument("--e2e-pairs", type=int, default=256,
                    help="pairs of the host-to-host (PCIe-inclusive) leg, rank 0 at N=1 only; 0 = skip")
    ap.add_argument("--io", choices=["auto", "none", "gather"], default="auto",
                    help="gather: after the headline run, an extra leg in which rank 0 owns every frame pair: frames go "
                         "to the ranks and flow comes back point to point over RCCL/xGMI, chunked and overlapped with "
                         "compute (nsof.dist.run_sharded_overlapped); printed as 'io_gather' next to the headline")
    ap.add_argument("--io-timeout", type=float, default=240.0,
                    help="seconds after which the scatter/compute/gather leg is abandoned (the headline line is printed regardless)")
    ap.add_argument("--sharded-legs", action="store_true",
                    help="run the two N > 1 legs (config5_sharded, config4_sharded) at N = 1 as well (they are on by default for N > 1)")
    ap.add_argument("--config4-pairs", type=int, default=0,
                    help="frame pairs per dataset in the config4_sharded leg (0 = every pair the reference's loops walk: 360 pairs, 645 calls)")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the joined events -> accumulator -> flow leg (BASELINE config 5) and its accumulator record")
    ap.add_argument("--no-fast-leg", action="store_true",
                    help="skip the extra leg that times the opt-in float polynomial expansion (NSOF_OPT_POLYEXP_F32)")
    ap.add_argument("--no-param-legs", action="store_true",
                    help="skip the extra legs that time the reference's other two parameter sets (B, C) on the same frames")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--mode", choices=["pairs", "sequence"], default="pairs",
                    help="pairs (headline): independent frame pairs; sequence: pairs+1 consecutive frames, the "
                         "per-frame work is shared between neighbouring pairs (nsof_farneback_u8_sequence_dev)")
    args = ap.parse_args()

    # ONE JSON line on stdout, whatever libraries print: RCCL writes a version banner to stdout when a process
    # group comes up.  Everything else that lands on file descriptor 1 is sent to stderr; the JSON line goes to the
    # real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # see nsof/_lib.py: streams that share a hardware queue serialise

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    import __graft_entry__ as ge
    ge.build_native()
    import nsof  # noqa: E402  (preloads torch's HIP runtime, then libnsof.so)
    import torch
    from nsof import _lib
    from nsof.farneback import PARAMS_A, PARAMS_B, PARAMS_C

    # NSOF_BENCH_REHEARSAL=1: a dry run of the N > 1 code path on a box with fewer GPUs than ranks -- the ranks share
    # the devices that exist and the two collectives (barrier, max of the elapsed time) run over gloo on host tensors.
    # Its numbers mean nothing (the ranks time-share one GPU); real runs use one GPU per rank and RCCL.
    rehearsal = os.environ.get("NSOF_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ   # under torch.distributed.run: always RCCL
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    p = {"A": PARAMS_A, "B": PARAMS_B, "C": PARAMS_C}[args.params]
    h, w, n = args.height, args.width, args.pairs
    ctx = nsof.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    prevs, nexts = synth_pairs_gpu(torch, dev, n, h, w, 1234 + 1000 * rank)
    flow = torch.empty((n, h, w, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)

    if args.mode == "sequence":   # pairs+1 frames: frame i+1 = "next" of pair i = "prev" of pair i+1
        frames = torch.cat([prevs, nexts[-1:]], 0).contiguous()

    def step():
        if args.mode == "sequence":
            nsof.farneback_sequence(frames, flow, n + 1, h, w, p, ctx=ctx)
        else:
            nsof.farneback_batch(prevs, nexts, flow, n, h, w, p, ctx=ctx)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ids = [_lib.K_PREP, _lib.K_POLYEXP, _lib.K_UPSAMPLE, _lib.K_UPDMAT, _lib.K_BLUR, _lib.K_ITERATE]
    if not args.no_prof:
        ctx.prof_enable(*kernel_ids)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()   # the stream the steps ran on (= torch's current stream) + the kernels' hand-over error word: raises on a lost carry
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    barrier()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    prof = {}
    if not args.no_prof:
        for k in kernel_ids:
            prof[k] = ctx.prof_collect(k)
        ctx.prof_enable()

    exit_code = 0
    out = None
    wd = None
    import threading as _threading
    state = {"printed": False, "lock": _threading.Lock()}
    if rank == 0:
        total_pairs = n * world * args.steps
        alg = algorithmic_bytes_per_pair(nsof, w, h, p)

        def roof(kid, prof=prof, steps=args.steps, alg=alg):
            ms, launches = prof[kid]
            if not launches:
                return None
            bytes_total = alg[kid] * n * steps                # this rank's launches moved this many algorithmic bytes
            per_launch = bytes_total / launches
            gbs = bytes_total / (ms * 1e-3) / 1e9
            return {"kernel": _lib.load().nsof_kernel_name(kid).decode(), "bound": "hbm", "achieved": round(gbs, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "traffic": None, "avg_launch_us": round(ms * 1e3 / launches, 2), "launches": launches,
                    "algorithmic_bytes_per_launch": int(per_launch)}

        out = {
            "metric": "1080p frame-pairs/sec (Farneback, HIP)" if (w, h) == (1920, 1080) else f"{w}x{h} frame-pairs/sec",
            "value": round(total_pairs / elapsed, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{w}x{h} u8 frame pairs, Farneback params {args.params} "
                                   f"(pyr_scale={p.pyr_scale}, levels={p.levels}, winsize={p.winsize}, "
                                   f"iterations={p.iterations}, poly_n={p.poly_n}, poly_sigma={p.poly_sigma}, flags=0)",
                       "mode": args.mode, "pairs_per_gpu_per_step": n, "global_pairs_per_step": n * world,
                       "rowsum_order": ("library order (one running sum per image row, NSOF_OPT_EXACT_ROWSUMS=1, the default): "
                                        "every stage in the reference's operation order" if ctx.get_option(_lib.OPT_EXACT_ROWSUMS)
                                        else "per-pixel window sums (NSOF_OPT_EXACT_ROWSUMS=0, the opt-in fast mode)"),
                       "parallelism": f"pairs sharded over {world} rank(s), no data-path collective"},
        }
        if rehearsal:
            out["rehearsal"] = "NSOF_BENCH_REHEARSAL=1: ranks share the GPUs present, gloo collectives -- not a measurement"
        if prof:
            dom = max(kernel_ids, key=lambda k: prof[k][0])
            out["roofline"] = roof(dom)
            out["roofline_polyexp"] = roof(_lib.K_POLYEXP)
            if out["roofline_polyexp"] and prof[_lib.K_POLYEXP][1]:
                # The expansion kernel of level 0 also does the level-0 pyramid stage's work (no such launch any more).
                # frac above prices the FUSED kernel's compulsory bytes (21 B/px there); on the separate stages' bytes
                # (24 B/px expansion + the pyramid stage's 5 B/px of level 0) the same launches read:
                alg_sep = algorithmic_bytes_per_pair(nsof, w, h, p, fused_level0=False)
                sep = (alg_sep[_lib.K_POLYEXP] + 2 * 5 * w * h) * n * args.steps
                ms_p = prof[_lib.K_POLYEXP][0]
                out["roofline_polyexp"]["separate_stage_bytes"] = {
                    "bytes_per_launch": int(sep / prof[_lib.K_POLYEXP][1]), "achieved": round(sep / (ms_p * 1e-3) / 1e9, 1),
                    "unit": "GB/s", "frac": round(sep / (ms_p * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "24 B/px expansion + 5 B/px level-0 pyramid stage: what these launches replace"}
            out["kernel_ms_per_step"] = {_lib.load().nsof_kernel_name(k).decode(): round(prof[k][0] / args.steps, 3)
                                         for k in kernel_ids}
            tj = os.path.join(ROOT, "profiles", "hbm_traffic.json")  # PMC-derived HBM bytes (see its _doc)
            if os.path.exists(tj):
                with open(tj) as f:
                    tr = json.load(f).get("kernels", {})
                for key in ("roofline", "roofline_polyexp"):
                    ent = tr.get(out[key]["kernel"]) if out.get(key) else None
                    if ent is not None:   # NOT counters of this run: a stored PMC ratio applied to this run's bytes
                        out[key]["traffic"] = int(out[key]["algorithmic_bytes_per_launch"] *
                                                  ent["traffic_over_algorithmic"])
                        out[key]["traffic_source"] = ("estimated: algorithmic bytes x PMC ratio "
                                                      f"{ent['traffic_over_algorithmic']} from profiles/hbm_traffic.json "
                                                      "(rocprofv3 --pmc passes over this bench's own launches, all "
                                                      "levels: scripts/prof_traffic_bench.sh, see its _doc)")
    # The legs every rank takes part in: scatter / compute / gather (rank 0 owns the pairs), BASELINE config 5 sharded
    # (accumulator row bands -> all-gather of the surface frames -> sharded pairs) and BASELINE config 4 sharded (the five
    # datasets' call list dealt round-robin).  The headline above is already measured: a watchdog makes sure it is printed
    # even if one of them -- RCCL traffic that a one-GPU box cannot rehearse -- hangs or fails on some rank (the line then
    # carries the error in place of that leg's numbers, and every rank leaves with exit code 4).
    import threading

    def arm_watchdog(leg):
        # what the watchdog prints: a private copy of what has been measured so far (the main thread keeps adding to `out`)
        snapshot = json.loads(json.dumps(out)) if rank == 0 else None

        def emergency():
            with state["lock"]:
                if rank == 0 and not state["printed"]:
                    snapshot[leg] = {"error": f"the {leg} leg did not finish within {args.io_timeout} s", "exit_code": 4}
                    os.write(real_stdout, (json.dumps(snapshot) + "\n").encode())
                    state["printed"] = True
                os._exit(4)

        t = threading.Timer(args.io_timeout, emergency)
        t.daemon = True
        t.start()
        return t

    def run_leg(key, fn):
        nonlocal wd
        if wd is not None:
            wd.cancel()
        wd = arm_watchdog(key)
        try:
            rec = fn()
        except Exception as e:   # noqa: BLE001 -- reported in the line; peers stuck in a collective are ended by their watchdogs
            rec = {"error": f"{type(e).__name__}: {e}"[:400]}
        if world == 1:
            wd.cancel()           # N > 1: stays armed until the next leg / the process group is torn down
        if rank == 0 and rec:
            out[key] = rec

    if (args.io == "gather" or (args.io == "auto" and world > 1)) and args.mode == "pairs":
        run_leg("io_gather", lambda: io_gather_leg(nsof, torch, dist if use_dist else None, ctx, p, dev, rank, world, n, h, w,
                                                   prevs, nexts))
    if (world > 1 or args.sharded_legs) and args.mode == "pairs":
        run_leg("config5_sharded", lambda: config5_sharded_leg(nsof, torch, dist if use_dist else None, ctx, dev, rank, world, rehearsal))
        run_leg("config4_sharded", lambda: config4_sharded_leg(nsof, torch, dist if use_dist else None, ctx, dev, rank, world, rehearsal,
                                                               args.config4_pairs))
    if rank == 0:
        if world == 1 and prof and args.mode == "pairs" and not args.no_fast_leg:
            out.update(fast_polyexp_leg(nsof, _lib, ctx, torch, p, prevs, nexts, flow, n, h, w, alg, roof_of=roof,
                                        prof=prof, steps=max(2, min(args.steps, 5))))
        if world == 1 and args.mode == "pairs" and not args.no_fast_leg:
            out.update(fast_rowsums_leg(nsof, _lib, ctx, torch, p, prevs, nexts, flow, n, h, w, max(2, min(args.steps, 5))))
        if world == 1 and args.mode == "pairs" and not args.no_param_legs:
            out["real_frames"] = real_frames_leg(nsof, _lib, ctx)
        if world == 1 and prof and args.mode == "pairs" and args.params == "A" and not args.no_param_legs:
            for name, q in (("B", PARAMS_B), ("C", PARAMS_C)):
                out["params_" + name] = params_leg(nsof, _lib, ctx, torch, q, prevs, nexts, flow, min(n, 128), h, w,
                                                   kernel_ids)
        if world == 1 and not args.no_config5 and args.mode == "pairs":
            out.update(config5_leg(nsof, torch, local_rank))
            out.update(config3_leg(nsof, torch, local_rank))
        if world == 1 and args.e2e_pairs > 0 and args.mode == "pairs":
            out.update(e2e_leg(nsof, p, prevs, nexts, flow, min(args.e2e_pairs, n), local_rank))
        if world == 1 and args.mode == "pairs" and not args.no_param_legs:
            out["sequence"] = sequence_leg(nsof, ctx, torch, p, prevs, nexts, flow, n, h, w, max(2, min(args.steps, 5)))
        if world == 1 and args.mode == "pairs" and not args.no_param_legs:
            out["single_call"] = single_call_leg(nsof, _lib, ctx, torch, p, prevs, nexts, h, w)
        if world == 1 and args.cpu_sample > 0 and args.mode == "pairs":
            out.update(cpu_leg(nsof, p, prevs, nexts, flow, min(args.cpu_sample, n)))
        # one verdict over every parity record of the line: the default mode must stay within 1e-4 everywhere
        for key in ("real_frames", "config5", "config3", "config5_sharded"):
            if isinstance(out.get(key), dict) and out[key].get("parity_ok") is False:
                out["parity_ok"] = False
        sys.stdout.flush()
        with state["lock"]:
            state["printed"] = True
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        if out.get("parity_ok") is False:
            print(f"bench: GPU flow differs from the CPU baseline by more than {out.get('epe_tolerance', 1e-4)} "
                  "(headline batch, real frames, config 3 or config 5)", file=sys.stderr)
            exit_code = 3
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if wd is not None:
        wd.cancel()
    ctx.close()
    if exit_code:
        sys.exit(exit_code)


def fast_polyexp_leg(nsof, _lib, ctx, torch, p, prevs, nexts, flow, n, h, w, alg, roof_of, prof, steps):
    """The opt-in tolerance mode of the north-star kernel (NSOF_OPT_POLYEXP_F32: float instead of double horizontal
    accumulation): its roofline on the same batch, the whole-step rate with it, and how far the flow moves from the
    default exact path (max-abs end-point difference over ALL pairs of the batch, computed on the device)."""
    flow_fast = torch.empty_like(flow)
    ctx.set_option(_lib.OPT_POLYEXP_F32, 1)
    try:
        nsof.farneback_batch(prevs, nexts, flow_fast, n, h, w, p, ctx=ctx)
        torch.cuda.synchronize()
        ctx.prof_enable(_lib.K_POLYEXP)
        t0 = time.perf_counter()
        for _ in range(steps):
            nsof.farneback_batch(prevs, nexts, flow_fast, n, h, w, p, ctx=ctx)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, launches = ctx.prof_collect(_lib.K_POLYEXP)
        ctx.prof_enable()
    finally:
        ctx.set_option(_lib.OPT_POLYEXP_F32, 0)
    prof = dict(prof)
    prof[_lib.K_POLYEXP] = (ms, launches)
    r = roof_of(_lib.K_POLYEXP, prof, steps, algorithmic_bytes_per_pair(nsof, w, h, p, fused_level0=False))
    diff = float((flow_fast - flow).abs().max().item())
    per_pair = (flow_fast - flow).abs().amax(dim=(1, 2, 3))
    return {"roofline_polyexp_fast": r, "fast_mode": {
        "option": "NSOF_OPT_POLYEXP_F32=1 (opt-in; default path is the exact one)", "value": round(n * steps / dt, 2),
        "unit": "pairs/s", "max_abs_epe_vs_exact_path": diff,
        "pairs_above_1e-4": int((per_pair > 1e-4).sum().item()), "pairs": n}}


def params_leg(nsof, _lib, ctx, torch, q, prevs, nexts, flow, k, h, w, kernel_ids, steps=3):
    """The reference's other parameter sets on the first k pairs of the same batch (B: autodriving / tabletennis /
    uavnew2, C: uav -- /root/reference/data/*/Parameters.txt): whole-step rate, per-kernel time, the roofline of the
    two dominant kernels on their own algorithmic bytes, and the first pair checked against the CPU oracle."""
    out_flow = torch.empty_like(flow[:k])   # `flow` keeps the headline result: the CPU leg checks it later
    nsof.farneback_batch(prevs, nexts, out_flow, k, h, w, q, ctx=ctx)
    torch.cuda.synchronize()
    ctx.prof_enable(*kernel_ids)
    t0 = time.perf_counter()
    for _ in range(steps):
        nsof.farneback_batch(prevs, nexts, out_flow, k, h, w, q, ctx=ctx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = {kid: ctx.prof_collect(kid) for kid in kernel_ids}
    ctx.prof_enable()
    alg = algorithmic_bytes_per_pair(nsof, w, h, q)
    name = _lib.load().nsof_kernel_name
    rec = {"workload": f"{w}x{h}, pyr_scale={q.pyr_scale}, levels={q.levels}, winsize={q.winsize}, "
                       f"iterations={q.iterations}, poly_n={q.poly_n}, poly_sigma={q.poly_sigma}",
           "value": round(k * steps / dt, 2), "unit": "pairs/s", "pairs": k,
           "kernel_ms_per_step": {name(kid).decode(): round(prof[kid][0] / steps, 3) for kid in kernel_ids},
           "roofline": {}}
    for kid in (_lib.K_ITERATE, _lib.K_POLYEXP, _lib.K_PREP):
        ms, launches = prof[kid]
        if launches:
            gbs = alg[kid] * k * steps / (ms * 1e-3) / 1e9
            rec["roofline"][name(kid).decode()] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                                                   "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                                   "avg_launch_us": round(ms * 1e3 / launches, 2)}
    if name(_lib.K_PREP).decode() in rec["roofline"]:
        rec["roofline"][name(_lib.K_PREP).decode()]["note"] = (
            "levels 1..L only (the generic-scale walk kernels, each re-reading the full frame): level 0 has no pyramid "
            "launch any more, the expansion kernel forms its image from the frame")
    fused = sum(alg[kid] for kid in kernel_ids if prof[kid][1])
    rec["fused_algorithmic_GBps"] = round(fused * k * steps / dt / 1e9, 1)
    rec["frac_of_hbm_peak"] = round(fused * k * steps / dt / 1e9 / HBM_PEAK_GBS, 4)
    from oracle import oracle as O  # noqa: N812  (the checker; only ever used in the CPU legs, tests and smoke())
    O.build()
    qa = [getattr(q, a) for a in ("pyr_scale", "levels", "winsize", "iterations", "poly_n", "poly_sigma", "flags")]
    ref = O.farneback(prevs[0].cpu().numpy(), nexts[0].cpu().numpy(), *qa)
    rec["max_abs_epe_vs_oracle_pair0"] = float(abs(out_flow[0].cpu().numpy() - ref).max())
    del out_flow
    torch.cuda.empty_cache()
    return rec


def sequence_leg(nsof, ctx, torch, p, prevs, nexts, flow, n, h, w, steps):
    """The reference's own workload shape: CONSECUTIVE frames of one video (/root/reference/optical_flow_seg.py walks
    frame i -> i+1), through nsof_farneback_u8_sequence_dev -- pyramid and polynomial expansion are computed once
    per frame instead of once per pair side.  n+1 frames -> n flow fields; flow i must equal the pair call on
    (frame i, frame i+1)."""
    frames = torch.cat([prevs, nexts[-1:]], 0).contiguous()   # frame i+1 = 'next' of pair i only for the last one ...
    out = torch.empty_like(flow)
    nsof.farneback_sequence(frames, out, n + 1, h, w, p, ctx=ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        nsof.farneback_sequence(frames, out, n + 1, h, w, p, ctx=ctx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # ... so the check uses the last pair, whose two frames are the batch's last (prev, next)
    same = bool(torch.equal(out[n - 1], flow[n - 1]))
    return {"value": round(n * steps / dt, 2), "unit": "pairs/s", "frames": n + 1,
            "path": "nsof_farneback_u8_sequence_dev (per-frame stages shared between neighbouring pairs)",
            "last_pair_identical_to_pair_call": same}


def single_call_leg(nsof, _lib, ctx, torch, p, prevs, nexts, h, w, reps=20):
    """What a caller that replaces ``cv2.calcOpticalFlowFarneback`` one frame at a time sees (the reference's scripts
    call it once per frame, /root/reference/optical_flow_seg.py:202-206): milliseconds per lone call, host numpy in
    -> host numpy out and device-resident, in the default mode (library row-sum order), with the opt-in per-pixel row sums
    (NSOF_OPT_EXACT_ROWSUMS=0) and with that mode's row bands (NSOF_OPT_ROW_BANDS)."""
    import numpy as np
    hp, hn = prevs[0].cpu().numpy(), nexts[0].cpu().numpy()
    kw = p.as_kwargs()
    one = torch.empty((1, h, w, 2), dtype=torch.float32, device=prevs.device)
    ids = [_lib.K_PREP, _lib.K_POLYEXP, _lib.K_UPSAMPLE, _lib.K_ITERATE]
    name = _lib.load().nsof_kernel_name
    rec, flows = {}, {}
    for key, bands, exact in (("default", 0, 1), ("fast_rowsums", 0, 0), ("row_bands", 1, 0)):
        ctx.set_option(_lib.OPT_ROW_BANDS, bands)
        ctx.set_option(_lib.OPT_EXACT_ROWSUMS, exact)   # row bands belong to the fast row-sum mode
        try:
            for _ in range(3):
                flows[key] = nsof.calcOpticalFlowFarneback(hp, hn, None, **kw, ctx=ctx)
            t0 = time.perf_counter()
            for _ in range(reps):
                nsof.calcOpticalFlowFarneback(hp, hn, None, **kw, ctx=ctx)
            host_ms = (time.perf_counter() - t0) / reps * 1e3
            nsof.farneback_batch(prevs, nexts, one, 1, h, w, p, ctx=ctx)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                nsof.farneback_batch(prevs, nexts, one, 1, h, w, p, ctx=ctx)
            torch.cuda.synchronize()
            dev_ms = (time.perf_counter() - t0) / reps * 1e3
            ctx.prof_enable(*ids)
            nsof.farneback_batch(prevs, nexts, one, 1, h, w, p, ctx=ctx)
            torch.cuda.synchronize()
            parts = {name(k).decode(): round(ctx.prof_collect(k)[0], 3) for k in ids}
            ctx.prof_enable()
        finally:
            ctx.set_option(_lib.OPT_ROW_BANDS, 0)
            ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 1)
        rec[key] = {"host_to_host_ms": round(host_ms, 3), "device_resident_ms": round(dev_ms, 3), "kernel_ms": parts}
    # the same lone call forced through the fused strip walker (NSOF_OPT_SMALL_BATCH_JOBS=0): what the small-batch form buys
    ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, 0)
    try:
        for _ in range(3):
            fx = nsof.calcOpticalFlowFarneback(hp, hn, None, **kw, ctx=ctx)
        t0 = time.perf_counter()
        for _ in range(reps):
            nsof.calcOpticalFlowFarneback(hp, hn, None, **kw, ctx=ctx)
        rec["default_fused_kernel_only"] = {"host_to_host_ms": round((time.perf_counter() - t0) / reps * 1e3, 3),
                                            "option": "NSOF_OPT_SMALL_BATCH_JOBS=0",
                                            "max_abs_vs_default": float(np.abs(fx - flows["default"]).max())}
    finally:
        ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, 64)
    rec["default"]["form"] = ("three wide kernels per iteration (calls with <= 64 strip jobs, NSOF_OPT_SMALL_BATCH_JOBS), "
                              "bit-identical to the fused kernel")
    from nsof import synth
    from nsof.farneback import PARAMS_B
    bp, bn = synth.make_pair(5, 801, 801)
    kb = PARAMS_B.as_kwargs()
    for _ in range(3):
        nsof.calcOpticalFlowFarneback(bp, bn, None, **kb, ctx=ctx)
    t0 = time.perf_counter()
    for _ in range(reps):
        nsof.calcOpticalFlowFarneback(bp, bn, None, **kb, ctx=ctx)
    rec["default_801x801_params_B"] = {"host_to_host_ms": round((time.perf_counter() - t0) / reps * 1e3, 3)}
    rec["row_bands"]["option"] = "NSOF_OPT_ROW_BANDS=1 (opt-in; column sums restart per band)"
    rec["row_bands"]["max_abs_vs_default"] = float(np.abs(flows["row_bands"] - flows["default"]).max())
    rec["workload"] = f"one {w}x{h} pair per call"
    return rec


def config3_leg(nsof, torch, local_rank):
    """BASELINE config 3 joined, on one GPU (1280x720 synthetic stream, 1 ms slices, scheme 1 split,
    active -6 V): events -> surface frames + gating maps every 33 slices -> device ROI rectangles -> Farneback (params A) on
    every ROI crop of consecutive surface frames as one work list (pipeline.events_to_roi_flows).  Two streams: the
    SURVEY stream as defined (200 k background events: every 20x20 block holds a recent event, so the gate opens the full
    frame) and two sparse variants (silent 0.5 V so idle devices decay below the threshold): 5 k background events -- a few
    hundred small ROIs per map, the many-small-crops stress of the work list -- and 300 background events, where the gate
    leaves the moving box and little else.  Parity: the first flow canvas against the oracle chain
    (oracle/accum_ref.c -> uint8(255 w) / block currents -> host gating mirror -> oracle/farneback_ref.c per crop)."""
    import numpy as np
    from nsof import gating, pipeline, synth
    from nsof.farneback import PARAMS_A
    from oracle import oracle as O  # noqa: N812
    H, W, every, ms = 720, 1280, 33, 20  # noqa: N806
    pa = [getattr(PARAMS_A, kk) for kk in ("pyr_scale", "levels", "winsize", "iterations", "poly_n", "poly_sigma", "flags")]
    # bug_compatible=False: pair (k, k+1) is gated by the map of frame k+1, as opticalFlow3D is written and as rounds 2-3
    # measured this leg (the library's default follows the shipped scripts, memimg2 := memimg1 -> frame k; both are tested:
    # tests/test_gating.py::test_config3_roi_flow_pipeline_vs_oracle_chain)
    cfg = gating.GatingConfig(MEMSIZE=ms, EXTEND_HEIGHT_UPPER=20, EXTEND_HEIGHT_LOWER=20, EXTEND_WIDTH_LEFT=20,
                              EXTEND_WIDTH_RIGHT=20, THRES=240, FLAG=1, farneback_params=PARAMS_A, bug_compatible=False)
    _, _, _, usable = O.host_cpu()
    nt = min(usable, 16)
    rec = {"workload": f"events -> scheme-1 surface -> 8-bit frames + gating maps every {every} slices -> device ROI -> "
                       f"Farneback A per ROI crop, {W}x{H}, one GPU; MEMSIZE {ms}, THRES 240, FLAG 1, extend 20 px",
           "epe_tolerance": 1e-4, "streams": {}}
    ok_all = True
    with nsof.Context(local_rank) as c:
        for name, n_bg, silent in (("survey_200k_background", 200_000, 0.0), ("sparse_5k_background", 5_000, 0.5),
                                     ("box_only_300_background", 300, 0.5)):
            x, y, p, t = synth.make_events(2024, W, H, n_background=n_bg)
            tm = {}
            kw = dict(slice_us=1000, active_v=-6.0, silent_v=silent, snapshot_every=every, ctx=c, timings=tm, max_rects=256)
            pipeline.events_to_roi_flows(x, y, p, t, (H, W), cfg, **kw)           # warm-up
            frames, rects, flows = pipeline.events_to_roi_flows(x, y, p, t, (H, W), cfg, **kw)
            n_fr = tm["frames"]
            total = tm["surface_and_gating_s"] + tm["flow_s"]
            # first pair against the oracle chain
            ws = [O.accum_slices_per_s(x, y, t, H, W, 1000, -6.0, silent, n_slices=(k + 1) * every, n_threads=nt)[1] for k in range(2)]
            f = [(np.float32(255.0) * w).astype(np.uint8) for w in ws]
            gfr = frames[:2].cpu().numpy()
            frames_equal = bool(np.array_equal(gfr[0], f[0]) and np.array_equal(gfr[1], f[1]))
            gi = 0 if cfg.bug_compatible else 1   # which frame's map gates the pair (the scripts: the first, seg.py:435)
            g = gating.current_to_gray(pipeline.surface_to_block_current(O.accum_resistance(ws[gi]), ms))
            tp = gating.update_transition_pic(g, np.zeros_like(g, dtype=np.float64), cfg.THRES).astype(np.uint8)
            n, _, stats, _ = gating.connectedComponentsWithStats(tp, cfg.CONNECT)
            want = [gating._roi(*[int(v) for v in stats[i, :4]], W, H, ms, ms, cfg) for i in range(1, n)]
            canvas = np.zeros((H, W, 2), np.float32)
            for (x0, y0, x1, y1) in want:
                canvas[y0:y1, x0:x1] = O.farneback(np.ascontiguousarray(f[0][y0:y1, x0:x1]),
                                                   np.ascontiguousarray(f[1][y0:y1, x0:x1]), *pa)
            got = flows[0].cpu().numpy()
            d = float(np.abs(got - canvas).max())
            ok = frames_equal and rects[gi] == want and d < 1e-4
            ok_all &= ok
            rec["streams"][name] = {
                "events": int(len(t)), "silent_v": silent, "frames": n_fr, "roi_calls": tm["roi_calls"],
                "roi_pixel_fraction": round(tm["roi_pixels"] / float((n_fr - 1) * H * W), 4),
                "surface_and_gating_ms": round(tm["surface_and_gating_s"] * 1e3, 2), "flow_ms": round(tm["flow_s"] * 1e3, 2),
                "flow_fields_per_s": round((n_fr - 1) / total, 1), "x_realtime": round(n_fr * every * 1e-3 / total, 1),
                "first_pair_vs_oracle_chain": {"surface_frames_equal": frames_equal, "rects_equal": bool(rects[gi] == want), "gated_by": "first frame of the pair (bug-compatible)" if gi == 0 else "second frame",
                                               "rects": len(want), "max_abs_epe_vs_oracle": d,
                                               "bit_identical": bool(np.array_equal(got, canvas))},
                "parity_ok": bool(ok)}
    rec["parity_ok"] = bool(ok_all)
    return {"config3": rec}


def config5_leg(nsof, torch, local_rank):
    """BASELINE config 5 joined, on one GPU: synthetic 3840x2160 stream @ 1 M events/s ->
    dense scheme-1 accumulator update of every 1 ms slice (events uploaded once) -> every 33 slices the surface as an
    8-bit frame -> Farneback (params A) between consecutive surface frames; nothing leaves HBM in between.
    "accumulator": slices/s with the roofline on SURVEY's definition (8 B/px/slice + 16 B/event: one read + one
    write of w per slice) and on the bytes the fused design actually has to move (per 33 slices ONE pass: w read + w
    write + two slice-mask words read = 16 B/px, the 8-bit frame written by the same pass 1 B/px, 4 B/event), the CPU oracle's rate
    for the same slices (1 thread / all cores) and the state parity after the sampled slices."""
    import numpy as np
    from nsof import pipeline, synth
    from oracle import oracle as O  # noqa: N812
    H, W, every = 2160, 3840, 33  # noqa: N806
    x, y, p, t = synth.make_event_stream_4k()
    out = {}
    with nsof.Context(local_rank) as c:
        tm, tmd = {}, {}
        # the roofline run of the accumulator half: the every-pixel pass per interval (dense=True), 17 B/px per 33 slices
        pipeline.events_to_flow_sequence(x, y, p, t, (H, W), snapshot_every=every, ctx=c, timings=tmd, dense=True)   # warm-up
        frames_d, _ = pipeline.events_to_flow_sequence(x, y, p, t, (H, W), snapshot_every=every, ctx=c, timings=tmd, dense=True)
        # the pipeline as it runs by default: frames as copy + patch of the previous one (nsof_accum_run_frames)
        pipeline.events_to_flow_sequence(x, y, p, t, (H, W), snapshot_every=every, ctx=c, timings=tm)   # warm-up
        frames, flows = pipeline.events_to_flow_sequence(x, y, p, t, (H, W), snapshot_every=every, ctx=c, timings=tm)
        frames_same = bool(torch.equal(frames, frames_d))
        del frames_d
        finite = bool(torch.isfinite(flows).all().item())
        n_sl, n_fr = tm["slices"], tm["frames"]
        rate = n_sl / tm["accumulator_s"]
        npx, n_ev = H * W, int((t < n_sl * 1000 + t[0]).sum())
        survey_bytes = 8.0 * npx * n_sl + 16.0 * n_ev
        fused_bytes = (n_sl / every) * (16.0 + 1.0) * npx + 4.0 * n_ev
        rate_dense = n_sl / tmd["accumulator_s"]
        # the tile walk: frames are write-only (1 B/px per frame), w is read and written once per run, an event is read twice
        # (x, y: 4 B) by the bucketing pass and its 2-byte record written and read once
        patch_bytes = (n_sl / every) * 1.0 * npx + 8.0 * npx + 12.0 * n_ev
        # parity + CPU rate on a bounded sample: the first k slices through the CPU oracle
        k = 20
        model, logical, physical, usable = O.host_cpu()
        cpu1, w_ref = O.accum_slices_per_s(x, y, t, H, W, 1000, -6.0, 0.0, n_slices=k, n_threads=1)
        cpua, ta = 0.0, usable
        for nt in sorted({usable, min(usable, 16)}, reverse=True):   # see cpu_leg: a box's CPU share may be smaller
            v, _ = O.accum_slices_per_s(x, y, t, H, W, 1000, -6.0, 0.0, n_slices=k, n_threads=nt)
            if v > cpua:
                cpua, ta = v, nt
        acc = nsof.Accumulator(H, W, 1, "split", -6.0, 0.0, ctx=c, dense=True)
        idx = nsof.accumulator.slice_index_array(t, 1000)
        acc.set_events(x, y, p, t, idx)
        acc.run(0, k)
        werr = float(np.abs(acc.w() - w_ref).max())
        acc.close()
        out["accumulator"] = {
            "value": round(rate, 1), "unit": "slices/s", "workload": f"{W}x{H} sensor, 1 M events/s, 1 ms slices, scheme-1 "
            f"update of every slice + a surface frame every {every} slices; events uploaded once.  value = the default path "
            "(silent voltage in the dead zone: nsof_accum_run_frames as a tile-persistent walk -- a wave owns 1024 pixels for the "
            "whole run, state / frame bytes / slice masks in LDS, frames write-only, two launches per run); "
            "dense_roofline_run = the every-pixel pass per interval (dense=True), same frames",
            "frames_identical_to_dense_run": frames_same,
            "roofline_tile_walk_bytes": {"bytes_per_slice": round(patch_bytes / n_sl), "achieved": round(patch_bytes / tm["accumulator_s"] / 1e9, 1),
                                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(patch_bytes / tm["accumulator_s"] / 1e9 / HBM_PEAK_GBS, 3),
                                         "note": "1 B/px per frame written + w once per run + 12 B per event: what this form has to move"},
            "dense_roofline_run": {"value": round(rate_dense, 1), "unit": "slices/s",
                                   "roofline_fused_bytes": {"bytes_per_slice": round(fused_bytes / n_sl),
                                                            "achieved": round(fused_bytes / tmd["accumulator_s"] / 1e9, 1), "peak": HBM_PEAK_GBS,
                                                            "unit": "GB/s", "frac": round(fused_bytes / tmd["accumulator_s"] / 1e9 / HBM_PEAK_GBS, 3)}},
            "slices": n_sl, "events": n_ev, "x_realtime": round(rate / 1000.0, 1),
            "roofline_survey_definition": {"bytes_per_slice": round(survey_bytes / n_sl), "achieved": round(survey_bytes / tm["accumulator_s"] / 1e9, 1),
                                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(survey_bytes / tm["accumulator_s"] / 1e9 / HBM_PEAK_GBS, 3),
                                           "note": "8 B/px/slice + 16 B/event (one read + one write of w per slice): the state of an untouched "
                                                   "pixel is not moved at all here, so this exceeds 1"},
            "roofline_fused_bytes": {"bytes_per_slice": round(fused_bytes / n_sl), "achieved": round(fused_bytes / tmd["accumulator_s"] / 1e9, 1),
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fused_bytes / tmd["accumulator_s"] / 1e9 / HBM_PEAK_GBS, 3),
                                     "note": "the every-pixel pass (dense_roofline_run): 17 B/px per 33-slice interval"},
            "cpu_baseline": {"value": round(cpu1, 2), "value_all_cores": round(cpua, 2), "unit": "slices/s", "cores": 1,
                             "cores_all": ta, "cpu_model": model, "kind": "port",
                             "sample": f"first {k} slices, oracle/accum_ref.c (gcc -O3 -march=native, OpenMP over pixels; "
                                       f"best of {usable} and 16 threads)"},
            "max_abs_w_vs_oracle": werr, "w_tolerance": 5e-7, "parity_ok": bool(werr <= 5e-7)}
        # parity of the JOINED pipeline's first pair: accumulator oracle (33 / 66 slices) -> uint8(255 w) -> Farneback oracle.
        # These frames are the low-texture class (a flat field plus sparse dots).
        from nsof.farneback import PARAMS_A
        pa = [getattr(PARAMS_A, kk) for kk in ("pyr_scale", "levels", "winsize", "iterations", "poly_n", "poly_sigma", "flags")]
        best_nt = ta
        _, w1 = O.accum_slices_per_s(x, y, t, H, W, 1000, -6.0, 0.0, n_slices=every, n_threads=best_nt)
        _, w2 = O.accum_slices_per_s(x, y, t, H, W, 1000, -6.0, 0.0, n_slices=2 * every, n_threads=best_nt)
        f1, f2 = (np.float32(255.0) * w1).astype(np.uint8), (np.float32(255.0) * w2).astype(np.uint8)
        gfr = frames[:2].cpu().numpy()
        frames_equal = bool(np.array_equal(gfr[0], f1) and np.array_equal(gfr[1], f2))
        ref5 = O.farneback(f1, f2, *pa)
        d5 = np.abs(flows[0].cpu().numpy() - ref5).max(-1)
        fast5 = nsof.calcOpticalFlowFarneback(gfr[0], gfr[1], None, *pa, ctx=c, exact=False)
        d5f = np.abs(fast5 - ref5).max(-1)
        c5_ok = frames_equal and float(d5.max()) < 1e-4
        out["config5"] = {"workload": "events -> accumulator -> surface frames -> Farneback A at 3840x2160, one GPU",
                          "first_pair_vs_oracle_chain": {
                              "surface_frames_equal": frames_equal,
                              "default": {"max_abs_epe_vs_oracle": float(d5.max()), "pixels_above_1e-4": int((d5 > 1e-4).sum()),
                                          "bit_identical": bool(np.array_equal(flows[0].cpu().numpy(), ref5))},
                              "fast_rowsums": {"max_abs_epe_vs_oracle": float(d5f.max()), "pixels_above_1e-4": int((d5f > 1e-4).sum())},
                              "oracle": "oracle/accum_ref.c (33 and 66 slices) -> uint8(255 w) -> oracle/farneback_ref.c"},
                          "parity_ok": c5_ok, "epe_tolerance": 1e-4,
                          "surface_frames": n_fr, "accumulator_ms": round(tm["accumulator_s"] * 1e3, 2),
                          "flow_ms": round(tm["flow_s"] * 1e3, 2),
                          "flow_pairs_per_s_4k": round((n_fr - 1) / tm["flow_s"], 1),
                          "stream_seconds_per_wall_second": round((n_sl / 1000.0) / (tm["accumulator_s"] + tm["flow_s"]), 2),
                          "flow_finite": finite}
    return out


def _max_over_ranks(torch, dist, value, dev, rehearsal):
    if dist is None or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if (rehearsal or dist.get_backend() == "gloo") else dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _gather_objects(dist, obj, world):
    if dist is None or world == 1:
        return [obj]
    outs = [None] * world
    dist.all_gather_object(outs, obj)
    return outs


def config5_sharded_leg(nsof, torch, dist, ctx, dev, rank, world, rehearsal):
    """BASELINE config 5 over the ranks of the job (BASELINE.json quotes it on 8 MI355X): the synthetic 3840x2160 stream @
    1 M events/s -> accumulator ROW BANDS (every rank filters the replicated stream to its rows, global slice grid) ->
    ALL-GATHER of the 8-bit surface frames (the pipeline's one exchange; RCCL over xGMI) -> the pairs of consecutive
    frames sharded in contiguous chunks through nsof_farneback_u8_sequence_dev
    (/root/reference/eventsim/event_mem_sim.py:164-228 -> /root/reference/optical_flow_seg.py:203).  Records per-stage
    time (max over ranks), all-gather bytes and rate, stream-seconds per wall-second, and on rank 0 the first pair
    against the CPU oracle chain.  UNMEASURED on real multi-GPU hardware by the builder (one-GPU boxes): the numbers of
    the driver's node are the first."""
    import numpy as np
    from nsof import pipeline, synth
    from nsof.farneback import PARAMS_A
    from oracle import oracle as O  # noqa: N812
    H, W, every = 2160, 3840, 33  # noqa: N806
    x, y, p, t = synth.make_event_stream_4k()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    st = {}
    pipeline.events_to_flow_sequence_sharded(x, y, p, t, (H, W), snapshot_every=every, ctx=ctx)   # warm-up (workspaces, RCCL channels)
    barrier()
    t0 = time.perf_counter()
    (lo, hi), frames, flows = pipeline.events_to_flow_sequence_sharded(x, y, p, t, (H, W), snapshot_every=every, ctx=ctx, stats=st)
    barrier()
    wall = _max_over_ranks(torch, dist, time.perf_counter() - t0, dev, rehearsal)
    per_rank = _gather_objects(dist, {k: (round(v, 5) if isinstance(v, float) else v) for k, v in st.items()}, world)
    finite = bool(torch.isfinite(flows).all().item()) if flows is not None else True
    n_fr = st["frames"]
    rec = None
    if rank == 0:
        ag_s = max(r["allgather_s"] for r in per_rank)
        ag_bytes = per_rank[0]["allgather_bytes_received"]
        rec = {"workload": f"events -> accumulator row bands -> all-gather of {n_fr} surface frames -> sharded pairs, Farneback A at "
                           f"{W}x{H}, {world} rank(s)",
               "backend": (st["backend"] or "none") + (" (RCCL)" if st["backend"] == "nccl" else ""), "world_size": world,
               "hardware_note": "rehearsal: ranks share one GPU, gloo collectives -- not a measurement" if rehearsal else
                                ("one rank: nothing is exchanged" if world == 1 else "measured on this node"),
               "surface_frames": n_fr, "slices": n_fr * every, "wall_s": round(wall, 4),
               "stream_seconds_per_wall_second": round(n_fr * every / 1000.0 / wall, 2),
               "bands_ms_max": round(max(r["bands_s"] for r in per_rank) * 1e3, 2),
               "allgather_ms_max": round(ag_s * 1e3, 2), "allgather_bytes_received_per_rank": ag_bytes,
               "allgather_GBps_per_rank": round(ag_bytes / ag_s / 1e9, 2) if ag_bytes and ag_s > 0 else None,
               "flow_ms_max": round(max(r["flow_s"] for r in per_rank) * 1e3, 2),
               "pairs_per_rank": [r["pairs"][1] - r["pairs"][0] for r in per_rank],
               "band_rows_per_rank": [r["band_rows"] for r in per_rank], "flow_finite": finite}
        # parity of rank 0's first pair: accumulator oracle (33 / 66 slices) -> uint8(255 w) -> Farneback oracle
        pa = [getattr(PARAMS_A, kk) for kk in ("pyr_scale", "levels", "winsize", "iterations", "poly_n", "poly_sigma", "flags")]
        _, _, _, usable = O.host_cpu()
        nt = max(1, min(usable, 16) // max(1, min(world, 8)))   # the ranks of one host share its cores
        _, w1 = O.accum_slices_per_s(x, y, t, H, W, 1000, -6.0, 0.0, n_slices=every, n_threads=nt)
        _, w2 = O.accum_slices_per_s(x, y, t, H, W, 1000, -6.0, 0.0, n_slices=2 * every, n_threads=nt)
        f1, f2 = (np.float32(255.0) * w1).astype(np.uint8), (np.float32(255.0) * w2).astype(np.uint8)
        gfr = frames[:2].cpu().numpy()
        frames_equal = bool(np.array_equal(gfr[0], f1) and np.array_equal(gfr[1], f2))
        d = float(np.abs(flows[0].cpu().numpy() - O.farneback(f1, f2, *pa)).max()) if hi > lo else None
        rec["first_pair_vs_oracle_chain"] = {"surface_frames_equal": frames_equal, "max_abs_epe_vs_oracle": d,
                                             "oracle": "oracle/accum_ref.c -> uint8(255 w) -> oracle/farneback_ref.c"}
        rec["parity_ok"] = bool(frames_equal and (d is None or d < 1e-4))
    barrier()
    return rec


def config4_sharded_leg(nsof, torch, dist, ctx, dev, rank, world, rehearsal, pairs_per_dataset=0):
    """BASELINE config 4 over the ranks of the job: every flow call of the reference's evaluation loops over grasp +
    autodriving + uav + uavnew2 + tabletennis (gated ROI calls + full-frame calls, each dataset's Parameters.txt:
    /root/reference/data/*/Parameters.txt:1-26, /root/reference/optical_flow_seg.py:390-496; gating rectangles from the
    reference's .mat stacks, frames synthetic at the real sizes) dealt ROUND-ROBIN over the ranks (nsof.workload.shard_calls:
    independent calls, no data-path collective); each rank runs its share as work lists, host memory to host memory.
    Whole-job calls/s = calls / max-over-ranks time.  UNMEASURED on real multi-GPU hardware by the builder."""
    import numpy as np
    from nsof import workload as wl
    with np.load(os.path.join(ROOT, "tests", "golden", "gating_stacks.npz")) as z:
        stacks = {k: z[k] for k in z.files}
    t0 = time.perf_counter()
    calls, _ = wl.mixed_workload(stacks, pairs_per_dataset=pairs_per_dataset or None)
    t_build = time.perf_counter() - t0
    mine = wl.shard_calls(calls, rank, world)
    wl.run_calls(mine[:8], ctx=ctx)                               # warm-up (workspace, pinned staging)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    wl.run_calls(mine, ctx=ctx)
    dt = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    wall = _max_over_ranks(torch, dist, dt, dev, rehearsal)
    chk = float(sum(float(np.abs(c.flow[::7, ::7]).sum()) for c in mine[::5]))
    per_rank = _gather_objects(dist, {"calls": len(mine), "megapixels": round(sum(c.prev.size for c in mine) / 1e6, 1),
                                      "seconds": round(dt, 4), "flow_checksum": chk,
                                      "finite": bool(all(np.isfinite(c.flow).all() for c in mine[::5]))}, world)
    if rank != 0:
        return None
    mpx = sum(c.prev.size for c in calls) / 1e6
    backend = dist.get_backend() if dist is not None else "none"
    return {"workload": "config 4: grasp+autodriving+uav+uavnew2+tabletennis, gated ROI + full-frame calls, dealt round-robin",
            "backend": backend + (" (RCCL)" if backend == "nccl" else ""), "world_size": world,
            "hardware_note": "rehearsal: ranks share one GPU, gloo collectives -- not a measurement" if rehearsal else
                             ("one rank" if world == 1 else "measured on this node"),
            "calls": len(calls), "roi_calls": sum(c.kind == "roi" for c in calls), "megapixels": round(mpx, 1),
            "pairs_per_dataset": pairs_per_dataset or "all", "build_workload_s": round(t_build, 2),
            "wall_s_max_over_ranks": round(wall, 4), "calls_per_s": round(len(calls) / wall, 1), "mpx_per_s": round(mpx / wall, 1),
            "per_rank": per_rank, "flow_finite": bool(all(r["finite"] for r in per_rank))}


def io_gather_leg(nsof, torch, dist, ctx, p, dev, rank, world, n, h, w, prevs, nexts, chunk=32):
    """Rank 0 owns all world*n frame pairs (it replicates its own batch): scatter -> compute -> gather with
    nsof.dist.run_sharded_overlapped (RCCL send/recv per peer, chunks of `chunk` pairs, copies overlapped with
    compute).  Whole-job pairs/s including both transfers; at world == 1 nothing moves."""
    from nsof import dist as nd
    n_total = n * world
    if rank == 0:
        prev_all = prevs.repeat(world, 1, 1) if world > 1 else prevs
        next_all = nexts.repeat(world, 1, 1) if world > 1 else nexts
    else:
        prev_all = next_all = None

    def compute_into(pv, nx, o):
        nsof.farneback_batch(pv.contiguous(), nx.contiguous(), o, pv.shape[0], h, w, p, ctx=ctx)   # ctx runs on torch's stream

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if dist is None:      # single process: the same code path needs a (1-rank) process group
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import torch.distributed as td
    rec = {}
    try:
        # rank 0 is the sink of every flow field: time it with an equal share and, for N > 1, with half a share
        for key, share in (("equal_shares", 1.0),) + ((("src_half_share", 0.5),) if world > 1 else ()):
            st = {}
            nd.run_sharded_overlapped(prev_all, next_all, n_total, (h, w), dev, None, chunk=chunk, compute_into=compute_into,
                                      src_share=share)   # warm-up
            sync()
            t0 = time.perf_counter()
            out = nd.run_sharded_overlapped(prev_all, next_all, n_total, (h, w), dev, None, chunk=chunk,
                                            compute_into=compute_into, src_share=share, stats=st)
            sync()
            dt = time.perf_counter() - t0
            if td.get_world_size() > 1:
                tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if td.get_backend() == "gloo" else dev)
                td.all_reduce(tt, op=td.ReduceOp.MAX)
                dt = float(tt.item())
            if rank == 0:
                rec[key] = {"value": round(n_total / dt, 1), "unit": "pairs/s", "pairs_per_rank": st["pairs_per_rank"],
                            "GB_moved": round(st["bytes_moved"] / 1e9, 2),
                            "link_GBps_into_rank0": round(st["pairs_moved"] * 8 * h * w / dt / 1e9, 1),
                            "flow_checksum": float(out[::max(1, n_total // 8)].double().abs().sum().item())}
        backend, wsz = td.get_backend(), td.get_world_size()
    finally:
        if dist is None:
            tdist.destroy_process_group()
    if rank != 0:
        return None
    best = max(rec, key=lambda k_: rec[k_]["value"])
    return {"value": rec[best]["value"], "unit": "pairs/s", "pairs": n_total, "chunk_pairs": chunk, "best": best,
            "backend": backend + (" (RCCL)" if backend == "nccl" else ""), "world_size": wsz,
            "device": torch.cuda.get_device_name(dev), "variants": rec,
            "path": "rank 0 -> send/recv frames -> compute on every rank -> send/recv flow -> rank 0 (point to point, "
                    "three-stage pipeline over chunks, preallocated staging)",
            "bytes_moved_per_pair": 2 * h * w + 8 * h * w if world > 1 else 0}


def fast_rowsums_leg(nsof, _lib, ctx, torch, p, prevs, nexts, flow, n, h, w, steps):
    """NSOF_OPT_EXACT_ROWSUMS=0 (opt-in): each pixel's box-filter window summed directly instead of the library's running
    row sum -- the same numbers to ~1e-16 in double; invisible on these textured frames, up to ~8e-4 px at rank-deficient
    windows of real footage (see real_frames).  Whole-step rate on the same batch and the distance to the default result."""
    out = torch.empty_like(flow)
    ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 0)
    try:
        nsof.farneback_batch(prevs, nexts, out, n, h, w, p, ctx=ctx)
        torch.cuda.synchronize()
        ctx.prof_enable(_lib.K_ITERATE)
        t0 = time.perf_counter()
        for _ in range(steps):
            nsof.farneback_batch(prevs, nexts, out, n, h, w, p, ctx=ctx)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, launches = ctx.prof_collect(_lib.K_ITERATE)
        ctx.prof_enable()
    finally:
        ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 1)
    return {"fast_rowsums": {"option": "NSOF_OPT_EXACT_ROWSUMS=0 (opt-in; not the library's row-sum order)",
                             "value": round(n * steps / dt, 1), "unit": "pairs/s", "pairs": n, "steps": steps,
                             "iterate_avg_launch_us": round(ms * 1e3 / max(launches, 1), 2),
                             "max_abs_vs_default_path": float((out - flow).abs().max().item())}}


def real_frames_leg(nsof, _lib, ctx):
    """Parity on the reference's OWN frames (committed under tests/golden/): the 801x801 autodriving pair with parameter set
    B (3x3 windows: the rank-deficient case) and the 1080x1920 grasp pair with set A, per mode: max-abs end-point error
    against the CPU oracle and the number of pixels above 1e-4.  parity_ok is about the default mode."""
    import numpy as np
    from nsof import gating
    from nsof.farneback import PARAMS_A, PARAMS_B
    from oracle import oracle as O  # noqa: N812
    try:
        from PIL import Image
    except ImportError:
        return {"skipped": "PIL not importable"}
    G = os.path.join(ROOT, "tests", "golden")  # noqa: N806

    def gray(path):
        return gating.frame_to_gray(np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[..., ::-1]), "RGB2GRAY")

    cases = {"autodriving_801x801_params_B": ([os.path.join(G, "frames", "autodriving", f"{k}.jpg") for k in (1, 2)], PARAMS_B),
             "grasp_1080x1920_params_A": ([os.path.join(G, "demo", f"grasp_{k}.jpg") for k in (1, 2)], PARAMS_A)}
    rec, ok = {}, True
    for name, (paths, q) in cases.items():
        if not all(os.path.exists(pp) for pp in paths):
            rec[name] = {"skipped": "frames not found"}
            continue
        a, b = gray(paths[0]), gray(paths[1])
        args = [getattr(q, k) for k in ("pyr_scale", "levels", "winsize", "iterations", "poly_n", "poly_sigma", "flags")]
        ref = O.farneback(a, b, *args)
        r = {}
        for mode, exact in (("default", True), ("fast_rowsums", False)):
            got = nsof.calcOpticalFlowFarneback(a, b, None, *args, ctx=ctx, exact=exact)
            d = np.abs(got - ref).max(-1)
            r[mode] = {"max_abs_epe_vs_oracle": float(d.max()), "pixels_above_1e-4": int((d > 1e-4).sum()),
                       "bit_identical": bool(np.array_equal(got, ref))}
        ok = ok and r["default"]["max_abs_epe_vs_oracle"] < 1e-4
        rec[name] = r
    rec["parity_ok"] = ok
    rec["epe_tolerance"] = 1e-4
    return rec


def e2e_leg(nsof, p, prevs, nexts, flow, k, local_rank):
    """What a caller of the drop-in sees: frames in ordinary (pageable) host arrays in, flow in host memory out
    (the reference times exactly this around its cv2 call, optical_flow_seg.py:202-206, 492-496).  k pairs through
    nsof_farneback_u8_batch -- upload, compute and download of consecutive chunks overlap on three streams; the
    flow fields are handed out in page-locked arrays (cv2 also returns arrays it allocated).  PCIe-inclusive: this is
    NEVER the headline value."""
    import numpy as np
    h, w = prevs.shape[1:]
    hp, hn = prevs[:k].cpu().numpy(), nexts[:k].cpu().numpy()
    pairs = [(hp[i], hn[i]) for i in range(k)]
    out = {}
    with nsof.Context(local_rank) as c2:
        pinned = [nsof.pinned_empty((h, w, 2), np.float32) for _ in range(k)]
        nsof.farneback_pairs(pairs, p, pinned, ctx=c2)        # warm-up: staging slots, workspace, page faults
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            nsof.farneback_pairs(pairs, p, pinned, ctx=c2)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        same = bool(np.array_equal(pinned[0], flow[0].cpu().numpy()) and np.array_equal(pinned[k - 1], flow[k - 1].cpu().numpy()))
        bytes_pair = 2 * h * w + 8 * h * w
        out["e2e"] = {"value": round(k / best, 1), "unit": "pairs/s", "pairs": k,
                      "path": "host numpy frames (pageable) -> nsof_farneback_u8_batch -> page-locked host flow arrays",
                      "bytes_per_pair": bytes_pair, "d2h_GBps": round(k * 8 * h * w / best / 1e9, 2),
                      "pcie_bound_pairs_per_s": round(55e9 / (8 * h * w), 1),
                      "frac_of_pcie_bound": round((k / best) / (55e9 / (8 * h * w)), 3),
                      "identical_to_device_resident": same}
        kk = min(k, 32)                                    # pageable outputs: one more host copy per flow field
        pageable = [np.empty((h, w, 2), np.float32) for _ in range(kk)]
        nsof.farneback_pairs(pairs[:kk], p, pageable, ctx=c2)
        t0 = time.perf_counter()
        nsof.farneback_pairs(pairs[:kk], p, pageable, ctx=c2)
        out["e2e"]["pageable_outputs_pairs_per_s"] = round(kk / (time.perf_counter() - t0), 1)
        nsof.calcOpticalFlowFarneback(hp[0], hn[0], None, **p.as_kwargs(), ctx=c2)   # warm-up: workspace, side stream, staging
        t0 = time.perf_counter()                           # one synchronous call per pair, the reference's pattern
        for i in range(min(k, 32)):
            nsof.calcOpticalFlowFarneback(hp[i], hn[i], None, **p.as_kwargs(), ctx=c2)
        out["e2e"]["one_call_per_pair_pairs_per_s"] = round(min(k, 32) / (time.perf_counter() - t0), 1)
    return out


def cpu_leg(nsof, p, prevs, nexts, flow, k):
    """CPU baseline on the first k pairs of this run's batch, timed on this box's host cores:
    cv2 itself when it is importable (``kind: reference``: 1 thread and all cores, version + IPP line), otherwise the
    C restatement oracle/farneback_ref.c built -O3 -march=native here (``kind: port``: 1 thread, and all cores with
    OpenMP over pairs).  Also the max-abs end-point error of the GPU flow against that CPU result."""
    import numpy as np
    from oracle import oracle as O  # noqa: N812  (the checker; only ever used here, in tests and in smoke())
    O.build()
    hp, hn, gf = prevs[:k].cpu().numpy(), nexts[:k].cpu().numpy(), flow[:k].cpu().numpy()
    args = [getattr(p, a) for a in ("pyr_scale", "levels", "winsize", "iterations", "poly_n", "poly_sigma", "flags")]
    model, logical, physical, usable = O.host_cpu()
    base = {"unit": "pairs/s", "cpu_model": model, "logical_cores": logical, "physical_cores": physical,
            "usable_cores": usable}
    out = {}
    try:
        import cv2
    except ImportError:
        cv2 = None
    if cv2 is not None:
        info = cv2.getBuildInformation()
        ipp = next((ln.strip() for ln in info.splitlines() if "IPP" in ln), "IPP: ?")
        kw = dict(zip(("pyr_scale", "levels", "winsize", "iterations", "poly_n", "poly_sigma", "flags"), args))
        res = {}
        for nt in (1, usable):
            cv2.setNumThreads(nt)
            cv2.calcOpticalFlowFarneback(hp[0], hn[0], None, **kw)
            t0 = time.perf_counter()
            refs = [cv2.calcOpticalFlowFarneback(hp[i], hn[i], None, **kw) for i in range(k)]
            res[nt] = k / (time.perf_counter() - t0)
        err = max(float(np.abs(refs[i] - gf[i]).max()) for i in range(k))
        out["cpu_baseline"] = dict(base, value=round(res[1], 4), cores=1, kind="reference",
                                   value_all_cores=round(res[usable], 4), cores_all=usable,
                                   sample=f"first {k} pairs of the timed batch, cv2 {cv2.__version__} ({ipp}), "
                                          f"setNumThreads(1) and setNumThreads({usable})")
        out["max_abs_epe_vs_cv2"] = err
        err_o = max(float(np.abs(O.farneback(hp[i], hn[i], *args) - gf[i]).max()) for i in range(min(k, 2)))
        out["max_abs_epe_vs_oracle"] = err_o
        err = max(err, err_o)
    else:
        O.farneback_many(hp[:1, :64, :64], hn[:1, :64, :64], *args, n_threads=1)   # build + load + warm
        t0 = time.perf_counter()
        refs = O.farneback_many(hp, hn, *args, n_threads=1)
        v1 = k / (time.perf_counter() - t0)
        # all-cores leg: OpenMP over pairs.  A box may expose more logical CPUs than its share lets a process use
        # (oversubscribed threads only thrash), so a 16-thread run is timed as well and the better one reported.
        va, ta, ka = 0.0, usable, k
        for nt in sorted({usable, min(usable, 16)}, reverse=True):
            kk = max(k, min(2 * nt, prevs.shape[0]))            # a couple of pairs per thread
            hpa, hna = (hp, hn) if kk == k else (prevs[:kk].cpu().numpy(), nexts[:kk].cpu().numpy())
            t0 = time.perf_counter()
            O.farneback_many(hpa, hna, *args, n_threads=nt)
            v = kk / (time.perf_counter() - t0)
            if v > va:
                va, ta, ka = v, nt, kk
        err = max(float(np.abs(refs[i] - gf[i]).max()) for i in range(k))
        out["cpu_baseline"] = dict(base, value=round(v1, 4), cores=1, kind="port",
                                   value_all_cores=round(va, 4), cores_all=ta,
                                   sample=f"first {k} pairs of the timed batch on 1 thread, first {ka} pairs on "
                                          f"{ta} threads (OpenMP over pairs; best of {usable} and 16 threads); CPU oracle oracle/farneback_ref.c "
                                          f"built gcc -O3 -march=native -ffp-contract=off on this host; restatement, "
                                          f"not OpenCV (cv2 is not importable on this image)")
        out["max_abs_epe_vs_oracle"] = err
        out["bit_identical_to_oracle"] = bool(all(np.array_equal(refs[i], gf[i]) for i in range(k)))
    out["epe_tolerance"] = 1e-4
    out["parity_ok"] = bool(err < 1e-4)
    return out


if __name__ == "__main__":
    main()
