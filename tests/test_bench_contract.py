"""bench.py prints ONE JSON line with the fields the driver and the judge read (task statement, sections 4 and
"How to work"): metric/value/unit/n_gpus/steps/warmup/ms_per_step/higher_is_better/scaling/vs_baseline/dtype/data/
config.workload plus the roofline and cpu_baseline objects.  Small batch so the test takes seconds."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_bench_json_contract():
    env = dict(os.environ, NSOF_SKIP_BUILD="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pairs", "8", "--steps", "2", "--warmup", "1",
                          "--cpu-sample", "1"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 0.01
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    assert d["max_abs_epe_vs_oracle"] <= d["epe_tolerance"]
    # round-2 records: host-to-host rate, opt-in float expansion, accumulator + joined config-5 pipeline
    e = d["e2e"]
    assert e["unit"] == "pairs/s" and e["value"] > 0 and e["identical_to_device_resident"] is True
    f = d["fast_mode"]
    assert f["value"] > 0 and 0 <= f["max_abs_epe_vs_exact_path"] < 0.5 and d["roofline_polyexp_fast"]["frac"] > 0
    a = d["accumulator"]
    assert a["unit"] == "slices/s" and a["value"] > 0 and a["parity_ok"] is True and a["cpu_baseline"]["value"] > 0
    assert a["roofline_fused_bytes"]["frac"] > 0 and a["roofline_survey_definition"]["bytes_per_slice"] > 6e7
    assert d["config5"]["flow_finite"] is True and d["config5"]["surface_frames"] >= 2
    assert d["parity_ok"] is True
    # round 3: the library's row-sum order is the default -> the headline batch, the reference's real frames and the first
    # pair of the joined config-5 pipeline are all bit-identical to the oracle; the per-pixel sums are a named fast mode
    assert "library order" in d["config"]["rowsum_order"] and d["bit_identical_to_oracle"] is True
    c3 = d["config3"]
    assert c3["parity_ok"] is True and len(c3["streams"]) == 3
    for srec in c3["streams"].values():
        assert srec["first_pair_vs_oracle_chain"]["bit_identical"] is True and srec["flow_fields_per_s"] > 0
    assert c3["streams"]["survey_200k_background"]["roi_pixel_fraction"] == 1.0
    c5 = d["config5"]
    assert c5["parity_ok"] is True and c5["first_pair_vs_oracle_chain"]["surface_frames_equal"] is True
    assert c5["first_pair_vs_oracle_chain"]["default"]["max_abs_epe_vs_oracle"] < 1e-4
    rf = d["real_frames"]
    assert rf["parity_ok"] is True
    for name in ("autodriving_801x801_params_B", "grasp_1080x1920_params_A"):
        assert rf[name]["default"]["bit_identical"] is True and rf[name]["default"]["pixels_above_1e-4"] == 0
    assert rf["autodriving_801x801_params_B"]["fast_rowsums"]["max_abs_epe_vs_oracle"] > 1e-4   # why it is opt-in
    fr = d["fast_rowsums"]
    assert fr["value"] > 0 and fr["max_abs_vs_default_path"] < 1e-4
    # the other two parameter sets of the reference's datasets, consecutive-frame mode, lone-call latency
    for key in ("params_B", "params_C"):
        q = d[key]
        assert q["unit"] == "pairs/s" and q["value"] > 0 and q["max_abs_epe_vs_oracle_pair0"] <= d["epe_tolerance"]
        assert 0 < q["roofline"]["iterate"]["frac"] < 1 and 0 < q["roofline"]["polyexp"]["frac"] < 1
    assert d["sequence"]["value"] > 0 and d["sequence"]["last_pair_identical_to_pair_call"] is True
    sc = d["single_call"]
    assert sc["default"]["host_to_host_ms"] > 0 and sc["row_bands"]["device_resident_ms"] > 0
    assert sc["fast_rowsums"]["host_to_host_ms"] > 0
    assert sc["row_bands"]["max_abs_vs_default"] < 1e-3
    assert sc["default_fused_kernel_only"]["max_abs_vs_default"] == 0.0 and sc["default_801x801_params_B"]["host_to_host_ms"] > 0


@pytest.mark.gpu
def test_bench_two_rank_path_rehearsal():
    """The N > 1 code path of bench.py as the driver launches it (torch.distributed.run, one process per rank), on a
    box with one GPU: NSOF_BENCH_REHEARSAL=1 lets the two ranks share the device and run the barrier / max-over-ranks
    over gloo.  Checks the plumbing (rank 0 prints exactly one line, whole-job value, ranks exit cleanly), not speed."""
    env = dict(os.environ, NSOF_SKIP_BUILD="1", NSOF_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--pairs", "8", "--config4-pairs", "6"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_pairs_per_step"] == 16 and d["scaling"] == "weak"
    assert abs(d["value"] - 16 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 0.01
    assert "rehearsal" in d and "cpu_baseline" not in d     # rank-0-only legs run at N = 1 only
    io = d["io_gather"]
    assert "error" not in io and io["value"] > 0 and io["world_size"] == 2 and io["backend"].startswith("gloo")
    # BASELINE configs 5 and 4 over the ranks of the job (VERDICT r3 Missing 1): row bands -> all-gather -> sharded pairs;
    # the five datasets' call list dealt round-robin
    c5 = d["config5_sharded"]
    assert "error" not in c5, c5
    assert c5["world_size"] == 2 and c5["backend"].startswith("gloo") and sum(c5["band_rows_per_rank"]) == 2160
    assert c5["allgather_bytes_received_per_rank"] == c5["surface_frames"] * 1080 * 3840 and c5["allgather_ms_max"] > 0
    assert sum(c5["pairs_per_rank"]) == c5["surface_frames"] - 1 and c5["stream_seconds_per_wall_second"] > 0
    assert c5["first_pair_vs_oracle_chain"]["surface_frames_equal"] is True and c5["parity_ok"] is True
    assert c5["first_pair_vs_oracle_chain"]["max_abs_epe_vs_oracle"] == 0.0
    c4 = d["config4_sharded"]
    assert "error" not in c4, c4
    assert c4["world_size"] == 2 and len(c4["per_rank"]) == 2 and sum(r["calls"] for r in c4["per_rank"]) == c4["calls"]
    assert abs(c4["per_rank"][0]["calls"] - c4["per_rank"][1]["calls"]) <= 1 and c4["calls_per_s"] > 0 and c4["flow_finite"]


@pytest.mark.gpu
def test_bench_prints_the_headline_when_the_gather_leg_hangs():
    """The N > 1 line must not depend on the scatter/compute/gather leg finishing: with a watchdog time of ~0 the leg is
    abandoned on every rank, rank 0 still prints exactly one line carrying the measured headline and the error."""
    env = dict(os.environ, NSOF_SKIP_BUILD="1", NSOF_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--pairs", "8", "--io-timeout", "0.01"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode != 0, "a hung leg must show in the exit status (every rank leaves with code 4)"
    assert "exitcode  : 4" in out.stderr or "exitcode: 4" in out.stderr or "exit code 4" in out.stderr.lower() or out.returncode in (1, 4), \
        out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "did not finish" in d["io_gather"]["error"]
    assert d["io_gather"]["exit_code"] == 4
