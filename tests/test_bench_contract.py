"""bench.py's output contract (one JSON line, the driver's keys, roofline and cpu_baseline objects), on a short run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_cli_parses_without_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in r.stdout


@pytest.mark.gpu
def test_bench_emits_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "64", "--warmup", "8",
                        "--crt-channels", "3000000", "--crt-blocks", "40", "--dropin-blocks", "60", "--crt-pcie-blocks", "30", "--next-rows-channels", "300000"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    j = json.loads(lines[-1])                                  # the JSON is the LAST line of stdout
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 64 and j["warmup"] == 8 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["reps"] >= 1 and j["regions"] >= 5 and j["drained_k_step_region_ms_per_step"] > 0 and j["ms_per_step_min"] <= j["ms_per_step"] <= j["ms_per_step_max"]
    assert j["config"]["baseline_config"] == 3 and j["roofline"]["launches_timed"] >= 200
    assert j["value"] > 0 and abs(j["value"] - 1024 * 0.02 / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
    roof = j["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and 0 < roof["frac"] < 1
    assert roof["algorithmic_bytes_per_block"] == 25920008
    pipe = roof["pipelined"]                                   # the forward transform alone, 4 blocks in flight
    assert pipe["blocks_timed"] == 2000 and 0 < pipe["forward_us_per_block"] < roof["forward_us_per_block"]
    assert abs(pipe["frac"] - pipe["achieved"] / roof["peak"]) < 1e-12
    cpu = j["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and cpu["value"] > 0 and "sample" in cpu
    crt = j["c_rt"]
    assert crt["sustained"] is True and crt["channels"] >= 2990000 and crt["worst_block_ms"] <= 20.0 and crt["blocks"] == 40
    assert crt["probes"] and crt["probes"][-1]["channels"] == crt["channels"]            # the ladder (one rung here)
    # round 3: the boundary and the host link are in the driver-run line
    assert roof["traffic_source"] is None or "committed profile" in roof["traffic_source"]
    assert "rocprof_frac" in roof and "fft_calibration" in cpu
    legs = j["dropin"]
    assert len(legs) == 3 and [x["threads"] for x in legs] == [1024, 1024, 2000]
    for x in legs:
        assert "error" not in x, x
        assert x["blocks"] == 60 and x["drops"] == 0 and 0 < x["ms_per_block"] < 20.0 and x["worst_block_gap_ms"] > 0
        assert x["host_profile"]["staged_hits"] > 0
    pc = j["c_rt_pcie"]
    assert len(pc) == 3 and all("error" not in x for x in pc), pc          # baseband, demodulated PCM, the same double-buffered
    assert pc[0]["d2h_bytes_per_channel"] == 1920 and pc[1]["d2h_bytes_per_channel"] == 481 and pc[2]["d2h_bytes_per_channel"] == 481
    assert "double-buffered" in pc[2]["loop"] and pc[2]["worst_latency_ms"] >= pc[2]["worst_block_ms"]
    nr = j["next_rows"]                                        # SURVEY 8f's rows behind every channel of one large bank
    assert [x["mode"] for x in nr] == ["linear", "pll", "fm"] and all("error" not in x for x in nr), nr
    assert all(x["channels"] == 297984 and x["fits_20ms"] and all(v > 0 for v in x["ns_per_channel"].values()) for x in nr)
    assert all(x["blocks"] == 30 and x["d2h_bytes_per_block"] == x["channels"] * x["d2h_bytes_per_channel"] for x in pc)
    # round 4: every scale leg compares sampled channels (incl. the highest index) with the oracle after its timed blocks ...
    assert crt["verified_channels"] >= 64 and crt["max_rel_err"] < 1e-5 and crt["highest_channel_checked"] >= crt["channels"] - 1024 - 3072
    assert all(x["verified_channels"] >= 64 and x["max_rel_err"] < 1e-5 for x in pc) and pc[1]["pcm_mismatches"] == 0 and pc[2]["pcm_mismatches"] == 0
    assert all(x["verified_channels"] >= 64 and x["max_rel_err"] < 1e-5 and x["pcm_mismatches"] == 0 and x["verification"]["status_mismatches"] == 0 for x in nr)
    # ... the boundary runs at wall-clock pace ...
    pl = j["dropin_paced"]
    assert len(pl) == 2 and [x["threads"] for x in pl] == [1024, 2000]
    for x in pl:
        assert "error" not in x, x
        pd = x["paced"]
        assert pd["block_drops"] == 0 and pd["skipped_blocks"] == 0 and pd["blocks_served_to_every_channel"] == pd["blocks_measured"]
        assert 0 < pd["latency_ms"]["p50"] < 20.0 and 19.0 < x["ms_per_block"] < 32.0          # one block per 20 ms on the front end's clock (+ start-up of 1000-2000 threads and tear-down, spread over only 60 blocks here)
    # ... the CPU leg is like for like, and the line says what each leg cost
    cc = cpu["c_rt_cpu"]
    assert cc["probes"] and "block_drops" in cc["probes"][0] and cpu["us_per_channel_block"] > 0 and cpu["fwd_fft_ms"] > 0
    assert j["quick"] is False and set(j["leg_seconds"]) >= {"c_rt", "cpu_baseline", "dropin", "dropin_paced", "c_rt_pcie", "next_rows"}
    assert roof["profiles_match_this_tree"]["kernel_sources_sha16"]


@pytest.mark.gpu
def test_bench_quick_mode_is_the_headline_only():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--quick"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.strip()][-1])
    assert j["quick"] is True and j["value"] > 0 and j["roofline"]["frac"] > 0
    assert j["c_rt"] is None and j["cpu_baseline"] is None and j["dropin"] is None and j["dropin_paced"] is None and j["c_rt_pcie"] is None and j["next_rows"] is None


@pytest.mark.gpu
@pytest.mark.parametrize("extra,cfg", [(["--config", "1"], 1), (["--config", "2"], 2), (["--config", "4"], 4), (["--config", "5"], 5)])
def test_bench_other_configs_run_on_one_gpu(extra, cfg):
    env = dict(os.environ)
    if cfg == 4:
        env["BENCH_FORCE_DIST"] = "1"          # one rank, but through the process group and the RCCL exchange behind the C ABI
        env["BENCH_ALL_EXCHANGES"] = "1"       # ... every hand-over: spectrum rows, whole slot, and (round 4) the block's samples
        env["MASTER_PORT"] = "29617"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-crt",
                        "--no-cpu-baseline", "--min-seconds", "0.05", "--no-crt-pcie", "--no-next-rows", "--dropin-blocks", "40"] + extra,
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.strip()][-1])
    assert j["config"]["baseline_config"] == cfg and j["n_gpus"] == 1 and j["value"] > 0
    if cfg == 1:                                        # the reference's plumbing case: a COMPLEX 2.4 MS/s master, one IQ channel
        assert j["config"]["channels_total"] == 1 and j["config"]["N"] == 60000 and j["roofline"]["algorithmic_bytes_per_block"] == 16 * 60000
        assert len(j["dropin"]) == 1 and "error" not in j["dropin"][0] and j["dropin"][0]["threads"] == 1 and j["dropin"][0]["drops"] == 0
    if cfg == 2:
        assert len(j["dropin"]) == 3 and all("error" not in x for x in j["dropin"])
    if cfg == 4:
        assert "RCCL" in j["exchange"] and "replicate" in j["legs"]
        assert {"samples", "subband", "broadcast", "replicate"} <= set(j["legs"]) | {"broadcast" if "whole spectrum slot" in j["exchange"] else "subband"}
        assert all(v["value"] > 0 for v in j["legs"].values())
    if cfg == 5:
        assert "replicas only" in j["exchange"]


def test_bench_refuses_to_run_fewer_ranks_than_asked():
    # --gpus N without a launcher self-spawns N ranks and fails LOUDLY when fewer GPUs are visible (here: none or one)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "2", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "only" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("args,port", [(["--config", "5", "--no-crt"], 29581),
                                       (["--config", "4", "--exchange", "replicate", "--crt-channels", "1500000", "--crt-blocks", "12"], 29582)])
def test_bench_two_ranks_control_flow_on_one_gpu(args, port):
    """The driver's N > 1 launch line with two ranks sharing this box's one GPU (gloo control plane: RCCL refuses two ranks per
    device, so the modes without a data-path collective): rendezvous, per-rank workloads and seeds, barrier + max-over-ranks
    timing, the gathered C_rt leg, ONE JSON line from rank 0 with the whole-job aggregate."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--min-seconds", "0.1", "--no-cpu-baseline"] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1                                          # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    per_rank = 1024
    assert j["config"]["channels_total"] == 2 * per_rank
    assert abs(j["value"] - 2 * per_rank * 0.02 / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
    if "--no-crt" not in args:
        assert j["c_rt"]["gpus"] == 2 and j["c_rt"]["channels"] >= 2 * 1490000 and j["c_rt"]["blocks"] == 12
