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


def _strict(line):
    """the driver's parser is a strict one: no NaN / Infinity tokens, one object"""
    def bad(tok):
        raise ValueError("non-standard JSON token %r" % tok)
    return json.loads(line, parse_constant=bad)


def _run_bench(args, tmp_path, timeout=900, env=None):
    detail = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--detail", detail] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    # the LAST line of stdout is the headline: compact (round 4's 23 KB line came back from the driver unparsed), strict JSON
    assert len(lines[-1]) < 6144, len(lines[-1])
    head = _strict(lines[-1])
    full = _strict(open(detail).read())
    assert head["detail"] == detail
    return head, full


def test_headline_of_the_committed_full_run_is_compact_and_strict():
    """bench.headline() over the detail object of the last full run committed under profiles/ (no GPU needed): under 6 KB, strict JSON,
    the contract keys + roofline + cpu_baseline, one-number summaries of the other legs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    for name in ("r06_bench_detail.json", "r05_bench_detail.json", "r04_bench.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            break
    full = json.load(open(path))
    line = b.headline(full, "gpurun_out/bench_detail.json")
    assert len(line) < 6144
    h = _strict(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in h, k
    assert set(h["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and set(h["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert "workload" in h["config"] and h["c_rt"]["channels"] > 0 and len(h["next_rows"]) == 3
    # a pathological detail object (NaN, numpy scalars, very long strings) still yields a strict line under the limit
    import numpy as np
    full["roofline"]["achieved"] = float("nan"); full["value"] = np.float64(full["value"]); full["config"]["workload"] = "x" * 300
    full["dropin"] = [{"error": "e" * 5000}] * 3
    h2 = _strict(b.headline(full, None))
    assert h2["roofline"]["achieved"] is None and len(b.headline(full, None)) < 6144


@pytest.mark.gpu
def test_bench_emits_one_json_line_with_the_contract_keys(tmp_path):
    h, j = _run_bench(["--gpus", "1", "--steps", "64", "--warmup", "8", "--crt-channels", "3000000", "--crt-blocks", "40", "--dropin-blocks", "60",
                       "--crt-pcie-blocks", "30", "--next-rows-channels", "300000"], tmp_path)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j and k in h, k
    # ---- the headline: what the driver parses
    assert h["n_gpus"] == 1 and h["steps"] == 64 and h["warmup"] == 8 and h["higher_is_better"] is True
    assert h["scaling"] == "weak" and h["vs_baseline"] is None and h["dtype"] == "f32" and h["data"] == "synthetic"
    assert "workload" in h["config"] and "model" not in h["config"] and h["config"]["baseline_config"] == 3
    assert h["value"] > 0 and abs(h["value"] - 1024 * 0.02 / (h["ms_per_step"] * 1e-3)) <= 1e-3 * h["value"]       # (5 significant digits in the headline)
    hr = h["roofline"]
    assert hr["bound"] == "hbm" and hr["unit"] == "GB/s" and hr["peak"] == 8000.0 and 0 < hr["frac"] < 1 and abs(hr["frac"] - hr["achieved"] / hr["peak"]) < 1e-4
    assert hr["algorithmic_bytes_per_block"] == 25920008 and hr["launches_timed"] >= 200 and 0 < hr["pipelined"]["frac"] < 1 and "traffic" in hr and "rocprof_frac" in hr
    hc = h["cpu_baseline"]
    assert hc["kind"] in ("reference", "port") and hc["cores"] >= 1 and hc["value"] > 0 and "sample" in hc and hc["c_rt_cpu"]["channels"] >= 0
    assert h["c_rt"]["sustained"] is True and h["c_rt"]["channels"] >= 2990000 and h["c_rt"]["worst_block_ms"] <= 20.0 and h["c_rt"]["verified_channels"] >= 64
    assert [x["threads"] for x in h["dropin"]] == [1024, 1024, 2000] and all(x["drops"] == 0 for x in h["dropin"])
    assert [x["threads"] for x in h["dropin_paced"]] == [1024, 2000]
    for x in h["dropin_paced"]:                              # round 5: block 0 is an ordinary block, and drops are drops wherever they fall
        assert x["drops"] == 0 and x["drops_first_8_blocks"] == 0 and 0 <= x["block0_ms"] < 20.0 and x["max_first_8_blocks_ms"] < 20.0, x
    hs = h["dropin_sharded"]
    assert len(hs) == 1 and hs[0]["devices"] == 2 and hs[0]["slaves_per_device"] == [1024, 1024] and hs[0]["threads"] == 2048, hs
    assert hs[0]["drops"] == 0 and hs[0]["mismatched_channels"] == 0 and hs[0]["verified_channels"] >= 24 and hs[0]["devices_distinct"] is False
    # (one host hiccup in a synchronous PCIe probe makes that probe's worst block late: the line says so; two of the three must hold)
    assert sum(bool(x["sustained"]) for x in h["c_rt_pcie"]) >= 2 and all(x["worst_block_ms"] > 0 for x in h["c_rt_pcie"]) and [x["mode"] for x in h["next_rows"]] == ["linear", "pll", "fm"]
    assert all(x["pcm_mismatches"] == 0 and x["ms_per_block"] < 20.0 for x in h["next_rows"]) and h["rccl_ranks"] == 0 and h["quick"] is False
    # ---- the detail file: everything else
    assert j["n_gpus"] == 1 and j["steps"] == 64 and j["warmup"] == 8 and j["higher_is_better"] is True
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
    assert crt["verified_channels"] >= 64 and crt["max_rel_err"] < 1e-5 and crt["highest_channel_checked"] >= crt["channels"] - 1024 - 3072
    assert all(x["verified_channels"] >= 64 and x["max_rel_err"] < 1e-5 for x in pc) and pc[1]["pcm_mismatches"] == 0 and pc[2]["pcm_mismatches"] == 0
    assert all(x["verified_channels"] >= 64 and x["max_rel_err"] < 1e-5 and x["pcm_mismatches"] == 0 and x["verification"]["status_mismatches"] == 0 for x in nr)
    pl = j["dropin_paced"]
    assert len(pl) == 2 and [x["threads"] for x in pl] == [1024, 2000]
    for x in pl:
        assert "error" not in x, x
        pd = x["paced"]
        assert pd["block_drops"] == 0 and pd["skipped_blocks"] == 0 and pd["blocks_served_to_every_channel"] == pd["blocks_measured"]
        assert 0 < pd["latency_ms"]["p50"] < 20.0 and 19.0 < x["ms_per_block"] < 32.0          # one block per 20 ms on the front end's clock (+ start-up of 1000-2000 threads and tear-down, spread over only 60 blocks here)
    cc = cpu["c_rt_cpu"]
    assert cc["probes"] and "block_drops" in cc["probes"][0] and cc["repeats"] == 3 and cpu["us_per_channel_block"] > 0 and cpu["fwd_fft_ms"] > 0 and cpu["throughput_bound_channels"] > 0
    assert j["quick"] is False and set(j["leg_seconds"]) >= {"c_rt", "cpu_baseline", "dropin", "dropin_paced", "dropin_sharded", "c_rt_pcie", "next_rows"}
    assert roof["profiles_match_this_tree"]["kernel_sources_sha16"]


@pytest.mark.gpu
def test_bench_crt_search_reports_a_sustained_count_and_the_mean_crossing(tmp_path):
    """the default C_rt search (no explicit ladder), with short rungs: a sustained count whatever the box, every block of it inside 20 ms,
    the count at which the MEAN block time crosses 20 ms beside it, and never more rungs than the search allows"""
    h, j = _run_bench(["--steps", "20", "--warmup", "5", "--crt-blocks", "60", "--no-next-rows", "--no-dropin", "--no-dropin-paced", "--no-crt-pcie",
                       "--no-cpu-baseline"], tmp_path)
    c = j["c_rt"]
    assert c["sustained"] is True and 10_000_000 < c["channels"] <= 22_001_024 and c["worst_block_ms"] <= 20.0 and c["verified_channels"] >= 64
    assert len(c["calibration"]) == 2 and 1 <= c["rungs"] <= 12 and c["probes"][0]["channels"] > 0.8 * c["mean_crossing_channels"]
    assert 0.9 * c["channels"] < c["mean_crossing_channels"] < 1.25 * c["channels"]
    assert h["c_rt"]["channels"] == c["channels"] and h["c_rt"]["mean_crossing_channels"] > 0


@pytest.mark.gpu
def test_bench_quick_mode_is_the_headline_only(tmp_path):
    h, j = _run_bench(["--steps", "20", "--warmup", "5", "--quick"], tmp_path, timeout=600)
    assert j["quick"] is True and j["value"] > 0 and j["roofline"]["frac"] > 0 and h["roofline"]["frac"] > 0
    assert j["c_rt"] is None and j["cpu_baseline"] is None and j["dropin"] is None and j["dropin_paced"] is None and j["c_rt_pcie"] is None and j["next_rows"] is None
    assert h["c_rt"] is None and h["cpu_baseline"] is None


@pytest.mark.gpu
@pytest.mark.parametrize("extra,cfg", [(["--config", "1"], 1), (["--config", "2"], 2), (["--config", "4"], 4), (["--config", "5"], 5)])
def test_bench_other_configs_run_on_one_gpu(extra, cfg, tmp_path):
    env = dict(os.environ)
    if cfg == 4:
        env["BENCH_FORCE_DIST"] = "1"          # one rank, but through the process group and the RCCL exchange behind the C ABI
        env["BENCH_ALL_EXCHANGES"] = "1"       # ... every hand-over: spectrum rows, whole slot, and (round 4) the block's samples
        env["MASTER_PORT"] = "29617"
    h, j = _run_bench(["--steps", "20", "--warmup", "5", "--no-crt", "--no-cpu-baseline", "--min-seconds", "0.05", "--no-crt-pcie", "--no-next-rows",
                       "--dropin-blocks", "40"] + extra, tmp_path, env=env)
    assert h["config"]["baseline_config"] == cfg and h["value"] > 0
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
def test_bench_two_ranks_control_flow_on_one_gpu(args, port, tmp_path):
    """The driver's N > 1 launch line with two ranks sharing this box's one GPU (gloo control plane: RCCL refuses two ranks per
    device, so the modes without a data-path collective): rendezvous, per-rank workloads and seeds, barrier + max-over-ranks
    timing, the gathered C_rt leg, ONE JSON line from rank 0 with the whole-job aggregate."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--min-seconds", "0.1", "--no-cpu-baseline", "--no-dropin", "--detail", str(tmp_path / "detail.json")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6144                 # rank 0 only
    h = _strict(lines[0])
    assert h["n_gpus"] == 2 and h["value"] > 0
    j = _strict(open(str(tmp_path / "detail.json")).read())
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    per_rank = 1024
    assert j["config"]["channels_total"] == 2 * per_rank
    assert abs(j["value"] - 2 * per_rank * 0.02 / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
    if "--no-crt" not in args:
        assert j["c_rt"]["gpus"] == 2 and j["c_rt"]["channels"] >= 2 * 1490000 and j["c_rt"]["blocks"] == 12


class _FakeBank:
    def __init__(self):
        self.active = 0

    def set_responses(self, c0, resp): pass
    def set_shifts(self, c0, shifts): pass
    def set_active(self, n): self.active = n
    def destroy(self): pass


class _FakeEng:
    def __init__(self):
        self.b = _FakeBank()

    def bank(self, P, olen, n, shared_rows=0):
        self.cap = n
        return self.b


@pytest.mark.parametrize("ns_per_channel,spike_at", [(0.93, None), (1.30, None), (0.93, 20_100_000), (2.4, None)])
def test_crt_search_decisions_on_a_modelled_device(ns_per_channel, spike_at):
    """bench.crt_leg's search (no GPU): a device whose block takes 0.4 ms + ns_per_channel x channels, optionally with one late block per
    rung above a channel count.  The search must report the largest grid count whose every block stays inside 20 ms -- stepping UP from
    its first rung on a fast device, DOWN on a slow one (never 'nothing sustained'), taking the bisection step -- and a mean-crossing
    count within 1 % of the model's."""
    import importlib.util
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    import __graft_entry__ as ge
    pkg = ge.load()
    eng = _FakeEng()
    wl = b.workload_for(3, 0, 1, 0)
    calls = {"n": 0}

    def run_one(job):
        calls["n"] += 1
        n = eng.b.active + wl["nch"]
        ms = 0.4 + ns_per_channel * 1e-6 * n
        if spike_at and n > spike_at and calls["n"] % 97 == 0:
            ms += 3.0
        return ms
    r = b.crt_leg(pkg, eng, wl, None, 120, run_one, verify=0)
    true_cross = (20.0 - 0.4) / (ns_per_channel * 1e-6)
    assert r["sustained"] is True and r["channels"] > 0 and r["worst_block_ms"] <= 20.0
    assert abs(r["mean_crossing_channels"] - true_cross) <= 0.01 * true_cross
    limit = min(true_cross, spike_at or 1e18, eng.cap + wl["nch"])
    assert r["channels"] <= limit and r["channels"] >= limit - 270_000 - 3072, (r["channels"], limit)     # within one 0.25 M grid step of the truth
    assert len(r["calibration"]) == 2 and 1 <= r["rungs"] <= 14


@pytest.mark.gpu
def test_a_leg_that_never_comes_back_costs_its_own_numbers_not_the_line(tmp_path):
    """round 5: one bench run (every engine stream CU-masked) never came back and cost the whole record.  Every leg now has a time budget
    and a watchdog THREAD (the main thread may sit inside a HIP call): the leg that overruns is named in the line, the run ends with rc 0
    and the legs that did finish are in it.  BENCH_TEST_HANG_LEG makes the named leg hang; the budget is cut to 4 s."""
    env = dict(os.environ, BENCH_TEST_HANG_LEG="c_rt", BENCH_LEG_BUDGET_S="4")
    h, j = _run_bench(["--quick", "--steps", "20", "--warmup", "5"], tmp_path, timeout=300, env=env)
    assert h["leg_timeouts"] == ["c_rt"] and h["headline_from"] == "local"
    assert h["value"] > 0 and h["roofline"]["frac"] > 0 and h["c_rt"] is None        # what ran before the hang is in the line
    # ... and when the HEADLINE leg itself hangs there is still one strict line, with a null value, rc 0
    env = dict(os.environ, BENCH_TEST_HANG_LEG="headline", BENCH_LEG_BUDGET_S="4")
    h, j = _run_bench(["--quick", "--steps", "20", "--warmup", "5"], tmp_path, timeout=300, env=env)
    assert h["leg_timeouts"] == ["headline"] and h["value"] is None and h["roofline"] is None
    # ... and when nothing ever comes up (the runtime, the engine): the start-up has a budget too
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--quick"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, BENCH_TEST_HANG_LEG="startup", BENCH_LEG_BUDGET_S="3"))
    assert r.returncode == 0, r.stderr[-1000:]
    h = _strict(r.stdout.strip().splitlines()[-1])
    assert h["leg_timeouts"] == ["startup"] and h["value"] is None and h["metric"].startswith("channels sustained")


def test_bench_startup_has_a_time_budget_too():
    """no GPU needed: the watchdog runs from the first line of main() -- a process that never gets as far as its first measurement (the
    runtime or the engine never coming up) still prints ONE strict line with a null value and rc 0 after the start-up budget"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--quick"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, BENCH_TEST_HANG_LEG="startup", BENCH_LEG_BUDGET_S="2"))
    assert r.returncode == 0, r.stderr[-1000:]
    h = _strict(r.stdout.strip().splitlines()[-1])
    assert h["leg_timeouts"] == ["startup"] and h["value"] is None and h["n_gpus"] == 1 and "giving up on it" in r.stderr
