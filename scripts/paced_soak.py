"""scripts/paced_soak.py [threads] [blocks] -- the filter.h drop-in at wall-clock pace for minutes, not seconds: BASELINE config 3's master (129.6 MS/s real), one pthread per
channel, the front end on its own 20 ms clock (tests/c/dropin_harness.c, 4 input blocks replayed).  Reports drops, skipped blocks and the latency distribution."""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
import test_dropin as T
nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
T._build_lib(); ol.build()
fs, L, M, olen = 129.6e6, 2592000, 648001, 240
N = L + M - 1
ring = 4
g = ol.SigGen(10.00002e6 / fs, 0.1, 0.01, ol.scale_ad(True, 1), True, seed=1)
x = g.generate(ring * L)
kinds = [(50 / 12000, 3000 / 12000), (-200 / 12000, 200 / 12000), (-5000 / 12000, 5000 / 12000)]
plan = []
for i in range(nthreads):
    shift = ol.compute_tuning(N, fs, 1e6 + (i % 1040) * 60e3 + (i % 40))[1]
    lo, hi = kinds[i % 3]
    plan.append((shift, shift, 10 ** 9, 10 ** 9, lo, hi, 11.0, lo, hi))
env = {"HARNESS_PACED_US": "20000", "HARNESS_INPUT_BLOCKS": str(ring), "HARNESS_KEEP": "0", "KA9Q_HIP_PROFILE": "1"}
with tempfile.TemporaryDirectory() as tmp:
    exe = os.path.join(tmp, "harness")
    T._build_harness(exe)
    open(os.path.join(tmp, "cfg.txt"), "w").write("%d %d %d %d %d %d %d\n" % (L, M, ol.REAL, olen, len(plan), nblocks, 65536))
    with open(os.path.join(tmp, "plan.bin"), "wb") as f:
        for p in plan:
            f.write(struct.pack("iiiiddddd", *p))
    x.tofile(os.path.join(tmp, "in.bin"))
    r = subprocess.run([exe, tmp], capture_output=True, text=True, timeout=nblocks * 0.02 + 300, env=dict(os.environ, **env))
    print("rc", r.returncode)
    meta = open(os.path.join(tmp, "meta.txt")).read().split()
    meta = dict(zip(meta[::2], meta[1::2]))
    lat = np.fromfile(os.path.join(tmp, "latency.bin"), np.int64).reshape(nblocks, 2)
    ms = lat[:, 0] / 1e6
    print("threads", nthreads, "blocks", nblocks, "drops", meta["drops"], "skipped", meta.get("skipped"), "served to every channel:", int((lat[:, 1] == nthreads).sum()), "of", nblocks)
    print("latency ms: p50 %.2f p99 %.2f p99.9 %.2f max %.2f (block %d); blocks over 10 ms: %d, over 20 ms: %d" % (np.percentile(ms, 50), np.percentile(ms, 99), np.percentile(ms, 99.9), ms.max(), int(ms.argmax()), int((ms > 10).sum()), int((ms > 20).sum())))
    print([ln[:300] for ln in r.stderr.splitlines() if "filter_hip profile" in ln])
