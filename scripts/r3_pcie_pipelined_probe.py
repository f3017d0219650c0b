#!/usr/bin/env python3
"""bench.py's double-buffered PCIe-inclusive leg (demodulated S16 PCM back to the host) at other channel counts.
usage: r3_pcie_pipelined_probe.py <channels> [<channels> ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import __graft_entry__ as ge
pkg = ge.load()
wl = bench.workload_for(3, 0, 1, 0)
pipelined = os.environ.get("PROBE_SYNC") != "1"          # PROBE_SYNC=1: the block-by-block loop (H2D -> kernels -> D2H -> sync)
for n in sys.argv[1:]:
    r = bench.crt_pcie_leg(pkg, wl, int(float(n)), 500, True, 0, pipelined)
    print(json.dumps({k: r[k] for k in ("channels", "blocks", "worst_block_ms", "mean_block_ms", "sustained", "d2h_GBps", "worst_latency_ms")}))
