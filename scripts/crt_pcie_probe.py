#!/usr/bin/env python3
"""C_rt with the host link in the loop: bench.py's `c_rt_pcie` leg on its own, for other channel counts.
usage: crt_pcie_probe.py <million channels> <baseband|demod> <sync|pipelined> [blocks]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import bench
pkg = ge.load()
n = int(float(sys.argv[1]) * 1e6)
demod = sys.argv[2] == "demod"
pipe = sys.argv[3] == "pipelined"
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 200
wl = bench.workload_for(3, 0, 1, 0)
print(json.dumps(bench.crt_pcie_leg(pkg, wl, n, blocks, demod, 0, pipe)))
