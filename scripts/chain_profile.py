#!/usr/bin/env python3
"""The SURVEY 8f chain (tuned chan_ifft + noise_est + demodulator) of ONE mode on a bank of N channels, nothing else: the workload of the
rocprofv3 --pmc passes of scripts/gpu/call.sh chainpmc (bench.py's own leg sits behind the headline loops: tens of thousands of launches
that a counter pass serialises -- round 5's passes did not finish in 140 s).  The instruction counts per channel do not depend on N.
usage: python scripts/chain_profile.py linear|pll|fm [channels]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import __graft_entry__ as ge

mode = sys.argv[1] if len(sys.argv) > 1 else "linear"
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
pkg = ge.load()
wl = bench.workload_for(3, 0, 1, 1024)
r = bench.next_rows_leg(pkg, wl, nch, 0, mode)
print(json.dumps({k: r[k] for k in ("channels", "mode", "pipelined_ms_per_block", "ns_per_channel", "pcm_mismatches", "verified_channels")}))
