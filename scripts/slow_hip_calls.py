#!/usr/bin/env python3
"""List the HIP API calls of a rocprofv3 --hip-trace run (csv) that took longer than 1 ms, in time order, with the copies / kernels that
were in flight around them.  usage: python scripts/slow_hip_calls.py <trace dir>"""
import csv
import glob
import os
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        rows.append((t0, t1, r.get("Function", "?"), r.get("Thread_Id", "?")))
rows.sort()
if not rows:
    print("no hip_api_trace.csv under", d, os.listdir(d) if os.path.isdir(d) else "")
    sys.exit(0)
base = rows[0][0]
# the block loop starts after set-up: report the slow calls of the LAST 60 % of the trace and, for context, of all of it
slow = [(t0, t1, fn, th) for t0, t1, fn, th in rows if t1 - t0 > 1_000_000]
print("%d HIP API calls, %d longer than 1 ms" % (len(rows), len(slow)))
for t0, t1, fn, th in slow:
    print("  t=%9.3f ms  %8.3f ms  %-36s thread %s" % ((t0 - base) / 1e6, (t1 - t0) / 1e6, fn, th))
# context of the slow calls that come after the set-up (the block loop): the calls of the same thread right before and after
last_setup = max([t0 for t0, t1, fn, th in rows if fn in ("hipHostRegister", "hipStreamCreateWithFlags", "hipExtStreamCreateWithCUMask")] or [base])
for t0, t1, fn, th in slow:
    if t0 < last_setup or fn in ("hipStreamDestroy", "hipHostFree", "hipFree"):
        continue
    mine = [r for r in rows if r[3] == th]
    i = mine.index((t0, t1, fn, th))
    print("context of the %.3f ms %s at t=%.3f ms:" % ((t1 - t0) / 1e6, fn, (t0 - base) / 1e6))
    for a0, a1, afn, _ in mine[max(0, i - 14):i + 4]:
        print("      t=%9.3f ms  %8.3f ms  %s%s" % ((a0 - base) / 1e6, (a1 - a0) / 1e6, afn, "   <====" if a0 == t0 else ""))
copies = []
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?") + " " + " ".join("%s=%s" % (k, v) for k, v in r.items() if k not in ("Start_Timestamp", "End_Timestamp", "Direction", "Kind", "Correlation_Id"))))
copies.sort()
big = [c for c in copies if c[1] - c[0] > 1_000_000]
print("%d memory copies, %d longer than 1 ms" % (len(copies), len(big)))
for t0, t1, k in big[:40]:
    print("  t=%9.3f ms  %8.3f ms  %s" % ((t0 - base) / 1e6, (t1 - t0) / 1e6, k))
print("all copies after set-up:")
for t0, t1, k in copies:
    if t0 >= last_setup:
        print("  t=%9.3f ms  %8.3f ms  %s" % ((t0 - base) / 1e6, (t1 - t0) / 1e6, k))
