#!/usr/bin/env python3
"""Summarise rocprofv3 output (rocpd sqlite .db from --kernel-trace, or the csv files of a --pmc pass)
into the small text tables committed under profiles/.   usage: rocprof_summary.py <dir-or-db> [...]"""
import csv, glob, os, sqlite3, sys
from collections import defaultdict

def tree_sha16():
    """fingerprint of the kernel sources a profile was taken from (bench.py compares it with the tree it runs)"""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("chz_kernels.h", "regfft.h", "chz_plan.h", "chz_launch.h"):
        h.update(open(os.path.join(root, "ka9q-radio_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]

def short(n):
    return n.replace("void chz::", "").replace("chz::", "").split("(")[0]

def db_summary(path):
    con = sqlite3.connect(path)
    rows = list(con.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
                            "max(grid_x), max(workgroup_x), max(vgpr_count), max(lds_size) from kernels group by name order by 6 desc"))
    tot = sum(r[5] for r in rows)
    print("# %s" % path)
    print("%-28s %7s %9s %9s %9s %6s %9s %6s %5s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct", "grid", "wg", "vgpr", "lds_B"))
    for r in rows:
        print("%-28s %7d %9.2f %9.2f %9.2f %6.1f %9d %6d %5d %7d" % (short(r[0])[:28], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                                   100.0 * r[5] / tot, r[6], r[7], r[8], r[9]))

def csv_summary(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
    for f in files:
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"]); c = row["Counter_Name"]
            acc[k][c] += float(row["Counter_Value"]); cnt[k][c] += 1
    print("# %s (per-dispatch averages)" % d)
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            print("    %-24s %16.1f   (n=%d)" % (c, acc[k][c] / cnt[k][c], cnt[k][c]))

def pmc_json(fetch_dir, write_dir, out):
    """forward-transform HBM traffic per block from separate FETCH_SIZE / WRITE_SIZE passes"""
    import json
    def avg(d, counter):
        acc = defaultdict(float); cnt = defaultdict(int)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == counter:
                    k = short(row["Kernel_Name"]); acc[k] += float(row["Counter_Value"]); cnt[k] += 1
        return {k: acc[k] / cnt[k] for k in acc}
    fe, wr = avg(fetch_dir, "FETCH_SIZE"), avg(write_dir, "WRITE_SIZE")
    kern = {}
    total = 0.0
    for k in fe:
        if not k.startswith("fwd_"):
            continue
        # counters are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced stream's bytes
        b = (2.0 * fe[k] + wr.get(k, 0.0)) * 1024.0
        kern[k] = {"FETCH_SIZE_KiB": fe[k], "WRITE_SIZE_KiB": wr.get(k, 0.0), "bytes_corrected": b}
        total += b
    json.dump({"forward_traffic_bytes_per_block": total, "kernels": kern, "kernel_sources_sha16": tree_sha16(),
               "note": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 wide reads)"},
              open(out, "w"), indent=1)
    print("wrote", out, total)

def kernels_json(dir4, dir1, out, tag, plain=""):
    """average kernel durations of the bench command under rocprofv3 --kernel-trace --stats (4 streams and 1 stream); `plain` = the detail file of
    the SAME command run unprofiled on the SAME box right before (round 6: boxes differ by 10 %, so a trace from one box and HIP-event
    durations from another do not say whether the two methods agree)"""
    import json
    same_box = None
    if plain and os.path.exists(plain):
        try:
            r = json.load(open(plain))["roofline"]
            same_box = {"hip_event_kernels_us": r["kernels_us"], "hip_event_forward_us": r["forward_us_per_block"], "frac": r["frac"]}
        except Exception as ex:
            same_box = {"error": str(ex)[:200]}
    def avgs(d):
        r = {}
        for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            con = sqlite3.connect(db)
            for name, n, avg in con.execute("select name, count(*), avg(end-start) from kernels group by name"):
                r[short(name)] = {"calls": n, "avg_us": avg / 1e3}
        return r
    k4, k1 = avgs(dir4), avgs(dir1)
    fwd1 = sum(v["avg_us"] for k, v in k1.items() if k.startswith("fwd_"))
    fwd4 = sum(v["avg_us"] for k, v in k4.items() if k.startswith("fwd_"))
    json.dump({"round": tag, "command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --quick", "kernel_sources_sha16": tree_sha16(),
               "forward_us_1_stream": fwd1, "forward_us_4_streams_sum_of_overlapping_kernels": fwd4,
               "kernels_1_stream": k1, "kernels_4_streams": k4,
               "same_box_unprofiled": same_box,
               "rocprof_over_hip_events": (fwd1 / same_box["hip_event_forward_us"]) if same_box and same_box.get("hip_event_forward_us") else None,
               "note": "CHZ_STREAMS=1: one kernel at a time, comparable with roofline.kernels_us; profiled runs are about 10 % slower than unprofiled ones"},
              open(out, "w"), indent=1)
    print("wrote", out, "forward 1 stream %.2f us" % fwd1)

if len(sys.argv) > 1 and sys.argv[1] == "--kernels-json":
    kernels_json(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "", sys.argv[6] if len(sys.argv) > 6 else ""); sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "--json":
    pmc_json(sys.argv[2], sys.argv[3], sys.argv[4]); sys.exit(0)

for a in sys.argv[1:]:
    if a.endswith(".db"):
        db_summary(a)
    elif os.path.isdir(a):
        dbs = glob.glob(os.path.join(a, "**", "*.db"), recursive=True)
        if dbs and not glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True):
            for d in dbs: db_summary(d)
        else:
            csv_summary(a)
