import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, __graft_entry__ as ge, bench, oracle_lib as ol
pkg = ge.load()
L, M, N = bench.L, bench.M, bench.N
plan = bench.channel_plan_config3(1024)
resp = np.stack([pkg.filterapi.design_response(300, 240, N, True, lo, hi, 11.0) for _, lo, hi in plan])
shifts = np.array([p[0] for p in plan], np.int32)
rings = {"siggen": bench.siggen_ring(ol), "normal": np.random.default_rng(0).standard_normal(8 * L).astype(np.float32)}
for rname, x in list(rings.items()) + list(rings.items())[::-1]:
    for respname, r in (("designed", resp), ("ones", np.ones((1024, 300), np.complex64))):
        eng = pkg.engine.Engine(L, M, pkg.engine.REAL, ring_blocks=8)
        eng.write(x[:8 * L - (M - 1)]); eng.write(x[8 * L - (M - 1):])
        b = eng.bank(300, 240, 1024); b.set_responses(0, r); b.set_shifts(0, shifts); b.set_active(1024)
        eng.set_notches([0], 0.01)
        eng.run_blocks(0, 200, graph=False)
        t = eng.run_blocks(200, 2000, graph=False)
        for rep in range(3):
            it = eng.run_blocks(0, 200, graph=False, instrument=True)
            print("%-7s %-8s %.2f us/blk | first %.1f cols %.1f rows %.1f chan %.1f" % (rname, respname, t.total_ms / 2000 * 1e3,
                  it.first_ms/it.first_n*1e3, it.cols_ms/it.cols_n*1e3, it.rows_ms/it.rows_n*1e3, it.chan_ms/it.chan_n*1e3))
        t1 = eng.run_blocks(0, 400, graph=False); lanes = eng.lanes
        os.environ["CHZ_STREAMS"] = "1"
        eng.close()
