import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as ge
if os.environ.get("WITH_TORCH"):
    import torch; torch.cuda.init(); torch.cuda.synchronize(); print("torch loaded")
pkg = ge.load()
L = 2592000; M = L // 4 + 1
x = np.random.default_rng(0).standard_normal(8 * L).astype(np.float32)
eng = pkg.engine.Engine(L, M, pkg.engine.REAL, ring_blocks=8)
eng.write(x[:8 * L - (M - 1)]); eng.write(x[8 * L - (M - 1):])
b = eng.bank(300, 240, 1024)
b.set_responses(0, np.ones((1024, 300), np.complex64)); b.set_shifts(0, 25000 + np.arange(1024) * 1500); b.set_active(1024)
eng.set_notches([0], 0.01)
def instr(tag):
    it = eng.run_blocks(0, 200, graph=False, instrument=True)
    print(tag, "first %.1f cols %.1f rows %.1f chan %.1f" % (it.first_ms/it.first_n*1e3, it.cols_ms/it.cols_n*1e3, it.rows_ms/it.rows_n*1e3, it.chan_ms/it.chan_n*1e3))
instr("cold")
eng.run_blocks(0, 200, graph=False); instr("after 200 blk")
t = eng.run_blocks(0, 4000, graph=False); print("4000 blk: %.2f us/blk" % (t.total_ms/4000*1e3)); instr("after 4000 blk")
instr("again")
t = eng.run_blocks(0, 20000, graph=False); print("20000 blk: %.2f us/blk" % (t.total_ms/20000*1e3)); instr("after 20000 blk")
import time; time.sleep(1.0); instr("after 1 s idle")
