"""scripts/many_small_engines.py -- what does an engine for a small master cost?  (wfm.c / stereod.c / rdsd.c / packetd.c / ctcss.c create one small private
master per channel or session; behind filter.h each becomes a full engine.)  Creates n engines of wfm's composite geometry (REAL, L = 7680, M = 7681, N = 15,360)
with three banks each, and reports device memory and creation time per engine."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ka9q-radio_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.cuda.init()
free0, total = torch.cuda.mem_get_info()
engs = []
t0 = time.time()
for i in range(n):
    e = pkg.engine.Engine(7680, 7681, 1, ring_blocks=8)          # 1 = REAL
    banks = [e.bank(1920, 960, 4), e.bank(1920, 960, 4), e.bank(1920, 960, 4, real=True)]
    engs.append((e, banks))
    if i in (0, 1, 3, 7, 15, 31, 63):
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info()
        print("%3d engines: %.1f MB of device memory each, %.0f ms each" % (i + 1, (free0 - free) / (i + 1) / 1e6, (time.time() - t0) / (i + 1) * 1e3), flush=True)
t1 = time.time()
for e, _ in engs:
    e.close()
print("destroy: %.0f ms each" % ((time.time() - t1) / n * 1e3))
