#!/usr/bin/env python3
"""What this MI355X really streams through HBM when the working set is far beyond the 256 MB Infinity Cache: device-to-device copies
of 8 ... 48 GB (read + write, the mix the C_rt leg has), a read-only reduction and a write-only fill.  The 6.29 TB/s 'copy rate' used
next to the forward transform comes from 13 MB buffers (cache-resident); this is the figure the channel kernel at 17 M channels
(41 GB of responses in, 33 GB of outputs out per block) has to be held against."""
import json, time
import torch
res = {}
dev = torch.device("cuda:0")
for gb in (8, 24, 48):
    n = gb * (1 << 30) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).fill_(1.0)
    b = torch.empty(n, dtype=torch.float32, device=dev)
    for name, fn, bytes_moved in (("copy", lambda: b.copy_(a), 2 * n * 4), ("read", lambda: a.sum(), n * 4), ("fill", lambda: b.fill_(2.0), n * 4)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res["%s_%dGB" % (name, gb)] = round(bytes_moved / dt / 1e12, 3)
    del a, b
    torch.cuda.empty_cache()
print(json.dumps({"unit": "TB/s (bytes read + bytes written per second)", **res}))
