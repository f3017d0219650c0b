#!/usr/bin/env python3
"""One all-reduce and one barrier through the engine's OWN communicator (chz_comm_create -> RCCL behind the C ABI), one rank per GPU:
the first thing scripts/first_multigpu_run.sh runs.  Launch: python -m torch.distributed.run --nproc-per-node N scripts/rccl_sanity.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import __graft_entry__ as ge

rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dist.init_process_group("gloo", rank=rank, world_size=world)          # control plane only: ships the RCCL id
pkg = ge.load()
uid = [pkg.engine.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
t0 = time.perf_counter()
comm = pkg.engine.Comm(rank, world, uid[0], device=local)
t1 = time.perf_counter()
got = comm.allreduce_max([float(rank + 1)])
comm.barrier()
t2 = time.perf_counter()
assert abs(got[0] - world) < 1e-12, got
print("rank %d/%d on device %d: communicator in %.2f s, all-reduce(max) = %g, barrier ok (%.3f s)" % (rank, world, local, t1 - t0, got[0], t2 - t1), flush=True)
comm.close()
dist.destroy_process_group()
