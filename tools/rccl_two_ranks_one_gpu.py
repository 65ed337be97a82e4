#!/usr/bin/env python3
"""Can RCCL run two ranks on ONE device?  (The pool's boxes have one GPU; the replay engine's RCCL branch needs >= 2 ranks to move a byte
between processes.)  Spawns two processes on cuda:0, all-gathers a small tensor, prints what happened.  python tools/rccl_two_ranks_one_gpu.py"""
import os, sys, traceback
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda:0"))
        x = torch.full((4,), float(rank + 1), device="cuda:0")
        out = torch.empty(8, device="cuda:0")
        dist.all_gather_into_tensor(out, x)
        torch.cuda.synchronize()
        print(f"rank {rank}: gathered {out.tolist()}", flush=True)
        dist.destroy_process_group()
    except Exception as e:   # noqa: BLE001 — the point is to see the message
        print(f"rank {rank}: FAILED {type(e).__name__}: {str(e)[:600]}", flush=True)
        traceback.print_exc(limit=1)
        sys.exit(3)


if __name__ == "__main__":
    print("torch", torch.__version__, "nccl/rccl", torch.cuda.nccl.version() if torch.cuda.is_available() else None, flush=True)
    mp.start_processes(worker, args=(29653,), nprocs=2, join=True, start_method="spawn")
