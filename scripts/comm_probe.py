#!/usr/bin/env python
"""Host and device cost of one bucket through the library's RCCL communicator (nst_comm_*) next to torch.distributed, on a
one-rank group (the collectives are identities: what is measured is everything around them).

    NST_DIST_FORCE=1 python scripts/comm_probe.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NST_DIST_FORCE", "1")
from neurst_amd.training.distributed import NativeComm, init_distributed  # noqa: E402
import torch.distributed as dist  # noqa: E402


def timed(fn, iters=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    host = (time.perf_counter() - t0) / iters * 1e6
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / iters * 1e6
    return round(host, 1), round(total, 1)


def main():
    init_distributed()
    comm = NativeComm()
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    for n in (1024, 32 << 20):
        x = torch.zeros(n, device="cuda")
        print(f"n = {n} floats")
        print("  native  allreduce + fence      (host us, total us):", timed(lambda: (comm.allreduce_bucket(x, [cur]), comm.fence(cur))))
        print("  native  allreduce, 2 producers + fence           :", timed(lambda: (comm.allreduce_bucket(x, [cur, side]), comm.fence(cur))))
        print("  native  allreduce only                           :", timed(lambda: comm.allreduce_bucket(x, [cur])))
        print("  native  fence only                               :", timed(lambda: comm.fence(cur)))

        def torch_path():
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                h = dist.all_reduce(x, async_op=True)
            h.wait()
            cur.wait_stream(side)
        print("  torch   all_reduce on a side stream + waits      :", timed(torch_path))
    # does any call BLOCK the host while the producer stream is busy?  (~20 ms of matrix products queued first)
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    x = torch.zeros(1 << 20, device="cuda")

    def busy():
        for _ in range(40):
            torch.mm(a, a)

    def host_us(fn):
        torch.cuda.synchronize()
        busy()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        return round((t1 - t0) * 1e6, 1), round((time.perf_counter() - t0) * 1e6, 1)
    for _ in range(2):
        print("busy producer: (host us of the call, us until the device is idle)")
        print("  native allreduce, no producers          :", host_us(lambda: comm.allreduce_bucket(x, [])))
        print("  native allreduce, producer = busy stream:", host_us(lambda: comm.allreduce_bucket(x, [cur])))
        print("  native fence                            :", host_us(lambda: comm.fence(cur)))
        print("  native allreduce + fence                :", host_us(lambda: (comm.allreduce_bucket(x, [cur]), comm.fence(cur))))
        print("  torch all_reduce on the busy stream     :", host_us(lambda: dist.all_reduce(x)))
    comm.destroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
