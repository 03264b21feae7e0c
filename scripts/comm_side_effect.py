#!/usr/bin/env python
"""Does creating an RCCL communicator change how fast unrelated kernels run?  Times a fixed workload (the library's LayerNorm
backward at the encoder shape, a torch elementwise pass, a torch matrix product) before / after nst_comm_init / after
nst_comm_destroy, with and without a torch.distributed process group in the process.

    python scripts/comm_side_effect.py [--torch-pg] [--second-torch-group]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--torch-pg", action="store_true")
    ap.add_argument("--second-torch-group", action="store_true")
    ap.add_argument("--no-native", action="store_true")
    args = ap.parse_args()
    from neurst_amd import kernels as K
    if args.torch_pg:
        os.environ["NST_DIST_FORCE"] = "1"
    from neurst_amd.training.distributed import NativeComm, init_distributed
    init_distributed()
    dev = "cuda:0"
    rows, d = 28800, 256
    x = torch.randn(rows, d, device=dev).bfloat16()
    dy = torch.randn(rows, d, device=dev).bfloat16()
    gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    y, mean, rstd = K.layernorm_fwd(x, gamma, beta, 1e-6)
    big = torch.randn(64 << 20, device=dev)
    a = torch.randn(4096, 4096, device=dev).bfloat16()

    def timed(fn, iters=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return round(s.elapsed_time(e) / iters * 1e3, 1)

    def workload(tag):
        dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
        r = {"ln_bwd_us": timed(lambda: K.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db)),
             "ln_fwd_us": timed(lambda: K.layernorm_fwd(x, gamma, beta, 1e-6)),
             "torch_mul_256MB_us": timed(lambda: big.mul_(1.0)),
             "torch_mm_4096_us": timed(lambda: torch.mm(a, a))}
        print(f"{tag:36s} {r}", flush=True)

    workload("before")
    if args.second_torch_group:
        import torch.distributed as dist
        g2 = dist.new_group([0])
        t = torch.zeros(8, device=dev)
        dist.all_reduce(t, group=g2)
        torch.cuda.synchronize()
        workload("after a second torch group")
    if not args.no_native:
        comm = NativeComm()
        workload("after nst_comm_init")
        t = torch.zeros(1024, device=dev)
        comm.allreduce_bucket(t, [torch.cuda.current_stream()])
        comm.fence(torch.cuda.current_stream())
        torch.cuda.synchronize()
        workload("after one bucket")
        comm.destroy()
        workload("after nst_comm_destroy")


if __name__ == "__main__":
    main()
