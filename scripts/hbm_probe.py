#!/usr/bin/env python
"""What plain streams reach on this box (a yardstick for the HBM-bound kernels): fill, copy and read-only reduce of 1.18 GB (the
size of the conv front end's largest activation), next to the shader clock rocm-smi reports while a matrix-bound kernel runs.

    python scripts/hbm_probe.py [--out gpurun_out/hbm_probe.json]
"""
import argparse
import json
import subprocess
import threading
import time

import torch


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def sclk():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            if "sclk" in line:
                return line.split(":")[-1].strip()
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    n = 128 * 450 * 40 * 256
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    y = torch.empty_like(x)
    nbytes = n * 2
    res = {"bytes": nbytes}
    res["fill_tb_per_s"] = nbytes / timed(lambda: x.zero_()) / 1e12
    res["fill_nonzero_tb_per_s"] = nbytes / timed(lambda: x.fill_(1.5)) / 1e12
    f32 = torch.empty(n // 2, dtype=torch.float32, device="cuda")
    res["fill_nonzero_f32_tb_per_s"] = nbytes / timed(lambda: f32.fill_(1.5)) / 1e12
    half = x[: n // 2]
    res["cast_f32_to_bf16_write_tb_per_s"] = (n // 2) * 2 / timed(lambda: half.copy_(f32)) / 1e12   # reads 4 B, writes 2 B per element
    res["copy_tb_per_s_read_plus_write"] = 2 * nbytes / timed(lambda: y.copy_(x)) / 1e12
    res["sum_tb_per_s"] = nbytes / timed(lambda: x.view(torch.int16).sum()) / 1e12
    # shader clock under a matrix-bound load (hipBLASLt 8192^3 bf16), sampled from another thread
    a8 = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    clocks, stop = [], []

    def sample():
        while not stop:
            c = sclk()
            if c:
                clocks.append(c)
            time.sleep(0.2)
    res["sclk_idle"] = sclk()
    th = threading.Thread(target=sample)
    th.start()
    t0 = time.time()
    it = 0
    while time.time() - t0 < 4.0:
        for _ in range(20):
            torch.mm(a8, a8)
        torch.cuda.synchronize()
        it += 20
    dt = time.time() - t0
    stop.append(1)
    th.join()
    res["mm_8192_tflops"] = 2 * 8192 ** 3 * it / dt / 1e12
    res["sclk_under_matrix_load"] = clocks
    print(json.dumps(res))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
