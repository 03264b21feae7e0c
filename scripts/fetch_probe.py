#!/usr/bin/env python3
"""Fetch ceiling of a CU (nst_probe_fetch): workgroups that only stream 32 KB pieces through the LDS-DMA ring of the stream
kernels (or plain register loads) and multiply nothing.  One JSON document on stdout / --out.

    shared  : every workgroup reads the same `span` bytes (L2-resident weight streams of the feed-forward / conv2 kernels)
    sliced  : 32 workgroups per region (the tiles of one split-K slice share their operand rows, as the weight gradients do)
    private : every workgroup its own region (operands streamed from HBM once)
Rates in TB/s over the chip and bytes per clock and CU at 2.4 GHz, for 1 and 2 workgroups per CU and ring depths 2..4."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd._lib import lib  # noqa: E402
from neurst_amd.kernels import check  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--steps", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    buf = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
    buf.random_(0, 255)
    sink = torch.zeros(4, device=dev)
    res = {"cus": cus, "steps": a.steps, "piece_bytes": 32768, "cases": []}
    stream = torch.cuda.current_stream().cuda_stream

    def run(name, wgs, ring, mode, span, wg_stride_of, group_mod=None, steps=None, pat=0):
        steps = steps or a.steps

        def launch():
            check(lib.nst_probe_fetch(buf.data_ptr(), wg_stride_of, span, steps, ring, mode, wgs, group_mod or wgs, pat,
                                      sink.data_ptr(), stream), "probe_fetch")
        for _ in range(2):
            launch()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            launch()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        us = sorted(ts)[len(ts) // 2]
        total = wgs * steps * 32768
        res["cases"].append({"pattern": name, "lane_map": ["contiguous", "reduction_major_swizzled", "row_major_swizzled",
                                                              "reduction_major_rows_4KB_apart", "row_major_rows_4KB_apart"][pat], "workgroups": wgs, "per_cu": wgs / cus, "ring": ring,
                             "loads": "lds_dma" if mode == 0 else "registers", "us": us, "tb_per_s": total / us / 1e6,
                             "bytes_per_clk_per_cu": total / us / 1e6 * 1e12 / 2.4e9 / cus})
        print(res["cases"][-1], flush=True)

    for mode in (0, 1):
        for per_cu in (1, 2):
            wgs = cus * per_cu
            for ring in ((2, 3, 4) if per_cu == 1 else (2,)):
                if ring * 32768 * per_cu > 160 * 1024:
                    continue
                run("shared_1MB", wgs, ring, mode, 1 << 20, 0)
                span = min(a.steps * 32768, buf.numel() // wgs // 32768 * 32768)   # (a region is re-read when it is shorter than the run)
                assert span * wgs <= buf.numel()
                run("private", wgs, ring, mode, span, span)
                # 8 regions of 16 MB, each read ONCE by the 32 (64) workgroups of an XCD in step: the weight gradient's slices
                run("sliced_per_xcd", wgs, ring, mode, 16 << 20, 16 << 20, group_mod=8, steps=512)
    for per_cu in (1, 2):   # the lane maps of the stream GEMM's tile images on the shared and the sliced stream
        for pat in (0, 1, 2, 3, 4):
            run("shared_1MB", cus * per_cu, 2, 0, 1 << 20, 0, pat=pat)
            run("sliced_per_xcd", cus * per_cu, 2, 0, 16 << 20, 16 << 20, group_mod=8, steps=512, pat=pat)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
