#!/usr/bin/env python3
"""Secondary benchmark (BASELINE config "Transformer-big WMT14 en-de, bf16"): training steps of the
text Transformer on synthetic parallel-text batches, same contract as bench.py (W warm-up + K timed steps between
barrier+sync, max over ranks, one JSON line).  Not the headline metric: `bench.py` stays the SpeechTransformer line.
usage: python scripts/bench_text.py [--model transformer_big] [--batch 64] [--src-len 64] [--trg-len 64] [--steps 10]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def flops_forward(p, B, S, L, V):
    d, H, ffn = p["modality.dim"], p["encoder.num_attention_heads"], p["encoder.filter_size"]
    Ne, Nd = p["encoder.num_layers"], p["decoder.num_layers"]
    M, Md, dh = B * S, B * L, d // H
    enc = Ne * (2 * M * d * 3 * d + 4 * B * H * S * S * dh + 2 * M * d * d + 4 * M * d * ffn)
    dec = Nd * (2 * Md * d * 3 * d + 4 * B * H * L * L * dh + 2 * Md * d * d + 2 * Md * d * d + 2 * M * d * 2 * d
                + 4 * B * H * L * S * dh + 2 * Md * d * d + 4 * Md * d * ffn)
    return enc + dec + 2 * Md * d * V


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="transformer_big")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=64, help="sentence pairs per GPU per step")
    ap.add_argument("--src-len", type=int, default=64)
    ap.add_argument("--trg-len", type=int, default=64)
    ap.add_argument("--vocab", type=int, default=32003)
    a = ap.parse_args()
    from neurst_amd.criterions import build_criterion
    from neurst_amd.data.datasets.synthetic_text import SyntheticTextDataset
    from neurst_amd.optimizers import build_lr_schedule, build_optimizer
    from neurst_amd.tasks import build_task
    from neurst_amd.training.distributed import GradientReducer, init_distributed
    from neurst_amd.training.train_step import TrainStep
    from neurst_amd.utils import compat
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    import torch.distributed as dist
    rank, local_rank, world = init_distributed()
    dev = f"cuda:{local_rank}"
    hp = get_hyper_parameters(a.model)
    task = build_task({"task.class": "translation", "task.params": {"src_vocab_size": a.vocab, "trg_vocab_size": a.vocab}})
    model = task.build_model(hp, device=dev, dtype="bfloat16" if a.dtype == "bf16" else "float32", seed=1234 + rank, init_seed=42)
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    opt = build_optimizer({"optimizer.class": hp["optimizer.class"], "optimizer.params": hp["optimizer.params"]})
    opt.bind(model.store)
    opt.learning_rate = build_lr_schedule({"lr_schedule.class": hp["lr_schedule.class"], "lr_schedule.params": hp["lr_schedule.params"]})
    reducer = GradientReducer(model.store)
    reducer.broadcast_parameters(0)
    step_fn = TrainStep(model, crit, opt, reducer, use_graph=os.environ.get("NST_TRAIN_GRAPH", "1") != "0")   # graph replay like bench.py
    ds = SyntheticTextDataset({"batch_per_gpu": a.batch, "src_len": a.src_len, "trg_len": a.trg_len, "src_vocab_size": a.vocab,
                               "trg_vocab_size": a.vocab, "seed": 1234})
    it = ds.build_iterator(map_func=lambda b: task.example_to_input(b, compat.ModeKeys.TRAIN), shard_id=rank,
                           total_shards=world, device=dev)
    batches = [next(it) for _ in range(4)]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for i in range(a.warmup):
        step_fn(batches[i % 4])
    barrier()
    t0 = time.perf_counter()
    loss = None
    for i in range(a.steps):
        loss = step_fn(batches[i % 4])
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # roofline pass (un-timed): two more steps with HIP events around every launch of the MFMA kernel families (every rank runs
    # them -- they contain the exchange -- rank 0 records), priced like bench.py's roofline object
    from neurst_amd import kernels as K
    fams = ("gemm", "gemm_wgrad_group", "ffn_fwd", "ffn_bwd", "attention_fwd", "attention_bwd")
    probe, psteps = {}, 2
    step_fn.use_graph = False
    if rank == 0:
        K.PROBE.start(fams)
    for i in range(psteps):
        step_fn(batches[i % 4])
    if rank == 0:
        probe = K.PROBE.stop()
    barrier()
    if rank != 0:
        return
    peak = 2500.0 if a.dtype == "bf16" else 157.3
    families = {}
    for name, rows in probe.items():
        ms, work, nbytes = sum(r[0] for r in rows), sum(r[1] for r in rows), sum(r[2] for r in rows)
        if ms <= 0:
            continue
        families[name] = {"bound": "mfma", "unit": "TFLOP/s", "peak": peak, "launches_per_step": len(rows) / psteps,
                          "ms_per_step": ms / psteps, "avg_launch_ms": ms / len(rows), "algorithmic_flops_per_step": work / psteps,
                          "achieved": work / (ms * 1e-3) / 1e12, "frac": work / (ms * 1e-3) / 1e12 / peak, "traffic": None}
        if nbytes > 0:
            families[name]["hbm_bound"] = {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                           "frac": nbytes / (ms * 1e-3) / 1e9 / 8000.0}
    roofline = None
    if families:
        top = max(families, key=lambda n: families[n]["ms_per_step"])
        roofline = dict(families[top], kernel=top, selected_as="largest share of in-step GPU time among the MFMA kernel families")
    tokens = world * a.batch * (a.src_len + a.trg_len) * a.steps
    fl = 3 * flops_forward(hp["model.params"], a.batch, a.src_len, a.trg_len, a.vocab)
    print(json.dumps({
        "metric": f"source+target tokens/sec, {a.model} training, whole job", "value": tokens / elapsed, "unit": "tokens/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"{a.model} train step: B={a.batch}/GPU x (S={a.src_len} + L={a.trg_len}) tokens, V={a.vocab}, "
                               f"label smoothing 0.1, Adam+Noam", "global_batch": world * a.batch, "seq_len": a.src_len,
                   "parallelism": f"dp{world}"},
        "model_tflops_per_s": fl * a.steps * world / elapsed / 1e12,
        "model_mfma_frac": fl * a.steps / elapsed / 1e12 / peak, "roofline": roofline, "roofline_families": families,
        "final_loss": float(loss),
        "params": int(model.store.total)}))


if __name__ == "__main__":
    main()
