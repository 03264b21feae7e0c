#!/usr/bin/env python3
"""Inference benchmark: beam-search decoding of the SpeechTransformer with the incremental K/V cache
on synthetic utterances (random-init weights, so hypotheses run to the step limit: every step is timed at full beam).
Reports utterances/s, generated tokens/s and the split encoder / per decoding step.  Not the headline metric.
usage: python scripts/bench_decode.py [--model speech_transformer_s] [--batch 32] [--frames 900] [--beam 4] [--max-len 75]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="speech_transformer_s")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=900)
    ap.add_argument("--beam", type=int, default=4)
    ap.add_argument("--max-len", type=int, default=75)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--graphs", action="store_true", help="replay one captured HIP graph per decoding step")
    a = ap.parse_args()
    from neurst_amd.layers.search import BeamSearch
    from neurst_amd.models import build_model
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    V, dev = 8008, "cuda:0"
    model = build_model(get_hyper_parameters(a.model), {"audio_feature_dim": 80, "audio_feature_channels": 1},
                        {"vocab_size": V, "eos_id": V - 1, "bos_id": V - 2, "unk_id": V - 3}, device=dev,
                        dtype="bfloat16" if a.dtype == "bf16" else "float32", seed=1234)
    g = torch.Generator().manual_seed(0)
    inputs = {"src": torch.randn(a.batch, a.frames, 80, 1, generator=g).to(dev),
              "src_length": torch.full((a.batch,), a.frames, dtype=torch.int64, device=dev)}
    search = BeamSearch(beam_size=a.beam, length_penalty=0.6, maximum_decode_length=a.max_len, extra_decode_length=0,
                        minimum_decode_length=a.max_len, use_graphs=a.graphs)   # EOS masked until the last step: fixed amount of work
    hyp, _ = search(model, inputs)                          # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.reps):
        hyp, scores = search(model, inputs)
    torch.cuda.synchronize()
    total = (time.time() - t0) / a.reps
    # encoder alone
    t0 = time.time()
    for _ in range(a.reps):
        model.get_symbols_to_logits_fn(inputs, beam_size=a.beam, decode_padded_length=a.max_len, use_graphs=a.graphs)
    torch.cuda.synchronize()
    enc = (time.time() - t0) / a.reps
    steps = int(hyp.shape[1])
    out = {"metric": "beam-search decoding, SpeechTransformer (synthetic, random weights)", "model": a.model, "dtype": a.dtype,
           "batch": a.batch, "frames": a.frames, "beam_size": a.beam, "decode_steps": steps, "hip_graphs": bool(a.graphs),
           "ms_per_batch": total * 1e3, "ms_encoder_and_cache": enc * 1e3, "ms_per_decode_step": (total - enc) / steps * 1e3,
           "utterances_per_s": a.batch / total, "generated_tokens_per_s": a.batch * steps / total,
           "beam_tokens_per_s": a.batch * a.beam * steps / total}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
