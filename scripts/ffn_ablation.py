#!/usr/bin/env python3
"""Where the time of the eight-wave feed-forward pair goes: the ablation twin of the library (make -C neurst_amd/csrc ablation)
launches the SAME kernel with stages compiled out (NST_FFN_DBG bits, nst_ffn.hip) at 28 800 rows x 2048 hidden units, forward
(dropout 0.1 / 0.1, gate bits) and backward (gate bits).  Results of an ablated launch are wrong by construction; the durations
are what is read.  Cold = a 768 MB fill in front of every launch (subtracted): the in-step condition.

    NST_LIBRARY=neurst_amd/lib/libneurst_hip_ablation.so python scripts/ffn_ablation.py [tag]
-> gpurun_out/<tag>_ffn_ablation.json
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from neurst_amd import kernels as K, _lib   # noqa: E402
from ffn_cost_model import cold, _time       # noqa: E402

STAGES = {1: "hidden-tile + gate-bit stores", 2: "weight DMA of the loop", 4: "weight fragment reads", 8: "MFMAs",
          16: "the loop's two barriers per chunk", 32: "mid-epilogue arithmetic (bias / ReLU / Philox / gate)",
          64: "P tile writes + reads"}
CASES = [0, 1, 2, 4, 8, 16, 32, 64, 3, 71, 76, 103, 119, 127]


def describe(bits):
    if bits == 0:
        return "the kernel as shipped"
    return "without: " + "; ".join(v for k, v in STAGES.items() if bits & k)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    assert "ablation" in _lib.LIB_PATH, "run with NST_LIBRARY=<...>/libneurst_hip_ablation.so"
    dev, M, d, F, p = "cuda:0", 28800, 256, 2048, 0.1
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, d, generator=g).to(dev).bfloat16()
    dy = torch.randn(M, d, generator=g).to(dev).bfloat16()
    w1 = (torch.randn(d, F, generator=g) * d ** -0.5).to(dev).bfloat16()
    w2 = (torch.randn(F, d, generator=g) * F ** -0.5).to(dev).bfloat16()
    w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
    b1, b2 = torch.zeros(F, device=dev), torch.zeros(d, device=dev)
    os.environ.pop("NST_FFN_DBG", None)
    y, h, bits = K.ffn_fwd(x, w1t, b1, w2t, b2, hidden_p=p, hidden_seed=1, hidden_site=1, out_p=p, out_seed=1, out_site=2,
                           save_gate_bits=True)
    fwd = lambda: K.ffn_fwd(x, w1t, b1, w2t, b2, hidden_p=p, hidden_seed=1, hidden_site=1, out_p=p, out_seed=1, out_site=2,   # noqa: E731
                            save_gate_bits=True)
    bwd = lambda: K.ffn_bwd(dy, h, w2, w1, hidden_p=p, gate_bits=bits)   # noqa: E731
    res = {"rows": M, "filter": F, "library": os.path.basename(_lib.LIB_PATH), "stages": {str(k): v for k, v in STAGES.items()},
           "mfma_floor_us_at_1.95GHz": round(4.0 * M * d * F / (256 * 4096 * 1.95e9) * 1e6, 1), "cases": []}
    for c in CASES:
        os.environ["NST_FFN_DBG"] = str(c)
        row = {"bits": c, "what": describe(c),
               "fwd_cold_us": round(cold(fwd), 1), "bwd_cold_us": round(cold(bwd), 1),
               "fwd_warm_us": round(_time(fwd), 1), "bwd_warm_us": round(_time(bwd), 1)}
        res["cases"].append(row)
        print(json.dumps(row), flush=True)
    os.environ.pop("NST_FFN_DBG", None)
    root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", f"{tag}_ffn_ablation.json"), "w") as fp:
        json.dump(res, fp, indent=1)


if __name__ == "__main__":
    main()
