#!/bin/bash
# HBM traffic and MFMA busy time of EVERY kernel of the training step (three rocprofv3 --pmc passes over eager steps) ->
# gpurun_out/<tag>_pmc_whole_step.json: per kernel family bytes per step and MFMA-busy share, and the step's totals.
#   gpurun --timeout 900 -- 'bash scripts/pmc_whole_step.sh r03'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-r03}
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  bash scripts/pmc_kernel.sh gpurun_out/${T}_pmc_whole_step_raw.json "" bench.py --eager --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > gpurun_out/${T}_pmc_whole_step.log 2>&1
python - "$T" "${2:-13.0}" <<'PY'
import json, re, sys
tag = sys.argv[1]
raw = json.load(open(f"gpurun_out/{tag}_pmc_whole_step_raw.json"))["summary"]
STEPS = 6.0   # 2 warm-up + 4 timed eager steps are all profiled
STEP_MS = float(sys.argv[2]) if len(sys.argv) > 2 else 13.0   # the graph-replay step time the totals are divided by


def fam(n):
    if "gemm256_group" in n or "g256_table" in n: return "weight-gradient GEMMs (grouped, 256 x 256 tiles)"
    if "dense_gemm_kernel_v3<unsigned short, float, 1, 1" in n: return "weight-gradient GEMMs"
    if "dense_gemm" in n: return "forward / input-gradient GEMMs"
    if "rowgemm" in n: return "whole-row products (product + the wrapper's LayerNorm stages)"
    if "ffn_slab_rows" in n: return "feed-forward pair"
    if "splitk" in n: return "split-K reduces"
    if "conv2" in n or "conv_splitk" in n: return "conv2"
    if "conv1" in n: return "conv1"
    if "ffn_pair" in n: return "feed-forward pair"
    if "attn" in n: return "attention"
    if "ln_" in n: return "LayerNorm"
    if "adam" in n: return "Adam"
    if "xent" in n: return "cross entropy"
    return "rest"


out = {}
for name, k in raw.items():
    f = out.setdefault(fam(name), {"hbm_bytes_per_step": 0.0, "mfma_busy_cycles_per_step": 0.0, "launches_per_step": 0.0})
    fetch = k.get("FETCH_SIZE", {}).get("sum", 0.0)
    write = k.get("WRITE_SIZE", {}).get("sum", 0.0)
    f["hbm_bytes_per_step"] += (2.0 * fetch + write) * 1024.0 / STEPS      # gfx950: FETCH_SIZE counts half of a wide read stream
    f["mfma_busy_cycles_per_step"] += k.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("sum", 0.0) / STEPS
    f["launches_per_step"] += k.get("FETCH_SIZE", {}).get("n", 0) / STEPS
tot_b = sum(v["hbm_bytes_per_step"] for v in out.values())
tot_m = sum(v["mfma_busy_cycles_per_step"] for v in out.values())
res = {"command": "scripts/pmc_whole_step.sh (bench.py --eager, 6 steps profiled, rocprofv3 --pmc passes FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES)",
       "correction": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024", "families": out,
       "hbm_bytes_per_step": tot_b, "mfma_busy_simd_cycles_per_step": tot_m,
       "step_ms_used": STEP_MS,
       "mfma_busy_fraction_of_the_step": tot_m / (1024.0 * STEP_MS * 1e-3 * 2.4e9),
       "hbm_rate_over_the_step_TB_per_s": tot_b / (STEP_MS * 1e-3) / 1e12}
json.dump(res, open(f"gpurun_out/{tag}_pmc_whole_step.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"]):
    print(f"{k:34s} {v['hbm_bytes_per_step'] / 1e9:7.2f} GB/step  mfma busy {v['mfma_busy_cycles_per_step'] / 1e6:9.1f} M SIMD-cycles  launches {v['launches_per_step']:.0f}")
print("total", round(tot_b / 1e9, 2), "GB/step;", round(res["hbm_rate_over_the_step_TB_per_s"], 2), f"TB/s over {STEP_MS} ms; MFMA busy", round(res["mfma_busy_fraction_of_the_step"], 3))
PY
