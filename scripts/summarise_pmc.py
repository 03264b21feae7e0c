#!/usr/bin/env python
"""Turn the raw per-kernel PMC tables of scripts/pmc_kernel.sh into the two small files bench.py reads:

  profiles/<tag>_pmc_ffn_gemm.json  {"summary": {...}}            MFMA utilisation of the FFN GEMM kernels
  profiles/<tag>_pmc_gemm.json      {"hbm_bytes_per_step": ...}   HBM bytes of the dense GEMM family in one train step
(both stamped with the git commit the counters were captured at: bench.py copies that stamp into its JSON line)

Counter arithmetic (MI355X_MICROARCH.md, rocprofv3 section):
  * SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over all SIMDs (32 per v_mfma_f32_32x32x16_bf16);
  * GRBM_GUI_ACTIVE counts busy cycles per XCD, summed over the 8 XCDs -> elapsed cycles = GRBM_GUI_ACTIVE / 8;
  * MFMA utilisation = MFMA busy cycles / (1024 SIMDs x elapsed cycles);
  * FETCH_SIZE (KiB) reports half of a wide coalesced read stream on gfx950 -> bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.
"""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS, XCDS = 1024, 8


def short(name):
    name = name.replace("void (anonymous namespace)::", "")
    return name.split("(")[0]


def ffn_summary(path):
    raw = json.load(open(path))["summary"]
    out = {}
    for kern, c in raw.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        busy, gui = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"], c["GRBM_GUI_ACTIVE"]["mean"]
        row = {"launches_profiled": c["SQ_VALU_MFMA_BUSY_CYCLES"]["n"], "mfma_busy_cycles_per_launch": busy,
               "elapsed_cycles_per_launch": gui / XCDS, "avg_launch_us": c.get("_us_pass0", {}).get("mean"),
               "mfma_utilisation": busy / (SIMDS * gui / XCDS)}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            row["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024
        if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c:
            row["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"]["mean"] / max(c["SQ_LDS_IDX_ACTIVE"]["mean"], 1)
        out[short(kern)] = row
    return out


def gemm_traffic(path, steps):
    raw = json.load(open(path))["summary"]
    total, launches, per_kernel, busy, gui = 0.0, 0, {}, 0.0, 0.0
    for kern, c in raw.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        b = (2 * c["FETCH_SIZE"]["sum"] + c["WRITE_SIZE"]["sum"]) * 1024
        total += b
        launches += c["FETCH_SIZE"]["n"]
        per_kernel[short(kern)] = {"launches_per_step": c["FETCH_SIZE"]["n"] / steps, "hbm_bytes_per_step": b / steps}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            per_kernel[short(kern)]["mfma_utilisation"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / (SIMDS * c["GRBM_GUI_ACTIVE"]["sum"] / XCDS)
            busy += c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"]
            gui += c["GRBM_GUI_ACTIVE"]["sum"]
    return {"kernel_filter": "dense_gemm_kernel_v3 (split-K reduce launches not included)", "steps_profiled": steps,
            "launches_per_step": launches / steps, "hbm_bytes_per_step": total / steps,
            "hbm_bytes_per_launch": total / max(launches, 1),
            "mfma_utilisation_in_step": busy / (SIMDS * gui / XCDS) if gui else None,
            "correction": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE reports half of a wide read stream)",
            "per_kernel": per_kernel}


def head_commit():
    import subprocess
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        return None


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--ffn", default=None)
    ap.add_argument("--gemm", default=None)
    ap.add_argument("--gemm-steps", type=int, default=6, help="train steps inside the profiled bench command (steps + warmup)")
    ap.add_argument("--commit", default=None, help="commit the counters were captured at (default: HEAD)")
    a = ap.parse_args()
    a.ffn = a.ffn or os.path.join(ROOT, "gpurun_out", f"{a.tag}_pmc_ffn_fused.json")
    a.gemm = a.gemm or os.path.join(ROOT, "gpurun_out", f"{a.tag}_pmc_gemm.json")
    commit = a.commit or head_commit()
    ffn = {"command": "scripts/pmc_kernel.sh ... ffn_pair scripts/ffn_bench.py --rows 28800 --iters 5 (stand-alone launches, M=28800 d=256 ffn=2048)",
           "formula": "mfma_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8)", "captured_at_commit": commit,
           "summary": ffn_summary(a.ffn)}
    json.dump(ffn, open(os.path.join(ROOT, "profiles", f"{a.tag}_pmc_ffn_gemm.json"), "w"), indent=1)
    g = gemm_traffic(a.gemm, a.gemm_steps)
    g["command"] = "scripts/pmc_kernel.sh ... dense_gemm_kernel_v3 bench.py --eager --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0"
    g["captured_at_commit"] = commit
    json.dump(g, open(os.path.join(ROOT, "profiles", f"{a.tag}_pmc_gemm.json"), "w"), indent=1)
    print(json.dumps(ffn["summary"], indent=1))
    print(json.dumps({k: v for k, v in g.items() if k != "per_kernel"}, indent=1))
    for k, v in g["per_kernel"].items():
        print(k[:90], v)
