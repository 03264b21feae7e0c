#!/bin/bash
# HBM traffic + MFMA busy of one hot kernel (default: the conv2 forward kernel, bench.py's roofline kernel), one counter
# per rocprofv3 pass as MI355X_MICROARCH.md prescribes -> gpurun_out/pmc_<target>_raw.json
# usage: scripts/pmc_conv2.sh [pmc_target.py target] [kernel-name substring]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
TARGET=${1:-conv2_fwd}
FILT=${2:-conv2_fwd_patch_kernel}
export PMC_TARGET=$TARGET PMC_FILT=$FILT
export TMPDIR=/tmp
mkdir -p gpurun_out
COUNTERS=${PMC_COUNTERS:-"FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"}
export COUNTERS
for c in $COUNTERS; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python $REPO/scripts/pmc_target.py $TARGET > /tmp/pmc_$c.log 2>&1)
done
python - <<'PY'
import csv, glob, json, os
FILT = os.environ['PMC_FILT']; TARGET = os.environ['PMC_TARGET']
out = {}
for c in os.environ["COUNTERS"].split():
    vals, durs = [], {}
    for f in glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if FILT in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"]))
    for f in glob.glob(f"/tmp/pmc_{c}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if FILT in r["Kernel_Name"]:
                durs.setdefault("us", []).append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    out[c] = {"values": vals, "mean": sum(vals) / max(len(vals), 1), "kernel_us": durs.get("us", [])}
json.dump(out, open(f"gpurun_out/pmc_{TARGET}_raw.json", "w"), indent=1)
for k, v in out.items():
    print(k, "n=%d mean=%.6g" % (len(v["values"]), v["mean"]), "kernel_us(mean)=%.1f" % (sum(v["kernel_us"]) / max(len(v["kernel_us"]), 1)))
PY
