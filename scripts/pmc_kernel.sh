#!/bin/bash
# PMC counters of the kernels whose name contains a substring, for an arbitrary python command.
#   scripts/pmc_kernel.sh <out.json> <kernel-name substring> <python args...>
# Counter groups (one rocprofv3 pass each; 8 SQ slots / 4 TCC slots per pass, FETCH_SIZE and WRITE_SIZE in separate passes --
# MI355X_MICROARCH.md "rocprofv3 PMC slots") come from PMC_GROUPS (";"-separated) or the default below.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
OUT=$1; FILT=$2; shift 2
export PMC_OUT=$OUT PMC_FILT=$FILT TMPDIR=/tmp
GROUPS_=${PMC_GROUPS:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM;FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum"}
export PMC_GROUPS_USED="$GROUPS_"
mkdir -p "$(dirname "$OUT")"
i=0
IFS=';' read -ra GR <<< "$GROUPS_"
for g in "${GR[@]}"; do
  rm -rf /tmp/pmc_pass_$i
  (cd /tmp && timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pmc_pass_$i -- python $(for a in "$@"; do if [ -e "$REPO/$a" ]; then echo "$REPO/$a"; else echo "$a"; fi; done) > /tmp/pmc_pass_$i.log 2>&1) || tail -3 /tmp/pmc_pass_$i.log
  i=$((i+1))
done
export PMC_NPASS=$i
python - <<'PY'
import csv, glob, json, os
FILT = os.environ["PMC_FILT"]
out = {"kernel_filter": FILT, "groups": os.environ["PMC_GROUPS_USED"], "counters": {}, "kernels": {}}
for i in range(int(os.environ["PMC_NPASS"])):
    for f in glob.glob(f"/tmp/pmc_pass_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if FILT in r["Kernel_Name"]:
                k = out["kernels"].setdefault(r["Kernel_Name"][:120], {})
                k.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob(f"/tmp/pmc_pass_{i}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if FILT in r["Kernel_Name"]:
                k = out["kernels"].setdefault(r["Kernel_Name"][:120], {})
                k.setdefault(f"_us_pass{i}", []).append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
summary = {}
for name, k in out["kernels"].items():
    summary[name] = {c: {"n": len(v), "mean": sum(v) / len(v), "sum": sum(v)} for c, v in k.items()}
out["summary"] = summary
out.pop("kernels")
json.dump(out, open(os.environ["PMC_OUT"], "w"), indent=1)
for name, k in summary.items():
    print(name[:100])
    for c, v in sorted(k.items()):
        print("   %-34s n=%-4d mean=%.6g" % (c, v["n"], v["mean"]))
PY
