#!/bin/bash
# Runs the kernel parity tests group by group (a GPU fault in one group must not hide the others) and
# leaves logs + error reports under gpurun_out/.   usage: scripts/gpu_kernel_suite.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
TAG=${1:-run}
: > gpurun_out/suite_${TAG}.summary
for grp in "mfma or lds_transpose" layernorm gemm colsum "attention" conv1 conv2 "embedding or scale_posenc or ls_xent or adam or errors"; do
  name=$(echo "$grp" | tr ' ' '_')
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "$grp" > gpurun_out/k_${TAG}_${name}.log 2>&1
  echo "[$TAG] $grp -> rc=$? : $(tail -n 1 gpurun_out/k_${TAG}_${name}.log)" | tee -a gpurun_out/suite_${TAG}.summary
done
grep -h -E "^FAILED|^ERROR" gpurun_out/k_${TAG}_*.log | head -60
