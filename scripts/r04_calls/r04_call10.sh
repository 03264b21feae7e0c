#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -k "oracle_fixture" > $O/c10_fixture_tests.log 2>&1
echo "fixture tests rc=$? $(tail -n 1 $O/c10_fixture_tests.log)"; grep -E "^FAILED|^ERROR|^E  " $O/c10_fixture_tests.log | head -20
python - <<'PY'
import json
d = json.load(open("gpurun_out/model_report.json"))
for k, v in sorted(d.items()):
    if "oracle_fixture" in k: print(k, v)
PY
