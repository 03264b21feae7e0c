#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
timeout 600 python scripts/fetch_probe.py --out gpurun_out/r03_fetch_ceiling.json 2>&1 | tail -24
