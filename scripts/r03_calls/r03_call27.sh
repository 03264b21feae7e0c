#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
bash scripts/ab_env.sh 2 NST_GRAPH_FORCE_CUTS 0 1 -- --steps 20 --warmup 5
timeout 600 python -m pytest tests/test_gpu_graph.py -x -q 2>&1 | tail -3
