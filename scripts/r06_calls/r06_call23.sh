#!/bin/bash
# RECORD of a round-6 experiment: the NST_* switch(es) this script sets existed only in the working tree of that experiment
# (removed with it; the library now warns about them).  Kept for the log under profiles/r06_history/; it does not re-run.
echo "$0: record of a removed experiment (see the header); not runnable against this tree" >&2; exit 1
# call 23: decoder weight-gradient group on the weight-gradient stream behind the decoder: text models, speech_transformer_m, ragged
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
for i in 1 2; do
  echo "base  end $(python scripts/bench_text.py --model transformer_base --batch 256 2>/dev/null | ms)  dec-side $(NST_WGRAD_DEC_SIDE=1 python scripts/bench_text.py --model transformer_base --batch 256 2>/dev/null | ms)"
  echo "big   end $(python scripts/bench_text.py --model transformer_big --batch 256 2>/dev/null | ms)  dec-side $(NST_WGRAD_DEC_SIDE=1 python scripts/bench_text.py --model transformer_big --batch 256 2>/dev/null | ms)"
  echo "st_m  end $(python bench.py --model speech_transformer_m --no-cpu-baseline --roofline-steps 0 2>/dev/null | ms)  dec-side $(NST_WGRAD_DEC_SIDE=1 python bench.py --model speech_transformer_m --no-cpu-baseline --roofline-steps 0 2>/dev/null | ms)"
done | tee gpurun_out/r06/c23_group_dec_other_models.log
