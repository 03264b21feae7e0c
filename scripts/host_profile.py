#!/usr/bin/env python3
"""cProfile of the host side of eager training steps (where do the ~13 ms of Python per step go).
    python scripts/host_profile.py [steps]"""
import cProfile
import io
import os
import pstats
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--eager", "--steps", sys.argv[1] if len(sys.argv) > 1 else "20", "--warmup", "3", "--no-cpu-baseline",
            "--roofline-steps", "0"]
import runpy  # noqa: E402

pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:9000])
