#!/usr/bin/env python3
"""Instruction statistics of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only), per basic block:
how many MFMA / VALU / SALU / LDS / VMEM instructions and SGPR spill moves (v_writelane / v_readlane) each block holds,
and which blocks are loops.  No GPU needed.

    python scripts/isa_stats.py file.s <substring of the mangled kernel name> [--blocks]
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op in ("v_readlane_b32", "v_writelane_b32"):
        return "sgpr_spill"
    if op.startswith("v_readfirstlane"):
        return "readfirstlane"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_load_lds") or op.startswith("buffer_load") and "lds" in op:
        return "lds_dma"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
        return "vmem_store"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_][\w$.]*:", l) and key in l.split(":")[0]:
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found")
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        blocks[cur].append((op, s))
    total = Counter()
    order = list(blocks)
    print(f"kernel {lines[start][:-1][:100]}: {len(blocks)} blocks")
    for name, ins in blocks.items():
        c = Counter(classify(op) for op, _ in ins)
        total.update(c)
        targets = [s.split()[-1] for op, s in ins if op.startswith(("s_cbranch", "s_branch"))]
        back = [t for t in targets if t in order and order.index(t) <= order.index(name)]
        if show_blocks and (len(ins) >= 20 or back):
            print(f"  {name:12s} n={len(ins):5d} {'LOOP->' + ','.join(back) if back else '':18s} " +
                  " ".join(f"{k}={v}" for k, v in sorted(c.items())))
    print("total:", " ".join(f"{k}={v}" for k, v in sorted(total.items())), "sum", sum(total.values()))


if __name__ == "__main__":
    main()
