#!/usr/bin/env python3
"""Static resources of every gfx950 kernel of libneurst_hip.so, read from the code-object metadata (no GPU needed):
VGPR (gfx950 metadata: the UNIFIED count, architectural + accumulation registers) / AGPR / SGPR counts, spills, scratch,
static LDS, max workgroup size -> profiles/<tag>_kernel_resources.json and a short table on stdout.  Register-limited
occupancy: waves per SIMD = min(8, floor(512 / vgpr_count)); dynamic LDS (the GEMM / conv rings) is set at launch and is not
in this metadata.

    python scripts/kernel_resources.py [tag] [comma-separated substrings of kernel names to print]
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neurst_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def demangle(names):
    tool = next((t for t in (os.path.join(LLVM, "llvm-cxxfilt"), "/usr/bin/c++filt") if os.path.exists(t)), None)
    if tool is None:
        return list(names)
    out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout
    return out.strip().splitlines()


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    tmp = tempfile.mkdtemp()
    kernels = []
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        co, elf = os.path.join(tmp, src + ".co"), os.path.join(tmp, src + ".elf")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only",
                               "-Wno-unused-result", "-Wno-pass-failed", "-c", os.path.join(CSRC, src), "-o", co],
                              stderr=subprocess.DEVNULL)
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + co,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], capture_output=True, text=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)", line)
            if not m:
                continue
            k, v = m.groups()
            if k == "name" and v.startswith("_Z"):
                cur["name"] = v
            elif k in FIELDS:
                cur[k] = int(v)
            if k == "wavefront_size":      # last field of a kernel record
                if "name" in cur:
                    cur["file"] = src
                    kernels.append(cur)
                cur = {}
    for k, d in zip(kernels, demangle([k["name"] for k in kernels])):
        k["demangled"] = d
        k["waves_per_simd"] = max(1, min(8, 512 // max(k.get("vgpr_count", 0), 1)))
    path = os.path.join(ROOT, "profiles", f"{tag}_kernel_resources.json")
    with open(path, "w") as fp:
        json.dump(kernels, fp, indent=1)
    vspill = [k for k in kernels if k.get("vgpr_spill_count", 0) or k.get("private_segment_fixed_size", 0)]
    sspill = [k for k in kernels if k.get("sgpr_spill_count", 0)]
    print(f"{len(kernels)} kernels -> {path}; VGPR spills / scratch: {len(vspill)}; SGPR spills (to VGPR lanes): {len(sspill)}")
    want = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    rows = [k for k in kernels if any(w in k["demangled"] for w in want)] if want else \
        sorted(kernels, key=lambda k: -k.get("vgpr_count", 0))[:12]
    for k in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", k["demangled"]).split("(")[0][:70]
        print(f"  {short:70s} vgpr {k.get('vgpr_count', 0):3d} agpr {k.get('agpr_count', 0):3d} sgpr {k.get('sgpr_count', 0):3d} "
              f"lds {k.get('group_segment_fixed_size', 0):6d} spill {k.get('vgpr_spill_count', 0)}/{k.get('sgpr_spill_count', 0)} "
              f"scratch {k.get('private_segment_fixed_size', 0)} waves/SIMD {k['waves_per_simd']}")


if __name__ == "__main__":
    main()
