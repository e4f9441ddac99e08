#!/usr/bin/env python
"""Per-instantiation register / spill / scratch table of a HIP translation unit, read from the code-object metadata the
compiler writes into the assembly (`hipcc ... -save-temps`): vgpr, sgpr, spilled vgprs / sgprs, scratch bytes per lane, LDS.

    python scripts/kernel_resources.py lantern_amd/csrc/search_kernel.hip [filter-regex]
"""
import os
import re
import subprocess
import sys
import tempfile

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wno-unused-function", "-x", "hip"]


def demangle(names):
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "/opt/rocm/llvm/bin/llvm-cxxfilt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
            return out[:len(names)]
        except Exception:
            continue
    return names


def table(asm_text, pattern=None):
    rows = []
    for b in asm_text.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))  # noqa: E731
        rows.append([name, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"),
                     g("group_segment_fixed_size"), g("max_flat_workgroup_size")])
    for r, d in zip(rows, demangle([r[0] for r in rows])):
        m = re.search(r"(k_\w+<[^>]*>)", d)
        r[0] = m.group(1) if m else d[:80]
    if pattern:
        rows = [r for r in rows if re.search(pattern, r[0])]
    return rows


def main():
    src = sys.argv[1]
    pattern = sys.argv[2] if len(sys.argv) > 2 else None
    if src.endswith(".s"):
        text = open(src).read()
    else:
        with tempfile.TemporaryDirectory() as tmp:
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", os.path.abspath(src), "-o", os.path.join(tmp, "x.o"), "-save-temps=obj"], cwd=tmp)
            asm = [f for f in os.listdir(tmp) if f.endswith(".s") and "amdgcn" in f][0]
            text = open(os.path.join(tmp, asm)).read()
    print("| kernel | vgpr | vgpr spills | sgpr | sgpr spills | scratch B/lane | static LDS | max threads |")
    print("|---|---|---|---|---|---|---|---|")
    for r in table(text, pattern):
        print("| `" + r[0] + "` | " + " | ".join(str(x) for x in r[1:]) + " |")


if __name__ == "__main__":
    main()
