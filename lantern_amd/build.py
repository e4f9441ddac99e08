"""Compile the HIP library in-tree: lantern_amd/lib/liblantern_gpu.so (gfx950 only).

    python -m lantern_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with
the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
LIB = os.path.join(OUT_DIR, "liblantern_gpu.so")

SOURCES = ["search_kernel.hip", "search_spec_kernel.hip", "search_adc_kernel.hip", "insert_kernel.hip", "insert_spec_kernel.hip", "kernels.hip", "bruteforce.hip", "grouping.hip", "index.cpp", "comm.cpp", "usearch_file.cpp", "scan_shim.cpp", "index_server.cpp", "scan_server.cpp", "mirror_cache.cpp", "node_tape.cpp"]
# every header under csrc/ is a dependency of every object (globbed, so a new header cannot be forgotten)
HEADERS = (sorted(h for h in os.listdir(CSRC) if h.endswith(".hpp")) + sorted("experimental/" + h for h in os.listdir(os.path.join(CSRC, "experimental")) if h.endswith(".hpp"))
           + ["../../include/lantern_gpu.h"])
# -ffp-contract=off: every fma in the kernels is explicit, so the reduction tree is exactly the
# one the oracle models.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-x", "hip"]


# LANTERN_BUILD_EXPERIMENTAL=1: also build the walk variants that lost their A/B (csrc/experimental/: the two-nodes-per-round walk
# and the one-wave walk, LANTERN_GPU_SPEC=3 / 4).  The default library -- what bench.py, the driver's tests and a deployment run --
# does not contain them.  Objects of the two builds live in separate directories; the library on disk is whichever was built last,
# and lantern_gpu_version() says which ("... +experimental").
EXPERIMENTAL = os.environ.get("LANTERN_BUILD_EXPERIMENTAL", "0") not in ("", "0")
EXPERIMENTAL_SOURCES = ["experimental/search_solo_kernel.hip"]
if EXPERIMENTAL:
    SOURCES = SOURCES + EXPERIMENTAL_SOURCES
    FLAGS = FLAGS + ["-DLGPU_EXPERIMENTAL=1"]
    OBJ_DIR = os.path.join(OUT_DIR, "obj_experimental")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path] + hdrs):
            cmd = [hipcc] + FLAGS + ["-I" + CSRC, "-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    marker = os.path.join(OUT_DIR, ".built_experimental")  # which of the two builds the library on disk is
    if force or _stale(LIB, objs) or os.path.exists(marker) != EXPERIMENTAL:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        if EXPERIMENTAL:
            open(marker, "w").close()
        elif os.path.exists(marker):
            os.remove(marker)
    # the cache model of bench.py's roofline.frac_dram_model (tools/cache_model.c: plain C, host only)
    cm_src, cm_lib = os.path.join(HERE, "tools", "cache_model.c"), os.path.join(OUT_DIR, "libcache_model.so")
    if os.path.exists(cm_src) and (force or _stale(cm_lib, [cm_src])):
        cmd = ["gcc", "-O2", "-Wall", "-shared", "-fPIC", "-o", cm_lib, cm_src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # the standalone binaries: `lantern_amd/lib/lantern-index-server --host H --port P --tmp-dir D` (external indexing
    # server) and `lantern_amd/lib/lantern-scan-server --index FILE --metric M --dim D --m M` (scan-side service)
    for src_name, bin_name in (("lantern_index_server.cpp", "lantern-index-server"), ("lantern_scan_server.cpp", "lantern-scan-server"),
                               ("lantern_scan_load.cpp", "lantern-scan-load"), ("lantern_index_load.cpp", "lantern-index-load")):
        tool_src = os.path.join(HERE, "tools", src_name)
        tool = os.path.join(OUT_DIR, bin_name)
        if os.path.exists(tool_src) and (force or _stale(tool, [tool_src, LIB])):
            cmd = [hipcc, "-O2", "-std=c++17", tool_src, "-o", tool, "-L" + OUT_DIR, "-llantern_gpu", "-Wl,-rpath,$ORIGIN", "-lpthread"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
