"""Every `file:line` citation of the reference in the ABI header, the oracle and the design documents must point into a
file that exists in the reference tree and has that many lines.  Runs where /root/reference is mounted (this
container); skipped elsewhere (the GPU box has no reference tree)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
DOCS = ["include/lantern_gpu.h", "oracle/lantern_oracle.h", "oracle/metrics.c", "oracle/hnsw.c", "DESIGN.md", "INTEGRATION.md",
        "lantern_amd/csrc/index.cpp", "lantern_amd/csrc/index_server.cpp", "lantern_amd/csrc/usearch_file.cpp",
        "lantern_amd/csrc/scan_shim.cpp", "lantern_amd/csrc/kernels.hip", "lantern_amd/csrc/walk.hpp", "lantern_amd/csrc/comm.hpp"]
CITE = re.compile(r"\b([A-Za-z_][A-Za-z0-9_/.\-]*\.(?:c|h|cpp|rs|sql|out|txt|toml|py|md))`?:(\d+)(?:-(\d+))?((?:,\d+(?:-\d+)?)*)")


def reference_files():
    by_base = {}
    for dirpath, _, files in os.walk(REF):
        if "/.git" in dirpath:
            continue
        for f in files:
            by_base.setdefault(f, []).append(os.path.join(dirpath, f))
    return by_base


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not mounted here")
def test_reference_citations_resolve():
    by_base = reference_files()
    lengths, bad, checked = {}, [], 0
    own = {os.path.basename(p) for p in DOCS} | {"bench.py", "capi.py", "sharded.py", "host_util.hpp", "device_common.hpp", "comm.cpp"}
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc), errors="ignore").read()
        for m in CITE.finditer(text):
            name = m.group(1)
            base = os.path.basename(name)
            if base in own or base not in by_base:
                continue  # a file of this repository, or not a reference file name
            cands = [p for p in by_base[base] if p.endswith("/" + name)] or by_base[base]
            spans = [(m.group(2), m.group(3))] + [tuple((s.split("-") + [None])[:2]) for s in m.group(4).split(",") if s]
            top = max(int(b or a) for a, b in spans)
            ok = False
            for p in cands:
                if p not in lengths:
                    lengths[p] = sum(1 for _ in open(p, errors="ignore"))
                ok = ok or top <= lengths[p]
            checked += 1
            if not ok:
                bad.append(f"{doc}: {m.group(0)} (longest candidate has {max(lengths[p] for p in cands)} lines)")
    assert checked > 150, checked
    assert not bad, "\n".join(bad[:20])
