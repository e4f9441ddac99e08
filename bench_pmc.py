"""bench_pmc.py -- one rocprofv3 counter pass over a re-executed leg of bench.py, in-run.

Counters are collected in passes of their own (`--pmc` only: gpurun refuses counter passes combined with trace domains), restricted
to one kernel by `--kernel-include-regex`; a pass holds what fits the block's slots (MI355X_MICROARCH.md "rocprofv3 PMC slots":
FETCH_SIZE takes three of the four TCC slots, WRITE_SIZE two -> separate passes; SQ has eight, GRBM two).  The child is this repo's
own bench.py in one of its `--*-child` modes: it rebuilds the same data from the same seeds, runs the launches and prints one JSON
line that holds `marker`.  A measurement harness: nothing of the search path imports it.
"""
from __future__ import annotations

import glob
import json
import os
import shutil
import sqlite3
import subprocess
import tempfile
import time


def rocprof():
    return shutil.which("rocprofv3")


def short_kernel_name(name: str) -> str:
    """`void lgpu::k_insert<3, 64, ...>(lgpu::InsertArgs)` -> `k_insert`"""
    head = name.replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
    return head.split("::")[-1].split(" ")[-1]


def run_pass(child_argv, kernel_like, kernel_regex, counters, marker, timeout=300, by_kernel=False):
    """-> {"child": the child's JSON line, "values": {counter: [per-dispatch values, dispatch order]}, "seconds", "command"} or {"error": ...}.
    by_kernel: "values" becomes {short kernel name: {counter: [values]}} over every kernel the regex admits (kernel_like is ignored)."""
    exe = rocprof()
    if not exe:
        return {"error": "rocprofv3 is not on PATH"}
    tmp = tempfile.mkdtemp(prefix="lantern_pmc_", dir="/tmp")
    cmd = [exe, "--kernel-include-regex", kernel_regex, "--pmc", *counters, "-d", tmp, "-o", "pmc", "--"] + list(child_argv)
    t0 = time.time()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        line = next((json.loads(l) for l in p.stdout.splitlines() if l.startswith("{") and marker in l), None)
        log_dir = os.environ.get("LANTERN_BENCH_PMC_LOG")
        if log_dir and (p.returncode != 0 or not line):  # debugging: the failed pass's whole output
            with open(os.path.join(log_dir, f"pmc_{'_'.join(counters)}.log"), "w") as f:
                f.write(p.stdout + "\n==== stderr\n" + p.stderr)
        if p.returncode != 0 or not line:
            return {"error": f"rc {p.returncode}: {(p.stderr or p.stdout)[-300:]}"}
        values = {} if by_kernel else {c: [] for c in counters}
        for db in glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True):
            cur = sqlite3.connect(db).cursor()
            cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            order = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else None)
            q = "select kernel_name, counter_name, value from counters_collection where kernel_name like ?" + (f" order by {order}" if order else "")
            for kname, name, v in cur.execute(q, ("%" if by_kernel else f"%{kernel_like}%",)):
                if name not in counters:
                    continue
                if by_kernel:
                    values.setdefault(short_kernel_name(kname), {c: [] for c in counters})[name].append(float(v))
                else:
                    values[name].append(float(v))
        if not values or not any(values.values()):
            return {"error": f"no counter rows for {kernel_regex} in the rocprofv3 output"}
        return {"child": line, "values": values, "seconds": time.time() - t0,
                "command": f"rocprofv3 --kernel-include-regex {kernel_regex} --pmc {' '.join(counters)} -- python bench.py {child_argv[2] if len(child_argv) > 2 else ''} ..."}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
