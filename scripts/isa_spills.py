#!/usr/bin/env python
"""Where a kernel's scratch (spilled-VGPR) traffic sits: every scratch_load / scratch_store of one kernel in a `-save-temps`
assembly file, with the loops (backward branches) that enclose it.

    python scripts/isa_spills.py file.s '<mangled-name-prefix>'
"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ":" in l.split()[0]][0]
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
labs = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labs[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labs and labs[m.group(1)] < i:
        loops.append((labs[m.group(1)], i))
print(len(body), "lines;", len(loops), "loops;", sum("s_barrier" in l for l in body), "barriers at", [i for i, l in enumerate(body) if "s_barrier" in l])
for i, l in enumerate(body):
    if "scratch_" in l:
        inside = sorted((b - a, a, b) for a, b in loops if a <= i <= b)
        print(i, l.strip().split(";")[0], "| innermost loops:", [(a, b) for _, a, b in inside[:3]])
print("largest loops:", sorted(loops, key=lambda x: x[0] - x[1])[:8])
