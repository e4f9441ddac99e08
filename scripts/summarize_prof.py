"""Condense rocprofv3 output (rocpd sqlite: kernel-trace --stats + separate --pmc passes) into small
text/JSON summaries.  Run on the box by scripts/history/profile_bench.sh, or locally on the pulled .db files:

    python scripts/summarize_prof.py gpurun_out/prof [profiles/r01]
"""
import glob
import json
import os
import sqlite3
import sys

out = sys.argv[1]
dest = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else os.path.join(out, "summary", "r")
os.makedirs(os.path.dirname(dest) or ".", exist_ok=True)


def dbs(sub):
    return sorted(glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True))


lines, summary = [], {}
for f in dbs("trace"):
    cur = sqlite3.connect(f).cursor()
    lines.append("## rocprofv3 --kernel-trace --stats  (python bench.py --no-cpu --steps 5 --warmup 2)")
    lines.append("| kernel | calls | total ms | avg us | % |")
    lines.append("|---|---|---|---|---|")
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 12"):
        lines.append(f"| `{name[:100]}` | {calls} | {total / 1e3:.3f} | {avg:.1f} | {pct:.2f} |")
    d = [r[0] for r in cur.execute("select (end-start) from kernels where name like '%k_search%' order by start")]
    if d:
        lines.append(f"\nk_search launches: {len(d)}; durations ms: " + ", ".join(f"{x / 1e6:.3f}" for x in d))
        summary["k_search_avg_ms_rocprof"] = sum(d) / len(d) / 1e6
    for r in cur.execute("select vgpr_count, accum_vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x from kernels where name like '%k_search%' limit 1"):
        lines.append(f"k_search resources: vgpr={r[0]} agpr={r[1]} sgpr={r[2]} lds={r[3]} B grid={r[4]} threads, workgroup={r[5]}")
pmc = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_dram"):
    for f in dbs(sub):
        cur = sqlite3.connect(f).cursor()
        for ctr, n, mean in cur.execute("select counter_name,count(*),avg(value) from counters_collection where kernel_name like '%k_search%' group by counter_name"):
            pmc[ctr] = {"launches": n, "mean_per_launch": mean}
summary["pmc_k_search"] = pmc
lines.append("\n## PMC counters (each in its own rocprofv3 --pmc pass), kernel k_search, mean per launch")
for k, v in pmc.items():
    lines.append(f"- {k}: {v['mean_per_launch']:.6g} over {v['launches']} launches")
if "FETCH_SIZE" in pmc:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE is in KiB and on gfx950 reports exactly 1/2 of the bytes of a wide
    # (16 B/lane) coalesced read -> x2.  WRITE_SIZE is uncalibrated there (and small here).
    rd = pmc["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
    wr = pmc.get("WRITE_SIZE", {}).get("mean_per_launch", 0.0) * 1024
    summary.update(hbm_read_bytes_per_launch_corrected=rd, hbm_write_bytes_per_launch_uncalibrated=wr, hbm_bytes_per_launch=rd + wr)
    lines.append(f"- HBM bytes per launch = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction) + WRITE_SIZE KiB x 1024 = {rd + wr:.5g} "
                 f"(read {rd:.5g}, write {wr:.5g})")
if "TCC_EA0_RDREQ_sum" in pmc and "TCC_EA0_RDREQ_DRAM_sum" in pmc and pmc["TCC_EA0_RDREQ_sum"]["mean_per_launch"] > 0:
    # which share of the L2's fabric-side read requests went on to DRAM (the rest was served by the 256 MiB Infinity Cache, or
    # peer / IO): applied to the corrected FETCH_SIZE bytes, so no request size has to be assumed
    share = pmc["TCC_EA0_RDREQ_DRAM_sum"]["mean_per_launch"] / pmc["TCC_EA0_RDREQ_sum"]["mean_per_launch"]
    summary["dram_share_of_fabric_reads"] = share
    lines.append(f"- TCC_EA0_RDREQ_DRAM_sum / TCC_EA0_RDREQ_sum = {share:.4f} of the fabric-side read requests are destined for DRAM")
    if "hbm_read_bytes_per_launch_corrected" in summary:
        summary["dram_read_bytes_per_launch"] = summary["hbm_read_bytes_per_launch_corrected"] * share
        lines.append(f"- DRAM read bytes per launch = corrected FETCH_SIZE bytes x that share = {summary['dram_read_bytes_per_launch']:.5g}")
if "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc:
    h, m = pmc["TCC_HIT_sum"]["mean_per_launch"], pmc["TCC_MISS_sum"]["mean_per_launch"]
    lines.append(f"- L2 hit rate = {h / (h + m):.3f}")
    summary["l2_hit_rate"] = h / (h + m)
for name in ("bench_trace.json",):
    try:
        txt = [l for l in open(os.path.join(out, name)) if l.startswith("{")][-1]
        b = json.loads(txt)
        summary["bench_line_under_trace"] = b
        r = b["roofline"]
        lines.append(f"\n## bench.py line of the traced run\nvalue={b['value']:.0f} {b['unit']}, ms_per_step={b['ms_per_step']:.3f}, "
                     f"roofline achieved={r['achieved']:.0f} GB/s (frac {r['frac']:.3f}), HIP-event avg launch {r['avg_launch_ms']:.3f} ms, "
                     f"algorithmic bytes/launch {r['algorithmic_bytes_per_launch']:.5g}")
    except Exception as e:  # noqa
        lines.append(f"(no bench line: {e})")
open(dest + "_kernel_stats.md", "w").write("\n".join(lines) + "\n")
json.dump(summary, open(dest + "_summary.json", "w"), indent=1)
print("\n".join(lines))
