"""The roofline objects of bench.py's JSON line name what each fraction is a fraction OF (CPU test on made-up figures: the functions
that assemble the line are plain arithmetic), and the counter-pass helper maps kernel names as the build pass relies on."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def test_roofline_names_its_fractions():
    bench = load("bench")
    alg_bytes, launch_s = 54.0e9, 6.6e-3
    model = {"dram_bytes_model": 5.3e9, "frac_dram_model": 5.3e9 / launch_s / 1e9 / 8000.0, "fabric_model_over_counters": 1.0}
    r = bench.roofline(alg_bytes / launch_s / 1e9, 48.9e9, "counters", launch_s, alg_bytes, launch_s, 1, 4, measured_here=True, unique=327000.0,
                       row_bytes=3072, list_bytes=128, expansions_per_launch=8192 * 71.0, model=model)
    # the contract's keys
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # `frac` is the algorithmic figure, says so, and the physical figures sit beside it under their own names
    assert r["frac"] == r["frac_algorithmic"] > 1.0 and "frac_algorithmic" in r["frac_is"] and "NOT a physical" in r["frac_is"]
    assert abs(r["frac_fabric"] - 48.9e9 / launch_s / 1e9 / 8000.0) < 1e-12
    assert r["frac_dram_model"] == model["frac_dram_model"] and r["dram_bytes_model"] == 5.3e9
    assert r["frac_cold_miss_lower_bound"] < r["frac_dram_model"] < r["frac_fabric"] < r["frac"]
    assert r["dram_bytes_per_launch"] is None and r["frac_dram"] is None  # no DRAM-side counter on this part
    assert "exceeds 1" in r["note"]
    # the random-row ceiling is this run's own (roofline.gather) when it was measured, the committed calibration's otherwise -- and says which
    assert r["gather_ceiling"] == bench.GATHER_CEILING_GBS and "another box" in r["gather_ceiling_source"]
    g = {"algorithmic_gbs": 5950.0, "dram_gbs": 5430.0, "measured_in_this_run": True, "infinity_cache": {"algorithmic_gbs": 7100.0}}
    rg = bench.roofline(alg_bytes / launch_s / 1e9, 48.9e9, "counters", launch_s, alg_bytes, launch_s, 1, 4, measured_here=True, model=model, gather=g)
    assert rg["gather_ceiling"] == 5950.0 and rg["gather_dram_ceiling"] == 5430.0 and "THIS run" in rg["gather_ceiling_source"] and rg["gather"] is g
    assert abs(rg["dram_model_over_gather_dram_ceiling"] - 5.3e9 / launch_s / 1e9 / 5430.0) < 1e-12
    # ... and the cache-served counterpart: the fabric rate of the walk against a gather over a table that fits the Infinity Cache
    assert rg["gather_infinity_cache_ceiling"] == 7100.0 and abs(rg["fabric_over_infinity_cache_gather"] - rg["achieved_fabric"] / 7100.0) < 1e-12
    assert r["gather_infinity_cache_ceiling"] is None and r["fabric_over_infinity_cache_gather"] is None
    assert bench.roofline(1.0, None, None, 1.0, 1.0, 1.0, 1, 1, gather={"error": "x"})["gather_ceiling"] == bench.GATHER_CEILING_GBS
    # without counters or a model the physical fields are null, never silently the algorithmic figure
    r2 = bench.roofline(6500.0, None, None, launch_s, 6500.0e9 * launch_s, launch_s, 1, 4)
    assert r2["frac_fabric"] is None and r2["frac_dram_model"] is None and r2["traffic"] is None and r2["frac"] == 6500.0 / 8000.0


def test_build_roofline_takes_counter_bytes_per_kernel():
    bench = load("bench")

    class A:
        dim, quant, M = 768, "f32", 16

    c = {"add_walk_evals": 3630 * 10**6, "add_expansions": 141 * 10**6, "add_reprunes": 7 * 10**6, "add_revlink_evals": 549 * 10**6}
    prof = {"batches": 234, "walk_ms": 1500.0, "connect_ms": 130.0, "group_ms": 18.0, "revlink_ms": 106.0, "exchange_ms": 0.0}
    traffic = {"k_insert": {"launches": 200, "fetch_bytes": 9.0e12}, "k_insert_spec": {"launches": 34, "fetch_bytes": 1.0e9},
               "k_connect": {"launches": 234, "fetch_bytes": 2.0e11}, "_source": "test", "_seconds": 1.0}
    out = bench.build_roofline(A, c, prof, 1.8, 1, traffic)
    w = out["walk"]
    assert w["traffic"] == 9.0e12 + 1.0e9 and 0.5 < w["traffic_over_algorithmic"] < 1.2 and w["frac_fabric"] > 0
    assert set(out["traffic_by_kernel"]) == {"k_insert", "k_insert_spec", "k_connect"}
    assert bench.build_roofline(A, c, prof, 1.8, 1, {"error": "no rocprofv3"})["traffic_error"] == "no rocprofv3"
    assert bench.build_roofline(A, c, prof, 1.8, 1, None)["walk"]["traffic"] is None


def test_short_kernel_names():
    pmc = load("bench_pmc")
    assert pmc.short_kernel_name("void lgpu::k_insert<3, 64, false>(lgpu::InsertArgs)") == "k_insert"
    assert pmc.short_kernel_name("lgpu::k_revlink_pairs(lgpu::RevlinkArgs) [clone .kd]") == "k_revlink_pairs"
    assert pmc.short_kernel_name("void lgpu::k_search<3, 64, false, 2, 1, 0>(lgpu::SearchArgs)") == "k_search"
    assert pmc.short_kernel_name("lgpu::(anonymous namespace)::k_gather_heads(lgpu::LinkReq const*, unsigned int)") == "k_gather_heads"
