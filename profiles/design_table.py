#!/usr/bin/env python3
"""The kernel table of DESIGN.md section 4 from a committed profile set: python profiles/design_table.py r06b
(bench line: HIP-event times; <tag>_lw_kernel_stats.md: rocprofv3 averages; pmc_traffic.json: PMC bytes)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1]
r = json.loads([ln for ln in open(os.path.join(HERE, f"{tag}_lw_bench.json")) if ln.startswith("{")][-1])
pmc = json.load(open(os.path.join(HERE, "pmc_traffic.json")))
assert pmc["round"] == tag, (pmc["round"], tag)
pk = r["roofline"]["per_kernel"]
rows = [("interpolation_kernel", "`interpolation_kernel` (interpolation.hip)", "`rrtmgp_interpolation`", 1297),
        ("tau_absorption_kernel", "`tau_slab_kernel` (tau_slab.h) + set-up, worklist", "`rrtmgp_compute_tau_absorption`", 3345),
        ("planck_source_kernel", "`planck_source_v9_kernel` (planck.hip)", "`rrtmgp_compute_Planck_source`", 4883),
        ("lw_noscat_seg_kernel", "`lw_noscat_seg_mixed_kernel<7, 8>` (solvers.hip) + reduction", "`rte_lw_solver_noscat`", 6331)]
print("| kernel (file) | API symbol | B/(col,lay) | alg GB | PMC GB | ms events (rocprofv3) | frac events (rocprofv3) |")
print("|---|---|---:|---:|---:|---:|---:|")
tot_alg = tot_pmc = 0.0
for key, name, sym, b in rows:
    v, p = pk[key], pmc["kernels"][key]
    us = p["rocprof_avg_us"]
    tot_alg += v["alg_GB"]; tot_pmc += p["hbm_GB_per_launch"]
    print(f"| {name} | {sym} | {b} | {v['alg_GB']:.2f} | {p['hbm_GB_per_launch']:.2f} | {v['avg_ms']:.2f} ({us / 1e3:.2f}) | "
          f"{v['frac']:.3f} ({v['alg_GB'] / (us * 1e-6) / 8000:.3f}) |")
ch = r["roofline"]["chain"]
pb = r["roofline"].get("profile_backed") or {}
print(f"| chain | | 15856 | {tot_alg:.2f} | {tot_pmc:.1f} | {r['ms_per_step']:.2f} step, kernels {ch['kernel_ms_per_step']:.2f} ({pb.get('chain_ms', 0):.2f}) | "
      f"{tot_alg / (r['ms_per_step'] * 1e-3) / 8000:.3f} step, {ch['frac']:.3f} ({pb.get('chain_frac', 0):.3f}) |")
c = r["config"]
print(f"\nvalue {r['value'] / 1e6:.3f} M columns/s; plain ABI {c['plain_abi_ms_per_step']} ms; deferred sources {c['deferred_sources_ms_per_step']} ms; "
      f"factored {c['factored_sources']['ms_per_step'] if isinstance(c.get('factored_sources'), dict) else None} ms; "
      f"cpu {r['cpu_baseline']['value']:.0f} ({r['cpu_baseline']['cores']} cores), {r['cpu_baseline']['value_1core']:.0f} (1 core); "
      f"host arrays {c.get('host_array_mode_columns_per_s')}")
for w in ("sw", "allsky"):
    q = json.loads([ln for ln in open(os.path.join(HERE, f"{tag}_{w}_bench.json")) if ln.startswith("{")][-1])
    print(w, f"{q['ms_per_step']:.2f} ms, {q['value'] / 1e6:.3f} M columns/s, chain frac {q['roofline']['chain']['frac']}, plain {q['config']['plain_abi_ms_per_step']}",
          {k: v["avg_ms"] for k, v in q["roofline"]["per_kernel"].items()})
