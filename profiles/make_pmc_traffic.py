#!/usr/bin/env python3
"""profiles/<tag>_pmc_summary.csv (FETCH_SIZE / WRITE_SIZE in KB per dispatch, from profiles/run_profiles.sh)
-> profiles/pmc_traffic.json, the per-launch HBM traffic bench.py replays as roofline.traffic.
usage: python profiles/make_pmc_traffic.py r02g"""
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# bench.py's kernel groups <- the kernels whose traffic belongs to them (the call's helpers included)
GROUPS = {
    "interpolation_kernel": ("interpolation_kernel",),
    "tau_absorption_kernel": ("tau_slab_kernel", "tau_absorption_worklist_kernel", "tile_geom2_kernel",
                              "tau_setup_kernel", "tau_absorption_kernel"),
    "planck_source_kernel": ("planck_source_v9_kernel", "planck_source_worklist_kernel", "planck_flags_kernel",
                             "planck_source_kernel", "relayout_gfast_kernel"),
    "lw_noscat_seg_kernel": ("lw_noscat_seg_kernel", "lw_noscat_seg_mixed_kernel", "reduce_parts_kernel"),
}


MAIN = {"interpolation_kernel": "interpolation_kernel", "tau_absorption_kernel": "tau_slab_kernel",
        "planck_source_kernel": "planck_source_v9_kernel", "lw_noscat_seg_kernel": "lw_noscat_seg_kernel"}


def rocprof_averages(tag):
    """avg_us per launch of every kernel in profiles/<tag>_lw_kernel_stats.md (rocprofv3 --kernel-trace --stats of the bench command)."""
    avg = {}
    path = os.path.join(HERE, f"{tag}_lw_kernel_stats.md")
    if not os.path.exists(path):
        return avg
    for line in open(path):
        c = [x.strip() for x in line.split("|")]
        if len(c) >= 6 and c[2].isdigit():
            name = c[1].split("<")[0]
            calls, total = avg.get(name, (0, 0.0))
            avg[name] = (calls + int(c[2]), total + float(c[3]))
    return avg


def main(tag):
    rows = list(csv.DictReader(open(os.path.join(HERE, f"{tag}_pmc_summary.csv"))))
    avg = rocprof_averages(tag)
    out = {}
    for group, names in GROUPS.items():
        rd = wr = 0.0
        for r in rows:
            if r["kernel"].split("<")[0] in names:
                rd += 2.0 * float(r["FETCH_SIZE"]) * 1024 / 1e9
                wr += float(r["WRITE_SIZE"]) * 1024 / 1e9
        out[group] = {"hbm_GB_per_launch": round(rd + wr, 3), "read_GB": round(rd, 3), "write_GB": round(wr, 3)}
        main_k = MAIN[group]
        if group == "lw_noscat_seg_kernel" and "lw_noscat_seg_mixed_kernel" in avg:  # (60 layers: segments of 7 and 8 layers, round 6)
            main_k = "lw_noscat_seg_mixed_kernel"
        if main_k in avg:  # the call's main kernel, and its helper kernels beside it (rocprofv3 averages per launch)
            calls, total = avg[main_k]
            out[group]["rocprof_avg_us"] = round(total / calls, 1)
            out[group]["rocprof_helpers_us"] = round(sum(avg[n][1] for n in names if n in avg and n != main_k) / calls, 1)
    doc = {
        "ncol": 100000, "round": tag,
        "source": f"profiles/{tag}_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of "
                  "`python bench.py --steps 1 --warmup 1 --no-cpu-baseline`, KB per dispatch)",
        "correction": "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts 128-byte read requests at 64 B, "
                      "MI355X_MICROARCH.md HBM section); Infinity-Cache hits are included in FETCH_SIZE",
        "kernels": out,
    }
    json.dump(doc, open(os.path.join(HERE, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
