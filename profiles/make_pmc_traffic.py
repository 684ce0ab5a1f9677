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
    "tau_absorption_kernel": ("tau_absorption_v9_kernel", "tau_absorption_worklist_kernel", "tile_geom2_kernel",
                              "tau_setup_kernel", "tau_absorption_kernel"),
    "planck_source_kernel": ("planck_source_v9_kernel", "planck_source_worklist_kernel", "planck_flags_kernel",
                             "planck_source_kernel", "relayout_gfast_kernel"),
    "lw_noscat_seg_kernel": ("lw_noscat_seg_kernel",),
}


def main(tag):
    rows = list(csv.DictReader(open(os.path.join(HERE, f"{tag}_pmc_summary.csv"))))
    out = {}
    for group, names in GROUPS.items():
        rd = wr = 0.0
        for r in rows:
            if r["kernel"].split("<")[0] in names:
                rd += 2.0 * float(r["FETCH_SIZE"]) * 1024 / 1e9
                wr += float(r["WRITE_SIZE"]) * 1024 / 1e9
        out[group] = {"hbm_GB_per_launch": round(rd + wr, 3), "read_GB": round(rd, 3), "write_GB": round(wr, 3)}
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
