#!/usr/bin/env python
"""Turn a rocprofv3 rocpd SQLite database (rocprofv3 --kernel-trace --stats -d DIR -o NAME) into a
small text summary that can be committed under profiles/.
usage: python profiles/summarize.py gpurun_out/prof/NAME_results.db profiles/r01_NAME_kernel_stats.md "<command that was profiled>"
"""
import sqlite3
import sys


def main(db_path, out_path, cmd=""):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_path, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\nsource db: {db_path} (view `top_kernels`; durations in microseconds)\n\n")
        f.write("| kernel | calls | total_us | avg_us | % |\n|---|---:|---:|---:|---:|\n")
        for name, calls, total, avg, pct in rows:
            short = name.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")
            if name.startswith("void (anonymous") or name.startswith("(anonymous"):
                short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            f.write(f"| {short[:70]} | {calls} | {total:.1f} | {avg:.1f} | {pct:.2f} |\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
