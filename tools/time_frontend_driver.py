"""GPU-box helper: the reference's UNCHANGED Fortran frontend (oracle/_ref/bin/ref_frontend_driver = k%load -> k%gas_optics
-> rte_lw / rte_sw per block of columns, host arrays) on the HIP library, host-mirror mode against staged mode, and the same
program on the reference's CPU kernels (one core).
usage: time_frontend_driver.py [lw|sw] [ncol] [block,block,...] [modes] [threads,threads,...]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import stream_io

kind = sys.argv[1] if len(sys.argv) > 1 else "lw"
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 98304
blocks = [int(b) for b in (sys.argv[3] if len(sys.argv) > 3 else "8192,32768").split(",")]
modes = (sys.argv[4] if len(sys.argv) > 4 else "mirror,staged").split(",")
threads = [int(t) for t in (sys.argv[5] if len(sys.argv) > 5 else "1").split(",")]
for bs in blocks:
    if ncol % bs:
        continue
    for nt in threads:
        for mode, r in stream_io.measure_frontend_driver(kind, ncol, bs, modes, threads=nt, env_extra={"REF_DRIVER_TIMING": "1"}).items():
            for rep in r.get("reports", [])[:2]:
                print("   ", rep)
            if nt == 1:
                for ln in r.get("detail", []):
                    print("   ", ln)
            print(f"{kind} {mode:7s} threads {nt:2d} block {bs:6d}: best {r['columns_per_s']:12.0f} columns/s   ({' | '.join(r['passes'])})", flush=True)
