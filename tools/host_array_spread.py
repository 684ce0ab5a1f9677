"""GPU-box helper: the unchanged Fortran frontend on host arrays (host-mirror mode, OpenMP build, 8 host threads), several program
starts with a few passes each; per start the pass times and what the cgroup's CPU controller says (cpu.max quota, periods throttled
during the start) -- diagnosis of the bimodal pass times of round 4 (0.06 s vs 0.28 s).
usage: host_array_spread.py [starts] [passes] [threads] [KEY=VALUE env for the driver ...]"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import stream_io

starts = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 4
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
extra = dict(kv.split("=", 1) for kv in sys.argv[4:])


def cpu_stat():
    out = {}
    for p in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        if os.path.exists(p):
            for ln in open(p):
                k, v = ln.split()
                out[k] = int(v)
            break
    return out


for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(p):
        print("cgroup", p, open(p).read().strip(), "| usable CPUs (affinity):", len(os.sched_getaffinity(0)))
print("driver env:", extra)
allr = []
for s in range(starts):
    a = cpu_stat()
    r = stream_io.measure_frontend_driver("lw", 98304, 4096, ("mirror",), nrep=nrep, threads=threads, env_extra=extra)["mirror"]
    b = cpu_stat()
    thr = {k: b.get(k, 0) - a.get(k, 0) for k in ("nr_periods", "nr_throttled", "throttled_usec")}
    allr += r["pass_rates"][1:]
    print(f"start {s}: passes " + " | ".join(p.split(",")[0] for p in r["passes"]) + f"   throttled: {thr}", flush=True)
allr.sort()
if allr:
    print(f"steady passes (every pass but each start's first): n {len(allr)}, median {allr[len(allr)//2]:.0f}, min {allr[0]:.0f}, max {allr[-1]:.0f} columns/s, max/min {allr[-1]/allr[0]:.2f}")
