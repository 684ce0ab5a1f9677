"""GPU-box helper: the reference's UNCHANGED Fortran frontend (oracle/_ref/bin/ref_frontend_driver = k%load -> k%gas_optics
-> rte_lw / rte_sw per block of columns, host arrays) on the HIP library, staged mode against host-mirror mode, and the same
program on the reference's CPU kernels (one core).  usage: time_frontend_driver.py [lw|sw] [ncol] [block,block,...] [modes]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from rte_rrtmgp_amd import kdist_load, synth
import stream_io

kind = sys.argv[1] if len(sys.argv) > 1 else "lw"
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 98304
blocks = [int(b) for b in (sys.argv[3] if len(sys.argv) > 3 else "8192,32768").split(",")]
modes = (sys.argv[4] if len(sys.argv) > 4 else "mirror,staged").split(",")
nlay = 60
ngpt, nbnd = (256, 16) if kind == "lw" else (224, 14)
gases = list(synth.GAS_NAMES)
raw = kdist_load.synth_raw(kind, ngpt=ngpt, nbnd=nbnd, nminor_lower=4 * nbnd, nminor_upper=2 * nbnd + 3)
kd = kdist_load.init_from_raw(raw, gases); kd.scalars.pop("gas_names")
atm = synth.make_atmosphere(ncol, nlay, seed=42, kdist=kd, ngas=kd.ngas)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
kf = os.path.join(d, "k.bin"); stream_io.write_kdist_stream(kf, raw, kind == "lw")
ref = None
for mode in modes:
    for bs in blocks:
        if ncol % bs:
            continue
        af, of = os.path.join(d, "a.bin"), os.path.join(d, "o.bin")
        if mode == "cpuref":
            n = min(ncol, 4096)  # a bounded sample for the one-core CPU run
            sub = synth.make_atmosphere(n, nlay, seed=42, kdist=kd, ngas=kd.ngas)
            stream_io.write_atmosphere_stream(af, sub, kind == "lw", block=min(bs, 32), checks=False, nrep=1)
            t0 = time.time()
            fl, out = stream_io.run_frontend_driver("ref_frontend_driver_cpuref", kf, af, of, gases, n, nlay, kind == "lw")
        else:
            stream_io.write_atmosphere_stream(af, atm, kind == "lw", block=bs, checks=False, nrep=3)
            env = {"RTE_HIP_HOST_MIRROR": "1" if mode == "mirror" else "0", "RTE_HIP_STAGING_REPORT": "1"}
            t0 = time.time()
            fl, out = stream_io.run_frontend_driver("ref_frontend_driver", kf, af, of, gases, ncol, nlay, kind == "lw", env=env)
            ref = ref or {}
            if bs not in ref:
                ref[bs] = fl
            for k in fl:  # same block size -> same kernels and reduction order: the modes must agree bit for bit
                assert np.array_equal(fl[k], ref[bs][k]), (mode, bs, k, float(np.max(np.abs(fl[k] - ref[bs][k]))))
        best = [ln for ln in out.splitlines() if "best columns/s" in ln][0].split(":")[1].strip()
        passes = " | ".join(ln.split(":")[1].strip() for ln in out.splitlines() if ln.startswith("pass"))
        rep = [ln for ln in getattr(stream_io, "last_stderr", "").splitlines() if "staging report" in ln]
        if rep:
            print("   ", rep[0])
        print(f"{kind} {mode:7s} block {bs:6d}: best {float(best):12.0f} columns/s   ({passes}; wall {time.time()-t0:.1f} s)", flush=True)
