"""The reference's UNCHANGED Fortran frontend built with OpenMP target offload (oracle/build_extern_offload.sh:
flang -fopenmp --offload-arch=gfx950, extern mode) on the HIP library: its own `!$omp target data` regions keep the arrays on
the device and the library resolves the mapped host addresses with omp_get_mapped_ptr.  Compares the fluxes with the same
driver on the reference's CPU kernels and prints the rate.   usage: python tools/check_offload_frontend.py [ncol] [block] [lw|sw]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stream_io
from rte_rrtmgp_amd import kdist_load, synth

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 512
block = int(sys.argv[2]) if len(sys.argv) > 2 else 512
kind = sys.argv[3] if len(sys.argv) > 3 else "lw"
nlay = 60
ngpt, nbnd = (256, 16) if kind == "lw" else (224, 14)
gases = list(synth.GAS_NAMES)
raw = kdist_load.synth_raw(kind, ngpt=ngpt, nbnd=nbnd, nminor_lower=4 * nbnd, nminor_upper=2 * nbnd + 3)
kd = kdist_load.init_from_raw(raw, gases); kd.scalars.pop("gas_names")
d = tempfile.mkdtemp(prefix="rte_off_")
kf, af, of = (os.path.join(d, n) for n in ("k.bin", "a.bin", "o.bin"))
stream_io.write_kdist_stream(kf, raw, kind == "lw")
atm = synth.make_atmosphere(ncol, nlay, seed=42, kdist=kd, ngas=kd.ngas)
res = {}
for name, binary, env, nrep in (("cpuref", "ref_frontend_driver_cpuref", {}, 1),
                                ("offload", "ref_frontend_driver_offload", {"OMP_TARGET_OFFLOAD": "MANDATORY", "RTE_HIP_STAGING_REPORT": "1", "REF_DRIVER_DEEP_SETUP_MB": "4096"}, 3),
                                ("offload, look-up off (staged)", "ref_frontend_driver_offload", {"OMP_TARGET_OFFLOAD": "MANDATORY", "RTE_HIP_OMP_MAPPED": "0", "REF_DRIVER_DEEP_SETUP_MB": "4096"}, 1)):
    n = min(ncol, 2048) if name == "cpuref" else ncol
    a = synth.make_atmosphere(n, nlay, seed=42, kdist=kd, ngas=kd.ngas)
    stream_io.write_atmosphere_stream(af, a, kind == "lw", block=min(block, n) if name != "cpuref" else 32, checks=False, nrep=nrep)
    t0 = time.time()
    try:
        fl, log = stream_io.run_frontend_driver(binary, kf, af, of, gases, n, nlay, kind == "lw", env=env)
    except AssertionError as e:
        print(name, "FAILED:", str(e)[-1500:])
        continue
    res[name] = fl
    best = [ln for ln in log.splitlines() if "best columns/s" in ln]
    print(f"{name}: {time.time() - t0:.1f} s wall; {best[0].strip() if best else ''}")
    for ln in stream_io.last_stderr.splitlines():
        if "staging report" in ln:
            print("   ", ln.strip()[:400])
if "cpuref" in res:
    for name in res:
        if name == "cpuref":
            continue
        n = res["cpuref"]["flux_up"].shape[0]
        for k in res["cpuref"]:
            a, b = res[name][k][:n], res["cpuref"][k]
            print(f"  {name} {k}: worst |diff| / max|ref| = {np.max(np.abs(a - b)) / np.max(np.abs(b)):.2e}")
