"""GPU-box helper: the one-pass SW gas optics (rte_hip_gas_optics_sw_2str) against the two-kernel chain, bit for bit, and timing."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
PREC = "sp" if "sp" in sys.argv[1:] else "dp"  # `python tools/check_fused_sw.py sp`: the single-precision library
lib = hiplib.load(PREC); xp = frontend.TorchArrays("cuda:0", PREC)
kd = synth.make_kdist("sw"); go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
for ncol, nlay in ((70, 24), (1200, 24), (5003, 60), (20000, 60)):
    atm = synth.make_atmosphere(ncol, nlay, seed=7, kdist=kd)
    args = [A(getattr(atm, k)) for k in ("play", "plev", "tlay", "col_gas", "col_dry")]
    g = torch.Generator(device="cuda").manual_seed(1)
    cl = tuple(xp.empty((ncol, nlay, kd.nbnd)).uniform_(0, 1, generator=g) for _ in range(3))
    for clouds in (None, cl):
        a = go.gas_optics_sw(ncol, nlay, *args, fuse_rayleigh=True, clouds_bybnd=clouds)
        ref = {k: a[k].clone() for k in ("tau", "ssa", "g")}
        b = go.gas_optics_sw(ncol, nlay, *args, fuse_rayleigh="all", clouds_bybnd=clouds)
        torch.cuda.synchronize()
        bad = {k: int((b[k] != ref[k]).sum()) for k in ref}
        rel = {k: float(((b[k] - ref[k]).abs() / ref[k].abs().clamp_min(1e-30)).max()) for k in ref}
        wl = hiplib.ext_call(lib, "rte_hip_stat", ["i"], 0)
        print(ncol, nlay, "clouds" if clouds else "clear", "mismatches", bad, "max rel", rel, "worklist", wl)
ncol = 100000; atm = synth.make_atmosphere(ncol, 60, seed=42, kdist=kd)
args = [A(getattr(atm, k)) for k in ("play", "plev", "tlay", "col_gas", "col_dry")]; bufs = {}
hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1)
for mode in (True, "all", True, "all"):
    go.gas_optics_sw(ncol, 60, *args, buffers=bufs, fuse_rayleigh=mode); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): go.gas_optics_sw(ncol, 60, *args, buffers=bufs, fuse_rayleigh=mode)
    torch.cuda.synchronize(); print("mode", mode, "%.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
