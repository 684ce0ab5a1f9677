"""GPU-box helper for an experiment build with -DLW_TIMING (python tools/fastbuild.py lwt:solvers.hip=-DLW_TIMING;
RTE_HIP_VARIANT=lwt python tools/time_lw_phases.py [factored]): where the waves of lw_noscat_seg_kernel spend their time, per
segment number and phase of a g-point, in s_memtime ticks per g-point (the timers cost time themselves: read the shares)."""
import ctypes, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1); hiplib.ext_call(lib, "rte_hip_share_geometry", ["i"], 1)
fact = len(sys.argv) > 1 and sys.argv[1] == "factored"
ncol, nlay = 100000, 60
kd = synth.make_kdist("lw"); atm = synth.make_atmosphere(ncol, nlay, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))
emis = xp.full((ncol, kd.ngpt), 0.98)
b, rb = {}, {}
go.gas_optics_lw(ncol, nlay, play, plev, tlay, tsfc, col_gas, tlev, atm.top_at_1, buffers=b, factored_sources=fact)
def solve():
    if fact:
        frontend.rte_lw_factored(lib, xp, ncol, nlay, kd.ngpt, kd.nbnd, go.t["band_lims_gpt"], atm.top_at_1, b["tau"], b["pfrac"],
                                 b["planck_lay"], b["planck_lev"], emis, b["sfc_src"], buffers=rb)
    else:
        frontend.rte_lw(lib, xp, ncol, nlay, kd.ngpt, atm.top_at_1, b["tau"], b["lay_src"], b["lev_src"], emis, b["sfc_src"], buffers=rb)
solve(); solve(); torch.cuda.synchronize()
out = np.zeros((8, 5), dtype=np.uint64); tm = lib.raw("rte_hip_lw_timing")
tm(out.ctypes.data_as(ctypes.c_void_p))
t0 = time.perf_counter(); solve(); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
tm(out.ctypes.data_as(ctypes.c_void_p))
per = out.astype(np.float64) / (-(-ncol // 64) * kd.ngpt)
names = ["issue next loads", "pass 1 (+ input wait)", "wait at the barrier", "chains", "pass 2"]
print("%s sources, one call %.2f ms; ticks per g-point and wave:" % ("factored" if fact else "ABI", ms))
print("%-24s" % "segment" + "".join("%9d" % s for s in range(8)) + "     mean")
for k, n in enumerate(names):
    print("%-24s" % n + "".join("%9.1f" % per[s, k] for s in range(8)) + "%9.1f" % per[:, k].mean())
print("%-24s" % "sum" + "".join("%9.1f" % per[s].sum() for s in range(8)) + "%9.1f" % per.sum(1).mean())
