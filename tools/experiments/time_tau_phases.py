"""GPU-box helper for an experiment build with -DTAU_TIMING (python tools/fastbuild.py taut:tau_absorption.hip=-DTAU_TIMING;
RTE_HIP_VARIANT=taut python tools/time_tau_phases.py): s_memtime ticks of the compute and loader waves of tau_absorption_v9_kernel per phase of
a stage = (512-column tile, layer, 16 g-points), 1e5 columns x 60 layers x 256 g-points (the timers cost time themselves: read the shares)."""
import ctypes, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1); hiplib.ext_call(lib, "rte_hip_share_geometry", ["i"], 1)
ncol, nlay = 100000, 60
kd = synth.make_kdist("lw"); atm = synth.make_atmosphere(ncol, nlay, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, tlay, col_gas = (A(getattr(atm, k)) for k in ("play", "tlay", "col_gas"))
st = go.interpolation(ncol, nlay, play, tlay, col_gas)
tau = xp.empty((ncol, nlay, kd.ngpt))
def run():
    lib.zero_array_3D(ncol, nlay, kd.ngpt, tau)
    go.compute_tau_absorption(ncol, nlay, st, play, tlay, col_gas, tau)
run(); run(); torch.cuda.synchronize()
out = np.zeros(16, dtype=np.uint64); tm = lib.raw("rte_hip_tau_timing")
tm(out.ctypes.data_as(ctypes.c_void_p))
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
tm(out.ctypes.data_as(ctypes.c_void_p))
nstage = -(-ncol // 512) * nlay * (kd.ngpt // 16)
o = out.astype(np.float64)
print("one call %.2f ms, %d stages; ticks per stage and wave:" % (ms, nstage))
if o[8:].sum() > 0:  # the DMA form (no loader waves): waves 0-3 store at the end of the stage, waves 4-7 after the next barrier
    names = ["waiting at the barrier", "previous stage's stores (rotated waves) + set-up", "major gather + FMAs", "minor species",
             "waiting for its own DMA pieces / weights", "DMA issue + row plan", "requests for the next stage + stores"]
    for half, off in (("waves 0-3", 0), ("waves 4-7 (rotated)", 8)):
        print(" ", half)
        for k, n in enumerate(names):
            print("    %-48s %9.1f" % (n, o[off + k] / (nstage * 4)))
        print("    sum %.1f" % (o[off:off + 7].sum() / (nstage * 4)))
else:
    names = ["compute: waiting at the barrier", "compute: previous stage's stores + set-up", "compute: major gather + FMAs",
             "compute: minor species, rest", "loader: requesting + waiting for table pieces", "loader: LDS writes", "loader: waiting at the barrier"]
    for k, n in enumerate(names):
        print("  %-48s %9.1f" % (n, o[k] / (nstage * (8 if k < 4 else 2))))
    print("  compute sum %.1f, loader sum %.1f" % (o[:4].sum() / (nstage * 8), o[4:7].sum() / (nstage * 2)))
