"""GPU-box helper for an experiment build with -DSW_TIMING (python tools/fastbuild.py swt:solvers.hip=-DSW_TIMING;
RTE_HIP_VARIANT=swt python tools/time_sw_phases.py): where the waves of sw_2stream_seg_kernel spend their time, per segment
number and phase of a g-point, in s_memtime ticks per g-point (the timers themselves cost time: read the shares, not the sum)."""
import ctypes, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
ncol, ngpt, nlay = 100000, 224, int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator(device="cuda").manual_seed(1)
def R(*sh, lo=0.0, hi=1.0):
    t = xp.empty(sh); t.uniform_(lo, hi, generator=g); return t
tau, ssa, gg = R(ncol, nlay, ngpt, hi=2), R(ncol, nlay, ngpt, hi=0.9), R(ncol, nlay, ngpt, hi=0.8)
mu0, alb, idir = R(ncol, nlay, lo=0.1, hi=0.9), R(ncol, ngpt, hi=0.3), R(ncol, ngpt, hi=100)
rb = {}
f = lambda: frontend.rte_sw(lib, xp, ncol, nlay, ngpt, False, tau, ssa, gg, mu0, idir, alb, alb, buffers=rb)
f(); f(); torch.cuda.synchronize()
out = np.zeros((8, 6), dtype=np.uint64); tm = lib.raw("rte_hip_sw_timing")
tm(out.ctypes.data_as(ctypes.c_void_p))
t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
tm(out.ctypes.data_as(ctypes.c_void_p))
tiles = (ncol + 63) // 64
per = out.astype(np.float64) / (tiles * ngpt)
names = ["pass 1 + composite", "wait at barrier 1", "beam + adding chain", "own layers + (A, B)", "wait at barrier 2", "final sweep"]
print("%d layers, one call %.2f ms; ticks per g-point and wave:" % (nlay, ms))
print("%-22s" % "segment" + "".join("%9d" % s for s in range(8)) + "     mean")
for k, n in enumerate(names):
    print("%-22s" % n + "".join("%9.1f" % per[s, k] for s in range(8)) + "%9.1f" % per[:, k].mean())
print("%-22s" % "sum" + "".join("%9.1f" % per[s].sum() for s in range(8)) + "%9.1f" % per.sum(1).mean())
