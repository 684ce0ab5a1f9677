"""GPU-box helper: lw_solver_noscat (broadband, 1 angle) per layer count -- the one-segment kernels (<= 80 layers), the
two-sub-segment kernel (81 ... 160) and the generic kernel.  usage: time_lw_layers.py [ncol] [ngpt] [nlay,nlay,...]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ngpt = int(sys.argv[2]) if len(sys.argv) > 2 else 128
layers = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "60,80,91,128,137").split(",")]
def rnd(*sh, scale=1.0, off=0.0):
    return torch.rand(*reversed(sh), dtype=torch.float64, device="cuda").mul_(scale).add_(off)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for nlay in layers:
    tau, lay, lev = rnd(ncol, nlay, ngpt, scale=2.0), rnd(ncol, nlay, ngpt, scale=10, off=1), rnd(ncol, nlay + 1, ngpt, scale=10, off=1)
    emis, sfc = rnd(ncol, ngpt, scale=0.2, off=0.8), rnd(ncol, ngpt, scale=10)
    bufs = {}
    fn = lambda: frontend.rte_lw(lib, xp, ncol, nlay, ngpt, False, tau, lay, lev, emis, sfc, buffers=bufs)
    t_seg = timed(fn)
    up = bufs["flux_up"].clone()
    hiplib.ext_call(lib, "rte_hip_force_generic_lw", ["i"], 1)
    t_gen = timed(fn)
    hiplib.ext_call(lib, "rte_hip_force_generic_lw", ["i"], 0)
    err = float(((bufs["flux_up"] - up).abs() / up.abs().clamp_min(1e-300)).max())
    gb = (3 * nlay + 1) * 8 * ncol * ngpt / 1e9
    print(f"lw_solver_noscat {ncol} x {nlay} x {ngpt}: production {t_seg:.2f} ms ({t_seg / nlay * 1e3:.1f} us per layer, {gb / t_seg:.2f} TB/s), "
          f"generic {t_gen:.2f} ms; max rel diff {err:.1e}")
    del tau, lay, lev, bufs; torch.cuda.empty_cache()
