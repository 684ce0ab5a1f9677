"""GPU-box helper (one line, for tools/variants.py A/B): rte_sw_solver_2stream, broadband, 1e5 columns x 224 g-points at 60 and
72 layers on random optical properties -- wall-clock per call -- and a checksum of the fluxes."""
import sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
ncol, ngpt = 100000, 224
g = torch.Generator(device="cuda").manual_seed(1)
def R(*sh, lo=0.0, hi=1.0):
    t = xp.empty(sh); t.uniform_(lo, hi, generator=g); return t
out = []
for nlay in (60, 72):
    tau, ssa, gg = R(ncol, nlay, ngpt, hi=2), R(ncol, nlay, ngpt, hi=0.9), R(ncol, nlay, ngpt, hi=0.8)
    mu0, alb, idir = R(ncol, nlay, lo=0.1, hi=0.9), R(ncol, ngpt, hi=0.3), R(ncol, ngpt, hi=100)
    rb = {}
    f = lambda: frontend.rte_sw(lib, xp, ncol, nlay, ngpt, False, tau, ssa, gg, mu0, idir, alb, alb, buffers=rb)
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    out.append("%d layers %.3f ms (sum up %.10e)" % (nlay, (time.perf_counter() - t0) / 5 * 1e3, float(rb["flux_up"].sum())))
    del tau, ssa, gg, rb; torch.cuda.empty_cache()
print("; ".join(out))
