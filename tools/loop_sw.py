"""GPU-box helper: run rte_sw_solver_2stream (broadband, 1e5 x 60 x 224) in a loop for N seconds, to sample clocks / power beside it."""
import sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
ncol, nlay, ngpt = 100000, 60, 224
g = torch.Generator(device="cuda").manual_seed(1)
def R(*sh, lo=0.0, hi=1.0):
    t = xp.empty(sh); t.uniform_(lo, hi, generator=g); return t
tau, ssa, gg = R(ncol, nlay, ngpt, hi=2), R(ncol, nlay, ngpt, hi=0.9), R(ncol, nlay, ngpt, hi=0.8)
mu0, alb, idir = R(ncol, nlay, lo=0.1, hi=0.9), R(ncol, ngpt, hi=0.3), R(ncol, ngpt, hi=100)
rb = {}
f = lambda: frontend.rte_sw(lib, xp, ncol, nlay, ngpt, False, tau, ssa, gg, mu0, idir, alb, alb, buffers=rb)
f(); torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(10): f()
    torch.cuda.synchronize(); n += 10
print("calls", n, "ms per call", (time.time() - t0) / n * 1e3)
