import sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
ncol, ngpt = 100000, 128
g = torch.Generator(device="cuda").manual_seed(1)
def R(*sh, lo=0.0, hi=1.0):
    t = xp.empty(sh); t.uniform_(lo, hi, generator=g); return t
for nlay in (60, 72, 80, 96, 100):
    tau, lay, lev = R(ncol, nlay, ngpt, hi=2), R(ncol, nlay, ngpt, lo=1, hi=10), R(ncol, nlay + 1, ngpt, lo=1, hi=10)
    emis, sfc = R(ncol, ngpt, lo=0.9), R(ncol, ngpt, hi=10)
    for gen in (0, 1):
        hiplib.ext_call(lib, "rte_hip_force_generic_lw", ["i"], gen)
        rb = {}
        f = lambda: frontend.rte_lw(lib, xp, ncol, nlay, ngpt, False, tau, lay, lev, emis, sfc, buffers=rb)
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize(); print(f"nlay {nlay} {'generic' if gen else 'segmented'}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
hiplib.ext_call(lib, "rte_hip_force_generic_lw", ["i"], 0)
