"""GPU-box helper: lw_solver_noscat with the surface Jacobian (broadband, 1 angle), 1e5 columns x 128 g-points."""
import sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
ncol, ngpt = 100000, 128
def rnd(*sh, scale=1.0, off=0.0):
    return torch.rand(*reversed(sh), dtype=torch.float64, device="cuda").mul_(scale).add_(off)
for nlay in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "60,72,80").split(",")]:
    tau, lay, lev = rnd(ncol, nlay, ngpt, scale=2.0), rnd(ncol, nlay, ngpt, scale=10, off=1), rnd(ncol, nlay + 1, ngpt, scale=10, off=1)
    emis, sfc, sj = rnd(ncol, ngpt, scale=0.2, off=0.8), rnd(ncol, ngpt, scale=10), rnd(ncol, ngpt)
    rb = {}
    f = lambda: frontend.rte_lw(lib, xp, ncol, nlay, ngpt, False, tau, lay, lev, emis, sfc, sfc_src_jac=sj, do_jacobians=True, buffers=rb)
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    print(f"{nlay} layers with Jacobian: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms (checksum {float(rb['flux_up_jac'].sum()):.10e})", flush=True)
    del tau, lay, lev, rb; torch.cuda.empty_cache()
