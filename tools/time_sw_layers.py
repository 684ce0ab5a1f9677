"""GPU-box helper: rte_sw_solver_2stream (broadband) at 1e5 columns x 112 g-points for several layer counts, the segmented
kernels against the generic kernel (rte_hip_force_generic_sw), random optical properties; ms per call and the relative
difference of the two results.  usage: time_sw_layers.py [nlay,nlay,...]"""
import sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
ncol, ngpt = 100000, 112
lays = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "60,72,80,88,91,96").split(",")]
g = torch.Generator(device="cuda").manual_seed(1)
def R(*sh, lo=0.0, hi=1.0):
    t = xp.empty(sh); t.uniform_(lo, hi, generator=g); return t
for nlay in lays:
    tau, ssa, gg = R(ncol, nlay, ngpt, hi=2), R(ncol, nlay, ngpt, hi=0.9), R(ncol, nlay, ngpt, hi=0.8)
    mu0, alb, idir = R(ncol, nlay, lo=0.1, hi=0.9), R(ncol, ngpt, hi=0.3), R(ncol, ngpt, hi=100)
    res = {}
    for mode in (0, 1):
        hiplib.ext_call(lib, "rte_hip_force_generic_sw", ["i"], mode)
        rb = {}
        f = lambda: frontend.rte_sw(lib, xp, ncol, nlay, ngpt, False, tau, ssa, gg, mu0, idir, alb, alb, buffers=rb)
        f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t0) / 3 * 1e3, rb["flux_up"].clone(), rb["flux_dn"].clone())
    hiplib.ext_call(lib, "rte_hip_force_generic_sw", ["i"], 0)
    d = max(float((res[0][i] - res[1][i]).abs().max() / res[1][i].abs().max()) for i in (1, 2))
    print(f"{nlay} layers: production {res[0][0]:.2f} ms ({res[0][0] / nlay * 1e3:.0f} us/layer), generic {res[1][0]:.2f} ms, rel. difference {d:.1e}", flush=True)
    del tau, ssa, gg, res, rb; torch.cuda.empty_cache()
