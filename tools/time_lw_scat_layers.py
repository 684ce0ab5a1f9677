"""GPU-box helper: the two LW solvers for scattering optical properties -- lw_solver_noscat with Tang rescaling (broadband)
and lw_solver_2stream (spectral output) -- at 1e5 columns x 112 g-points for several layer counts, segmented kernels against
the generic ones (rte_hip_force_generic_lw).  usage: time_lw_scat_layers.py [nlay,nlay,...]"""
import sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
ncol, ngpt = 100000, 112
lays = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "60,72,80,91,96").split(",")]
g = torch.Generator(device="cuda").manual_seed(1)
def R(*sh, lo=0.0, hi=1.0):
    t = xp.empty(sh); t.uniform_(lo, hi, generator=g); return t
for nlay in lays:
    tau, ssa, gg = R(ncol, nlay, ngpt, hi=2), R(ncol, nlay, ngpt, hi=0.9), R(ncol, nlay, ngpt, hi=0.8)
    lay, lev = R(ncol, nlay, ngpt, lo=1, hi=11), R(ncol, nlay + 1, ngpt, lo=1, hi=11)
    emis, sfc = R(ncol, ngpt, lo=0.8, hi=1.0), R(ncol, ngpt, hi=10)
    for what, kw, keys in (("rescaling", {}, ("flux_up", "flux_dn")), ("2-stream", {"use_2stream": True}, ("gpt_flux_up", "gpt_flux_dn"))):
        res = {}
        for mode in (0, 1):
            hiplib.ext_call(lib, "rte_hip_force_generic_lw", ["i"], mode)
            rb = {}
            f = lambda: frontend.rte_lw(lib, xp, ncol, nlay, ngpt, False, tau, lay, lev, emis, sfc, ssa=ssa, g=gg, buffers=rb, **kw)
            f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): f()
            torch.cuda.synchronize()
            res[mode] = ((time.perf_counter() - t0) / 3 * 1e3, [rb[k].clone() for k in keys])
            del rb
        hiplib.ext_call(lib, "rte_hip_force_generic_lw", ["i"], 0)
        d = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(res[0][1], res[1][1]))
        print(f"{nlay} layers {what:9s}: production {res[0][0]:.2f} ms ({res[0][0] / nlay * 1e3:.0f} us/layer), generic {res[1][0]:.2f} ms, rel. difference {d:.1e}", flush=True)
        del res; torch.cuda.empty_cache()
    del tau, ssa, gg, lay, lev; torch.cuda.empty_cache()
