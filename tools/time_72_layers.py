import sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
ncol, nlay, ngpt = 100000, 72, 224
g = torch.Generator(device="cuda").manual_seed(1)
def R(*sh, lo=0.0, hi=1.0):
    t = xp.empty(sh); t.uniform_(lo, hi, generator=g); return t
tau, ssa, gg = R(ncol, nlay, ngpt, hi=2), R(ncol, nlay, ngpt, hi=0.9), R(ncol, nlay, ngpt, hi=0.8)
lay, lev = R(ncol, nlay, ngpt, lo=1, hi=10), R(ncol, nlay + 1, ngpt, lo=1, hi=10)
emis, sfc, inc = R(ncol, ngpt, lo=0.9), R(ncol, ngpt, hi=10), R(ncol, ngpt)
mu0, alb, idir = R(ncol, nlay, lo=0.1, hi=0.9), R(ncol, ngpt, hi=0.3), R(ncol, ngpt, hi=100)
fu, fd = xp.empty((ncol, nlay + 1, ngpt)), xp.empty((ncol, nlay + 1, ngpt))
def timed(name, f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms")
for gen in (0, 1):
    hiplib.ext_call(lib, "rte_hip_force_generic_lw", ["i"], gen); hiplib.ext_call(lib, "rte_hip_force_generic_sw", ["i"], gen)
    tag = "generic" if gen else "segmented"
    timed(f"72 layers, {tag}: rte_lw_solver_2stream", lambda: lib.rte_lw_solver_2stream(ncol, nlay, ngpt, False, tau, ssa, gg, lay, lev, emis, sfc, inc, fu, fd))
    rb = {}
    timed(f"72 layers, {tag}: rte_sw (2-stream, broadband)", lambda: frontend.rte_sw(lib, xp, ncol, nlay, ngpt, False, tau, ssa, gg, mu0, idir, alb, alb, buffers=rb))
hiplib.ext_call(lib, "rte_hip_force_generic_lw", ["i"], 0); hiplib.ext_call(lib, "rte_hip_force_generic_sw", ["i"], 0)
rb2 = {}
timed("72 layers: rte_lw (no scattering, broadband, 1 angle)", lambda: frontend.rte_lw(lib, xp, ncol, nlay, ngpt, False, tau, lay, lev, emis, sfc, buffers=rb2))
rb3 = {}
timed("72 layers: rte_lw with Tang rescaling (2str clouds, no-scattering solver)", lambda: frontend.rte_lw(lib, xp, ncol, nlay, ngpt, False, tau, lay, lev, emis, sfc, ssa=ssa, g=gg, buffers=rb3))
