"""GPU-box helper: timings of the remaining solver entry points at benchmark size (1e5 x 60 x 256/224):
lw_solver_2stream (spectral output, as the ABI defines it) and sw_solver_noscat."""
import ctypes, sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
ncol, nlay, ngpt = 100000, 60, 256
g = torch.Generator(device="cuda").manual_seed(1)
def R(*sh):  # Fortran-ordered arrays the way frontend.TorchArrays lays them out
    t = xp.empty(sh)
    t.uniform_(0.0, 1.0, generator=g)
    return t
tau, ssa, gg = R(ncol, nlay, ngpt).mul_(2), R(ncol, nlay, ngpt).mul_(0.9), R(ncol, nlay, ngpt).mul_(0.8)
lay, lev = R(ncol, nlay, ngpt).mul_(10).add_(1), R(ncol, nlay + 1, ngpt).mul_(10).add_(1)
emis, sfc, inc = R(ncol, ngpt).mul_(0.1).add_(0.9), R(ncol, ngpt).mul_(10), R(ncol, ngpt)
mu0 = R(ncol, nlay).mul_(0.8).add_(0.1)
fu, fd = xp.empty((ncol, nlay + 1, ngpt)), xp.empty((ncol, nlay + 1, ngpt))
def timed(name, f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {dt * 1e3:.2f} ms")
timed("rte_lw_solver_2stream (spectral out)", lambda: lib.rte_lw_solver_2stream(ncol, nlay, ngpt, False, tau, ssa, gg, lay, lev, emis, sfc, inc, fu, fd))
timed("rte_sw_solver_noscat", lambda: lib.rte_sw_solver_noscat(ncol, nlay, ngpt, False, tau, mu0, inc, fu))
