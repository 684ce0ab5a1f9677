import sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
ncol, nlay = 100000, 60
kd = synth.make_kdist("sw"); atm = synth.make_atmosphere(ncol, nlay, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, plev, tlay, col_gas, col_dry = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "col_gas", "col_dry"))
mu0 = xp.full((ncol, nlay), 0.86); alb = xp.full((ncol, kd.ngpt), 0.06)
bufs, rb = {}, {}
def t(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); th = time.perf_counter() - t0; torch.cuda.synchronize(); return th * 1e3, (time.perf_counter() - t0) * 1e3
go_f = lambda: go.gas_optics_sw(ncol, nlay, play, plev, tlay, col_gas, col_dry, buffers=bufs)
rt_f = lambda: frontend.rte_sw(lib, xp, ncol, nlay, kd.ngpt, False, bufs["tau"], bufs["ssa"], bufs["g"], mu0, bufs["toa_src"], alb, alb, buffers=rb)
for i in range(4):
    print("step", i, "gas_optics_sw (host ms, total ms)", tuple(round(x, 2) for x in t(go_f)), "rte_sw", tuple(round(x, 2) for x in t(rt_f)))
