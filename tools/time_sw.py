"""GPU-box helper: per-kernel HIP-event timings of the SW chain (gas optics + sw_solver_2stream), config 3."""
import ctypes, sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
import os
lib = hiplib.load(); hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1)
if os.environ.get("RTE_SW_MIXED"): hiplib.ext_call(lib, "rte_hip_sw_mixed_segments", ["i"], int(os.environ["RTE_SW_MIXED"]))  # A/B: 0 = segments of 8 + 8 layers
xp = frontend.TorchArrays("cuda:0")
ncol = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100000
nlay = 60
kd = synth.make_kdist("sw"); atm = synth.make_atmosphere(ncol, nlay, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, plev, tlay, col_gas, col_dry = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "col_gas", "col_dry"))
mu0 = xp.full((ncol, nlay), 0.86); alb = xp.full((ncol, kd.ngpt), 0.06)
bufs, rb = {}, {}
def step():
    go.gas_optics_sw(ncol, nlay, play, plev, tlay, col_gas, col_dry, buffers=bufs, fuse_rayleigh=(True if "--chain" in sys.argv else "all"))
    frontend.rte_sw(lib, xp, ncol, nlay, kd.ngpt, False, bufs["tau"], bufs["ssa"], bufs["g"], mu0, bufs["toa_src"], alb, alb, buffers=rb)
step(); step(); step(); torch.cuda.synchronize()
hiplib.ext_call(lib, "rte_hip_profile_reset", []); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 1)
t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 0)
n = hiplib.ext_call(lib, "rte_hip_profile_count", []); out = {}
for i in range(n):
    buf = ctypes.create_string_buffer(128); cnt, ms = ctypes.c_longlong(0), ctypes.c_double(0)
    lib.raw("rte_hip_profile_get")(ctypes.c_int(i), buf, ctypes.c_int(128), ctypes.byref(cnt), ctypes.byref(ms))
    out[buf.value.decode()] = (int(cnt.value) // 3, round(ms.value / 3, 3))
print("SW step %.2f ms -> %.3g columns/s" % (dt * 1e3, ncol / dt)); print(out)
print("flux_dn sfc mean", float(rb["flux_dn"][0].mean()), "flux_up toa mean", float(rb["flux_up"][-1].mean()))
