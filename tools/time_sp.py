"""GPU-box helper: the LW chain of bench.py in the single-precision build (-DRTE_USE_SP), per-kernel HIP-event times."""
import ctypes, sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load("sp"); hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1); xp = frontend.TorchArrays("cuda:0", "sp")
ncol = 100000
kd = synth.make_kdist("lw"); atm = synth.make_atmosphere(ncol, 60, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))
emis = xp.full((ncol, kd.ngpt), 0.98); bufs, rb = {}, {}
def step():
    go.gas_optics_lw(ncol, 60, play, plev, tlay, tsfc, col_gas, tlev, atm.top_at_1, buffers=bufs)
    frontend.rte_lw(lib, xp, ncol, 60, kd.ngpt, atm.top_at_1, bufs["tau"], bufs["lay_src"], bufs["lev_src"], emis, bufs["sfc_src"], buffers=rb)
for _ in range(3): step()
torch.cuda.synchronize()
hiplib.ext_call(lib, "rte_hip_profile_reset", []); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 1)
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 0)
n = hiplib.ext_call(lib, "rte_hip_profile_count", []); out = {}
for i in range(n):
    buf = ctypes.create_string_buffer(128); cnt, ms = ctypes.c_longlong(0), ctypes.c_double(0)
    lib.raw("rte_hip_profile_get")(ctypes.c_int(i), buf, ctypes.c_int(128), ctypes.byref(cnt), ctypes.byref(ms))
    out[buf.value.decode()] = round(ms.value / max(1, cnt.value), 3)
print("single precision LW step %.2f ms -> %.3g columns/s" % (dt * 1e3, ncol / dt)); print(out)
