"""GPU-box helper: the LW gas-optics kernels on a g128-shaped table (16 bands of 8 g-points): production kernels
(8-wide stages) vs direct-gather kernels."""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1); xp = frontend.TorchArrays("cuda:0")
ncol = 100000
kd = synth.make_kdist("lw", ngpt=128, nbnd=16); atm = synth.make_atmosphere(ncol, 60, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))
for direct in (0, 1):
    hiplib.ext_call(lib, "rte_hip_force_direct_gather", ["i"], direct)
    bufs = {}
    go.gas_optics_lw(ncol, 60, play, plev, tlay, tsfc, col_gas, tlev, False, buffers=bufs)
    hiplib.ext_call(lib, "rte_hip_profile_reset", []); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 1)
    for _ in range(3): go.gas_optics_lw(ncol, 60, play, plev, tlay, tsfc, col_gas, tlev, False, buffers=bufs)
    torch.cuda.synchronize(); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 0)
    n = hiplib.ext_call(lib, "rte_hip_profile_count", []); out = {}
    for i in range(n):
        buf = ctypes.create_string_buffer(128); cnt, ms = ctypes.c_longlong(0), ctypes.c_double(0)
        lib.raw("rte_hip_profile_get")(ctypes.c_int(i), buf, ctypes.c_int(128), ctypes.byref(cnt), ctypes.byref(ms))
        out[buf.value.decode()] = round(ms.value / max(1, cnt.value), 3)
    print("direct" if direct else "production", out)
hiplib.ext_call(lib, "rte_hip_force_direct_gather", ["i"], 0)
