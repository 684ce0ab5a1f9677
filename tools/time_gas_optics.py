"""GPU-box helper: per-kernel HIP-event timings of the LW gas-optics chain at bench size."""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1); xp = frontend.TorchArrays("cuda:0")
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
kd = synth.make_kdist("lw"); atm = synth.make_atmosphere(ncol, 60, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))
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
print(out)
