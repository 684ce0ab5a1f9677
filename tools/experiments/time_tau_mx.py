"""compute_tau_absorption at benchmark size (1e5 x 60 x 256, deferred zero fill), variant 9 or 10 (argv[1]), climate argv[2]:
per-kernel HIP-event times (rte_hip_profile_*) and, in an -DMX_TIMING build, the clocks the matrix-core kernel's column and
matrix waves spend working / waiting at the stage barriers."""
import ctypes, os, sys
sys.path.insert(0, ".")
import torch
import rte_rrtmgp_amd  # noqa
from rte_rrtmgp_amd import frontend, hiplib, synth
v = int(sys.argv[1]) if len(sys.argv) > 1 else 10
climate = sys.argv[2] if len(sys.argv) > 2 else "rce"
lib = hiplib.load(); hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1); xp = frontend.TorchArrays("cuda:0")
hiplib.ext_call(lib, "rte_hip_tau_variant", ["i"], v)
ncol = 100000
kd = synth.make_kdist("lw"); atm = synth.make_atmosphere(ncol, 60, seed=42, kdist=kd, climate=climate)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, tlay, col_gas = A(atm.play), A(atm.tlay), A(atm.col_gas)
st = go.interpolation(ncol, 60, play, tlay, col_gas)
tau = xp.zeros((ncol, 60, kd.ngpt))
def step():
    lib.zero_array_3D(ncol, 60, kd.ngpt, tau)
    go.compute_tau_absorption(ncol, 60, st, play, tlay, col_gas, tau)
for _ in range(3): step()
torch.cuda.synchronize()
try:
    tm = lib.raw("rte_hip_mx_timing")
except Exception:
    tm = None
buf4 = (ctypes.c_ulonglong * 4)()
if tm: tm(buf4)
hiplib.ext_call(lib, "rte_hip_profile_reset", []); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 1)
N = 5
for _ in range(N): step()
torch.cuda.synchronize(); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 0)
n = hiplib.ext_call(lib, "rte_hip_profile_count", []); out = {}
for i in range(n):
    b = ctypes.create_string_buffer(128); cnt, ms = ctypes.c_longlong(0), ctypes.c_double(0)
    lib.raw("rte_hip_profile_get")(ctypes.c_int(i), b, ctypes.c_int(128), ctypes.byref(cnt), ctypes.byref(ms))
    out[b.value.decode()] = round(ms.value / max(1, cnt.value), 3)
extra = ""
if tm:
    tm(buf4)
    waves = 8 * ((ncol + 511) // 512) * 60 * N
    extra = " | clocks per wave and stage: column busy %.0f wait %.0f, matrix busy %.0f wait %.0f" % tuple(x / waves / 16 for x in buf4)
print(f"variant {v} {climate}: {out}{extra} checksum {float(tau.sum()):.12e}")
