"""GPU-box helper: A/B the kernel variants selectable at run time (tau 7/9, Planck 7/9) in ONE process,
interleaved, so that box-to-box and run-to-run noise cancels."""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1); xp = frontend.TorchArrays("cuda:0")
ncol = 100000
kd = synth.make_kdist("lw"); atm = synth.make_atmosphere(ncol, 60, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))
bufs = {}
def run(tv, pv, reps=3):
    hiplib.ext_call(lib, "rte_hip_planck_variant", ["i"], pv)
    go.gas_optics_lw(ncol, 60, play, plev, tlay, tsfc, col_gas, tlev, False, buffers=bufs)
    hiplib.ext_call(lib, "rte_hip_profile_reset", []); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 1)
    for _ in range(reps): go.gas_optics_lw(ncol, 60, play, plev, tlay, tsfc, col_gas, tlev, False, buffers=bufs)
    torch.cuda.synchronize(); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 0)
    n = hiplib.ext_call(lib, "rte_hip_profile_count", []); out = {}
    for i in range(n):
        buf = ctypes.create_string_buffer(128); cnt, ms = ctypes.c_longlong(0), ctypes.c_double(0)
        lib.raw("rte_hip_profile_get")(ctypes.c_int(i), buf, ctypes.c_int(128), ctypes.byref(cnt), ctypes.byref(ms))
        out[buf.value.decode()] = round(ms.value / max(1, cnt.value), 3)
    tau = sum(v for k, v in out.items() if k.startswith("tau_")) + out.get("relayout_gfast_kernel", 0)
    pl = sum(v for k, v in out.items() if k.startswith("planck_"))
    return round(tau, 3), round(pl, 3)
for rnd in range(3):
    for tv, pv in ((7, 7), (9, 9)):
        print("round", rnd, "variants tau/planck", tv, pv, "-> tau total ms, planck total ms:", run(tv, pv))
