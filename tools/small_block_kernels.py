"""GPU-box helper: per-kernel event times of the LW chain at small column counts (library profile scopes)."""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
for name in ("rte_hip_defer_zero", "rte_hip_share_geometry"):
    hiplib.ext_call(lib, name, ["i"], 1)
NLAY = 60
kd = synth.make_kdist("lw"); go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
for B in [int(x) for x in sys.argv[1:]] or [1024, 4096, 100000]:
    atm = synth.make_atmosphere(B, NLAY, seed=42, kdist=kd)
    play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))
    emis = xp.full((B, kd.ngpt), 0.98); bufs, rb = {}, {}
    def chain():
        go.gas_optics_lw(B, NLAY, play, plev, tlay, tsfc, col_gas, tlev, atm.top_at_1, buffers=bufs)
        frontend.rte_lw(lib, xp, B, NLAY, kd.ngpt, atm.top_at_1, bufs["tau"], bufs["lay_src"], bufs["lev_src"], emis, bufs["sfc_src"], buffers=rb)
    for _ in range(3): chain()
    torch.cuda.synchronize()
    hiplib.ext_call(lib, "rte_hip_profile_reset", []); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 1)
    for _ in range(5): chain()
    torch.cuda.synchronize(); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 0)
    n = hiplib.ext_call(lib, "rte_hip_profile_count", []); out = {}
    for i in range(n):
        buf = ctypes.create_string_buffer(128); cnt, ms = ctypes.c_longlong(0), ctypes.c_double(0)
        lib.raw("rte_hip_profile_get")(ctypes.c_int(i), buf, ctypes.c_int(128), ctypes.byref(cnt), ctypes.byref(ms))
        out[buf.value.decode()] = round(ms.value / 5 * 1e3, 1)
    print("ncol", B, "us per chain:", {k: v for k, v in sorted(out.items(), key=lambda kv: -kv[1]) if v > 0}, "sum", round(sum(out.values()), 1), flush=True)
