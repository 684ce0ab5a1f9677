"""GPU-box helper: does the time of the store-bound kernels depend on WHERE their output arrays lie?  The LW gas optics at bench
size, the output arrays (tau, lay_src, lev_src) allocated anew several times in ONE process -- earlier ones kept alive, so every
round gets other physical memory -- and then once more after everything was freed.  Prints the kernels' HIP-event times per round
with the arrays' addresses (docs/lab-notebook.md, round 6: Planck is bimodal across processes on one box, 5.05 / 5.8 ms)."""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1); xp = frontend.TorchArrays("cuda:0")
ncol = 100000
kd = synth.make_kdist("lw"); atm = synth.make_atmosphere(ncol, 60, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))


def timed(bufs, n=3):
    go.gas_optics_lw(ncol, 60, play, plev, tlay, tsfc, col_gas, tlev, False, buffers=bufs)
    hiplib.ext_call(lib, "rte_hip_profile_reset", []); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 1)
    for _ in range(n): go.gas_optics_lw(ncol, 60, play, plev, tlay, tsfc, col_gas, tlev, False, buffers=bufs)
    torch.cuda.synchronize(); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 0)
    out = {}
    for i in range(hiplib.ext_call(lib, "rte_hip_profile_count", [])):
        buf = ctypes.create_string_buffer(128); cnt, ms = ctypes.c_longlong(0), ctypes.c_double(0)
        lib.raw("rte_hip_profile_get")(ctypes.c_int(i), buf, ctypes.c_int(128), ctypes.byref(cnt), ctypes.byref(ms))
        out[buf.value.decode()] = round(ms.value / max(1, cnt.value), 3)
    return out


keep = []
shared = {}
for rnd in range(5):
    bufs = dict(shared)  # (interpolation state shared: only the 3-D outputs move)
    o = timed(bufs)
    if not shared:
        shared = {k: v for k, v in bufs.items() if k not in ("tau", "lay_src", "lev_src")}
    print("round %d  tau %.3f  planck %.3f  interp %.3f   lay_src @ %#x  lev_src @ %#x  tau @ %#x  reserved %.1f GB" % (
        rnd, o["tau_absorption_kernel"], o["planck_source_kernel"], o["interpolation_kernel"], bufs["lay_src"].data_ptr(),
        bufs["lev_src"].data_ptr(), bufs["tau"].data_ptr(), torch.cuda.memory_reserved() / 1e9), flush=True)
    keep.append(bufs)
del keep, bufs
torch.cuda.empty_cache()
for rnd in range(2):
    bufs = dict(shared)
    o = timed(bufs)
    print("after free %d  tau %.3f  planck %.3f  interp %.3f   lay_src @ %#x  lev_src @ %#x" % (
        rnd, o["tau_absorption_kernel"], o["planck_source_kernel"], o["interpolation_kernel"], bufs["lay_src"].data_ptr(), bufs["lev_src"].data_ptr()), flush=True)
    del bufs
    torch.cuda.empty_cache()
