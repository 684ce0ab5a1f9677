"""GPU-box helper: a map of the device memory by speed.  22 chunks of 12.3 GB are allocated one after the other (plain hipMalloc:
the driver hands out physical memory in its own order); for every neighbouring pair (chunk i = lay_src, chunk i + 1 = lev_src) the
Planck kernel is timed, and a memset of chunk i beside it."""
import ctypes, sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1); xp = frontend.TorchArrays("cuda:0")
ncol, nlay = 100000, 60
kd = synth.make_kdist("lw"); atm = synth.make_atmosphere(ncol, nlay, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))


def timed(bufs, n=2):
    go.gas_optics_lw(ncol, nlay, play, plev, tlay, tsfc, col_gas, tlev, False, buffers=bufs)
    hiplib.ext_call(lib, "rte_hip_profile_reset", []); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 1)
    for _ in range(n): go.gas_optics_lw(ncol, nlay, play, plev, tlay, tsfc, col_gas, tlev, False, buffers=bufs)
    torch.cuda.synchronize(); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 0)
    out = {}
    for i in range(hiplib.ext_call(lib, "rte_hip_profile_count", [])):
        buf = ctypes.create_string_buffer(128); cnt, ms = ctypes.c_longlong(0), ctypes.c_double(0)
        lib.raw("rte_hip_profile_get")(ctypes.c_int(i), buf, ctypes.c_int(128), ctypes.byref(cnt), ctypes.byref(ms))
        out[buf.value.decode()] = round(ms.value / max(1, cnt.value), 3)
    return out


bufs = {}
timed(bufs)  # interpolation state, tau etc. from torch's allocator
shared = {k: v for k, v in bufs.items() if k not in ("lay_src", "lev_src")}
del bufs
torch.cuda.empty_cache()
free, total = torch.cuda.mem_get_info()
nchunk = int((free - (6 << 30)) // (ncol * (nlay + 1) * kd.ngpt * 8))
print("free %.1f GB of %.1f: %d chunks" % (free / 1e9, total / 1e9, nchunk), flush=True)
chunks = [torch.empty((kd.ngpt, nlay + 1, ncol), dtype=torch.float64, device="cuda") for _ in range(nchunk)]
for i in range(nchunk - 1):
    b = dict(shared)
    b["lay_src"] = chunks[i].flatten()[: ncol * nlay * kd.ngpt].view(kd.ngpt, nlay, ncol)
    b["lev_src"] = chunks[i + 1]
    o = timed(b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); chunks[i].zero_(); e1.record(); torch.cuda.synchronize()
    print("chunk %2d @ %#x  planck %.3f  tau %.3f  memset %.3f ms (%.2f TB/s)" % (i, chunks[i].data_ptr(), o["planck_source_kernel"], o["tau_absorption_kernel"],
          e0.elapsed_time(e1), chunks[i].numel() * 8 / e0.elapsed_time(e1) / 1e9), flush=True)
