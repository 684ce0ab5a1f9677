"""GPU-box helper: the same question as planck_placement.py with the 3-D output arrays in memory from
hipExtMallocWithFlags(hipDeviceMallocContiguous) (physically contiguous: large page-table fragments) against torch's own
allocations, alternating in one process."""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1); xp = frontend.TorchArrays("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipFree.argtypes = [ctypes.c_void_p]
ncol, nlay = 100000, 60
kd = synth.make_kdist("lw"); atm = synth.make_atmosphere(ncol, nlay, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))


class Ext:
    def __init__(self, shape_f, flag):
        n = 1
        for s in shape_f: n *= s
        p = ctypes.c_void_p()
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p), n * 8, flag)
        assert rc == 0 and p.value, rc
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": tuple(reversed(shape_f)), "typestr": "<f8", "data": (p.value, False), "version": 2}


def timed(bufs, n=3):
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


shared, keep = {}, []
for rnd in range(6):
    bufs = dict(shared)
    kind = ("torch", "contiguous", "plain hipMalloc")[rnd % 3]
    if kind != "torch":
        exts = {k: Ext(sh, 0x4 if kind == "contiguous" else 0x0) for k, sh in (("tau", (ncol, nlay, kd.ngpt)), ("lay_src", (ncol, nlay, kd.ngpt)), ("lev_src", (ncol, nlay + 1, kd.ngpt)))}
        for k, e in exts.items():
            bufs[k] = torch.as_tensor(e, device="cuda")
        keep.append(exts)
    o = timed(bufs)
    if not shared:
        shared = {k: v for k, v in bufs.items() if k not in ("tau", "lay_src", "lev_src")}
    print("round %d %-16s tau %.3f  planck %.3f  interp %.3f   lay_src @ %#x" % (rnd, kind, o["tau_absorption_kernel"], o["planck_source_kernel"],
          o["interpolation_kernel"], bufs["lay_src"].data_ptr()), flush=True)
    keep.append(bufs)
