"""GPU-box helper: lay_src and lev_src inside ONE physically contiguous allocation (hipDeviceMallocContiguous), lev_src at a chosen
distance behind lay_src: is Planck's time a function of that distance (then the 5.0 / 5.8 ms modes are channel aliasing between the
two arrays and a padding fixes the mode), the same in every process?"""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1); xp = frontend.TorchArrays("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
ncol, nlay = 100000, 60
kd = synth.make_kdist("lw"); atm = synth.make_atmosphere(ncol, nlay, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))
n_lay, n_lev = ncol * nlay * kd.ngpt * 8, ncol * (nlay + 1) * kd.ngpt * 8
flag = 0x4 if "--plain" not in sys.argv else 0x0
big = ctypes.c_void_p()
assert hip.hipExtMallocWithFlags(ctypes.byref(big), 2 * n_lev + (1 << 31), flag) == 0 and big.value


class View:
    def __init__(self, ptr, shape_f):
        self.__cuda_array_interface__ = {"shape": tuple(reversed(shape_f)), "typestr": "<f8", "data": (ptr, False), "version": 2}


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


shared = {}
PADS = (4096, 4096) if "--quick" in sys.argv else (0, 4096, 65536, 1 << 20, (1 << 21) + 4096, 3 << 20, 1 << 24, (1 << 26) + (1 << 16), 1 << 28, (1 << 30) + 12345 * 512, 0, 1 << 20)
for pad in PADS:
    bufs = dict(shared)
    bufs["lay_src"] = torch.as_tensor(View(big.value, (ncol, nlay, kd.ngpt)), device="cuda")
    bufs["lev_src"] = torch.as_tensor(View(big.value + n_lay + pad, (ncol, nlay + 1, kd.ngpt)), device="cuda")
    o = timed(bufs)
    if not shared:
        shared = {k: v for k, v in bufs.items() if k not in ("lay_src", "lev_src")}
    print("pad %11d  planck %.3f  tau %.3f  (%s, %s)" % (pad, o["planck_source_kernel"], o["tau_absorption_kernel"], "plain" if flag == 0 else "contiguous", __import__("os").environ.get("RTE_HIP_VARIANT", "product")), flush=True)
