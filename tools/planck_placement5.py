"""GPU-box helper: the LW chain with its three 3-D arrays (tau, lay_src, lev_src) in memory of different physical layouts, in one
process: torch's allocator, hipDeviceMallocContiguous, and 2 MB pieces mapped in shuffled / in allocation order
(tools/scatter_alloc.hip).  Per-kernel HIP-event times."""
import ctypes, os, sys
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1); xp = frontend.TorchArrays("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
sc = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libscatter_alloc.so"))
sc.scatter_alloc.restype = ctypes.c_void_p
sc.scatter_alloc.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_int]
ncol, nlay = 100000, 60
kd = synth.make_kdist("lw"); atm = synth.make_atmosphere(ncol, nlay, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))
emis = xp.full((ncol, kd.ngpt), 0.98)


class View:
    def __init__(self, ptr, shape_f):
        self.__cuda_array_interface__ = {"shape": tuple(reversed(shape_f)), "typestr": "<f8", "data": (ptr, False), "version": 2}


def alloc(kind, shape_f, seed):
    n = 8
    for s in shape_f: n *= s
    if kind == "torch":
        return xp.empty(shape_f)
    if kind == "contiguous":
        p = ctypes.c_void_p()
        assert hip.hipExtMallocWithFlags(ctypes.byref(p), n, 0x4) == 0
        ptr = p.value
    else:
        ptr = sc.scatter_alloc(n, (int(kind.split(":")[1]) << 20) if ":" in kind else 0, seed, 0 if kind.startswith("inorder") else 1)
        assert ptr
    return torch.as_tensor(View(ptr, shape_f), device="cuda")


def timed(bufs, rb, n=3):
    def step():
        go.gas_optics_lw(ncol, nlay, play, plev, tlay, tsfc, col_gas, tlev, False, buffers=bufs)
        frontend.rte_lw(lib, xp, ncol, nlay, kd.ngpt, False, bufs["tau"], bufs["lay_src"], bufs["lev_src"], emis, bufs["sfc_src"], buffers=rb)
    step()
    hiplib.ext_call(lib, "rte_hip_profile_reset", []); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 1)
    for _ in range(n): step()
    torch.cuda.synchronize(); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 0)
    out = {}
    for i in range(hiplib.ext_call(lib, "rte_hip_profile_count", [])):
        buf = ctypes.create_string_buffer(128); cnt, ms = ctypes.c_longlong(0), ctypes.c_double(0)
        lib.raw("rte_hip_profile_get")(ctypes.c_int(i), buf, ctypes.c_int(128), ctypes.byref(cnt), ctypes.byref(ms))
        out[buf.value.decode()] = round(ms.value / max(1, cnt.value), 3)
    return out


shared, rb, keep = {}, {}, []
kinds = sys.argv[1:] or ["torch", "contiguous", "scatter", "inorder", "scatter:16", "torch", "scatter", "contiguous"]
for rnd, kind in enumerate(kinds):
    bufs = dict(shared)
    for k, sh in (("tau", (ncol, nlay, kd.ngpt)), ("lay_src", (ncol, nlay, kd.ngpt)), ("lev_src", (ncol, nlay + 1, kd.ngpt))):
        bufs[k] = alloc(kind, sh, 17 * rnd + len(k))
    o = timed(bufs, rb)
    if not shared:
        shared = {k: v for k, v in bufs.items() if k not in ("tau", "lay_src", "lev_src")}
    print("%-12s interp %.3f  tau %.3f  planck %.3f  solver %.3f   sum %.3f" % (kind, o["interpolation_kernel"], o["tau_absorption_kernel"], o["planck_source_kernel"],
          o["lw_noscat_seg_kernel"], o["interpolation_kernel"] + o["tau_absorption_kernel"] + o["planck_source_kernel"] + o["lw_noscat_seg_kernel"]), flush=True)
    keep.append(bufs)
