"""GPU-box helper: rte_lw_solver_noscat at benchmark size for several g-point group counts (rte_hip_seg_groups)."""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
ncol, nlay, ngpt = 100000, 60, 256
g = torch.Generator(device="cuda").manual_seed(1)
def R(*sh):
    t = xp.empty(sh); t.uniform_(0.0, 1.0, generator=g); return t
tau, lay, lev = R(ncol, nlay, ngpt).mul_(2), R(ncol, nlay, ngpt).mul_(10).add_(1), R(ncol, nlay + 1, ngpt).mul_(10).add_(1)
emis, sfc = R(ncol, ngpt).mul_(0.1).add_(0.9), R(ncol, ngpt).mul_(10)
rb = {}
f = lambda: frontend.rte_lw(lib, xp, ncol, nlay, ngpt, False, tau, lay, lev, emis, sfc, buffers=rb)
for groups in [int(a) for a in sys.argv[1:]] or [0, 2, 4, 8, 16, 32]:
    hiplib.ext_call(lib, "rte_hip_lw_sfc_lds", ["i"], 0 if groups >= 100 else 1)
    groups = groups % 100 if groups >= 0 else groups  # negative: g-points per block given directly (uneven last group)
    hiplib.ext_call(lib, "rte_hip_seg_groups", ["i"], groups)
    f(); f(); torch.cuda.synchronize()
    hiplib.ext_call(lib, "rte_hip_profile_reset", []); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 1)
    for _ in range(5): f()
    torch.cuda.synchronize(); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 0)
    n = hiplib.ext_call(lib, "rte_hip_profile_count", []); out = {}
    for i in range(n):
        buf = ctypes.create_string_buffer(128); cnt, ms = ctypes.c_longlong(0), ctypes.c_double(0)
        lib.raw("rte_hip_profile_get")(ctypes.c_int(i), buf, ctypes.c_int(128), ctypes.byref(cnt), ctypes.byref(ms))
        out[buf.value.decode()] = round(ms.value / max(1, cnt.value), 3)
    print("groups", groups, out, "checksum", float(rb["flux_up"].sum()))
