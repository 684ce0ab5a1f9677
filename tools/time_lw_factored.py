"""GPU-box helper: the LW step of bench.py (1e5 columns x 60 layers x 256 g-points, deferred zero fill, shared geometry) with the
sources as the ABI defines them and FACTORED (rte_hip_compute_Planck_source_factored -> rte_hip_lw_solver_noscat_factored):
ms per step, per-kernel HIP-event times, and whether the fluxes are the same bits."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
hiplib.set_stream(lib, torch.cuda.current_stream().cuda_stream)
hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1); hiplib.ext_call(lib, "rte_hip_share_geometry", ["i"], 1)
import os
if os.environ.get("LW_SFC_LDS"): hiplib.ext_call(lib, "rte_hip_lw_sfc_lds", ["i"], int(os.environ["LW_SFC_LDS"]))
ncol, nlay = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 60
kd = synth.make_kdist("lw"); atm = synth.make_atmosphere(ncol, nlay, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))
emis = xp.full((ncol, kd.ngpt), 0.98)
bufs, rb, bufs_f, rb_f = {}, {}, {}, {}
def step_abi():
    go.gas_optics_lw(ncol, nlay, play, plev, tlay, tsfc, col_gas, tlev, atm.top_at_1, buffers=bufs)
    frontend.rte_lw(lib, xp, ncol, nlay, kd.ngpt, atm.top_at_1, bufs["tau"], bufs["lay_src"], bufs["lev_src"], emis, bufs["sfc_src"], buffers=rb)
def step_fac():
    go.gas_optics_lw(ncol, nlay, play, plev, tlay, tsfc, col_gas, tlev, atm.top_at_1, buffers=bufs_f, factored_sources=True)
    frontend.rte_lw_factored(lib, xp, ncol, nlay, kd.ngpt, kd.nbnd, go.t["band_lims_gpt"], atm.top_at_1, bufs_f["tau"], bufs_f["pfrac"],
                             bufs_f["planck_lay"], bufs_f["planck_lev"], emis, bufs_f["sfc_src"], buffers=rb_f)
def prof(fn, n=5):
    import ctypes
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    hiplib.ext_call(lib, "rte_hip_profile_reset", []); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 1)
    for _ in range(3): fn()
    torch.cuda.synchronize(); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 0)
    out = {}
    get = lib.raw("rte_hip_profile_get")
    for i in range(hiplib.ext_call(lib, "rte_hip_profile_count", [])):
        name = ctypes.create_string_buffer(128); n_ = ctypes.c_longlong(); tot = ctypes.c_double()
        get(ctypes.c_int(i), name, ctypes.c_int(128), ctypes.byref(n_), ctypes.byref(tot))
        out[name.value.decode()] = tot.value / 3
    return ms, out
for rep in range(2):
    for tag, fn in (("ABI sources", step_abi), ("factored   ", step_fac)):
        ms, k = prof(fn)
        print("%s %.2f ms/step  " % (tag, ms) + "  ".join("%s %.2f" % (n.replace("_kernel", ""), v) for n, v in sorted(k.items(), key=lambda kv: -kv[1]) if v > 0.03))
print("fluxes identical:", bool(torch.equal(rb["flux_up"], rb_f["flux_up"]) and torch.equal(rb["flux_dn"], rb_f["flux_dn"])),
      " max |diff| up %.3e" % float((rb["flux_up"] - rb_f["flux_up"]).abs().max()))
