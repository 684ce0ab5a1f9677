"""compute_tau_absorption: the matrix-core kernel (rte_hip_tau_variant(10), csrc/tau_mx.h) against the specialised-wave
slab kernel (9) and the direct-gather kernels -- worst relative differences on several tables / atmospheres, then HIP-event
times at benchmark size (1e5 x 60 x 256), overwrite (deferred zero fill) and accumulate."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import rte_rrtmgp_amd  # noqa
from rte_rrtmgp_amd import frontend, hiplib, synth
import cases

hip = hiplib.load()
hiplib.set_stream(hip, torch.cuda.current_stream().cuda_stream)
xp = frontend.TorchArrays("cuda:0")
A = xp.asarray


def variant(v):
    hiplib.ext_call(hip, "rte_hip_tau_variant", ["i"], v)


def tau_of(kd, atm, ncol, nlay, start, mode):
    go = frontend.GasOptics(hip, kd, xp)
    play, tlay, col_gas = A(atm.play), A(atm.tlay), A(atm.col_gas)
    st = go.interpolation(ncol, nlay, play, tlay, col_gas)
    hiplib.ext_call(hip, "rte_hip_force_direct_gather", ["i"], 1 if mode == "direct" else 0)
    variant(10 if mode == "mx" else 9)
    tau = xp.full((ncol, nlay, kd.ngpt), start)
    go.compute_tau_absorption(ncol, nlay, st, play, tlay, col_gas, tau)
    out = xp.to_numpy(tau).copy()
    hiplib.ext_call(hip, "rte_hip_force_direct_gather", ["i"], 0)
    variant(9)
    return out


if "--time-only" not in sys.argv:
    for label, kdargs, ncol, nlay, atmargs in [
        ("4 bands x 16, 1100 x 24", dict(ngpt=64, nbnd=4, nminor_lower=11, nminor_upper=7), 1100, 24, dict(seed=77)),
        ("4 bands, top_at_1", dict(ngpt=64, nbnd=4, nminor_lower=11, nminor_upper=7), 1100, 24, dict(seed=78, top_at_1=True)),
        ("g256 even, 2000 x 60", dict(), 2000, 60, dict(seed=5)),
        ("g256 ragged minors, 1100 x 24", dict(minor_distribution="ragged"), 1100, 24, dict(seed=8)),
        ("g256 ragged, top_at_1, sites", dict(minor_distribution="ragged"), 1500, 30, dict(seed=9, top_at_1=True, climate="sites")),
        ("wide bands 2 x 32", dict(ngpt=64, nbnd=2), 700, 19, dict(seed=3)),
        ("512 columns exactly", dict(ngpt=64, nbnd=4), 512, 12, dict(seed=4)),
    ]:
        kd = synth.make_kdist("lw", **kdargs)
        atm = synth.make_atmosphere(ncol, nlay, kdist=kd, **atmargs)
        for start in (0.125,):
            ref = tau_of(kd, atm, ncol, nlay, start, "direct")
            v9 = tau_of(kd, atm, ncol, nlay, start, "v9")
            mx = tau_of(kd, atm, ncol, nlay, start, "mx")
            print(f"{label:36s} start {start}: |v9 - direct| {cases.rel_err(v9, ref):.2e}   |mx - direct| {cases.rel_err(mx, ref):.2e}"
                  f"   elementwise worst mx {np.max(np.abs(mx - ref) / np.abs(ref)):.2e}", flush=True)

# ---- timing at benchmark size
ncol, nlay = 100000, 60
for climate in ("rce", "sites"):
    kd = synth.make_kdist("lw")
    atm = synth.make_atmosphere(ncol, nlay, seed=42, kdist=kd, climate=climate)
    go = frontend.GasOptics(hip, kd, xp)
    play, tlay, col_gas = A(atm.play), A(atm.tlay), A(atm.col_gas)
    st = go.interpolation(ncol, nlay, play, tlay, col_gas)
    tau = xp.zeros((ncol, nlay, kd.ngpt))
    for defer in (1, 0):
        hiplib.ext_call(hip, "rte_hip_defer_zero", ["i"], defer)
        for v in (9, 10, 9, 10):
            variant(v)
            ts = []
            for rep in range(8):
                if defer:
                    hip.zero_array_3D(ncol, nlay, kd.ngpt, tau)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                go.compute_tau_absorption(ncol, nlay, st, play, tlay, col_gas, tau)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            print(f"{climate:6s} defer_zero={defer} variant {v:2d}: min {min(ts[2:]):.3f} ms  median {sorted(ts[2:])[3]:.3f} ms (whole call, events on the null stream)", flush=True)
    hiplib.ext_call(hip, "rte_hip_defer_zero", ["i"], 0)
variant(9)
