"""GPU-box helper: run compute_tau_absorption at benchmark size in a loop for N seconds (to sample clocks / power beside it with rocm-smi)."""
import sys, time
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1); hiplib.ext_call(lib, "rte_hip_share_geometry", ["i"], 1)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
ncol, nlay = 100000, 60
kd = synth.make_kdist("lw"); atm = synth.make_atmosphere(ncol, nlay, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, tlay, col_gas = (A(getattr(atm, k)) for k in ("play", "tlay", "col_gas"))
st = go.interpolation(ncol, nlay, play, tlay, col_gas)
tau = xp.empty((ncol, nlay, kd.ngpt))
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(20):
        lib.zero_array_3D(ncol, nlay, kd.ngpt, tau)
        go.compute_tau_absorption(ncol, nlay, st, play, tlay, col_gas, tau)
    torch.cuda.synchronize(); n += 20
print("calls", n, "ms per call", (time.time() - t0) / n * 1e3)
