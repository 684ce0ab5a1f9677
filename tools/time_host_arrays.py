"""GPU-box helper: the LW chain with HOST arrays (what the unchanged Fortran frontend passes): every call stages its
arrays through the device arena and copies its outputs back -- the PCIe-inclusive rate of the drop-in boundary."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); xp = frontend.NumpyArrays()
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
kd = synth.make_kdist("lw"); atm = synth.make_atmosphere(ncol, 60, seed=42, kdist=kd)
go = frontend.GasOptics(lib, kd, xp); A = xp.asarray
play, plev, tlay, tlev, tsfc, col_gas = (A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"))
emis = xp.full((ncol, kd.ngpt), 0.98); bufs, rb = {}, {}
def step():
    go.gas_optics_lw(ncol, 60, play, plev, tlay, tsfc, col_gas, tlev, atm.top_at_1, buffers=bufs)
    frontend.rte_lw(lib, xp, ncol, 60, kd.ngpt, atm.top_at_1, bufs["tau"], bufs["lay_src"], bufs["lev_src"], emis, bufs["sfc_src"], buffers=rb)
step()
t0 = time.perf_counter(); n = 3
for _ in range(n): step()
dt = (time.perf_counter() - t0) / n
print("host arrays (pageable), %d columns: %.1f ms per step -> %.3g columns/s; flux_up toa mean %.6f" % (ncol, dt * 1e3, ncol / dt, float(np.mean(rb["flux_up"][:, -1]))))
