"""GPU-box helper: production vs direct kernels of the gas-optics calls over many shapes (fast check, not a test)."""
import itertools, sys
import numpy as np
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib, synth
hip = hiplib.load(); xp = frontend.TorchArrays("cuda:0"); A = xp.asarray
worst = 0.0
SHARE = "share" in sys.argv[1:]  # production runs with the opt-in driver modes: deferred zero fill + geometry sharing (masks
                                 # from the interpolation call, ragged last blocks included)
RAGGED = "ragged" in sys.argv[1:]  # tables with 0 ... 8 minor intervals per band and regime (the tail pass of the tau kernel)
NFLAV = 2 if "fewflav" in sys.argv[1:] else 10  # two flavors: nearly every stage keeps the previous stage's weights (tau_slab.h)
for kind, ncol, nlay, top in itertools.product(("lw", "sw"), (512, 513, 1023, 1537), (1, 2, 7, 33, 64, 65, 100), (False, True)):
    kd = (synth.make_kdist(kind, ngpt=128, nbnd=8, nflav=NFLAV, minor_distribution="ragged") if RAGGED
          else synth.make_kdist(kind, ngpt=64, nbnd=4, nflav=NFLAV))
    atm = synth.make_atmosphere(ncol, nlay, seed=ncol + nlay, kdist=kd, top_at_1=top)
    outs = []
    for direct in (0, 1):
        hiplib.ext_call(hip, "rte_hip_force_direct_gather", ["i"], direct)
        hiplib.ext_call(hip, "rte_hip_share_geometry", ["i"], 1 if (SHARE and not direct) else 0)
        hiplib.ext_call(hip, "rte_hip_defer_zero", ["i"], 1 if (SHARE and not direct) else 0)
        go = frontend.GasOptics(hip, kd, xp)
        if kind == "lw":
            b = go.gas_optics_lw(ncol, nlay, A(atm.play), A(atm.plev), A(atm.tlay), A(atm.tsfc), A(atm.col_gas), A(atm.tlev), top)
            keys = ("tau", "lay_src", "lev_src", "sfc_src", "sfc_src_jac")
        else:
            b = go.gas_optics_sw(ncol, nlay, A(atm.play), A(atm.plev), A(atm.tlay), A(atm.col_gas), A(atm.col_dry))
            keys = ("tau_abs", "tau_rayleigh", "tau", "ssa")
        outs.append({k: np.array(xp.to_numpy(b[k])) for k in keys})
    hiplib.ext_call(hip, "rte_hip_force_direct_gather", ["i"], 0)
    hiplib.ext_call(hip, "rte_hip_share_geometry", ["i"], 0)
    hiplib.ext_call(hip, "rte_hip_defer_zero", ["i"], 0)
    for k in outs[0]:
        den = np.max(np.abs(outs[1][k])); err = float(np.max(np.abs(outs[0][k] - outs[1][k])) / (den if den else 1))
        worst = max(worst, err)
        if not err <= 1e-12: print("MISMATCH", kind, ncol, nlay, top, k, err)
print("worst relative difference production vs direct:", worst)
