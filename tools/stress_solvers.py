"""GPU-box helper: segmented vs generic solver kernels over many shapes (fast check, not a test)."""
import itertools, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib
hip = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
g = torch.Generator(device="cuda").manual_seed(7)
def R(*sh, lo=0.0, hi=1.0):
    t = xp.empty(sh); t.uniform_(lo, hi, generator=g); return t
worst = {}
for ncol, nlay, ngpt, top in itertools.product((1, 63, 64, 65, 200), (1, 2, 7, 8, 9, 16, 33, 63, 64, 65, 72, 73, 80, 81, 96, 97, 128, 129, 137, 144, 145, 159, 160, 161), (1, 5, 16, 37), (False, True)):
    tau, ssa, gg = R(ncol, nlay, ngpt, hi=3.0), R(ncol, nlay, ngpt, hi=0.999), R(ncol, nlay, ngpt, lo=-0.3, hi=0.9)
    lay, lev = R(ncol, nlay, ngpt, lo=1, hi=10), R(ncol, nlay + 1, ngpt, lo=1, hi=10)
    emis, sfc, inc = R(ncol, ngpt, lo=0.8, hi=1.0), R(ncol, ngpt, hi=10), R(ncol, ngpt)
    mu0 = R(ncol, nlay, lo=-0.2, hi=1.0); adir, adif, idir = R(ncol, ngpt), R(ncol, ngpt), R(ncol, ngpt, hi=100)
    res = []
    for gen in (0, 1):
        hiplib.ext_call(hip, "rte_hip_force_generic_lw", ["i"], gen); hiplib.ext_call(hip, "rte_hip_force_generic_sw", ["i"], gen)
        o = {}
        r = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, top, tau, lay, lev, emis, sfc, inc_flux=inc, buffers={})
        o["lw.up"], o["lw.dn"] = r["flux_up"].clone(), r["flux_dn"].clone()
        # three angles + Jacobian (broadband), and spectral output with two angles + Jacobian
        sj = sfc * 0.1
        r = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, top, tau, lay, lev, emis, sfc, inc_flux=inc, n_gauss_angles=3, sfc_src_jac=sj, do_jacobians=True, buffers={})
        o["lw3j.up"], o["lw3j.dn"], o["lw3j.jac"] = r["flux_up"].clone(), r["flux_dn"].clone(), r["flux_up_jac"].clone()
        r = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, top, tau, lay, lev, emis, sfc, inc_flux=inc, n_gauss_angles=2, sfc_src_jac=sj, do_jacobians=True, do_broadband=False, buffers={})
        o["lws.up"], o["lws.dn"], o["lws.jac"] = r["gpt_flux_up"].clone(), r["gpt_flux_dn"].clone(), r["flux_up_jac"].clone()
        r = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, top, tau, lay, lev, emis, sfc, ssa=ssa, g=gg, use_2stream=True, inc_flux=inc, buffers={})
        o["lw2.up"], o["lw2.dn"] = r["flux_up"].clone(), r["flux_dn"].clone()
        r = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, top, tau, lay, lev, emis, sfc, ssa=ssa, g=gg, inc_flux=inc, buffers={})
        o["lwr.up"], o["lwr.dn"] = r["flux_up"].clone(), r["flux_dn"].clone()
        r = frontend.rte_sw(hip, xp, ncol, nlay, ngpt, top, tau, ssa, gg, mu0, idir, adir, adif, inc_flux_dif=inc, buffers={})
        o["sw.up"], o["sw.dn"], o["sw.dir"] = r["flux_up"].clone(), r["flux_dn"].clone(), r["flux_dir"].clone()
        r = frontend.rte_sw(hip, xp, ncol, nlay, ngpt, top, tau, ssa, gg, mu0, idir, adir, adif, inc_flux_dif=inc, do_broadband=False, buffers={})
        o["sws.up"], o["sws.dn"], o["sws.dir"] = r["gpt_flux_up"].clone(), r["gpt_flux_dn"].clone(), r["gpt_flux_dir"].clone()
        res.append(o)
    hiplib.ext_call(hip, "rte_hip_force_generic_lw", ["i"], 0); hiplib.ext_call(hip, "rte_hip_force_generic_sw", ["i"], 0)
    for k in res[0]:
        den = float(res[1][k].abs().max()); err = float((res[0][k] - res[1][k]).abs().max()) / (den if den else 1.0)
        worst[k] = max(worst.get(k, 0.0), err)
        if not err <= 1e-11: print("MISMATCH", ncol, nlay, ngpt, top, k, err)
print("worst relative differences segmented vs generic:", {k: float(f"{v:.2e}") for k, v in worst.items()})
