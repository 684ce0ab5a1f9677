"""Seeded parity cases shared by the oracle tests, the golden-fixture generator and the GPU tests.

``run_suite(lib, xp, case)`` drives every kernel of the hot path through the C ABI of ``lib``
(any library exporting include/rte_rrtmgp_kernels.h: reference build, C restatement, HIP) with
arrays from backend ``xp`` and returns a flat ``{name: numpy array}`` dict of all outputs.
"""
from __future__ import annotations

import hashlib
import os
import sys
from dataclasses import dataclass

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import rte_rrtmgp_amd  # noqa: E402
from rte_rrtmgp_amd import frontend, synth  # noqa: E402


@dataclass
class Case:
    name: str
    kind: str  # "lw" | "sw"
    ncol: int
    nlay: int
    top_at_1: bool
    kd_kwargs: dict
    seed: int = 11


TINY = dict(ngpt=16, nbnd=4, ntemp=14, npres=12, neta=5, nflav=6, ngas=8, nminor_lower=7, nminor_upper=5)
MID = dict(ngpt=64, nbnd=8, nminor_lower=20, nminor_upper=12)

CASES = {
    "lw_tiny_sfc1": Case("lw_tiny_sfc1", "lw", 5, 10, False, TINY),
    "lw_tiny_top1": Case("lw_tiny_top1", "lw", 5, 10, True, TINY),
    "sw_tiny_sfc1": Case("sw_tiny_sfc1", "sw", 5, 10, False, TINY),
    "sw_tiny_top1": Case("sw_tiny_top1", "sw", 5, 10, True, TINY),
    "lw_mid_ragged": Case("lw_mid_ragged", "lw", 70, 33, False, MID, seed=5),   # ncol not a multiple of 64
    "lw_mid_top1": Case("lw_mid_top1", "lw", 67, 72, True, MID, seed=6),        # nlay > 64: 9-layer segments
    "sw_mid_ragged": Case("sw_mid_ragged", "sw", 70, 33, True, MID, seed=7),
    "lw_g256": Case("lw_g256", "lw", 3, 60, False, {}, seed=8),                  # the benchmark table shapes
    "sw_g224": Case("sw_g224", "sw", 3, 60, False, {}, seed=9),
}


def make_inputs(case: Case):
    kd = synth.make_kdist(case.kind, seed=1234, **case.kd_kwargs)
    if case.kd_kwargs.get("npres", 59) != 59:
        # tiny pressure grid: keep the synthetic atmosphere inside it
        pass
    atm = synth.make_atmosphere(case.ncol, case.nlay, seed=case.seed, top_at_1=case.top_at_1,
                                ngas=kd.ngas, kdist=kd)
    rng = np.random.default_rng(case.seed + 100)
    ex = {
        "sfc_emis": synth.F(rng.uniform(0.9, 1.0, size=(case.ncol, kd.ngpt))),
        "inc_flux": synth.F(rng.uniform(0.0, 3.0, size=(case.ncol, kd.ngpt))),
        "ssa": synth.F(rng.uniform(0.0, 0.95, size=(case.ncol, case.nlay, kd.ngpt))),
        "g": synth.F(rng.uniform(-0.2, 0.9, size=(case.ncol, case.nlay, kd.ngpt))),
        "mu0": synth.F(np.repeat(rng.uniform(0.15, 1.0, size=(case.ncol, 1)), case.nlay, axis=1)),
        "sfc_alb_dir": synth.F(rng.uniform(0.02, 0.5, size=(case.ncol, kd.ngpt))),
        "sfc_alb_dif": synth.F(rng.uniform(0.02, 0.5, size=(case.ncol, kd.ngpt))),
        "lw_Ds": synth.F(rng.uniform(1.2, 1.9, size=(case.ncol, kd.ngpt))),
    }
    return kd, atm, ex


def inputs_digest(kd, atm, ex) -> str:
    h = hashlib.sha256()
    for k in sorted(kd.arrays):
        h.update(np.ascontiguousarray(kd.arrays[k]).tobytes())
    for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas"):
        h.update(np.ascontiguousarray(getattr(atm, k)).tobytes())
    for k in sorted(ex):
        h.update(np.ascontiguousarray(ex[k]).tobytes())
    return h.hexdigest()


def run_suite(lib, xp, case: Case, inputs=None, which="all"):
    """Run the kernels; returns {name: numpy array}.  ``which``: 'all' or 'core' (the benchmark chain)."""
    kd, atm, ex = inputs if inputs is not None else make_inputs(case)
    A = xp.asarray
    out = {}
    ncol, nlay, ngpt = case.ncol, case.nlay, kd.ngpt
    go = frontend.GasOptics(lib, kd, xp)
    play, plev, tlay, tlev, tsfc = A(atm.play), A(atm.plev), A(atm.tlay), A(atm.tlev), A(atm.tsfc)
    col_gas, col_dry = A(atm.col_gas), A(atm.col_dry)

    def grab(prefix, d, names):
        for n in names:
            out[prefix + n] = np.array(xp.to_numpy(d[n]))

    if case.kind == "lw":
        b = go.gas_optics_lw(ncol, nlay, play, plev, tlay, tsfc, col_gas, tlev, case.top_at_1)
        st = b["interp"]
        for n in ("jtemp", "jpress", "tropo", "jeta", "col_mix", "fmajor", "fminor"):
            out["interp." + n] = np.array(xp.to_numpy(getattr(st, n)))
        grab("", b, ["tau", "lay_src", "lev_src", "sfc_src", "sfc_src_jac"])
        emis = A(ex["sfc_emis"])
        # the benchmark configuration: broadband, one Gauss angle
        r = frontend.rte_lw(lib, xp, ncol, nlay, ngpt, case.top_at_1, b["tau"], b["lay_src"], b["lev_src"],
                            emis, b["sfc_src"])
        grab("lw1.", r, ["flux_up", "flux_dn"])
        if which == "core":
            return out
        # 3 angles, Jacobian, incident flux, broadband
        r = frontend.rte_lw(lib, xp, ncol, nlay, ngpt, case.top_at_1, b["tau"], b["lay_src"], b["lev_src"],
                            emis, b["sfc_src"], n_gauss_angles=3, inc_flux=A(ex["inc_flux"]),
                            sfc_src_jac=b["sfc_src_jac"], do_jacobians=True)
        grab("lw3j.", r, ["flux_up", "flux_dn", "flux_up_jac"])
        # spectral output, 2 angles
        r = frontend.rte_lw(lib, xp, ncol, nlay, ngpt, case.top_at_1, b["tau"], b["lay_src"], b["lev_src"],
                            emis, b["sfc_src"], n_gauss_angles=2, do_broadband=False)
        grab("lw2s.", r, ["gpt_flux_up", "gpt_flux_dn"])
        # spectral output, 1 angle, user secants, Jacobian
        r = frontend.rte_lw(lib, xp, ncol, nlay, ngpt, case.top_at_1, b["tau"], b["lay_src"], b["lev_src"],
                            emis, b["sfc_src"], lw_Ds=A(ex["lw_Ds"]), do_broadband=False,
                            sfc_src_jac=b["sfc_src_jac"], do_jacobians=True)
        grab("lwDs.", r, ["gpt_flux_up", "gpt_flux_dn", "flux_up_jac"])
        # rescaling (Tang) with synthetic ssa, g; broadband + Jacobian, then spectral
        ssa, g = A(ex["ssa"]), A(ex["g"])
        r = frontend.rte_lw(lib, xp, ncol, nlay, ngpt, case.top_at_1, b["tau"], b["lay_src"], b["lev_src"],
                            emis, b["sfc_src"], ssa=ssa, g=g, sfc_src_jac=b["sfc_src_jac"], do_jacobians=True,
                            n_gauss_angles=2)
        grab("lwresc.", r, ["flux_up", "flux_dn", "flux_up_jac"])
        r = frontend.rte_lw(lib, xp, ncol, nlay, ngpt, case.top_at_1, b["tau"], b["lay_src"], b["lev_src"],
                            emis, b["sfc_src"], ssa=ssa, g=g, do_broadband=False)
        grab("lwrescs.", r, ["gpt_flux_up", "gpt_flux_dn"])
        # two-stream LW (spectral + reduce)
        r = frontend.rte_lw(lib, xp, ncol, nlay, ngpt, case.top_at_1, b["tau"], b["lay_src"], b["lev_src"],
                            emis, b["sfc_src"], ssa=ssa, g=g, use_2stream=True, inc_flux=A(ex["inc_flux"]))
        grab("lw2str.", r, ["gpt_flux_up", "gpt_flux_dn", "flux_up", "flux_dn"])
        # net flux reducers
        net = xp.empty((ncol, nlay + 1))
        lib.rte_net_broadband_full(ncol, nlay + 1, ngpt, r["gpt_flux_dn"], r["gpt_flux_up"], net)
        out["net_full"] = np.array(xp.to_numpy(net))
        net2 = xp.empty((ncol, nlay + 1))
        lib.rte_net_broadband_precalc(ncol, nlay + 1, r["flux_dn"], r["flux_up"], net2)
        out["net_precalc"] = np.array(xp.to_numpy(net2))
    else:
        b = go.gas_optics_sw(ncol, nlay, play, plev, tlay, col_gas, col_dry)
        st = b["interp"]
        for n in ("jtemp", "jpress", "tropo", "jeta", "col_mix", "fmajor", "fminor"):
            out["interp." + n] = np.array(xp.to_numpy(getattr(st, n)))
        grab("", b, ["tau_abs", "tau_rayleigh", "tau", "ssa", "g", "toa_src"])
        mu0 = A(ex["mu0"])
        adir, adif = A(ex["sfc_alb_dir"]), A(ex["sfc_alb_dif"])
        r = frontend.rte_sw(lib, xp, ncol, nlay, ngpt, case.top_at_1, b["tau"], b["ssa"], b["g"], mu0,
                            b["toa_src"], adir, adif)
        grab("sw.", r, ["flux_up", "flux_dn", "flux_dir"])
        if which == "core":
            return out
        r = frontend.rte_sw(lib, xp, ncol, nlay, ngpt, case.top_at_1, b["tau"], b["ssa"], b["g"], mu0,
                            b["toa_src"], adir, adif, inc_flux_dif=A(ex["inc_flux"]), do_broadband=False)
        grab("sws.", r, ["gpt_flux_up", "gpt_flux_dn", "gpt_flux_dir"])
        # strongly scattering cloudy-like layers
        r = frontend.rte_sw(lib, xp, ncol, nlay, ngpt, case.top_at_1, b["tau"], A(ex["ssa"]), A(ex["g"]), mu0,
                            b["toa_src"], adir, adif)
        grab("swc.", r, ["flux_up", "flux_dn", "flux_dir"])
        r = frontend.rte_sw(lib, xp, ncol, nlay, ngpt, case.top_at_1, b["tau"], None, None, mu0, b["toa_src"],
                            None, None, noscat=True)
        grab("swn.", r, ["gpt_flux_dir", "flux_dir"])
    run_optprops(lib, xp, case, out)
    # array utilities
    z = xp.full((ncol, 3, 2, 2), 7.0)
    lib.zero_array_4D(ncol, 3, 2, 2, z)
    out["zero4"] = np.array(xp.to_numpy(z))
    s = xp.empty((ncol, 7))
    lib.set_to_scalar_2D(ncol, 7, s, 2.5)
    out["set2"] = np.array(xp.to_numpy(s))
    return out


def run_optprops(lib, xp, case: Case, out: dict):
    """Optical-properties arithmetic, cloud look-up-table optics and column subsetting (elementwise
    kernels of the all-sky path) on small seeded arrays."""
    rng = np.random.default_rng(case.seed + 555)
    ncol, nlay, ngpt, nbnd, nm1, nm2 = case.ncol, min(case.nlay, 9), 12, 3, 3, 4
    lims = xp.asarray(np.array([[1, 5, 9], [4, 8, 12]], dtype=np.int32))
    A = xp.asarray
    F = synth.F

    def r3(n3, lo=0.0, hi=1.0):
        return F(rng.uniform(lo, hi, size=(ncol, nlay, n3)))

    base = {"tau": r3(ngpt, 0, 3), "ssa": r3(ngpt, 0, 0.99), "g": r3(ngpt, -0.5, 0.9),
            "p": F(rng.uniform(-0.5, 0.9, size=(nm1, ncol, nlay, ngpt)))}
    for tag, n3 in (("g", ngpt), ("b", nbnd)):  # operand 2 on g-points / on bands
        op2 = {"tau": r3(n3, 0, 2), "ssa": r3(n3, 0, 0.99), "g": r3(n3, -0.5, 0.9),
               "p": F(rng.uniform(-0.5, 0.9, size=(nm2, ncol, nlay, n3)))}
        sfx, extra = ("", ()) if tag == "g" else ("_bybnd", (nbnd, lims))
        pre = "increment" if tag == "g" else "inc"

        def call(name, scal, arrs):
            dev = [A(x) if isinstance(x, np.ndarray) else x for x in arrs]
            getattr(lib, f"rte_{pre}_{name}{sfx}")(ncol, nlay, ngpt, *scal, *dev, *extra)
            return dev

        d = call("1scalar_by_1scalar", (), [base["tau"].copy(), op2["tau"]]); out[f"op.{tag}.1s1s"] = np.array(xp.to_numpy(d[0]))
        d = call("1scalar_by_2stream", (), [base["tau"].copy(), op2["tau"], op2["ssa"]]); out[f"op.{tag}.1s2s"] = np.array(xp.to_numpy(d[0]))
        d = call("1scalar_by_nstream", (), [base["tau"].copy(), op2["tau"], op2["ssa"]]); out[f"op.{tag}.1sns"] = np.array(xp.to_numpy(d[0]))
        d = call("2stream_by_1scalar", (), [base["tau"].copy(), base["ssa"].copy(), op2["tau"]])
        out[f"op.{tag}.2s1s.tau"], out[f"op.{tag}.2s1s.ssa"] = (np.array(xp.to_numpy(x)) for x in d[:2])
        d = call("2stream_by_2stream", (), [base["tau"].copy(), base["ssa"].copy(), base["g"].copy(), op2["tau"], op2["ssa"], op2["g"]])
        out[f"op.{tag}.2s2s.tau"], out[f"op.{tag}.2s2s.ssa"], out[f"op.{tag}.2s2s.g"] = (np.array(xp.to_numpy(x)) for x in d[:3])
        d = call("2stream_by_nstream", (nm2,), [base["tau"].copy(), base["ssa"].copy(), base["g"].copy(), op2["tau"], op2["ssa"], op2["p"]])
        out[f"op.{tag}.2sns.g"] = np.array(xp.to_numpy(d[2]))
        d = call("nstream_by_1scalar", (), [base["tau"].copy(), base["ssa"].copy(), op2["tau"]])
        out[f"op.{tag}.ns1s.ssa"] = np.array(xp.to_numpy(d[1]))
        d = call("nstream_by_2stream", (nm1,), [base["tau"].copy(), base["ssa"].copy(), base["p"].copy(), op2["tau"], op2["ssa"], op2["g"]])
        out[f"op.{tag}.ns2s.p"], out[f"op.{tag}.ns2s.ssa"] = np.array(xp.to_numpy(d[2])), np.array(xp.to_numpy(d[1]))
        d = call("nstream_by_nstream", (nm1, nm2), [base["tau"].copy(), base["ssa"].copy(), base["p"].copy(), op2["tau"], op2["ssa"], op2["p"]])
        out[f"op.{tag}.nsns.p"], out[f"op.{tag}.nsns.tau"] = np.array(xp.to_numpy(d[2])), np.array(xp.to_numpy(d[0]))
    # delta scaling
    t, s_, g_ = A(base["tau"].copy()), A(base["ssa"].copy()), A(base["g"].copy())
    lib.rte_delta_scale_2str_k(ncol, nlay, ngpt, t, s_, g_)
    out["ds.tau"], out["ds.ssa"], out["ds.g"] = (np.array(xp.to_numpy(x)) for x in (t, s_, g_))
    t, s_, g_ = A(base["tau"].copy()), A(base["ssa"].copy()), A(base["g"].copy())
    lib.rte_delta_scale_2str_f_k(ncol, nlay, ngpt, t, s_, g_, A(r3(ngpt, 0, 0.8)))
    out["dsf.tau"], out["dsf.ssa"], out["dsf.g"] = (np.array(xp.to_numpy(x)) for x in (t, s_, g_))
    # column subsets (colS..colE, 1-based inclusive)
    cS, cE = 2, max(2, ncol - 1)
    nc = cE - cS + 1
    o = xp.empty((nc, nlay, ngpt)); lib.rte_extract_subset_dim1_3d(ncol, nlay, ngpt, A(base["tau"]), cS, cE, o)
    out["sub.3d"] = np.array(xp.to_numpy(o))
    o = xp.empty((nm1, nc, nlay, ngpt)); lib.rte_extract_subset_dim2_4d(nm1, ncol, nlay, ngpt, A(base["p"]), cS, cE, o)
    out["sub.4d"] = np.array(xp.to_numpy(o))
    o = xp.empty((nc, nlay, ngpt)); lib.rte_extract_subset_absorption_tau(ncol, nlay, ngpt, A(base["tau"]), A(base["ssa"]), cS, cE, o)
    out["sub.abs"] = np.array(xp.to_numpy(o))
    # cloud optics from tables
    nsteps = 20
    mask = F(rng.random((ncol, nlay)) < 0.6, np.bool_)
    lwp, re = F(rng.uniform(0, 80, (ncol, nlay))), F(rng.uniform(2.5, 21.0, (ncol, nlay)))
    tabs = [F(rng.uniform(0.01, 1.0, (nsteps, ngpt))) for _ in range(3)]
    outs = [xp.empty((ncol, nlay, ngpt)) for _ in range(3)]
    lib.rrtmgp_compute_cld_from_table(ncol, nlay, ngpt, A(mask), A(lwp), A(re), nsteps, 1.0, 2.5, *[A(x) for x in tabs], *outs)
    out["cld.tau"], out["cld.taussa"], out["cld.taussag"] = (np.array(xp.to_numpy(x)) for x in outs)
    return out


def rel_err(a, b):
    """max |a-b| / max|b| (arrays may be int/bool: exact compare -> 0 or inf)."""
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind in "ib" or b.dtype.kind in "ib":
        return 0.0 if np.array_equal(a, b) else float("inf")
    den = np.max(np.abs(b))
    if den == 0:
        return float(np.max(np.abs(a)))
    return float(np.max(np.abs(a - b)) / den)


def elem_err(a, b, floor_frac=1e-8):
    """Elementwise relative error  max |a-b| / max(|b|, floor)  with the absolute floor taken PER PLANE of the last
    (g-point / band) axis: floor = floor_frac * max|b[..., g]|.  Unlike rel_err (one max-norm over the whole array),
    an O(1) relative error in a weak g-point -- optical depths and sources span many decades across g-points --
    cannot hide behind the strong ones; the floor only forgives values that are negligible inside their own plane."""
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind in "ib" or b.dtype.kind in "ib":
        return 0.0 if np.array_equal(a, b) else float("inf")
    if b.size == 0:
        return 0.0
    a, b = a.astype(np.float64), b.astype(np.float64)
    if b.ndim >= 2:
        plane_max = np.max(np.abs(b).reshape(-1, b.shape[-1]), axis=0)
        floor = (floor_frac * plane_max).reshape((1,) * (b.ndim - 1) + (-1,))
    else:
        floor = floor_frac * np.max(np.abs(b))
    den = np.maximum(np.abs(b), np.maximum(floor, 1e-300))
    return float(np.max(np.abs(a - b) / den))
