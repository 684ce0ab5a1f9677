"""The small computations either side of the hot path: Planck function on a
wavenumber grid, by-band flux reductions, dry-air column amounts, column gas amounts, level temperatures, optimal
transport angles, band -> g-point expansion, secant fill, the 1scl / nstr branches of combine_abs_and_rayleigh and the
RFMIP-SW boundary conditions.

CPU (``-m "not gpu"``): the C restatements (oracle/glue_oracle.c) against (a) the reference build where the reference
exposes the routine (rte_compute_Planck_source_1D/2D; get_layer_number / get_layer_mass through the wrapper symbols of
oracle/ref_wrappers.F90) and (b) the explicit formulas written out in numpy here.
GPU (``-m gpu``): the HIP kernels (csrc/glue.hip) against the C restatements, on device-resident arrays.
"""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from rte_rrtmgp_amd import cabi, frontend, hiplib  # noqa: E402

M_DRY, GRAV = 0.028964, 9.80665  # rte/kernels/mo_gas_optics_constants.F90:32-35


def _inputs(ncol=70, nlay=12, ngas=5, nbnd=4, gpb=8, seed=5):
    rng = np.random.default_rng(seed)
    F = np.asfortranarray
    ngpt = nbnd * gpb
    plev = F(np.sort(rng.uniform(10.0, 101000.0, (ncol, nlay + 1)), axis=1)[:, ::-1])
    play = F(0.5 * (plev[:, 1:] + plev[:, :-1]))
    tlay = F(rng.uniform(180.0, 320.0, (ncol, nlay)))
    vmr = F(rng.uniform(1e-6, 2e-2, (ncol, nlay, ngas)))
    band_lims = F(np.stack([1 + gpb * np.arange(nbnd), gpb * (1 + np.arange(nbnd))]).astype(np.int32))
    return dict(ncol=ncol, nlay=nlay, ngas=ngas, nbnd=nbnd, ngpt=ngpt, plev=plev, play=play, tlay=tlay, vmr=vmr,
                vmr_gcl=F(np.ascontiguousarray(np.moveaxis(vmr, 2, 0))), band_lims=band_lims,
                mol_weights=F(rng.uniform(0.002, 0.05, ngas)),
                tau=F(rng.uniform(0.0, 0.5, (ncol, nlay, ngpt))), tau_ray=F(rng.uniform(0.0, 0.1, (ncol, nlay, ngpt))),
                fit=F(np.stack([rng.uniform(0.1, 0.4, nbnd), rng.uniform(1.5, 1.8, nbnd)])),
                nus=F(np.linspace(10.0, 3000.0, 9)), dnus=F(np.full(9, 25.0)),
                gpt_up=F(rng.uniform(0.0, 5.0, (ncol, nlay + 1, ngpt))), gpt_dn=F(rng.uniform(0.0, 5.0, (ncol, nlay + 1, ngpt))),
                per_band=F(rng.uniform(0.0, 1.0, (nbnd, ncol))), Ds=F(np.array([1.3, 1.7, 2.4])),
                tsi=F(rng.uniform(1300.0, 1400.0, ncol)), toa=F(rng.uniform(0.1, 9.0, (ncol, ngpt))),
                sza=F(rng.uniform(0.0, 120.0, ncol)), usecol=F(rng.uniform(0.0, 120.0, ncol) < 90.0), alb=F(rng.uniform(0, 1, ncol)))


def run_glue(lib, xp, d):
    """Every glue entry point once; returns {name: numpy array}."""
    A = xp.asarray
    E = lambda *a, **k: hiplib.ext_call(lib, *a, **k)  # noqa: E731
    ncol, nlay, ngas, nbnd, ngpt = d["ncol"], d["nlay"], d["ngas"], d["nbnd"], d["ngpt"]
    out = {}
    plev, play, tlay, vmr, bl = A(d["plev"]), A(d["play"]), A(d["tlay"]), A(d["vmr"]), A(d["band_lims"])
    # Planck function on a wavenumber grid (reference C ABI)
    src2 = xp.empty((ncol, nlay, 9)); lib.rte_compute_Planck_source_2D(ncol, nlay, 9, A(d["nus"]), A(d["dnus"]), tlay, src2)
    src1 = xp.empty((ncol, 9)); lib.rte_compute_Planck_source_1D(ncol, 9, A(d["nus"]), A(d["dnus"]), A(d["tlay"][:, 0]), src1)
    out["planck2d"], out["planck1d"] = src2, src1
    # by-band reductions (reference C ABI)
    up, dn = A(d["gpt_up"]), A(d["gpt_dn"])
    bu, bd, bn, bn2 = (xp.empty((ncol, nlay + 1, nbnd)) for _ in range(4))
    lib.rte_sum_byband(ncol, nlay + 1, ngpt, nbnd, bl, up, bu)
    lib.rte_sum_byband(ncol, nlay + 1, ngpt, nbnd, bl, dn, bd)
    lib.rte_net_byband_full(ncol, nlay + 1, ngpt, nbnd, bl, dn, up, bn)
    lib.net_byband_precalc(ncol, nlay + 1, nbnd, bd, bu, bn2)
    out.update(byband_up=bu, byband_dn=bd, byband_net=bn, byband_net_precalc=bn2)
    # dry-air column amounts, layer masses, column gas amounts
    col_dry = xp.empty((ncol, nlay)); E("rte_hip_get_layer_number", "iiaadda", ncol, nlay, A(d["vmr"][:, :, 0]), plev, M_DRY, GRAV, col_dry)
    mass = xp.empty((ngas, ncol, nlay)); E("rte_hip_get_layer_mass", "iiiaaadda", ncol, nlay, ngas, A(d["vmr_gcl"]), plev, A(d["mol_weights"]), M_DRY, GRAV, mass)
    col_gas = xp.empty((ncol, nlay, ngas + 1)); E("rte_hip_col_gas_fill", "iiiaaa", ncol, nlay, ngas, vmr, col_dry, col_gas)
    out.update(col_dry=col_dry, layer_mass=mass, col_gas=col_gas)
    tlev = xp.empty((ncol, nlay + 1)); E("rte_hip_tlev_interp", "iiaaaa", ncol, nlay, play, plev, tlay, tlev)
    out["tlev"] = tlev
    ang = xp.empty((ncol, ngpt)); E("rte_hip_compute_optimal_angles", "iiiiaaaa", ncol, nlay, ngpt, nbnd, bl, A(d["tau"]), A(d["fit"]), ang)
    out["optimal_angles"] = ang
    t1 = xp.empty((ncol, nlay, ngpt)); E("rte_hip_combine_abs_and_rayleigh_1scl", "iiiaaa", ncol, nlay, ngpt, A(d["tau"]), A(d["tau_ray"]), t1)
    tn, sn, pn = xp.empty((ncol, nlay, ngpt)), xp.empty((ncol, nlay, ngpt)), xp.empty((3, ncol, nlay, ngpt))
    E("rte_hip_combine_abs_and_rayleigh_nstr", "iiiiaaaaa", ncol, nlay, ngpt, 3, A(d["tau"]), A(d["tau_ray"]), tn, sn, pn)
    out.update(comb1_tau=t1, combn_tau=tn, combn_ssa=sn, combn_p=pn)
    ex = xp.empty((ncol, ngpt)); E("rte_hip_expand_and_transpose", "iiiaaa", ncol, nbnd, ngpt, bl, A(d["per_band"]), ex)
    sec = xp.empty((ncol, ngpt, 3)); E("rte_hip_secants_fill", "iiiaa", ncol, ngpt, 3, A(d["Ds"]), sec)
    out.update(expand=ex, secants=sec)
    toa = A(d["toa"].copy(order="F")); E("rte_hip_rfmip_sw_toa_renorm", "iiaa", ncol, ngpt, A(d["tsi"]), toa)
    mu0 = xp.empty((ncol,)); E("rte_hip_rfmip_sw_mu0", "iaaa", ncol, A(d["sza"]), A(d["usecol"]), mu0)
    albs = xp.empty((nbnd, ncol)); E("rte_hip_broadcast_cols", "iiaa", nbnd, ncol, A(d["alb"]), albs)
    fu, fd = A(d["gpt_up"][:, :, 0].copy(order="F")), A(d["gpt_dn"][:, :, 0].copy(order="F"))
    E("rte_hip_mask_columns", "iiaaa", ncol, nlay + 1, A(d["usecol"]), fu, fd)
    out.update(toa=toa, mu0=mu0, alb_spec=albs, masked_up=fu, masked_dn=fd)
    xp.sync()
    return {k: np.array(xp.to_numpy(v)) for k, v in out.items()}


def numpy_formulas(d):
    """The same quantities written out in numpy from the reference's formulas (file:line in glue_oracle.c)."""
    ncol, nlay, ngas, nbnd, ngpt = d["ncol"], d["nlay"], d["ngas"], d["nbnd"], d["ngpt"]
    h, c, k = 6.626075540e-34, 2.99792458e8, 1.380649e-23
    B = lambda T, nu: 100.0 * 2.0 * h * (nu * 100.0) ** 3 * c ** 2 / (np.exp(h * c * nu * 100.0 / (k * T)) - 1.0)  # noqa: E731
    out = {"planck2d": B(d["tlay"][:, :, None], d["nus"][None, None, :]) * d["dnus"],
           "planck1d": B(d["tlay"][:, 0, None], d["nus"][None, :]) * d["dnus"]}
    bl = d["band_lims"]
    seg = lambda a: np.stack([a[:, :, bl[0, b] - 1:bl[1, b]].sum(axis=2) for b in range(nbnd)], axis=2)  # noqa: E731
    out.update(byband_up=seg(d["gpt_up"]), byband_dn=seg(d["gpt_dn"]), byband_net=seg(d["gpt_dn"] - d["gpt_up"]))
    out["byband_net_precalc"] = out["byband_dn"] - out["byband_up"]
    h2o, plev, play, tlay = d["vmr"][:, :, 0], d["plev"], d["play"], d["tlay"]
    dp = np.abs(plev[:, :-1] - plev[:, 1:])
    fact = 1.0 / (1.0 + h2o)
    m_air = (M_DRY + 0.018016 * h2o) * fact
    col_dry = 10.0 * dp * 6.02214076e23 * fact / (1000.0 * m_air * 100.0 * GRAV)
    out["col_dry"] = col_dry
    out["layer_mass"] = d["vmr_gcl"] * (d["mol_weights"][:, None, None] / M_DRY) * dp[None] / GRAV
    out["col_gas"] = np.concatenate([col_dry[:, :, None], d["vmr"] * col_dry[:, :, None]], axis=2)
    tlev = np.empty((ncol, nlay + 1))
    tlev[:, 0] = tlay[:, 0] + (plev[:, 0] - play[:, 0]) * (tlay[:, 1] - tlay[:, 0]) / (play[:, 1] - play[:, 0])
    tlev[:, nlay] = tlay[:, -1] + (plev[:, nlay] - play[:, -1]) * (tlay[:, -1] - tlay[:, -2]) / (play[:, -1] - play[:, -2])
    tlev[:, 1:nlay] = (play[:, :-1] * tlay[:, :-1] * (plev[:, 1:nlay] - play[:, 1:]) +
                       play[:, 1:] * tlay[:, 1:] * (play[:, :-1] - plev[:, 1:nlay])) / (plev[:, 1:nlay] * (play[:, :-1] - play[:, 1:]))
    out["tlev"] = tlev
    band_of = np.repeat(np.arange(nbnd), ngpt // nbnd)
    out["optimal_angles"] = d["fit"][0, band_of][None, :] * np.exp(-d["tau"].sum(axis=1)) + d["fit"][1, band_of][None, :]
    t = d["tau"] + d["tau_ray"]
    out.update(comb1_tau=t, combn_tau=t, combn_ssa=d["tau_ray"] / t)
    p = np.zeros((3, ncol, nlay, ngpt)); p[1] = 0.1
    out["combn_p"] = p
    out["expand"] = d["per_band"].T[:, band_of]
    out["secants"] = np.broadcast_to(d["Ds"][None, None, :], (ncol, ngpt, 3))
    out["toa"] = d["toa"] * (d["tsi"] / d["toa"].sum(axis=1))[:, None]
    out["mu0"] = np.where(d["usecol"], np.cos(d["sza"] * np.arccos(-1.0) / 180.0), 1.0)
    out["alb_spec"] = np.broadcast_to(d["alb"][None, :], (nbnd, ncol))
    out["masked_up"] = np.where(d["usecol"][:, None], d["gpt_up"][:, :, 0], 0.0)
    out["masked_dn"] = np.where(d["usecol"][:, None], d["gpt_dn"][:, :, 0], 0.0)
    return out


def _assert_close(a, b, tol, label):
    for k in b:
        scale = np.maximum(np.abs(b[k]), 1e-300)
        err = float(np.max(np.abs(a[k] - b[k]) / scale))  # elementwise relative
        assert err <= tol, (label, k, err)


def test_c_restatement_matches_formulas():
    d = _inputs()
    got = run_glue(O.load_c(), frontend.NumpyArrays(), d)
    _assert_close(got, numpy_formulas(d), 2e-13, "oracle vs numpy formulas")


def test_c_restatement_matches_reference_build():
    ref = O.load_ref()
    if ref is None:
        pytest.skip("reference build absent")
    d = _inputs(seed=9)
    xp = frontend.NumpyArrays()
    got = run_glue(O.load_c(), xp, d)
    ncol, nlay, ngas = d["ncol"], d["nlay"], d["ngas"]
    src2 = xp.empty((ncol, nlay, 9)); ref.rte_compute_Planck_source_2D(ncol, nlay, 9, d["nus"], d["dnus"], d["tlay"], src2)
    src1 = xp.empty((ncol, 9)); ref.rte_compute_Planck_source_1D(ncol, 9, d["nus"], d["dnus"], np.asfortranarray(d["tlay"][:, 0]), src1)
    _assert_close(got, {"planck2d": src2, "planck1d": src1}, 1e-14, "oracle vs reference Planck")
    dll = ctypes.CDLL(ref.path) if hasattr(ref, "path") else None
    if dll is None or not hasattr(dll, "rte_ref_get_layer_number"):
        pytest.skip("reference wrappers absent from the reference build")
    I = lambda v: ctypes.byref(ctypes.c_int(v))  # noqa: E731
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    col_dry = xp.empty((ncol, nlay)); h2o = np.asfortranarray(d["vmr"][:, :, 0])
    dll.rte_ref_get_layer_number(I(ncol), I(nlay), P(h2o), P(d["plev"]), P(col_dry))
    mass = xp.empty((ngas, ncol, nlay))
    dll.rte_ref_get_layer_mass(I(ncol), I(nlay), I(ngas), P(d["vmr_gcl"]), P(d["plev"]), P(d["mol_weights"]),
                               ctypes.byref(ctypes.c_double(M_DRY)), P(mass))
    assert np.array_equal(got["col_dry"], col_dry)
    assert np.array_equal(got["layer_mass"], mass)


# ---- the frontend's own glue loops, pinned against the reference (tests/golden/glue_frontend.npz: recorded by
# oracle/glue_recorder.c from a run of the reference's Fortran frontend on its CPU kernels; generator beside the fixture)
def _frontend_fixture():
    f = np.load(os.path.join(ROOT, "tests", "golden", "glue_frontend.npz"))
    return {k: np.asfortranarray(f[k]) for k in f.files}


def _glue_on_fixture(lib, xp, fx):
    """col_gas, tlev, optimal-angle secants and the expanded emissivity from the inputs the reference frontend was given."""
    A = xp.asarray
    E = lambda *a, **k: hiplib.ext_call(lib, *a, **k)  # noqa: E731
    ncol, nlay, _, _, ngpt, nbnd, ngas = (int(v) for v in fx["meta"])
    col_gas = xp.empty((ncol, nlay, ngas + 1)); E("rte_hip_col_gas_fill", "iiiaaa", ncol, nlay, ngas, A(fx["vmr"]), A(fx["col_dry"]), col_gas)
    tlev = xp.empty((ncol, nlay + 1)); E("rte_hip_tlev_interp", "iiaaaa", ncol, nlay, A(fx["play"]), A(fx["plev"]), A(fx["tlay"]), tlev)
    ang = xp.empty((ncol, ngpt))
    E("rte_hip_compute_optimal_angles", "iiiiaaaa", ncol, nlay, ngpt, nbnd, A(fx["band_lims_gpt"]), A(fx["ref_tau"]), A(fx["optimal_angle_fit"]), ang)
    ex = xp.empty((ncol, ngpt)); E("rte_hip_expand_and_transpose", "iiiaaa", ncol, nbnd, ngpt, A(fx["band_lims_gpt"]), A(fx["sfc_emis_bnd"]), ex)
    xp.sync()
    return {k: np.array(xp.to_numpy(v)) for k, v in (("col_gas", col_gas), ("tlev", tlev), ("Ds", ang), ("sfc_emis_gpt", ex))}


def _check_against_frontend(got, fx, label):
    # products, sums and one division per element, no transcendental function: the same bits as the Fortran frontend's
    assert np.array_equal(got["col_gas"], fx["ref_col_gas"]), label
    assert np.array_equal(got["sfc_emis_gpt"], fx["ref_sfc_emis_gpt"]), label
    assert float(np.max(np.abs(got["tlev"] - fx["ref_tlev"]) / fx["ref_tlev"])) <= 4e-16, label
    # D = fit1 exp(-sum tau) + fit2: libm / device exp
    assert float(np.max(np.abs(got["Ds"] - fx["ref_Ds"][:, :, 0]) / fx["ref_Ds"][:, :, 0])) <= 1e-14, label
    assert fx["ref_Ds"].min() >= 1.0 and np.ptp(fx["ref_Ds"]) > 0.01 and np.ptp(fx["ref_sfc_emis_gpt"], axis=1).min() > 0  # a real test


def test_c_restatement_matches_the_reference_frontend():
    fx = _frontend_fixture()
    _check_against_frontend(_glue_on_fixture(O.load_c(), frontend.NumpyArrays(), fx), fx, "glue_oracle.c vs the reference frontend")


@pytest.mark.gpu
def test_hip_glue_matches_the_reference_frontend():
    fx = _frontend_fixture()
    _check_against_frontend(_glue_on_fixture(hiplib.load(), frontend.TorchArrays("cuda:0"), fx), fx, "csrc/glue.hip vs the reference frontend")


# ---- the RFMIP-SW driver's boundary conditions, pinned against the reference's own statements (tests/golden/rfmip_sw_glue.npz:
# examples/rfmip-clear-sky/rrtmgp_rfmip_sw.F90:269-318, :331-337 cut out of the reference file where it lies and compiled in a
# C-callable frame by oracle/build_rfmip_sw_glue.sh; generator beside the fixture)
def _rfmip_sw_on_fixture(lib, xp):
    f = np.load(os.path.join(ROOT, "tests", "golden", "rfmip_sw_glue.npz"))
    fx = {k: np.asfortranarray(f[k]) for k in f.files}
    b = int(np.asarray(fx["block"]).reshape(-1)[0]) - 1
    A = xp.asarray
    E = lambda *a, **k: hiplib.ext_call(lib, *a, **k)  # noqa: E731
    ncol, ngpt = fx["in_toa_flux"].shape
    nbnd, nlev = fx["out_sfc_alb_spec"].shape[0], fx["in_flux_up"].shape[1]
    usecol = np.asfortranarray(fx["in_usecol"][:, b] != 0)
    toa = A(fx["in_toa_flux"].copy(order="F")); E("rte_hip_rfmip_sw_toa_renorm", "iiaa", ncol, ngpt, A(np.asfortranarray(fx["in_tsi"][:, b])), toa)
    mu0 = xp.empty((ncol,)); E("rte_hip_rfmip_sw_mu0", "iaaa", ncol, A(np.asfortranarray(fx["in_sza"][:, b])), A(usecol), mu0)
    albs = xp.empty((nbnd, ncol)); E("rte_hip_broadcast_cols", "iiaa", nbnd, ncol, A(np.asfortranarray(fx["in_albedo"][:, b])), albs)
    fu, fd = A(np.asfortranarray(fx["in_flux_up"][:, :, b])), A(np.asfortranarray(fx["in_flux_dn"][:, :, b]))
    E("rte_hip_mask_columns", "iiaaa", ncol, nlev, A(usecol), fu, fd)
    xp.sync()
    got = {k: np.array(xp.to_numpy(v)) for k, v in (("toa", toa), ("mu0", mu0), ("alb", albs), ("fu", fu), ("fd", fd))}
    return got, fx, b


def _check_rfmip_sw(got, fx, b, label):
    # the renormalisation sums 24 terms and divides: the reference's  sum(toa_flux, dim=2)  may associate differently (<= 2 ulp)
    assert float(np.max(np.abs(got["toa"] - fx["out_toa_flux"]) / fx["out_toa_flux"])) <= 5e-16, label
    assert np.allclose(got["toa"].sum(axis=1), fx["in_tsi"][:, b], rtol=1e-14, atol=0), label  # what the block is for
    assert float(np.max(np.abs(got["mu0"] - fx["out_mu0"]))) <= 2e-16, label                       # libm / device cos
    assert np.array_equal(got["mu0"][fx["in_usecol"][:, b] == 0], np.ones(int((fx["in_usecol"][:, b] == 0).sum()))), label
    assert np.array_equal(got["alb"], fx["out_sfc_alb_spec"]), label
    assert np.array_equal(got["fu"], fx["out_flux_up"][:, :, b]) and np.array_equal(got["fd"], fx["out_flux_dn"][:, :, b]), label
    assert (fx["out_flux_up"][fx["in_usecol"][:, b] == 0, :, b] == 0).all() and (fx["in_usecol"][:, b] == 0).sum() >= 3  # a real test


def test_c_restatement_matches_the_reference_rfmip_sw_boundary_block():
    got, fx, b = _rfmip_sw_on_fixture(O.load_c(), frontend.NumpyArrays())
    _check_rfmip_sw(got, fx, b, "glue_oracle.c vs rrtmgp_rfmip_sw.F90")


@pytest.mark.gpu
def test_hip_glue_matches_the_reference_rfmip_sw_boundary_block():
    got, fx, b = _rfmip_sw_on_fixture(hiplib.load(), frontend.TorchArrays("cuda:0"))
    _check_rfmip_sw(got, fx, b, "csrc/glue.hip vs rrtmgp_rfmip_sw.F90")


@pytest.mark.gpu
def test_hip_glue_matches_oracle():
    import torch

    hip = hiplib.load()
    for seed, shape in ((5, {}), (6, dict(ncol=1000, nlay=33, ngas=8, nbnd=16, gpb=16))):
        d = _inputs(seed=seed, **shape)
        ref = run_glue(O.load_c(), frontend.NumpyArrays(), d)
        got = run_glue(hip, frontend.TorchArrays("cuda:0"), d)
        exact = {k: v for k, v in ref.items() if k in ("byband_up", "byband_dn", "byband_net", "byband_net_precalc", "col_dry",
                                                       "layer_mass", "col_gas", "tlev", "comb1_tau", "combn_tau", "combn_ssa",
                                                       "combn_p", "expand", "secants", "toa", "alb_spec", "masked_up", "masked_dn")}
        for k, v in exact.items():  # no transcendental function involved: bit-identical
            assert np.array_equal(got[k], v), k
        _assert_close(got, {k: v for k, v in ref.items() if k not in exact}, 1e-13, "HIP vs oracle")
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_device_resident_rfmip_like_driver():
    """A device-resident RFMIP-style LW driver that never touches the host between kernels: col_dry and col_gas from
    volume mixing ratios, tlev from tlay, gas optics, optimal transport angles feeding lw_Ds, by-band and net fluxes
    -- against the same sequence on the C oracle with host arrays (fluxes 1e-10)."""
    from rte_rrtmgp_amd import synth

    kd = synth.make_kdist("lw", ngpt=64, nbnd=4)
    ncol, nlay = 600, 20
    atm = synth.make_atmosphere(ncol, nlay, seed=4, kdist=kd)
    res = {}
    for name, lib, xp in (("oracle", O.load_c(), frontend.NumpyArrays()), ("hip", hiplib.load(), frontend.TorchArrays("cuda:0"))):
        A = xp.asarray
        E = lambda *a, lib=lib: hiplib.ext_call(lib, *a)  # noqa: E731
        go = frontend.GasOptics(lib, kd, xp)
        plev, play, tlay = A(atm.plev), A(atm.play), A(atm.tlay)
        vmr = A(atm.vmr)
        col_dry = xp.empty((ncol, nlay))
        E("rte_hip_get_layer_number", "iiaadda", ncol, nlay, A(np.asfortranarray(atm.vmr[:, :, kd.idx_h2o - 1])), plev, M_DRY, GRAV, col_dry)
        col_gas = xp.empty((ncol, nlay, kd.ngas + 1)); E("rte_hip_col_gas_fill", "iiiaaa", ncol, nlay, kd.ngas, vmr, col_dry, col_gas)
        tlev = xp.empty((ncol, nlay + 1)); E("rte_hip_tlev_interp", "iiaaaa", ncol, nlay, play, plev, tlay, tlev)
        b = go.gas_optics_lw(ncol, nlay, play, plev, tlay, A(atm.tsfc), col_gas, tlev, atm.top_at_1)
        Ds = xp.empty((ncol, kd.ngpt)); E("rte_hip_compute_optimal_angles", "iiiiaaaa", ncol, nlay, kd.ngpt, kd.nbnd, go.t["band_lims_gpt"], b["tau"], go.t["optimal_angle_fit"], Ds)
        emis_bnd = A(np.asfortranarray(np.full((kd.nbnd, ncol), 0.98)))
        emis = xp.empty((ncol, kd.ngpt)); E("rte_hip_expand_and_transpose", "iiiaaa", ncol, kd.nbnd, kd.ngpt, go.t["band_lims_gpt"], emis_bnd, emis)
        rb = frontend.rte_lw(lib, xp, ncol, nlay, kd.ngpt, atm.top_at_1, b["tau"], b["lay_src"], b["lev_src"], emis, b["sfc_src"],
                             lw_Ds=Ds, do_broadband=False)
        bu = xp.empty((ncol, nlay + 1, kd.nbnd)); lib.rte_sum_byband(ncol, nlay + 1, kd.ngpt, kd.nbnd, go.t["band_lims_gpt"], rb["gpt_flux_up"], bu)
        net = xp.empty((ncol, nlay + 1)); lib.rte_net_broadband_full(ncol, nlay + 1, kd.ngpt, rb["gpt_flux_dn"], rb["gpt_flux_up"], net)
        xp.sync()
        res[name] = {k: np.array(xp.to_numpy(v)) for k, v in dict(col_gas=col_gas, tlev=tlev, Ds=Ds, byband_up=bu, net=net).items()}
    for k in ("col_gas", "tlev"):
        assert np.array_equal(res["hip"][k], res["oracle"][k]), k
    for k in ("Ds", "byband_up", "net"):
        err = np.max(np.abs(res["hip"][k] - res["oracle"][k])) / np.max(np.abs(res["oracle"][k]))
        assert err <= 1e-10, (k, err)
    assert res["hip"]["Ds"].min() >= 1.0  # mo_rte_lw.F90:214-216
