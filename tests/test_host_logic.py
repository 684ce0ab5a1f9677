"""CPU tests of the host-side logic: synthetic-table consistency rules, the
frontend mirror's buffer/decoy handling and analytic known-answer tests restated from the
reference's own data-free unit tests, run through the C oracle."""
import numpy as np
import pytest

import cases
from oracle import oracle as O
from rte_rrtmgp_amd import frontend, synth


@pytest.mark.parametrize("kind", ["lw", "sw"])
def test_synthetic_kdist_obeys_consistency_rules(kind):
    kd = synth.make_kdist(kind)
    a = kd.arrays
    bl = a["band_lims_gpt"]
    assert bl[0, 0] == 1 and bl[1, -1] == kd.ngpt and np.all(bl[0, 1:] == bl[1, :-1] + 1)
    gf = a["gpoint_flavor"]
    for b in range(kd.nbnd):  # flavor constant within a band (mo_gas_optics_rrtmgp_kernels.F90:384)
        s, e = bl[0, b] - 1, bl[1, b]
        assert np.all(gf[:, s:e] == gf[:, s:s + 1])
    assert a["flavor"].min() >= 0 and a["flavor"].max() <= kd.ngas
    assert not np.any((a["flavor"][0] == 0) & (a["flavor"][1] == 0))
    assert np.all(np.diff(a["press_ref"]) < 0) and np.all(np.diff(a["temp_ref"]) > 0)
    assert kd.press_ref_log_delta < 0
    for reg in ("lower", "upper"):
        lims, start = a[f"minor_limits_gpt_{reg}"], a[f"kminor_start_{reg}"]
        n = lims.shape[1]
        widths = lims[1] - lims[0] + 1
        assert np.all(start == 1 + np.concatenate([[0], np.cumsum(widths)[:-1]]))
        assert a[f"kminor_{reg}"].shape == (kd.ntemp, kd.neta, widths.sum())
        assert a[f"idx_minor_{reg}"].min() >= 1 and a[f"idx_minor_{reg}"].max() <= kd.ngas
        assert n == a[f"scale_by_complement_{reg}"].size
    if kind == "lw":
        pf = a["planck_frac"]
        for b in range(kd.nbnd):
            s, e = bl[0, b] - 1, bl[1, b]
            assert np.allclose(pf[..., s:e].sum(-1), 1.0)
        assert np.all(np.diff(a["totplnk"], axis=0) > 0)


def test_atmosphere_is_inside_the_tables():
    kd = synth.make_kdist("lw")
    for top in (False, True):
        atm = synth.make_atmosphere(40, 60, seed=1, top_at_1=top, kdist=kd)
        assert atm.play.min() > kd.press_ref_min and atm.play.max() < kd.press_ref_max
        for t in (atm.tlay, atm.tlev, atm.tsfc):
            assert t.min() > kd.temp_ref_min and t.max() < kd.temp_ref_max
        d = np.diff(atm.play, axis=1)
        assert np.all(d > 0) if top else np.all(d < 0)


def _gray_equilibrium(lib, xp, ncol=8, nlay=16, top_at_1=True):
    """Gray radiative equilibrium known answer, restating reference
    tests/rte_lw_solver_unit_tests.F90:241-343: OLR = 2 sigma T^4 / (2 + D tau)."""
    sigma, D = 5.670374419e-8, 1.66
    total_tau = np.array([0.1, 1.0, 10.0, 50.0] * 2)[:ncol]
    sfc_t = np.array([285.0] * 4 + [310.0] * 4)[:ncol]
    olr = (2.0 * sigma * sfc_t ** 4) / (2.0 + D * total_tau)
    tau = np.repeat((total_tau / nlay)[:, None], nlay, axis=1)
    edges = np.concatenate([np.zeros((ncol, 1)), np.cumsum(tau, axis=1)], axis=1)  # tau from the top
    lev = 0.5 / np.pi * olr[:, None] * (1.0 + D * edges)
    lay = 0.5 * (lev[:, :-1] + lev[:, 1:])
    if not top_at_1:
        tau, lev, lay = tau[:, ::-1], lev[:, ::-1], lay[:, ::-1]
    A = xp.asarray
    r = frontend.rte_lw(lib, xp, ncol, nlay, 1, top_at_1, A(tau.reshape(ncol, nlay, 1, order="F")),
                        A(lay.reshape(ncol, nlay, 1, order="F")), A(lev.reshape(ncol, nlay + 1, 1, order="F")),
                        xp.full((ncol, 1), 1.0), A((sigma / np.pi * sfc_t ** 4).reshape(ncol, 1)),
                        lw_Ds=xp.full((ncol, 1), D))
    up, dn = xp.to_numpy(r["flux_up"]), xp.to_numpy(r["flux_dn"])
    return olr, up, dn


@pytest.mark.parametrize("top_at_1", [True, False])
def test_gray_radiative_equilibrium_oracle(top_at_1):
    olr, up, dn = _gray_equilibrium(O.load_c(), frontend.NumpyArrays(), top_at_1=top_at_1)
    toa = 0 if top_at_1 else -1
    assert np.allclose(up[:, toa], olr, rtol=2e-4)  # discretisation error of the 16-layer problem
    net = up - dn
    assert np.allclose(net, net[:, :1], rtol=2e-4)  # net flux constant with height


def test_beer_lambert_direct_beam_oracle():
    """Thin-atmosphere direct beam = Beer-Lambert (reference tests/rte_sw_solver_unit_tests.F90:123-133)."""
    lib, xp = O.load_c(), frontend.NumpyArrays()
    ncol, nlay = 8, 16
    total_tau = np.array([1e-4, 1e-2] * 4)
    tau = np.repeat((total_tau / nlay)[:, None], nlay, axis=1).reshape(ncol, nlay, 1, order="F")
    mu0 = np.where(np.arange(ncol) < 4, 1.0, 0.5)
    mu0l = np.repeat(mu0[:, None], nlay, axis=1)
    toa = np.full((ncol, 1), 1360.0)
    r = frontend.rte_sw(lib, xp, ncol, nlay, 1, True, xp.asarray(tau), None, None, xp.asarray(mu0l),
                        xp.asarray(toa), None, None, noscat=True)
    fdir = r["flux_dir"]
    assert np.allclose(fdir[:, -1], 1360.0 * mu0 * np.exp(-total_tau / mu0), rtol=1e-13)


def test_decoys_are_never_written():
    """Broadband mode must not touch the spectral flux arguments (they are 1-element decoys here)."""
    lib, xp = O.load_c(), frontend.NumpyArrays()
    case = cases.CASES["lw_tiny_sfc1"]
    out = cases.run_suite(lib, xp, case, which="core")
    assert np.isfinite(out["lw1.flux_up"]).all()
