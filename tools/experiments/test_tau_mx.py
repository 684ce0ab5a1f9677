"""GPU tests of the matrix-core form of compute_tau_absorption (rte_hip_tau_variant(10), csrc/tau_mx.h): columns sorted by
LUT key per (tile, layer, flavor), the gathers as v_mfma_f64_16x16x4_f64 products.  Opt-in (the specialised-wave kernel stays
the default: DESIGN.md section 4.2b has the measurements); it must give the oracle's optical depths on every table shape the
production path accepts -- ragged last tiles, both orientations, accumulate and overwrite, more than four minor intervals
per band (sub-stages), bands wider than a stage, unordered site-like columns (many keys per tile), the by-band operand."""
import numpy as np
import pytest

import cases
from rte_rrtmgp_amd import frontend, hiplib, synth

pytestmark = pytest.mark.gpu
RTOL_GAS = 1e-12


@pytest.fixture(scope="module")
def hip():
    return hiplib.load()


@pytest.fixture(scope="module")
def oracle_c():
    from oracle import oracle as O

    return O.load_c()


def _tau(lib, xp, kd, atm, ncol, nlay, start, variant=None, defer=False, bybnd=None):
    A = xp.asarray
    go = frontend.GasOptics(lib, kd, xp)
    play, tlay, col_gas = A(atm.play), A(atm.tlay), A(atm.col_gas)
    st = go.interpolation(ncol, nlay, play, tlay, col_gas)
    if variant is not None:
        hiplib.ext_call(lib, "rte_hip_tau_variant", ["i"], variant)
        hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 1 if defer else 0)
    try:
        tau = xp.full((ncol, nlay, kd.ngpt), start)
        if defer:
            lib.zero_array_3D(ncol, nlay, kd.ngpt, tau)
        go.compute_tau_absorption(ncol, nlay, st, play, tlay, col_gas, tau, tau_bybnd=(A(bybnd) if bybnd is not None else None))
        return np.array(xp.to_numpy(tau))
    finally:
        if variant is not None:
            hiplib.ext_call(lib, "rte_hip_tau_variant", ["i"], 9)
            hiplib.ext_call(lib, "rte_hip_defer_zero", ["i"], 0)


CASES = [
    ("4 bands of 16, uneven minors", dict(ngpt=64, nbnd=4, nminor_lower=11, nminor_upper=7), 1100, 24, dict(seed=77)),
    ("the same, top_at_1", dict(ngpt=64, nbnd=4, nminor_lower=11, nminor_upper=7), 1100, 24, dict(seed=78, top_at_1=True)),
    ("g256", dict(), 1300, 60, dict(seed=5)),
    ("g256, ragged minors (sub-stages)", dict(minor_distribution="ragged"), 1100, 24, dict(seed=8)),
    ("g256 ragged, top_at_1, site-like columns", dict(minor_distribution="ragged"), 1500, 30, dict(seed=9, top_at_1=True, climate="sites")),
    ("bands of 32 g-points", dict(ngpt=64, nbnd=2), 700, 19, dict(seed=3)),
    ("exactly one tile", dict(ngpt=64, nbnd=4), 512, 12, dict(seed=4)),
]


@pytest.mark.parametrize("label,kdargs,ncol,nlay,atmargs", CASES, ids=[c[0] for c in CASES])
def test_matrix_core_tau_matches_the_oracle(hip, oracle_c, label, kdargs, ncol, nlay, atmargs):
    kd = synth.make_kdist("lw", **kdargs)
    atm = synth.make_atmosphere(ncol, nlay, kdist=kd, **atmargs)
    xp = frontend.TorchArrays("cuda:0")
    ref = _tau(oracle_c, frontend.NumpyArrays(), kd, atm, ncol, nlay, 0.125)
    acc = _tau(hip, xp, kd, atm, ncol, nlay, 0.125, variant=10)              # accumulate onto a non-zero tau
    assert cases.rel_err(acc, ref) <= RTOL_GAS, label
    v9 = _tau(hip, xp, kd, atm, ncol, nlay, 7.0, variant=9, defer=True)
    assert hiplib.ext_call(hip, "rte_hip_stat", ["i"], 3) == 9
    ovw = _tau(hip, xp, kd, atm, ncol, nlay, 7.0, variant=10, defer=True)    # recorded zero fill: tau is overwritten
    assert hiplib.ext_call(hip, "rte_hip_stat", ["i"], 3) == 10, "the matrix-core kernel did not run"
    assert cases.rel_err(ovw, ref - 0.125) <= RTOL_GAS, label
    assert cases.elem_err(ovw, v9, 1e-8) <= 1e-11, label


def test_matrix_core_tau_with_the_byband_operand(hip, oracle_c):
    """rte_hip_compute_tau_absorption_inc_bybnd (the all-sky LW increment folded into the kernel) on the matrix-core path."""
    kd = synth.make_kdist("lw")
    ncol, nlay = 900, 20
    atm = synth.make_atmosphere(ncol, nlay, seed=21, kdist=kd)
    rng = np.random.default_rng(3)
    bybnd = np.asfortranarray(rng.uniform(0.0, 2.0, size=(ncol, nlay, kd.nbnd)))
    xp = frontend.TorchArrays("cuda:0")
    mx = _tau(hip, xp, kd, atm, ncol, nlay, 0.0, variant=10, defer=True, bybnd=bybnd)
    v9 = _tau(hip, xp, kd, atm, ncol, nlay, 0.0, variant=9, defer=True, bybnd=bybnd)
    ref = _tau(oracle_c, frontend.NumpyArrays(), kd, atm, ncol, nlay, 0.0)
    ref = ref + np.repeat(bybnd, kd.ngpt // kd.nbnd, axis=2)
    assert cases.rel_err(v9, ref) <= RTOL_GAS
    assert cases.rel_err(mx, ref) <= RTOL_GAS
