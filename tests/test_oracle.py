"""CPU tests of the oracle (no GPU): the plain-C restatement against
  (1) the committed golden fixtures = outputs of the reference's own Fortran kernels, and
  (2) the reference build itself (oracle/_ref) when it is available (build container).
Tolerance: the restatement follows the reference expression by expression, so it is required to
agree to 1e-13 relative (in practice it is bit-identical)."""
import numpy as np
import pytest

import cases
import golden_util
from oracle import oracle as O
from rte_rrtmgp_amd import frontend

RTOL = 1e-13


@pytest.fixture(scope="module")
def clib():
    lib = O.load_c()
    lib.raw("rte_oracle_set_lw2str_bugcompat")(1)  # fixtures hold the reference default-kernel behaviour
    yield lib
    lib.raw("rte_oracle_set_lw2str_bugcompat")(0)


@pytest.mark.parametrize("name", list(cases.CASES))
def test_c_oracle_matches_golden(clib, name):
    case = cases.CASES[name]
    inp = cases.make_inputs(case)
    out = cases.run_suite(clib, frontend.NumpyArrays(), case, inp)
    golden_util.compare(name, out, inp, RTOL)


@pytest.mark.parametrize("name", ["lw_tiny_top1", "sw_tiny_sfc1", "lw_mid_ragged", "sw_mid_ragged"])
def test_c_oracle_matches_reference_build(clib, name):
    ref = O.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not available (no /root/reference, no prebuilt binary)")
    case = cases.CASES[name]
    inp = cases.make_inputs(case)
    xp = frontend.NumpyArrays()
    a = cases.run_suite(clib, xp, case, inp)
    b = O.big_stack(cases.run_suite, ref, xp, case, inp)
    for k in b:
        assert cases.rel_err(a[k], b[k]) <= RTOL, k


def test_tiny_fixture_inputs_are_self_contained():
    """The tiny fixtures carry their full inputs: regenerate-from-seed must reproduce them bit for bit."""
    z = golden_util.load("lw_tiny_sfc1")
    kd, atm, ex = cases.make_inputs(cases.CASES["lw_tiny_sfc1"])
    assert np.array_equal(z["in.kd.kmajor"], kd.arrays["kmajor"])
    assert np.array_equal(z["in.atm.play"], atm.play)
    assert np.array_equal(z["in.ex.ssa"], ex["ssa"])


def test_lw2stream_bugcompat_switch_matters():
    """Reference default-kernel quirk: g-point 1's level source used everywhere."""
    lib = O.load_c()
    case = cases.CASES["lw_tiny_sfc1"]
    inp = cases.make_inputs(case)
    xp = frontend.NumpyArrays()
    lib.raw("rte_oracle_set_lw2str_bugcompat")(0)
    good = cases.run_suite(lib, xp, case, inp)
    lib.raw("rte_oracle_set_lw2str_bugcompat")(1)
    quirk = cases.run_suite(lib, xp, case, inp)
    lib.raw("rte_oracle_set_lw2str_bugcompat")(0)
    assert cases.rel_err(good["lw2str.gpt_flux_up"], quirk["lw2str.gpt_flux_up"]) > 1e-3
    # g-point 1 itself is unaffected
    assert np.array_equal(good["lw2str.gpt_flux_up"][:, :, 0], quirk["lw2str.gpt_flux_up"][:, :, 0])
