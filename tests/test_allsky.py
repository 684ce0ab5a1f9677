"""All-sky assembly (examples/all-sky/rrtmgp_allsky.F90:336-404): cloud optics from
tables, liquid + ice combination, delta scaling, band-wise increment of the gas optical properties, solvers.
CPU: the C oracle against the reference's own kernels through the same host mirror.  GPU: the HIP library
against the oracle."""
import numpy as np
import pytest

from rte_rrtmgp_amd import frontend, synth


def _setup(kind, ncol, nlay, top_at_1=False):
    kd = synth.make_kdist(kind, ngpt=64, nbnd=4)
    atm = synth.make_atmosphere(ncol, nlay, seed=21, kdist=kd, top_at_1=top_at_1)
    tb = synth.make_cloud_optics(kd.nbnd)
    cl = synth.make_cloud_field(atm, tb)
    assert (cl["lwp"] > 0).any() and (cl["iwp"] > 0).any() and ((cl["lwp"] > 0) & (cl["iwp"] > 0)).any()
    return kd, atm, tb, cl


def _run(lib, xp, kind, kd, atm, tb, cl, ncol, nlay):
    A = xp.asarray
    go, co = frontend.GasOptics(lib, kd, xp), frontend.CloudOptics(lib, tb, xp)
    a = {k: A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas", "col_dry")}
    a["top_at_1"] = atm.top_at_1
    c = {k: A(v) for k, v in cl.items()}
    if kind == "lw":
        gb, cb, rb = frontend.allsky_lw(lib, xp, go, co, ncol, nlay, a, c, xp.full((ncol, kd.ngpt), 0.98))
        keys = [("cld_tau", cb), ("tau", gb), ("flux_up", rb), ("flux_dn", rb)]
    else:
        mu0, alb = xp.full((ncol, nlay), 0.86), xp.full((ncol, kd.ngpt), 0.06)
        gb, cb, rb = frontend.allsky_sw(lib, xp, go, co, ncol, nlay, a, c, mu0, alb)
        keys = [("cld_tau", cb), ("cld_ssa", cb), ("cld_g", cb), ("tau", gb), ("ssa", gb), ("g", gb), ("flux_up", rb),
                ("flux_dn", rb), ("flux_dir", rb)]
    return {k: np.array(xp.to_numpy(d[k])) for k, d in keys}


def _rel(a, b):
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den else 1.0))


@pytest.mark.parametrize("kind", ["lw", "sw"])
def test_oracle_matches_reference_kernels(kind):
    from oracle import oracle as O

    try:
        ref_lib = O.load_ref()
    except Exception:
        pytest.skip("oracle/_ref (reference build) not available")
    ncol, nlay = 12, 20
    kd, atm, tb, cl = _setup(kind, ncol, nlay)
    xp = frontend.NumpyArrays()
    got = _run(O.load_c(), xp, kind, kd, atm, tb, cl, ncol, nlay)
    ref = O.big_stack(_run, ref_lib, xp, kind, kd, atm, tb, cl, ncol, nlay)
    for k in ref:
        assert _rel(got[k], ref[k]) <= 1e-13, k
    assert got["flux_up"].max() > 0 and np.isfinite(got["flux_dn"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,top_at_1", [("lw", False), ("sw", False), ("sw", True)])
def test_hip_matches_oracle(kind, top_at_1):
    from oracle import oracle as O
    from rte_rrtmgp_amd import hiplib

    hip = hiplib.load()
    ncol, nlay = 70, 24
    kd, atm, tb, cl = _setup(kind, ncol, nlay, top_at_1)
    ref = _run(O.load_c(), frontend.NumpyArrays(), kind, kd, atm, tb, cl, ncol, nlay)
    out = _run(hip, frontend.TorchArrays("cuda:0"), kind, kd, atm, tb, cl, ncol, nlay)
    for k in ref:
        assert _rel(out[k], ref[k]) <= 1e-12, k
