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


def _run(lib, xp, kind, kd, atm, tb, cl, ncol, nlay, fuse=True):
    A = xp.asarray
    go, co = frontend.GasOptics(lib, kd, xp), frontend.CloudOptics(lib, tb, xp)
    a = {k: A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas", "col_dry")}
    a["top_at_1"] = atm.top_at_1
    c = {k: A(v) for k, v in cl.items()}
    if kind == "lw":
        gb, cb, rb = frontend.allsky_lw(lib, xp, go, co, ncol, nlay, a, c, xp.full((ncol, kd.ngpt), 0.98), fuse=fuse)
        keys = [("cld_tau", cb), ("tau", gb), ("flux_up", rb), ("flux_dn", rb)]
    else:
        mu0, alb = xp.full((ncol, nlay), 0.86), xp.full((ncol, kd.ngpt), 0.06)
        gb, cb, rb = frontend.allsky_sw(lib, xp, go, co, ncol, nlay, a, c, mu0, alb, fuse=fuse)
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


@pytest.mark.gpu
@pytest.mark.parametrize("kind,nlay,top_at_1", [("lw", 137, False), ("sw", 137, True), ("sw", 91, False)])
def test_hip_matches_oracle_at_host_model_layer_counts(kind, nlay, top_at_1):
    """The all-sky chain at the layer counts host models bring (91, 137 levels), 1 100 columns (the production gas-optics
    kernels): the wide solver paths -- two sub-segments per wave (LW), twelve layers per wave or the two-part solve (SW) --
    behind the same calls, against the oracle."""
    from oracle import oracle as O
    from rte_rrtmgp_amd import hiplib

    hip = hiplib.load()
    ncol = 1100
    kd, atm, tb, cl = _setup(kind, ncol, nlay, top_at_1)
    ref = O.big_stack(_run, O.load_c(), frontend.NumpyArrays(), kind, kd, atm, tb, cl, ncol, nlay)
    out = _run(hip, frontend.TorchArrays("cuda:0"), kind, kd, atm, tb, cl, ncol, nlay)
    # (SW with clouds: the two-part solve and the oracle's layer-by-layer recurrence round differently where the upward flux
    #  is a small difference of large terms -- a few 1e-11 of the largest flux; the full-size all-sky comparison asserts 1e-8)
    for k in ref:
        assert _rel(out[k], ref[k]) <= (1e-11 if kind == "lw" else 1e-9), k


@pytest.mark.gpu
@pytest.mark.parametrize("kind,ncol", [("lw", 70), ("sw", 70), ("lw", 1200), ("sw", 1200)])
def test_fused_extension_kernels_match_the_unfused_chain(kind, ncol):
    """The library's fused extension kernels (cloud optics in one pass; band-wise cloud increment inside
    compute_tau_absorption; Rayleigh + combine + increment in one pass) give exactly the arrays of the chain of
    reference-ABI kernels they replace -- same operations in the same order on the same doubles: bit-identical.
    70 columns run the direct kernels, 1200 the production ones."""
    from rte_rrtmgp_amd import hiplib

    hip = hiplib.load()
    nlay = 24
    kd, atm, tb, cl = _setup(kind, ncol, nlay)
    xp = frontend.TorchArrays("cuda:0")
    fused = _run(hip, xp, kind, kd, atm, tb, cl, ncol, nlay, fuse=True)
    unfused = _run(hip, xp, kind, kd, atm, tb, cl, ncol, nlay, fuse=False)
    for k in unfused:
        assert np.array_equal(fused[k], unfused[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("ncol", [70, 1200, 5003])
def test_one_pass_sw_gas_optics_matches_the_chain(ncol):
    """``fuse="all"`` (rte_hip_gas_optics_sw_2str): compute_tau_absorption + compute_tau_rayleigh + combine + the band-wise
    cloud increment in one pass.  The same operations on the same doubles as the chain of kernels, so the arrays are
    bit-identical wherever both run the slab kernel; the Rayleigh rows make the staged box a little larger, so a few
    (tile, layer, band) entries more go to the direct-gather code, whose sums of the same terms differ by rounding:
    1e-14 elementwise, and most values exactly equal.  Clear sky (the SW bench chain) and with clouds."""
    import torch

    from rte_rrtmgp_amd import hiplib

    hip = hiplib.load()
    nlay = 24
    kd, atm, tb, cl = _setup("sw", ncol, nlay)
    xp = frontend.TorchArrays("cuda:0")
    chain = _run(hip, xp, "sw", kd, atm, tb, cl, ncol, nlay, fuse=True)
    one = _run(hip, xp, "sw", kd, atm, tb, cl, ncol, nlay, fuse="all")
    for k in chain:
        err = np.max(np.abs(one[k] - chain[k]) / np.maximum(np.abs(chain[k]), 1e-300))
        assert err <= 1e-14, (k, err)
        assert np.mean(one[k] == chain[k]) > 0.98, k
    # clear sky: GasOptics alone
    go = frontend.GasOptics(hip, kd, xp)
    A = xp.asarray
    args = [A(getattr(atm, k)) for k in ("play", "plev", "tlay", "col_gas", "col_dry")]
    a = {k: v.clone() for k, v in go.gas_optics_sw(ncol, nlay, *args, fuse_rayleigh=True).items() if k in ("tau", "ssa", "g")}
    b = go.gas_optics_sw(ncol, nlay, *args, fuse_rayleigh="all")
    torch.cuda.synchronize()
    for k in a:
        err = float(((b[k] - a[k]).abs() / a[k].abs().clamp_min(1e-300)).max())
        assert err <= 1e-14, (k, err)
        assert float((b[k] == a[k]).double().mean()) > 0.98, k


def _golden():
    import importlib.util
    import os

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_allsky_golden", os.path.join(here, "make_allsky_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, np.load(os.path.join(here, "allsky_72.npz"))


@pytest.mark.parametrize("kind", ["lw", "sw"])
def test_oracle_matches_allsky_golden(kind):
    """The C oracle on the all-sky chain at the real shape of BASELINE configs[3] (72 layers, g256 / g224 tables)
    against the committed fixture, which holds the REFERENCE kernels' results (tests/golden/make_allsky_golden.py)."""
    from oracle import oracle as O

    mk, z = _golden()
    kd, atm, tb, cl = mk.setup(kind)
    assert str(z[f"{kind}.__digest__"]) == mk.digest(kd, atm, tb, cl), "fixture was made from different inputs"
    got = mk.run(O.load_c(), frontend.NumpyArrays(), kind, kd, atm, tb, cl, mk.NCOL)
    for k, v in got.items():
        ref = z[f"{kind}.{k}|full"] if v.ndim == 2 else z[f"{kind}.{k}|sample"]
        mine = v if v.ndim == 2 else v.ravel(order="F")[::mk.SAMPLE]
        assert _rel(mine, ref) <= 1e-12, k


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["lw", "sw"])
def test_hip_matches_allsky_golden(kind):
    """The HIP chain against the same fixture.  The fixture's 24 columns are tiled 30 times (720 columns: columns are
    independent, so every copy must reproduce the golden column) -- that puts the call on the PRODUCTION kernels: slab
    gas optics with g256 / g224 tables and the 9-layers-per-wave segmented solvers that `bench.py --workload allsky`
    runs at 72 layers."""
    from rte_rrtmgp_amd import hiplib

    mk, z = _golden()
    rep = 30
    kd, atm, tb, cl = mk.setup(kind)
    assert str(z[f"{kind}.__digest__"]) == mk.digest(kd, atm, tb, cl)
    tile = lambda a: np.asfortranarray(np.concatenate([a] * rep, axis=0))  # noqa: E731
    for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas", "col_dry", "vmr"):
        setattr(atm, k, tile(getattr(atm, k)))
    atm.ncol = mk.NCOL * rep
    cl = {k: tile(v) for k, v in cl.items()}
    out = mk.run(hiplib.load(), frontend.TorchArrays("cuda:0"), kind, kd, atm, tb, cl, mk.NCOL * rep)
    for k, v in out.items():
        if v.ndim == 2:  # broadband fluxes: every copy of the 24 columns against the golden values
            ref = z[f"{kind}.{k}|full"]
            for r in range(rep):
                blk = v[r * mk.NCOL:(r + 1) * mk.NCOL]
                assert _rel(blk, ref) <= 1e-10, (k, r)
                assert np.max(np.abs(blk - ref) / np.maximum(np.abs(ref), 1e-4 * np.abs(ref).max())) <= 1e-8, (k, r)
        else:  # optical properties: the first copy at the fixture's sample points
            first = v[:mk.NCOL].ravel(order="F")[::mk.SAMPLE]
            assert _rel(first, z[f"{kind}.{k}|sample"]) <= 1e-12, k
            assert np.array_equal(v[:mk.NCOL], v[mk.NCOL:2 * mk.NCOL]), k  # copies are bit-identical
