"""GPU parity tests (run with -m gpu on an MI355X): every kernel of the hot path, called through
the C ABI of librte_rrtmgp_hip.so, against the CPU oracle on the same seeded inputs and against
the committed golden fixtures (outputs of the reference's own Fortran kernels).

Tolerances (relative to the array's max magnitude):
  * integer / logical outputs (jtemp, jpress, jeta, tropo): bit-exact;
  * floating point: the contract from BASELINE.json's north_star is 1e-6 relative on fluxes;
    what is asserted here is much tighter -- RTOL_GAS for the gas-optics arrays (no
    transcendental except one log) and RTOL_FLUX for solver outputs (device exp vs libm exp,
    different summation order of the broadband reduction).
"""
import numpy as np
import pytest

import cases
import golden_util
from rte_rrtmgp_amd import frontend, hiplib

pytestmark = pytest.mark.gpu

RTOL_GAS = 1e-12
RTOL_FLUX = 1e-10
NORTH_STAR_RTOL = 1e-6


def _tol(name):
    return RTOL_GAS if name.startswith(("interp.", "tau", "lay_src", "lev_src", "sfc_src", "ssa", "g", "toa")) else RTOL_FLUX


@pytest.fixture(scope="module")
def hip():
    lib = hiplib.load()  # raises if the extension is missing: no fallback
    assert hiplib.ext_call(lib, "rte_hip_device_count", []) >= 1
    return lib


@pytest.fixture(scope="module")
def oracle_c():
    from oracle import oracle as O

    return O.load_c()


# elementwise metric (cases.elem_err): per-element relative error, absolute floor per g-point plane.  Fluxes are sums
# and differences of terms of either sign, so the floor is higher there (1e-4 of the plane's maximum).
ETOL_GAS, ETOL_FLUX = 1e-11, 1e-8


def _check(out, ref, label):
    assert set(ref) <= set(out)
    worst = (0.0, None)
    for k in ref:
        e = cases.rel_err(out[k], ref[k])
        assert e <= _tol(k), f"{label}: {k} rel err {e:.3e} > {_tol(k):.1e}"
        gas = _tol(k) == RTOL_GAS
        ee = cases.elem_err(out[k], ref[k], 1e-8 if gas else 1e-4)
        assert ee <= (ETOL_GAS if gas else ETOL_FLUX), f"{label}: {k} elementwise rel err {ee:.3e}"
        assert np.isfinite(np.asarray(out[k], dtype=np.float64)).all(), k
        if e > worst[0]:
            worst = (e, k)
    return worst


@pytest.mark.parametrize("name", list(cases.CASES))
def test_device_pointers_match_oracle(hip, oracle_c, name):
    case = cases.CASES[name]
    inp = cases.make_inputs(case)
    ref = cases.run_suite(oracle_c, frontend.NumpyArrays(), case, inp)
    out = cases.run_suite(hip, frontend.TorchArrays("cuda:0"), case, inp)
    worst = _check(out, ref, name)
    print(f"{name}: worst {worst}")


@pytest.mark.parametrize("name", ["lw_tiny_top1", "sw_tiny_sfc1", "lw_mid_ragged"])
def test_host_pointers_are_staged(hip, oracle_c, name):
    """What the unchanged Fortran frontend does: host arrays in, host arrays out."""
    case = cases.CASES[name]
    inp = cases.make_inputs(case)
    ref = cases.run_suite(oracle_c, frontend.NumpyArrays(), case, inp)
    out = cases.run_suite(hip, frontend.NumpyArrays(), case, inp)
    _check(out, ref, name + "(host)")


@pytest.mark.parametrize("name", list(cases.CASES))
def test_matches_golden_fixtures(hip, name):
    """Against outputs of the reference's own kernels (bug-compatible lw_2stream switch on)."""
    case = cases.CASES[name]
    inp = cases.make_inputs(case)
    hiplib.ext_call(hip, "rte_hip_set_lw2str_bugcompat", ["i"], 1)
    try:
        out = cases.run_suite(hip, frontend.TorchArrays("cuda:0"), case, inp)
    finally:
        hiplib.ext_call(hip, "rte_hip_set_lw2str_bugcompat", ["i"], 0)
    worst = golden_util.compare(name, out, inp, RTOL_FLUX)
    assert worst[0] <= NORTH_STAR_RTOL


@pytest.mark.parametrize("name", ["sw_tiny_sfc1", "sw_mid_ragged", "sw_g224"])
def test_segmented_and_generic_sw_solvers_agree(hip, oracle_c, name):
    """The production (segmented, composite-chained) and the generic SW two-stream kernels are two
    implementations of one recurrence: they must agree to rounding -- also with night-time columns
    (mu0 <= 0), a diffuse top boundary condition and strongly scattering layers."""
    import numpy as np
    case = cases.CASES[name]
    inp = cases.make_inputs(case)
    xp = frontend.TorchArrays("cuda:0")
    a = cases.run_suite(hip, xp, case, inp)
    hiplib.ext_call(hip, "rte_hip_force_generic_sw", ["i"], 1)
    try:
        b = cases.run_suite(hip, xp, case, inp)
    finally:
        hiplib.ext_call(hip, "rte_hip_force_generic_sw", ["i"], 0)
    for k in ("sw.flux_up", "sw.flux_dn", "sw.flux_dir", "swc.flux_up", "swc.flux_dn", "swc.flux_dir"):
        assert cases.rel_err(a[k], b[k]) <= 1e-12, k
    # sws: spectral output from the segmented kernel.  Single g-points of clear-sky gas optics: optically very thin
    # layers make 1 - exp(-2 k tau) lose digits (inherent in the reference's formulas, :1028-1031), and the two kernels'
    # exponentials differ by an ulp -- a few 1e-11 of the largest flux, where the band sums above agree to 1e-12
    for k in ("sws.gpt_flux_up", "sws.gpt_flux_dn", "sws.gpt_flux_dir"):
        assert cases.rel_err(a[k], b[k]) <= 1e-10, k
    # direct call: night columns + diffuse boundary condition, broadband, against the oracle
    rng = np.random.default_rng(11)
    F = lambda *sh: np.asfortranarray(rng.random(sh))
    A = xp.asarray
    # 72 / 75 / 88 / 91 / 96: nine ... twelve layers per wave (91 and 137 levels are what host models bring)
    # 97 ... 704: sw_solver_2stream as 2 ... 8 windows of layers on the segmented kernel (every window but the first twice)
    for nlay, top_at_1 in ((27, False), (27, True), (72, False), (75, True), (81, True), (88, False), (91, True), (96, False), (100, False),
                           (137, True), (185, False), (192, True), (200, False), (300, True), (445, False), (704, True)):
        ncol, ngpt = 70, 16
        tau, ssa, g = F(ncol, nlay, ngpt) * 3.0, F(ncol, nlay, ngpt) * 0.999, F(ncol, nlay, ngpt) * 0.9 - 0.1
        mu0 = np.asfortranarray(np.repeat((rng.random(ncol) * 1.2 - 0.2)[:, None], nlay, axis=1))  # some <= 0
        adir, adif, idir, idif = F(ncol, ngpt), F(ncol, ngpt), F(ncol, ngpt) * 100, F(ncol, ngpt) * 10
        ref = frontend.rte_sw(oracle_c, frontend.NumpyArrays(), ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, idir, adir, adif,
                              inc_flux_dif=idif)
        out = frontend.rte_sw(hip, xp, ncol, nlay, ngpt, top_at_1, A(tau), A(ssa), A(g), A(mu0), A(idir), A(adir), A(adif),
                              inc_flux_dif=A(idif))
        for k in ("flux_up", "flux_dn", "flux_dir"):
            assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-12, (k, nlay, top_at_1)
        # the same with spectral output: the segmented kernel stores the fluxes of the levels each wave owns
        ref = frontend.rte_sw(oracle_c, frontend.NumpyArrays(), ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, idir, adir, adif,
                              inc_flux_dif=idif, do_broadband=False)
        out = frontend.rte_sw(hip, xp, ncol, nlay, ngpt, top_at_1, A(tau), A(ssa), A(g), A(mu0), A(idir), A(adir), A(adif),
                              inc_flux_dif=A(idif), do_broadband=False)
        for k in ("gpt_flux_up", "gpt_flux_dn", "gpt_flux_dir"):
            assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-11, (k, nlay, top_at_1)
        # the LW two-stream solver on the same optical properties (segmented kernel) against the oracle
        lay, lev = F(ncol, nlay, ngpt) * 10 + 1, F(ncol, nlay + 1, ngpt) * 10 + 1
        emis, sfc, inc = F(ncol, ngpt) * 0.2 + 0.8, F(ncol, ngpt) * 10, F(ncol, ngpt)
        ref = frontend.rte_lw(oracle_c, frontend.NumpyArrays(), ncol, nlay, ngpt, top_at_1, tau, lay, lev, emis, sfc, ssa=ssa, g=g,
                              use_2stream=True, inc_flux=inc)
        out = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, top_at_1, A(tau), A(lay), A(lev), A(emis), A(sfc), ssa=A(ssa), g=A(g),
                              use_2stream=True, inc_flux=A(inc))
        for k in ("gpt_flux_up", "gpt_flux_dn"):
            assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-12, (k, nlay, top_at_1)
        # the no-scattering solver with Tang rescaling (what rte_lw does with 2-stream cloudy optical properties
        # by default), broadband, two quadrature angles, with Jacobian: segmented three-sweep kernel vs oracle
        sj = F(ncol, ngpt)
        ref = frontend.rte_lw(oracle_c, frontend.NumpyArrays(), ncol, nlay, ngpt, top_at_1, tau, lay, lev, emis, sfc, ssa=ssa, g=g,
                              inc_flux=inc, n_gauss_angles=2, sfc_src_jac=sj, do_jacobians=True)
        out = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, top_at_1, A(tau), A(lay), A(lev), A(emis), A(sfc), ssa=A(ssa), g=A(g),
                              inc_flux=A(inc), n_gauss_angles=2, sfc_src_jac=A(sj), do_jacobians=True)
        for k in ("flux_up", "flux_dn", "flux_up_jac"):
            assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-12, (k, nlay, top_at_1)
        # the no-scattering solver with spectral output, three angles + Jacobian (segmented kernel, angles accumulated in
        # the caller's arrays) vs oracle
        ref = frontend.rte_lw(oracle_c, frontend.NumpyArrays(), ncol, nlay, ngpt, top_at_1, tau, lay, lev, emis, sfc, inc_flux=inc,
                              n_gauss_angles=3, sfc_src_jac=sj, do_jacobians=True, do_broadband=False)
        out = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, top_at_1, A(tau), A(lay), A(lev), A(emis), A(sfc), inc_flux=A(inc),
                              n_gauss_angles=3, sfc_src_jac=A(sj), do_jacobians=True, do_broadband=False)
        for k in ("gpt_flux_up", "gpt_flux_dn", "flux_up_jac"):
            assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-12, (k, nlay, top_at_1)


@pytest.mark.parametrize("nlay,top_at_1", [(81, True), (91, False), (128, True), (137, False), (144, True), (160, False), (170, True),
                                           (176, False)])
def test_lw_noscat_with_more_than_80_layers(hip, oracle_c, nlay, top_at_1):
    """Host models at 91 / 128 / 137 levels: the reference has no layer limit (rte/kernels/mo_rte_solver_kernels.F90:697-743).
    81 ... 176 layers run on the two-sub-segment kernel (8 waves x 2 x 8 ... 11 layers): broadband with three angles,
    incident flux and Jacobian, and spectral output, against the oracle; partial last waves (81, 91, 137) included."""
    import numpy as np

    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    rng = np.random.default_rng(nlay)
    F = lambda *sh: np.asfortranarray(rng.random(sh))
    ncol, ngpt = 70, 16
    tau = F(ncol, nlay, ngpt) * 2.0
    lay, lev = F(ncol, nlay, ngpt) * 10 + 1, F(ncol, nlay + 1, ngpt) * 10 + 1
    emis, sfc, inc, sj = F(ncol, ngpt) * 0.2 + 0.8, F(ncol, ngpt) * 10, F(ncol, ngpt), F(ncol, ngpt)
    for kw, keys in ((dict(n_gauss_angles=3, do_jacobians=True), ("flux_up", "flux_dn", "flux_up_jac")),
                     (dict(n_gauss_angles=2, do_broadband=False), ("gpt_flux_up", "gpt_flux_dn")),
                     (dict(), ("flux_up", "flux_dn"))):
        ref = frontend.rte_lw(oracle_c, frontend.NumpyArrays(), ncol, nlay, ngpt, top_at_1, tau, lay, lev, emis, sfc, inc_flux=inc,
                              sfc_src_jac=sj, **kw)
        out = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, top_at_1, A(tau), A(lay), A(lev), A(emis), A(sfc), inc_flux=A(inc),
                              sfc_src_jac=A(sj), **kw)
        for k in keys:
            assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-12, (k, nlay, top_at_1, kw)


@pytest.mark.gpu
@pytest.mark.parametrize("nlay,top_at_1", [(177, True), (200, False), (256, True), (257, False), (300, True), (352, False), (400, True),
                                           (513, False), (1024, True)])
def test_lw_noscat_with_177_to_1024_layers(hip, oracle_c, nlay, top_at_1):
    """Columns of 177 ... 1024 layers: ceil(nlay / 128) windows of layers on the two-sub-segment kernel (top to bottom for the
    downward radiance at the interfaces, the last window with the surface, then bottom to top over the upward radiance of the
    window below).  Broadband with three angles, incident flux and Jacobian, and with one angle, against the oracle; 2 ... 8
    windows, uneven splits, both orientations.  The generic kernel would pass too: the kernel that ran is checked by name."""
    import numpy as np

    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    rng = np.random.default_rng(nlay)
    F = lambda *sh: np.asfortranarray(rng.random(sh))
    ncol, ngpt = 70, 16
    tau = F(ncol, nlay, ngpt) * (20.0 / nlay)
    lay, lev = F(ncol, nlay, ngpt) * 10 + 1, F(ncol, nlay + 1, ngpt) * 10 + 1
    emis, sfc, inc, sj = F(ncol, ngpt) * 0.2 + 0.8, F(ncol, ngpt) * 10, F(ncol, ngpt), F(ncol, ngpt)
    hiplib.ext_call(hip, "rte_hip_profile_reset", [])
    hiplib.ext_call(hip, "rte_hip_profile_enable", ["i"], 1)
    try:
        for kw, keys in ((dict(n_gauss_angles=3, do_jacobians=True), ("flux_up", "flux_dn", "flux_up_jac")),
                         (dict(), ("flux_up", "flux_dn"))):
            ref = frontend.rte_lw(oracle_c, frontend.NumpyArrays(), ncol, nlay, ngpt, top_at_1, tau, lay, lev, emis, sfc, inc_flux=inc,
                                  sfc_src_jac=sj, **kw)
            out = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, top_at_1, A(tau), A(lay), A(lev), A(emis), A(sfc), inc_flux=A(inc),
                                  sfc_src_jac=A(sj), **kw)
            for k in keys:
                assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-12, (k, nlay, top_at_1, kw)
    finally:
        hiplib.ext_call(hip, "rte_hip_profile_enable", ["i"], 0)
    import ctypes
    names = []
    for i in range(hiplib.ext_call(hip, "rte_hip_profile_count", [])):
        buf = ctypes.create_string_buffer(128); cnt, ms = ctypes.c_longlong(0), ctypes.c_double(0)
        hip.raw("rte_hip_profile_get")(ctypes.c_int(i), buf, ctypes.c_int(128), ctypes.byref(cnt), ctypes.byref(ms))
        if cnt.value:
            names.append(buf.value.decode())
    assert "lw_noscat_seg2_kernel" in names, names


@pytest.mark.parametrize("nlay,top_at_1", [(100, False), (112, True), (128, False), (137, True), (144, False)])
def test_lw_rescaling_with_up_to_144_layers(hip, oracle_c, nlay, top_at_1):
    """rte_lw's default for two-stream (cloudy) optical properties -- lw_solver_noscat with Tang rescaling -- on the segmented
    three-sweep kernel with 14 / 16 / 18 layers per wave (97 ... 144 layers: IFS-137-class host models), two angles, incident
    flux and Jacobian, against the oracle."""
    import numpy as np

    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    rng = np.random.default_rng(1000 + nlay)
    F = lambda *sh: np.asfortranarray(rng.random(sh))
    ncol, ngpt = 70, 16
    tau, ssa, g = F(ncol, nlay, ngpt) * 2.0, F(ncol, nlay, ngpt) * 0.999, F(ncol, nlay, ngpt) * 0.9 - 0.1
    lay, lev = F(ncol, nlay, ngpt) * 10 + 1, F(ncol, nlay + 1, ngpt) * 10 + 1
    emis, sfc, inc, sj = F(ncol, ngpt) * 0.2 + 0.8, F(ncol, ngpt) * 10, F(ncol, ngpt), F(ncol, ngpt)
    for kw, keys in ((dict(n_gauss_angles=2, do_jacobians=True), ("flux_up", "flux_dn", "flux_up_jac")), (dict(), ("flux_up", "flux_dn"))):
        ref = frontend.rte_lw(oracle_c, frontend.NumpyArrays(), ncol, nlay, ngpt, top_at_1, tau, lay, lev, emis, sfc, ssa=ssa, g=g,
                              inc_flux=inc, sfc_src_jac=sj, **kw)
        out = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, top_at_1, A(tau), A(lay), A(lev), A(emis), A(sfc), ssa=A(ssa), g=A(g),
                              inc_flux=A(inc), sfc_src_jac=A(sj), **kw)
        for k in keys:
            assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-12, (k, nlay, top_at_1, kw)


@pytest.mark.parametrize("nlay,top_at_1", [(27, False), (60, True), (72, False), (75, True), (91, False)])
def test_byband_fluxes_from_the_segmented_kernels(hip, oracle_c, nlay, top_at_1):
    """By-band fluxes (ty_fluxes_byband, rte/extensions/mo_fluxes_byband.F90:46-137): the reference reduces the spectral
    arrays with rte_sum_byband; the extensions rte_hip_lw_solver_noscat_byband / rte_hip_sw_solver_2stream_byband accumulate per
    band inside the segmented kernels (one block per column tile and band).  Bands of unequal width, LW with two angles and an
    incident flux, SW with night columns and a diffuse boundary condition, against the oracle's spectral arrays reduced by the
    oracle's rte_sum_byband."""
    import numpy as np

    xp, xo = frontend.TorchArrays("cuda:0"), frontend.NumpyArrays()
    A = xp.asarray
    rng = np.random.default_rng(100 + nlay)
    F = lambda *sh: np.asfortranarray(rng.random(sh))
    ncol, ngpt, nbnd = 70, 40, 3
    bl = np.asfortranarray(np.array([[1, 9, 25], [8, 24, 40]], dtype=np.int32))  # 8 + 16 + 16 g-points
    tau, ssa, g = F(ncol, nlay, ngpt) * 3.0, F(ncol, nlay, ngpt) * 0.999, F(ncol, nlay, ngpt) * 0.9 - 0.1
    lay, lev = F(ncol, nlay, ngpt) * 10 + 1, F(ncol, nlay + 1, ngpt) * 10 + 1
    emis, sfc, inc = F(ncol, ngpt) * 0.2 + 0.8, F(ncol, ngpt) * 10, F(ncol, ngpt)
    ref = frontend.rte_lw_byband(oracle_c, xo, ncol, nlay, ngpt, nbnd, bl, top_at_1, tau, lay, lev, emis, sfc, n_gauss_angles=2, inc_flux=inc)
    out = frontend.rte_lw_byband(hip, xp, ncol, nlay, ngpt, nbnd, A(bl), top_at_1, A(tau), A(lay), A(lev), A(emis), A(sfc),
                                 n_gauss_angles=2, inc_flux=A(inc))
    for k in ("bb_up", "bb_dn"):
        assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-12, ("lw", k)
    mu0 = np.asfortranarray(np.repeat((rng.random(ncol) * 1.2 - 0.2)[:, None], nlay, axis=1))  # some <= 0
    adir, adif, idir, idif = F(ncol, ngpt), F(ncol, ngpt), F(ncol, ngpt) * 100, F(ncol, ngpt) * 10
    ref = frontend.rte_sw_byband(oracle_c, xo, ncol, nlay, ngpt, nbnd, bl, top_at_1, tau, ssa, g, mu0, idir, adir, adif, inc_flux_dif=idif)
    out = frontend.rte_sw_byband(hip, xp, ncol, nlay, ngpt, nbnd, A(bl), top_at_1, A(tau), A(ssa), A(g), A(mu0), A(idir), A(adir), A(adif),
                                 inc_flux_dif=A(idif))
    for k in ("bb_up", "bb_dn", "bb_dir"):
        assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-12, ("sw", k)


@pytest.mark.parametrize("name", ["lw_mid_ragged", "lw_mid_top1", "lw_g256"])
def test_segmented_and_generic_lw_solvers_agree(hip, name):
    """The production (segmented) and the generic LW kernels are two implementations of one
    recurrence: they must agree to rounding."""
    case = cases.CASES[name]
    inp = cases.make_inputs(case)
    xp = frontend.TorchArrays("cuda:0")
    a = cases.run_suite(hip, xp, case, inp)
    hiplib.ext_call(hip, "rte_hip_force_generic_lw", ["i"], 1)
    try:
        b = cases.run_suite(hip, xp, case, inp)
    finally:
        hiplib.ext_call(hip, "rte_hip_force_generic_lw", ["i"], 0)
    for k in ("lw1.flux_up", "lw1.flux_dn", "lw3j.flux_up", "lw3j.flux_dn", "lw3j.flux_up_jac",
              "lw2s.gpt_flux_up", "lw2s.gpt_flux_dn", "lwDs.gpt_flux_up", "lwDs.gpt_flux_dn", "lwDs.flux_up_jac"):  # spectral output too
        assert cases.rel_err(a[k], b[k]) <= 1e-13, k
    # the two-stream solver (segmented with projective composites vs thread-per-column generic kernel)
    for k in ("lw2str.gpt_flux_up", "lw2str.gpt_flux_dn"):
        assert cases.rel_err(a[k], b[k]) <= 1e-12, k


def test_tau_absorption_paths_agree(hip, oracle_c):
    """The production tau kernel (LUT slab staged in LDS from g-fastest re-laid-out tables, FMAs) and
    the native-layout direct-gather kernel evaluate the same sums: equal to rounding; both match the
    oracle.  1100 columns x 16-wide bands exercises the production path, its ragged last tile and --
    with the wide temperature spread of the synthetic atmosphere -- the overflow worklist."""
    import torch
    from rte_rrtmgp_amd import synth

    kd = synth.make_kdist("lw", ngpt=64, nbnd=4, nminor_lower=11, nminor_upper=7)  # 16-wide bands
    ncol, nlay = 1100, 24
    atm = synth.make_atmosphere(ncol, nlay, seed=77, kdist=kd)
    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    go = frontend.GasOptics(hip, kd, xp)
    play, tlay, col_gas = A(atm.play), A(atm.tlay), A(atm.col_gas)
    st = go.interpolation(ncol, nlay, play, tlay, col_gas)

    def run():
        tau = xp.full((ncol, nlay, kd.ngpt), 0.125)  # non-zero start: tau is intent(inout)
        go.compute_tau_absorption(ncol, nlay, st, play, tlay, col_gas, tau)
        return xp.to_numpy(tau).copy()

    t_fast = run()
    hiplib.ext_call(hip, "rte_hip_force_direct_gather", ["i"], 1)
    t_dir = run()
    hiplib.ext_call(hip, "rte_hip_force_direct_gather", ["i"], 0)
    assert cases.rel_err(t_fast, t_dir) <= 1e-13
    xn = frontend.NumpyArrays()
    gon = frontend.GasOptics(oracle_c, kd, xn)
    stn = gon.interpolation(ncol, nlay, atm.play, atm.tlay, atm.col_gas)
    taun = xn.full((ncol, nlay, kd.ngpt), 0.125)
    gon.compute_tau_absorption(ncol, nlay, stn, atm.play, atm.tlay, atm.col_gas, taun)
    assert cases.rel_err(t_fast, taun) <= RTOL_GAS
    torch.cuda.synchronize()


@pytest.mark.parametrize("nbnd,ngpt,top_at_1", [(2, 64, False), (3, 96, True), (4, 64, True), (8, 64, False), (16, 128, True)])
def test_wide_bands_on_production_kernels(hip, oracle_c, nbnd, ngpt, top_at_1):
    """Bands wider than the 16 g-point stage (32 g-points) and NARROWER ones (8 g-points, the shape of the
    reduced g128 / g112 k-distributions, handled with 8-wide stages): the production tau / Planck / Rayleigh
    kernels walk them in several stages with the same band metadata.  Against the oracle and the direct
    kernels, LW and SW tables, 700 columns (ragged last tile), both vertical orientations (the surface
    layer is the first or the last one the Planck kernel visits)."""
    from rte_rrtmgp_amd import synth

    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    ncol, nlay = 700, 19
    for kind in ("lw", "sw"):
        kd = synth.make_kdist(kind, ngpt=ngpt, nbnd=nbnd)
        atm = synth.make_atmosphere(ncol, nlay, seed=3, kdist=kd, top_at_1=top_at_1)
        outs = {}
        for mode in ("fast", "direct", "oracle"):
            if mode == "oracle":
                lib, arr, conv = oracle_c, frontend.NumpyArrays(), (lambda v: v)
            else:
                lib, arr, conv = hip, xp, A
            hiplib.ext_call(hip, "rte_hip_force_direct_gather", ["i"], 1 if mode == "direct" else 0)
            try:
                go = frontend.GasOptics(lib, kd, arr)
                if kind == "lw":
                    b = go.gas_optics_lw(ncol, nlay, conv(atm.play), conv(atm.plev), conv(atm.tlay), conv(atm.tsfc),
                                         conv(atm.col_gas), conv(atm.tlev), atm.top_at_1)
                    keys = ("tau", "lay_src", "lev_src", "sfc_src", "sfc_src_jac")
                else:
                    b = go.gas_optics_sw(ncol, nlay, conv(atm.play), conv(atm.plev), conv(atm.tlay), conv(atm.col_gas),
                                         conv(atm.col_dry))
                    keys = ("tau_abs", "tau_rayleigh", "tau", "ssa")
                outs[mode] = {k: np.array(arr.to_numpy(b[k])) for k in keys}
            finally:
                hiplib.ext_call(hip, "rte_hip_force_direct_gather", ["i"], 0)
        for k in outs["oracle"]:
            assert cases.rel_err(outs["fast"][k], outs["direct"][k]) <= 1e-13, (kind, k)
            assert cases.rel_err(outs["fast"][k], outs["oracle"][k]) <= RTOL_GAS, (kind, k)

@pytest.mark.gpu
@pytest.mark.parametrize("top_at_1", [False, True])
def test_ragged_minor_intervals_on_production_kernels(hip, oracle_c, top_at_1):
    """Real coefficient files are ragged: 1 ... 9 minor-absorber intervals per band and regime (whatever
    reduce_minor_arrays leaves, rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:1790-1907).  The production tau kernels keep the
    column amounts of 4 intervals per stage in registers and run a band's further intervals in a tail pass: g256 / g224-shaped
    tables with 0 ... 8 intervals per band, 1100 columns, LW (tau), SW chain (tau_abs), and the one-pass SW gas optics
    (tau, ssa, g), against the oracle and the direct kernels."""
    from rte_rrtmgp_amd import synth

    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    ncol, nlay = 1100, 24
    for kind in ("lw", "sw"):
        kd = synth.make_kdist(kind, minor_distribution="ragged")
        gpb = kd.ngpt // kd.nbnd
        per_band = np.bincount((np.asarray(kd.arrays["minor_limits_gpt_lower"])[0] - 1) // gpb, minlength=kd.nbnd)
        assert per_band.max() > 4 and per_band.min() < 4
        atm = synth.make_atmosphere(ncol, nlay, seed=8, kdist=kd, top_at_1=top_at_1)
        outs = {}
        for mode in ("fast", "direct", "oracle", "onepass"):
            if mode == "onepass" and kind == "lw":
                continue
            if mode == "oracle":
                lib, arr, conv = oracle_c, frontend.NumpyArrays(), (lambda v: v)
            else:
                lib, arr, conv = hip, xp, A
            hiplib.ext_call(hip, "rte_hip_force_direct_gather", ["i"], 1 if mode == "direct" else 0)
            try:
                go = frontend.GasOptics(lib, kd, arr)
                if kind == "lw":
                    b = go.gas_optics_lw(ncol, nlay, conv(atm.play), conv(atm.plev), conv(atm.tlay), conv(atm.tsfc),
                                         conv(atm.col_gas), conv(atm.tlev), atm.top_at_1)
                    keys = ("tau", "lay_src")
                else:
                    b = go.gas_optics_sw(ncol, nlay, conv(atm.play), conv(atm.plev), conv(atm.tlay), conv(atm.col_gas),
                                         conv(atm.col_dry), fuse_rayleigh=("all" if mode == "onepass" else False))
                    keys = ("tau", "ssa", "g") if mode == "onepass" else ("tau_abs", "tau", "ssa", "g")
                outs[mode] = {k: np.array(arr.to_numpy(b[k])) for k in keys}
            finally:
                hiplib.ext_call(hip, "rte_hip_force_direct_gather", ["i"], 0)
        for k in outs["oracle"]:
            assert cases.rel_err(outs["fast"][k], outs["direct"][k]) <= 1e-13, (kind, k)
            assert cases.rel_err(outs["fast"][k], outs["oracle"][k]) <= RTOL_GAS, (kind, k)
            if "onepass" in outs and k in outs["onepass"]:
                assert cases.rel_err(outs["onepass"][k], outs["oracle"][k]) <= RTOL_GAS, (kind, k, "one-pass")


@pytest.mark.parametrize("kind,top_at_1", [("lw", False), ("lw", True), ("sw", False), ("sw", True)])
def test_production_kernels_at_the_real_table_shape_against_the_oracle(hip, oracle_c, kind, top_at_1):
    """The production (slab) kernels at the REAL table shapes -- g256: 16 bands of 16 g-points, 10 flavors, 64 + 36 minor
    intervals; g224: 14 bands -- compared array by array and element by element with the C oracle itself (not through the
    direct kernels, not through fluxes): 1024 columns x 60 layers = two full 512-column tiles per layer, tropopause layers
    with both regimes in one tile, both vertical orientations.  tau / lay_src / lev_src / sfc_src / sfc_src_jac (LW) and
    tau_abs / tau_rayleigh / tau / ssa / g (SW: the two-kernel chain, and tau / ssa of the one-pass form)."""
    from rte_rrtmgp_amd import synth

    kd = synth.make_kdist(kind)
    ncol, nlay = 1024, 60
    atm = synth.make_atmosphere(ncol, nlay, seed=21, kdist=kd, top_at_1=top_at_1)
    xp, xn = frontend.TorchArrays("cuda:0"), frontend.NumpyArrays()
    A = xp.asarray
    outs = {}
    for mode, lib, arr, conv in (("hip", hip, xp, A), ("oracle", oracle_c, xn, (lambda v: v))):
        go = frontend.GasOptics(lib, kd, arr)
        if kind == "lw":
            b = go.gas_optics_lw(ncol, nlay, conv(atm.play), conv(atm.plev), conv(atm.tlay), conv(atm.tsfc), conv(atm.col_gas),
                                 conv(atm.tlev), atm.top_at_1)
            keys = ("tau", "lay_src", "lev_src", "sfc_src", "sfc_src_jac")
        else:
            b = go.gas_optics_sw(ncol, nlay, conv(atm.play), conv(atm.plev), conv(atm.tlay), conv(atm.col_gas), conv(atm.col_dry))
            keys = ("tau_abs", "tau_rayleigh", "tau", "ssa", "g")
        outs[mode] = {k: np.array(arr.to_numpy(b[k])) for k in keys if b.get(k) is not None}
    assert hiplib.ext_call(hip, "rte_hip_stat", ["i"], 2) in (1, 2)  # (the tile geometry of the slab kernel was derived: not rerouted)
    for k, ref in outs["oracle"].items():
        got = outs["hip"][k]
        assert np.isfinite(got).all(), (kind, k)
        assert cases.rel_err(got, ref) <= RTOL_GAS, (kind, k, cases.rel_err(got, ref))
        assert cases.elem_err(got, ref, 1e-8) <= ETOL_GAS, (kind, k, cases.elem_err(got, ref, 1e-8))
    if kind == "sw":  # the one-pass form (absorption + Rayleigh + combine in the slab kernel)
        go = frontend.GasOptics(hip, kd, xp)
        b = go.gas_optics_sw(ncol, nlay, A(atm.play), A(atm.plev), A(atm.tlay), A(atm.col_gas), A(atm.col_dry), fuse_rayleigh="all")
        for k in ("tau", "ssa"):
            got = np.array(xp.to_numpy(b[k]))
            assert cases.elem_err(got, outs["oracle"][k], 1e-8) <= ETOL_GAS, (kind, k, "one-pass")


@pytest.mark.gpu
@pytest.mark.parametrize("kind,nflav,top_at_1", [("lw", 2, False), ("lw", 3, True), ("sw", 2, True), ("lw", 1, False)])
def test_bands_that_share_a_flavor_keep_their_weights(hip, oracle_c, kind, nflav, top_at_1):
    """Tables with FEW flavors: the slab kernel walks a single-regime tile's bands sorted by flavor and keeps the flavor weights
    in registers across stages of one flavor -- with 1 ... 3 flavors over 14 / 16 bands nearly every stage does, in the table
    order too (tiles at the tropopause, both regimes' flavors unchanged from band to band).  tau (SW: tau_abs, tau_rayleigh,
    and tau / ssa of the one-pass form) element by element against the C oracle."""
    from rte_rrtmgp_amd import synth

    kd = synth.make_kdist(kind, nflav=nflav)
    ncol, nlay = 1024, 60
    atm = synth.make_atmosphere(ncol, nlay, seed=5, kdist=kd, top_at_1=top_at_1)
    xp, xn = frontend.TorchArrays("cuda:0"), frontend.NumpyArrays()
    A = xp.asarray
    outs = {}
    for mode, lib, arr, conv in (("hip", hip, xp, A), ("oracle", oracle_c, xn, (lambda v: v))):
        go = frontend.GasOptics(lib, kd, arr)
        if kind == "lw":
            b = go.gas_optics_lw(ncol, nlay, conv(atm.play), conv(atm.plev), conv(atm.tlay), conv(atm.tsfc), conv(atm.col_gas),
                                 conv(atm.tlev), atm.top_at_1)
            keys = ("tau", "lay_src")
        else:
            b = go.gas_optics_sw(ncol, nlay, conv(atm.play), conv(atm.plev), conv(atm.tlay), conv(atm.col_gas), conv(atm.col_dry))
            keys = ("tau_abs", "tau_rayleigh", "tau", "ssa")
        outs[mode] = {k: np.array(arr.to_numpy(b[k])) for k in keys if b.get(k) is not None}
    assert hiplib.ext_call(hip, "rte_hip_stat", ["i"], 2) in (1, 2)  # (the slab kernel ran: its tile geometry was derived)
    for k, ref in outs["oracle"].items():
        got = outs["hip"][k]
        assert np.isfinite(got).all(), (kind, k)
        assert cases.elem_err(got, ref, 1e-8) <= ETOL_GAS, (kind, nflav, k, cases.elem_err(got, ref, 1e-8))
    if kind == "sw":
        go = frontend.GasOptics(hip, kd, xp)
        b = go.gas_optics_sw(ncol, nlay, A(atm.play), A(atm.plev), A(atm.tlay), A(atm.col_gas), A(atm.col_dry), fuse_rayleigh="all")
        for k in ("tau", "ssa"):
            assert cases.elem_err(np.array(xp.to_numpy(b[k])), outs["oracle"][k], 1e-8) <= ETOL_GAS, (kind, k, "one-pass")


@pytest.mark.gpu
def test_call_graph_replays_the_chain(hip):
    """rte_hip_graph_begin / _end / _launch: the LW chain (gas optics + rte_lw_solver_noscat, deferred zero fill and shared
    geometry as the bench runs it) captured once as a hipGraph and replayed on CHANGED inputs in the same arrays gives, bit
    for bit, what the calls themselves give on those inputs."""
    import torch
    from rte_rrtmgp_amd import synth

    kd = synth.make_kdist("lw")
    ncol, nlay = 1024, 60
    atm = synth.make_atmosphere(ncol, nlay, seed=3, kdist=kd)
    atm2 = synth.make_atmosphere(ncol, nlay, seed=4, kdist=kd)
    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    names = ("play", "plev", "tlay", "tlev", "tsfc", "col_gas")
    dev = {k: A(getattr(atm, k)) for k in names}
    alt = {k: A(getattr(atm2, k)) for k in names}
    first = {k: v.clone() for k, v in dev.items()}
    emis = xp.full((ncol, kd.ngpt), 0.98)
    go = frontend.GasOptics(hip, kd, xp)
    bufs, rb = {}, {}
    for name in ("rte_hip_defer_zero", "rte_hip_share_geometry"):
        hiplib.ext_call(hip, name, ["i"], 1)
    try:
        def chain():
            go.gas_optics_lw(ncol, nlay, dev["play"], dev["plev"], dev["tlay"], dev["tsfc"], dev["col_gas"], dev["tlev"], atm.top_at_1,
                             buffers=bufs)
            frontend.rte_lw(hip, xp, ncol, nlay, kd.ngpt, atm.top_at_1, bufs["tau"], bufs["lay_src"], bufs["lev_src"], emis,
                            bufs["sfc_src"], buffers=rb)

        def result():
            hiplib.ext_call(hip, "rte_hip_sync", [])
            return {k: np.array(xp.to_numpy(rb[k])) for k in ("flux_up", "flux_dn")} | {"tau": np.array(xp.to_numpy(bufs["tau"]))}

        chain()
        want1 = result()
        for k in names:
            dev[k].copy_(alt[k])
        chain()
        want2 = result()
        assert np.abs(want1["flux_up"] - want2["flux_up"]).max() > 0
        g = hiplib.CallGraph(hip, chain)
        try:
            for inputs, want in ((first, want1), (alt, want2), (first, want1)):
                for k in names:
                    dev[k].copy_(inputs[k])
                for t in (bufs["tau"], rb["flux_up"], rb["flux_dn"]):
                    t.fill_(-1.0)
                torch.cuda.synchronize()
                g.launch()
                got = result()
                for k in want:
                    assert np.array_equal(got[k], want[k]), k
            # a graph addresses the library's scratch arena and persistent buffers: once those are freed (here: rte_hip_release;
            # the same after a larger call that makes the arena grow) the graph must refuse to launch, not run on freed memory
            hiplib.ext_call(hip, "rte_hip_release", [])
            with pytest.raises(RuntimeError, match="stale"):
                g.launch()
            chain()  # (and the context works on: the calls themselves re-allocate)
            got = result()
            for k in want1:
                assert np.array_equal(got[k], want1[k]), k
        finally:
            g.close()
    finally:
        for name in ("rte_hip_defer_zero", "rte_hip_share_geometry"):
            hiplib.ext_call(hip, name, ["i"], 0)


def test_plans_follow_tables_changed_in_place(hip, oracle_c):
    """The host-side plans of the production kernels are cached per table ADDRESS; the device-side plan guards must
    notice tables whose CONTENTS changed behind those addresses (no rte_hip_invalidate_plans()): the call then runs
    on the direct kernels, and the next one on a rebuilt plan.  Checked for the minor-interval metadata
    (tau_absorption) and for band limits that no longer have the cached stage alignment (Planck, tau)."""
    import torch
    from rte_rrtmgp_amd import synth

    xp, xn = frontend.TorchArrays("cuda:0"), frontend.NumpyArrays()
    ncol, nlay = 1100, 18

    def both(kd, atm, go):
        A = xp.asarray
        args = [A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tsfc", "col_gas", "tlev")]
        b = go.gas_optics_lw(ncol, nlay, *args, atm.top_at_1)
        gon = frontend.GasOptics(oracle_c, kd, xn)
        bn = gon.gas_optics_lw(ncol, nlay, atm.play, atm.plev, atm.tlay, atm.tsfc, atm.col_gas, atm.tlev, atm.top_at_1)
        for k in ("tau", "lay_src", "lev_src", "sfc_src"):
            assert cases.rel_err(xp.to_numpy(b[k]), bn[k]) <= RTOL_GAS, k

    kd = synth.make_kdist("lw", ngpt=64, nbnd=4, nminor_lower=11, nminor_upper=7)
    atm = synth.make_atmosphere(ncol, nlay, seed=21, kdist=kd)
    go = frontend.GasOptics(hip, kd, xp)
    both(kd, atm, go)  # builds the plans
    # 1. other absorbers / scalings behind the same device addresses
    rng = np.random.default_rng(3)
    for reg in ("lower", "upper"):
        n = kd.arrays[f"idx_minor_{reg}"].shape[0]
        kd.arrays[f"idx_minor_{reg}"][:] = rng.integers(1, kd.ngas + 1, n)
        kd.arrays[f"minor_scales_with_density_{reg}"][:] = ~kd.arrays[f"minor_scales_with_density_{reg}"]
        go.t[f"idx_minor_{reg}"].copy_(torch.from_numpy(kd.arrays[f"idx_minor_{reg}"]))
        go.t[f"minor_scales_with_density_{reg}"].copy_(torch.from_numpy(kd.arrays[f"minor_scales_with_density_{reg}"]))
    both(kd, atm, go)  # guard fires: direct kernels
    both(kd, atm, go)  # rebuilt plan
    # 2. band limits that break the 16-alignment the cached stage width assumed: bands of 24 + 8 + 16 + 16 g-points
    bl = np.asfortranarray(np.array([[1, 25, 33, 49], [24, 32, 48, 64]], dtype=np.int32))
    kd.arrays["band_lims_gpt"][:] = bl
    go.t["band_lims_gpt"].copy_(torch.from_numpy(np.ascontiguousarray(bl.T)))
    gb = np.repeat(np.arange(1, 5), [24, 8, 16, 16]).astype(np.int32)
    kd.arrays["gpoint_bands"][:] = gb
    go.t["gpoint_bands"].copy_(torch.from_numpy(gb))
    for reg in ("lower", "upper"):  # minor intervals stay whole bands
        lims = kd.arrays[f"minor_limits_gpt_{reg}"]
        band_of = (lims[0] - 1) // 16
        lims[0], lims[1] = bl[0][band_of], bl[1][band_of]
        kd.arrays[f"kminor_start_{reg}"][:] = 1 + np.concatenate([[0], np.cumsum(lims[1] - lims[0] + 1)[:-1]])
        go.t[f"minor_limits_gpt_{reg}"].copy_(torch.from_numpy(np.ascontiguousarray(lims.T)))
        go.t[f"kminor_start_{reg}"].copy_(torch.from_numpy(kd.arrays[f"kminor_start_{reg}"]))
    gf = kd.arrays["gpoint_flavor"]
    gf[:, :] = gf[:, bl[0][gb - 1] - 1]  # flavors stay constant within the new bands
    go.t["gpoint_flavor"].copy_(torch.from_numpy(np.ascontiguousarray(gf.T)))
    both(kd, atm, go)
    both(kd, atm, go)
    torch.cuda.synchronize()


def test_pinned_host_arrays_are_synchronous(hip, oracle_c):
    """Host-VISIBLE memory (pinned: hipHostMalloc / torch pin_memory) is addressed by the kernels in place, but it is
    the caller's host array: the call must have finished when it returns (no stale fluxes read by the host)."""
    import torch

    ncol, nlay, ngpt = 3000, 40, 64
    rng = np.random.default_rng(8)

    def pinned(shape, fill):
        t = torch.empty(tuple(reversed(shape)), dtype=torch.float64).pin_memory()
        a = t.numpy().T  # Fortran-ordered view of the pinned buffer
        a[...] = fill
        return t, a

    keep = []
    arrs = {}
    for name, shape, fill in (("tau", (ncol, nlay, ngpt), rng.uniform(0.0, 2.0, (ncol, nlay, ngpt))),
                              ("lay", (ncol, nlay, ngpt), rng.uniform(1.0, 9.0, (ncol, nlay, ngpt))),
                              ("lev", (ncol, nlay + 1, ngpt), rng.uniform(1.0, 9.0, (ncol, nlay + 1, ngpt))),
                              ("emis", (ncol, ngpt), 0.97), ("sfc", (ncol, ngpt), 5.0), ("inc", (ncol, ngpt), 0.0),
                              ("Ds", (ncol, ngpt, 1), 1.66), ("up", (ncol, nlay + 1), -1.0), ("dn", (ncol, nlay + 1), -1.0)):
        t, a = pinned(shape, fill)
        keep.append(t)
        arrs[name] = a
    w = np.array([1.0])
    for _ in range(3):
        arrs["up"][...] = -1.0
        hip.rte_lw_solver_noscat(ncol, nlay, ngpt, False, 1, arrs["Ds"], w, arrs["tau"], arrs["lay"], arrs["lev"], arrs["emis"],
                                 arrs["sfc"], arrs["inc"], arrs["tau"], arrs["tau"], True, arrs["up"], arrs["dn"], False,
                                 arrs["sfc"], arrs["up"], False, arrs["tau"], arrs["tau"])
        got_up = arrs["up"].copy()  # read immediately: no torch / HIP synchronisation by the caller
        assert (got_up > 0).all()
    F = np.asfortranarray
    ru, rd = F(np.empty((ncol, nlay + 1))), F(np.empty((ncol, nlay + 1)))
    oracle_c.rte_lw_solver_noscat(ncol, nlay, ngpt, False, 1, F(arrs["Ds"]), w, F(arrs["tau"]), F(arrs["lay"]), F(arrs["lev"]),
                                  F(arrs["emis"]), F(arrs["sfc"]), F(arrs["inc"]), F(arrs["tau"]), F(arrs["tau"]), True, ru, rd, False,
                                  F(arrs["sfc"]), ru.copy(order="F"), False, F(arrs["tau"]), F(arrs["tau"]))
    assert cases.rel_err(got_up, ru) <= RTOL_FLUX


def test_tau_accumulates_onto_device_and_pinned_buffers(hip):
    """compute_tau_absorption is intent(inout): it adds to what `tau` holds.  On device memory the production kernel
    does that with a hardware floating-point atomic add: (incoming value + the optical depth computed onto zeros), bit for
    bit -- one addition of the same two numbers.  Host-visible (pinned) memory, where such atomics are not defined, goes
    to the direct kernels (plain read - add - write, the reference's association): the same sum to rounding."""
    import torch
    from rte_rrtmgp_amd import synth

    kd = synth.make_kdist("lw", ngpt=64, nbnd=4)
    ncol, nlay = 1100, 24  # >= 512 columns: the production kernel
    atm = synth.make_atmosphere(ncol, nlay, seed=19, kdist=kd)
    xn, xp = frontend.NumpyArrays(), frontend.TorchArrays("cuda:0")
    rng = np.random.default_rng(4)
    incoming = np.asfortranarray(rng.uniform(0.0, 3.0, (ncol, nlay, kd.ngpt)))
    F = np.asfortranarray
    go_n = frontend.GasOptics(hip, kd, xn)
    st = go_n.interpolation(ncol, nlay, F(atm.play), F(atm.tlay), F(atm.col_gas), None)
    # (1) onto zeros, pageable host array (staged through the device arena)
    zero = F(np.zeros((ncol, nlay, kd.ngpt)))
    go_n.compute_tau_absorption(ncol, nlay, st, F(atm.play), F(atm.tlay), F(atm.col_gas), zero)
    assert zero.max() > 0
    expect = incoming + zero
    # (2) onto the incoming values in pinned host memory (addressed in place)
    t = torch.empty((kd.ngpt, nlay, ncol), dtype=torch.float64).pin_memory()
    pinned = t.numpy().T
    pinned[...] = incoming
    go_n.compute_tau_absorption(ncol, nlay, st, F(atm.play), F(atm.tlay), F(atm.col_gas), pinned)
    assert cases.rel_err(pinned, expect) <= 1e-13
    # (3) onto the incoming values in device memory
    go_d = frontend.GasOptics(hip, kd, xp)
    A = xp.asarray
    st_d = go_d.interpolation(ncol, nlay, A(atm.play), A(atm.tlay), A(atm.col_gas), None)
    dev = A(incoming)
    go_d.compute_tau_absorption(ncol, nlay, st_d, A(atm.play), A(atm.tlay), A(atm.col_gas), dev)
    torch.cuda.synchronize()
    assert np.array_equal(xp.to_numpy(dev), expect)


def test_plain_abi_accumulate_looks_whether_tau_is_zero(hip):
    """Without the deferred zero fill compute_tau_absorption must accumulate, but the frontend has just zeroed tau
    (mo_gas_optics_rrtmgp.F90:637,679): the call reads the array once and runs the overwriting kernel when every element is
    zero, the accumulating one otherwise (TauV5::nonzero).  Same bits as the accumulating kernel alone on a zero array; one
    non-zero element anywhere -- or a NaN -- and the whole call accumulates."""
    import torch
    from rte_rrtmgp_amd import synth

    kd = synth.make_kdist("lw", ngpt=64, nbnd=4)
    ncol, nlay = 1100, 24
    atm = synth.make_atmosphere(ncol, nlay, seed=23, kdist=kd)
    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    go = frontend.GasOptics(hip, kd, xp)
    play, tlay, col_gas = A(atm.play), A(atm.tlay), A(atm.col_gas)
    st = go.interpolation(ncol, nlay, play, tlay, col_gas)

    def run(start, check):
        hiplib.ext_call(hip, "rte_hip_tau_zero_check", ["i"], 1 if check else 0)
        try:
            tau = A(start)
            go.compute_tau_absorption(ncol, nlay, st, play, tlay, col_gas, tau)
            return xp.to_numpy(tau).copy()
        finally:
            hiplib.ext_call(hip, "rte_hip_tau_zero_check", ["i"], 1)

    zero = np.zeros((ncol, nlay, kd.ngpt), order="F")
    base = run(zero, False)
    assert base.max() > 0 and np.array_equal(run(zero, True), base)
    one = zero.copy(order="F"); one[ncol - 1, nlay - 1, kd.ngpt - 1] = 2.5; one[0, 0, 0] = -0.0
    got = run(one, True)
    assert np.array_equal(got, run(one, False)) and got[ncol - 1, nlay - 1, kd.ngpt - 1] == 2.5 + base[ncol - 1, nlay - 1, kd.ngpt - 1]
    bad = zero.copy(order="F"); bad[5, 3, 7] = np.nan
    got = run(bad, True)
    assert np.isnan(got[5, 3, 7]) and np.array_equal(np.nan_to_num(got), np.nan_to_num(run(bad, False)))
    torch.cuda.synchronize()


def test_fused_rayleigh_combine_matches_unfused(hip, oracle_c):
    """rte_hip_tau_rayleigh_combine_2str (compute_tau_rayleigh + the 2-stream branch of combine_abs_and_rayleigh in one
    pass, in place on the absorption optical depth) against the unfused ABI calls -- bit-identical -- and the oracle;
    production kernel (1100 columns, 16- and 8-wide bands) and the direct kernel (small call)."""
    from rte_rrtmgp_amd import synth

    xp, xn = frontend.TorchArrays("cuda:0"), frontend.NumpyArrays()
    A = xp.asarray
    for ncol, nbnd, ngpt in ((1100, 4, 64), (1100, 8, 64), (90, 4, 64)):
        kd = synth.make_kdist("sw", ngpt=ngpt, nbnd=nbnd)
        nlay = 22
        atm = synth.make_atmosphere(ncol, nlay, seed=13, kdist=kd)
        go = frontend.GasOptics(hip, kd, xp)
        args = [A(getattr(atm, k)) for k in ("play", "plev", "tlay", "col_gas", "col_dry")]
        un = go.gas_optics_sw(ncol, nlay, *args)
        fu = go.gas_optics_sw(ncol, nlay, *args, fuse_rayleigh=True)
        assert "tau_rayleigh" not in fu
        for k in ("tau", "ssa", "g"):
            assert np.array_equal(xp.to_numpy(fu[k]), xp.to_numpy(un[k])), (ncol, nbnd, k)
        bn = frontend.GasOptics(oracle_c, kd, xn).gas_optics_sw(ncol, nlay, atm.play, atm.plev, atm.tlay, atm.col_gas, atm.col_dry)
        for k in ("tau", "ssa", "g"):
            assert cases.rel_err(xp.to_numpy(fu[k]), bn[k]) <= RTOL_GAS, (ncol, nbnd, k)
            assert cases.elem_err(xp.to_numpy(fu[k]), bn[k]) <= ETOL_GAS, (ncol, nbnd, k)



def test_tau_rayleigh_paths_agree(hip, oracle_c):
    """The production Rayleigh kernel (whole (T, eta) plane of a band staged in LDS, layers walked by the
    block) keeps the reference's association: bit-identical to the direct-gather kernel and to the oracle.
    1100 columns exercises its ragged last tile; both vertical orientations."""
    from rte_rrtmgp_amd import synth

    kd = synth.make_kdist("sw")
    ncol, nlay = 1100, 17
    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    for top_at_1 in (False, True):
        atm = synth.make_atmosphere(ncol, nlay, seed=31, kdist=kd, top_at_1=top_at_1)
        go = frontend.GasOptics(hip, kd, xp)
        play, tlay, col_gas, col_dry = A(atm.play), A(atm.tlay), A(atm.col_gas), A(atm.col_dry)
        st = go.interpolation(ncol, nlay, play, tlay, col_gas)

        def run():
            t = xp.full((ncol, nlay, kd.ngpt), -1.0)
            go.compute_tau_rayleigh(ncol, nlay, st, col_dry, col_gas, t)
            return xp.to_numpy(t).copy()

        t_fast = run()
        hiplib.ext_call(hip, "rte_hip_force_direct_gather", ["i"], 1)
        t_dir = run()
        hiplib.ext_call(hip, "rte_hip_force_direct_gather", ["i"], 0)
        assert np.array_equal(t_fast, t_dir)
        xn = frontend.NumpyArrays()
        gon = frontend.GasOptics(oracle_c, kd, xn)
        stn = gon.interpolation(ncol, nlay, atm.play, atm.tlay, atm.col_gas)
        tn = xn.full((ncol, nlay, kd.ngpt), -1.0)
        gon.compute_tau_rayleigh(ncol, nlay, stn, atm.col_dry, atm.col_gas, tn)
        assert cases.rel_err(t_fast, tn) <= RTOL_GAS


def test_deferred_zero_fill(hip, oracle_c):
    """rte_hip_defer_zero(1): zero_array on a device buffer is recorded, consumed by
    compute_tau_absorption on that buffer (overwrite instead of memset + accumulate), and materialised
    by any other library entry.  Results must not change."""
    import torch
    from rte_rrtmgp_amd import synth

    xp = frontend.TorchArrays("cuda:0")
    hiplib.ext_call(hip, "rte_hip_defer_zero", ["i"], 1)
    try:
        # (1) the LW chain on the production tau kernel (ncol >= 512, g256 table) and on the small-problem kernel
        for name in ("lw_mid_ragged", "lw_g256"):
            case = cases.CASES[name]
            inp = cases.make_inputs(case)
            ref = cases.run_suite(oracle_c, frontend.NumpyArrays(), case, inp, which="core")
            out = cases.run_suite(hip, xp, case, inp, which="core")
            for k in ref:
                assert cases.rel_err(out[k], ref[k]) <= _tol(k), k
        kd = synth.make_kdist("lw")
        ncol, nlay = 700, 20
        atm = synth.make_atmosphere(ncol, nlay, seed=5, kdist=kd)
        go = frontend.GasOptics(hip, kd, xp)
        A = xp.asarray
        play, tlay, col_gas = A(atm.play), A(atm.tlay), A(atm.col_gas)
        st = go.interpolation(ncol, nlay, play, tlay, col_gas)
        tau = xp.full((ncol, nlay, kd.ngpt), 3.0)
        hip.zero_array_3D(ncol, nlay, kd.ngpt, tau)      # deferred
        go.compute_tau_absorption(ncol, nlay, st, play, tlay, col_gas, tau)  # consumes it
        hiplib.ext_call(hip, "rte_hip_defer_zero", ["i"], 0)
        tau2 = xp.full((ncol, nlay, kd.ngpt), 3.0)
        hip.zero_array_3D(ncol, nlay, kd.ngpt, tau2)     # executed
        go.compute_tau_absorption(ncol, nlay, st, play, tlay, col_gas, tau2)
        # two instantiations of the kernel (with / without the tau read): same sums, possibly different rounding
        assert float(((tau - tau2).abs() / tau2.abs().clamp_min(1e-300)).max()) <= 1e-14
        hiplib.ext_call(hip, "rte_hip_defer_zero", ["i"], 1)
        # (2) a deferred fill that nobody consumes is materialised by the next library call
        z = xp.full((ncol, 7, 3), 5.0)
        hip.zero_array_3D(ncol, 7, 3, z)
        s = xp.empty((ncol, 7))
        hip.rte_sum_broadband(ncol, 7, 3, z, s)
        torch.cuda.synchronize()
        assert float(s.abs().max()) == 0.0 and float(z.abs().max()) == 0.0
    finally:
        hiplib.ext_call(hip, "rte_hip_defer_zero", ["i"], 0)


@pytest.mark.parametrize("name", ["lw_tiny_top1", "sw_tiny_sfc1", "lw_mid_ragged"])
def test_single_precision_build(name):
    """-DRTE_USE_SP build (the reference's RTE_ENABLE_SP): same kernels with Float = float, checked
    against the single-precision C oracle; tolerance is the reference's own SP failure threshold
    scale (examples/CMakeLists.txt:1-5 uses 0.35 W/m2 absolute, ~1e-3 relative)."""
    from oracle import oracle as O

    hip_sp = hiplib.load("sp")
    case = cases.CASES[name]
    inp = cases.make_inputs(case)
    ref = cases.run_suite(O.load_c("sp"), frontend.NumpyArrays("sp"), case, inp, which="core")
    out = cases.run_suite(hip_sp, frontend.TorchArrays("cuda:0", "sp"), case, inp, which="core")
    for k in ref:
        if ref[k].dtype.kind in "ib":
            # float32 truncation can move an index by one exactly at a cell boundary; require near-total agreement
            assert np.mean(out[k] == ref[k]) > 0.999, k
        else:
            assert cases.rel_err(out[k], ref[k]) <= 1e-3, (k, cases.rel_err(out[k], ref[k]))


def test_single_precision_production_kernels():
    """The production gas-optics kernels (LDS slabs, loader / compute waves) in the -DRTE_USE_SP build at a
    size that takes them (700 columns): against the same build's direct kernels (1e-5: float rounding of a
    different association) and the single-precision oracle."""
    from oracle import oracle as O
    from rte_rrtmgp_amd import synth

    hip_sp = hiplib.load("sp")
    xp = frontend.TorchArrays("cuda:0", "sp")
    A = xp.asarray
    ncol, nlay = 700, 21
    for kind in ("lw", "sw"):
        kd = synth.make_kdist(kind, ngpt=64, nbnd=4)
        atm = synth.make_atmosphere(ncol, nlay, seed=17, kdist=kd)
        outs = {}
        for mode in ("fast", "direct", "oracle"):
            if mode == "oracle":
                lib, arr = O.load_c("sp"), frontend.NumpyArrays("sp")
                conv = arr.asarray  # float64 inputs must become float32 before they cross the C ABI
            else:
                lib, arr, conv = hip_sp, xp, A
            hiplib.ext_call(hip_sp, "rte_hip_force_direct_gather", ["i"], 1 if mode == "direct" else 0)
            try:
                go = frontend.GasOptics(lib, kd, arr)
                if kind == "lw":
                    b = go.gas_optics_lw(ncol, nlay, conv(atm.play), conv(atm.plev), conv(atm.tlay), conv(atm.tsfc),
                                         conv(atm.col_gas), conv(atm.tlev), atm.top_at_1)
                    keys = ("tau", "lay_src", "lev_src", "sfc_src")
                else:
                    b = go.gas_optics_sw(ncol, nlay, conv(atm.play), conv(atm.plev), conv(atm.tlay), conv(atm.col_gas),
                                         conv(atm.col_dry))
                    keys = ("tau_abs", "tau_rayleigh", "tau")
                outs[mode] = {k: np.array(arr.to_numpy(b[k])) for k in keys}
            finally:
                hiplib.ext_call(hip_sp, "rte_hip_force_direct_gather", ["i"], 0)
        for k in outs["oracle"]:
            assert cases.rel_err(outs["fast"][k], outs["direct"][k]) <= 1e-5, (kind, k, cases.rel_err(outs["fast"][k], outs["direct"][k]))
            assert cases.rel_err(outs["fast"][k], outs["oracle"][k]) <= 1e-3, (kind, k)


def test_gray_radiative_equilibrium_on_device(hip):
    from test_host_logic import _gray_equilibrium

    for top in (True, False):
        olr, up, dn = _gray_equilibrium(hip, frontend.TorchArrays("cuda:0"), top_at_1=top)
        toa = 0 if top else -1
        assert np.allclose(up[:, toa], olr, rtol=2e-4)
        net = up - dn
        assert np.allclose(net, net[:, :1], rtol=2e-4)


def test_size_independent_properties(hip):
    """Size-independent properties (the full-size configurations are compared with the reference kernels value by
    value in tests/test_fullsize_oracle.py): (1) column-subset invariance -- a big batch made of a tile repeated must
    reproduce the tile's fluxes exactly in every copy (reference tests/rte_lw_solver_unit_tests.F90
    :139-144); (2) vertical-flip invariance (:150-165)."""
    import torch
    from rte_rrtmgp_amd import synth

    kd = synth.make_kdist("lw")
    tile, reps, nlay = 100, 41, 60  # 4100 columns, not a multiple of 64
    atm = synth.make_atmosphere(tile, nlay, seed=21, kdist=kd)
    xp = frontend.TorchArrays("cuda:0")

    def run(atm_np, ncol, top_at_1):
        A = xp.asarray
        go = frontend.GasOptics(hip, kd, xp)
        b = go.gas_optics_lw(ncol, nlay, A(atm_np["play"]), A(atm_np["plev"]), A(atm_np["tlay"]), A(atm_np["tsfc"]),
                             A(atm_np["col_gas"]), A(atm_np["tlev"]), top_at_1)
        r = frontend.rte_lw(hip, xp, ncol, nlay, kd.ngpt, top_at_1, b["tau"], b["lay_src"], b["lev_src"],
                            xp.full((ncol, kd.ngpt), 0.98), b["sfc_src"])
        return xp.to_numpy(r["flux_up"]).copy(), xp.to_numpy(r["flux_dn"]).copy()

    base = {k: getattr(atm, k) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas")}
    up1, dn1 = run(base, tile, False)
    big = {k: np.asfortranarray(np.concatenate([v] * reps, axis=0)) for k, v in base.items()}
    upN, dnN = run(big, tile * reps, False)
    # every copy equals the small-batch result to rounding (copies may take different kernel paths --
    # LDS slab with FMAs, overflow worklist, small-problem kernel -- which differ by a few ulp)
    for r_ in range(reps):
        assert cases.rel_err(upN[r_ * tile:(r_ + 1) * tile], up1) <= 1e-13
        assert cases.rel_err(dnN[r_ * tile:(r_ + 1) * tile], dn1) <= 1e-13
    flip = {k: (np.asfortranarray(v[:, ::-1]) if v.ndim >= 2 and k != "tsfc" else v) for k, v in base.items()}
    upF, dnF = run(flip, tile, True)
    assert cases.rel_err(upF[:, ::-1], up1) <= 1e-13 and cases.rel_err(dnF[:, ::-1], dn1) <= 1e-13
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("nlay", [91, 137])
def test_single_precision_solvers_at_host_model_layer_counts(oracle_c, nlay):
    """The wide solver paths (eleven / twelve layers per wave, two sub-segments, the two-part two-stream solves) in the
    -DRTE_USE_SP build: fluxes against the DOUBLE-precision oracle on the same inputs, within 2e-4 of the largest flux or three
    times what the single-precision ORACLE loses on them (lw_two_stream's differences of nearly equal terms cost 1e-2 in float)."""
    import numpy as np

    hip_sp = hiplib.load("sp")
    xs = frontend.TorchArrays("cuda:0", "sp")
    xo = frontend.NumpyArrays()
    rng = np.random.default_rng(7 + nlay)
    F = lambda *sh: np.asfortranarray(rng.random(sh))
    ncol, ngpt, top = 70, 16, nlay == 91
    tau, ssa, g = F(ncol, nlay, ngpt) * 0.5, F(ncol, nlay, ngpt) * 0.9, F(ncol, nlay, ngpt) * 0.8
    lay, lev = F(ncol, nlay, ngpt) * 10 + 1, F(ncol, nlay + 1, ngpt) * 10 + 1
    emis, sfc, inc = F(ncol, ngpt) * 0.2 + 0.8, F(ncol, ngpt) * 10, F(ncol, ngpt)
    mu0 = np.asfortranarray(np.repeat((rng.random(ncol) * 0.8 + 0.2)[:, None], nlay, axis=1))
    adir, adif, idir = F(ncol, ngpt) * 0.5, F(ncol, ngpt) * 0.5, F(ncol, ngpt) * 100
    S = xs.asarray
    runs = {
        "lw noscat": (lambda lib, x, c: frontend.rte_lw(lib, x, ncol, nlay, ngpt, top, c(tau), c(lay), c(lev), c(emis), c(sfc), inc_flux=c(inc)),
                      ("flux_up", "flux_dn")),
        "lw rescaled": (lambda lib, x, c: frontend.rte_lw(lib, x, ncol, nlay, ngpt, top, c(tau), c(lay), c(lev), c(emis), c(sfc), ssa=c(ssa),
                                                          g=c(g), inc_flux=c(inc)), ("flux_up", "flux_dn")),
        "lw 2-stream": (lambda lib, x, c: frontend.rte_lw(lib, x, ncol, nlay, ngpt, top, c(tau), c(lay), c(lev), c(emis), c(sfc), ssa=c(ssa),
                                                          g=c(g), inc_flux=c(inc), use_2stream=True), ("gpt_flux_up", "gpt_flux_dn")),
        "sw 2-stream": (lambda lib, x, c: frontend.rte_sw(lib, x, ncol, nlay, ngpt, top, c(tau), c(ssa), c(g), c(mu0), c(idir), c(adir), c(adif)),
                        ("flux_up", "flux_dn", "flux_dir")),
    }
    from oracle import oracle as O

    osp, xo_sp = O.load_c("sp"), frontend.NumpyArrays("sp")
    for name, (fn, keys) in runs.items():
        ref = fn(oracle_c, xo, lambda a: a)
        ref_sp = fn(osp, xo_sp, xo_sp.asarray)  # what single precision costs on these inputs (lw_two_stream: 1e-2)
        out = fn(hip_sp, xs, S)
        for k in keys:
            e_hip = cases.rel_err(xs.to_numpy(out[k]).astype(np.float64), ref[k])
            e_sp = cases.rel_err(np.asarray(ref_sp[k], dtype=np.float64), ref[k])
            assert e_hip <= max(2e-4, 3.0 * e_sp), (name, k, nlay, e_hip, e_sp)


@pytest.mark.parametrize("nlay,top_at_1,nmus,do_jac", [(60, True, 1, False), (60, False, 2, True), (19, True, 1, True),
                                                        (64, False, 1, False), (72, True, 3, False), (77, False, 1, True),
                                                        (100, True, 1, False)])
def test_factored_lw_sources_give_the_same_bits(hip, nlay, top_at_1, nmus, do_jac):
    """The factored LW path (extensions rte_hip_compute_Planck_source_factored -> rte_hip_lw_solver_noscat_factored): the Planck
    fraction per g-point and the Planck function per band instead of lay_source / lev_source.  The factors, expanded with
    rte_hip_expand_factored_sources, are the ABI call's arrays bit for bit (production and direct Planck kernels); the solver on
    the factors returns the fluxes [and the Jacobian] of the ABI solver on the arrays bit for bit -- both orientations, full and
    partial last segments (60 / 19 / 77 layers), 8 / 9 / 10 layers per wave, several angles, a ragged last column tile; 100
    layers take the expand-and-call-the-ABI route (-2)."""
    from rte_rrtmgp_amd import synth
    import torch

    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    ncol = 700
    kd = synth.make_kdist("lw", ngpt=64, nbnd=4)
    atm = synth.make_atmosphere(ncol, nlay, seed=11, kdist=kd, top_at_1=top_at_1)
    go = frontend.GasOptics(hip, kd, xp)
    args = (ncol, nlay, A(atm.play), A(atm.plev), A(atm.tlay), A(atm.tsfc), A(atm.col_gas), A(atm.tlev), atm.top_at_1)
    rng = np.random.default_rng(5)
    emis = A(np.asfortranarray(rng.uniform(0.9, 1.0, (ncol, kd.ngpt))))
    inc = A(np.asfortranarray(rng.uniform(0.0, 5.0, (ncol, kd.ngpt))))
    for direct in (0, 1):
        hiplib.ext_call(hip, "rte_hip_force_direct_gather", ["i"], direct)
        try:
            ref = go.gas_optics_lw(*args, buffers={})
            fac = go.gas_optics_lw(*args, buffers={}, factored_sources=True)
            lay, lev = xp.empty((ncol, nlay, kd.ngpt)), xp.empty((ncol, nlay + 1, kd.ngpt))
            go.expand_factored_sources(ncol, nlay, fac["pfrac"], fac["planck_lay"], fac["planck_lev"], lay, lev)
        finally:
            hiplib.ext_call(hip, "rte_hip_force_direct_gather", ["i"], 0)
        for k, v in (("lay_src", lay), ("lev_src", lev), ("sfc_src", fac["sfc_src"]), ("sfc_src_jac", fac["sfc_src_jac"]), ("tau", fac["tau"])):
            assert torch.equal(v, ref[k]), (k, direct)
        assert float(ref["lev_src"].min()) > 0
    r0 = frontend.rte_lw(hip, xp, ncol, nlay, kd.ngpt, atm.top_at_1, ref["tau"], ref["lay_src"], ref["lev_src"], emis, ref["sfc_src"],
                         n_gauss_angles=nmus, inc_flux=inc, sfc_src_jac=ref["sfc_src_jac"], do_jacobians=do_jac, buffers={})
    r1 = frontend.rte_lw_factored(hip, xp, ncol, nlay, kd.ngpt, kd.nbnd, go.t["band_lims_gpt"], atm.top_at_1, fac["tau"], fac["pfrac"],
                                  fac["planck_lay"], fac["planck_lev"], emis, fac["sfc_src"], n_gauss_angles=nmus, inc_flux=inc,
                                  sfc_src_jac=fac["sfc_src_jac"], do_jacobians=do_jac, buffers={})
    for k in ("flux_up", "flux_dn") + (("flux_up_jac",) if do_jac else ()):
        assert torch.equal(r0[k], r1[k]), (k, float((r0[k] - r1[k]).abs().max()))
    assert float(r0["flux_up"].min()) > 0


@pytest.mark.parametrize("nlay,top_at_1,do_broadband", [(60, True, True), (72, False, True), (37, True, True), (100, True, True),
                                                         (60, False, False)])
def test_implicit_asymmetry_parameter_of_clear_sky_sw(hip, nlay, top_at_1, do_broadband):
    """Clear-sky SW optical properties have g = 0 (combine_abs_and_rayleigh, mo_gas_optics_rrtmgp.F90:1983-2002).  With
    ``implicit_g`` the one-pass SW gas optics does not store that array and rte_sw_solver_2stream is called with g == NULL: tau and
    ssa are the same bits, and so are the fluxes -- on the instance of the segmented kernel that reads nothing for g (8 and 9 layers
    per wave, broadband) and on the paths that get an array of zeros from the library (more layers, spectral output)."""
    from rte_rrtmgp_amd import synth
    import torch

    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    ncol = 700
    kd = synth.make_kdist("sw", ngpt=64, nbnd=4)
    atm = synth.make_atmosphere(ncol, nlay, seed=13, kdist=kd, top_at_1=top_at_1)
    go = frontend.GasOptics(hip, kd, xp)
    args = (ncol, nlay, A(atm.play), A(atm.plev), A(atm.tlay), A(atm.col_gas), A(atm.col_dry))
    ref = go.gas_optics_sw(*args, buffers={}, fuse_rayleigh="all")
    imp = go.gas_optics_sw(*args, buffers={}, fuse_rayleigh="all", implicit_g=True)
    assert imp["g"] is None and float(ref["g"].abs().max()) == 0.0
    for k in ("tau", "ssa", "toa_src"):
        assert torch.equal(ref[k], imp[k]), k
    rng = np.random.default_rng(3)
    mu0 = np.repeat(rng.uniform(-0.2, 1.0, (ncol, 1)), nlay, axis=1)  # night columns among them
    alb = A(np.asfortranarray(rng.uniform(0.05, 0.4, (ncol, kd.ngpt))))
    out = []
    for b in (ref, imp):
        r = frontend.rte_sw(hip, xp, ncol, nlay, kd.ngpt, atm.top_at_1, b["tau"], b["ssa"], b["g"], A(np.asfortranarray(mu0)),
                            b["toa_src"], alb, alb, do_broadband=do_broadband, buffers={})
        out.append(r)
    keys = ("flux_up", "flux_dn", "flux_dir") if do_broadband else ("gpt_flux_up", "gpt_flux_dn", "gpt_flux_dir")
    for k in keys:
        assert torch.equal(out[0][k], out[1][k]), (k, float((out[0][k] - out[1][k]).abs().max()))
    assert float(out[0][keys[0]].max()) > 0


@pytest.mark.parametrize("nlay,top_at_1,nmus,do_jac", [(60, False, 1, False), (72, True, 2, True), (33, True, 1, False)])
def test_deferred_lw_sources_through_the_reference_symbols(hip, nlay, top_at_1, nmus, do_jac):
    """rte_hip_defer_sources(1) (RTE_HIP_DEFER_SOURCES=1 for an unchanged binary): rrtmgp_compute_Planck_source on device arrays
    leaves the factored sources and a record, rte_lw_solver_noscat on exactly these arrays solves from them -- fluxes identical
    BIT FOR BIT to the plain chain.  Misuse: the arrays handed to another entry point (lw_solver_2stream), read after
    rte_hip_sync, or a second solve on them all find lay_source / lev_source expanded, bit-identical to the plain call."""
    import torch
    from rte_rrtmgp_amd import synth

    kd = synth.make_kdist("lw")
    ncol = 1500
    atm = synth.make_atmosphere(ncol, nlay, seed=31, kdist=kd, top_at_1=top_at_1)
    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    go = frontend.GasOptics(hip, kd, xp)
    args = [A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tsfc", "col_gas", "tlev")]
    emis = xp.full((ncol, kd.ngpt), 0.98)

    def chain(solve=True, second=None):
        b = {}
        go.gas_optics_lw(ncol, nlay, *args, atm.top_at_1, buffers=b)
        rb = {}
        if solve:
            frontend.rte_lw(hip, xp, ncol, nlay, kd.ngpt, atm.top_at_1, b["tau"], b["lay_src"], b["lev_src"], emis, b["sfc_src"], buffers=rb,
                            n_gauss_angles=nmus, do_jacobians=do_jac, sfc_src_jac=b["sfc_src_jac"] if do_jac else None)
        if second == "solve_again":
            rb2 = {}
            frontend.rte_lw(hip, xp, ncol, nlay, kd.ngpt, atm.top_at_1, b["tau"], b["lay_src"], b["lev_src"], emis, b["sfc_src"], buffers=rb2)
            rb["again_up"] = rb2["flux_up"]
        if second == "two_stream":
            ssa, g = xp.full((ncol, nlay, kd.ngpt), 0.0), xp.full((ncol, nlay, kd.ngpt), 0.0)
            rb2 = {}
            frontend.rte_lw(hip, xp, ncol, nlay, kd.ngpt, atm.top_at_1, b["tau"], b["lay_src"], b["lev_src"], emis, b["sfc_src"], buffers=rb2,
                            ssa=ssa, g=g, use_2stream=True)
            rb["two_up"] = rb2["flux_up"]
        hiplib.ext_call(hip, "rte_hip_sync", [])
        torch.cuda.synchronize()
        out = {k: xp.to_numpy(v).copy() for k, v in rb.items() if k in ("flux_up", "flux_dn", "flux_up_jac", "again_up", "two_up") and hasattr(v, "detach")}
        out["lay_src"], out["lev_src"] = xp.to_numpy(b["lay_src"]).copy(), xp.to_numpy(b["lev_src"]).copy()
        return out

    plain = {m: chain(True, m) for m in (None, "solve_again", "two_stream")}
    plain_nosolve = chain(False)
    try:
        hiplib.ext_call(hip, "rte_hip_defer_sources", ["i"], 1)
        for m in (None, "solve_again", "two_stream"):
            got = chain(True, m)
            for k, ref in plain[m].items():
                assert np.array_equal(got[k], ref), (m, k)
        got = chain(False)  # never solved: rte_hip_sync materialises
        for k in ("lay_src", "lev_src"):
            assert np.array_equal(got[k], plain_nosolve[k]), k
    finally:
        hiplib.ext_call(hip, "rte_hip_defer_sources", ["i"], 0)


def test_deferred_lw_sources_two_source_objects_and_reuse_after_the_solve(hip):
    """(1) Planck(A), Planck(B) on other, LARGER arrays, solve(A), solve(B): B's call must not take A's factors away
    (A is expanded first; both solves bit-identical to the plain chain).  (2) After the solve the record is no longer trusted:
    the caller overwrites lay_source with its own data (a torch kernel, not the library) and solves again -- the library must
    take the arrays as they are, and nothing may be expanded into the caller's data."""
    import torch
    from rte_rrtmgp_amd import synth

    kd = synth.make_kdist("lw")
    nlay = 60
    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    go = frontend.GasOptics(hip, kd, xp)

    def inputs(ncol, seed):
        atm = synth.make_atmosphere(ncol, nlay, seed=seed, kdist=kd)
        return atm, [A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tsfc", "col_gas", "tlev")], xp.full((ncol, kd.ngpt), 0.98)

    def solve(ncol, atm, b, emis):
        rb = {}
        frontend.rte_lw(hip, xp, ncol, nlay, kd.ngpt, atm.top_at_1, b["tau"], b["lay_src"], b["lev_src"], emis, b["sfc_src"], buffers=rb)
        return rb

    def run(interleaved):
        (na, nb) = (700, 1900)
        atm_a, args_a, em_a = inputs(na, 5)
        atm_b, args_b, em_b = inputs(nb, 6)
        ba, bb = {}, {}
        go.gas_optics_lw(na, nlay, *args_a, atm_a.top_at_1, buffers=ba)
        if interleaved:
            go.gas_optics_lw(nb, nlay, *args_b, atm_b.top_at_1, buffers=bb)
            ra = solve(na, atm_a, ba, em_a)
            rbb = solve(nb, atm_b, bb, em_b)
        else:
            ra = solve(na, atm_a, ba, em_a)
            go.gas_optics_lw(nb, nlay, *args_b, atm_b.top_at_1, buffers=bb)
            rbb = solve(nb, atm_b, bb, em_b)
        hiplib.ext_call(hip, "rte_hip_sync", [])
        torch.cuda.synchronize()
        return {k: xp.to_numpy(v).copy() for k, v in (("a_up", ra["flux_up"]), ("a_dn", ra["flux_dn"]), ("b_up", rbb["flux_up"]),
                                                       ("b_dn", rbb["flux_dn"]), ("a_lay", ba["lay_src"]), ("a_lev", ba["lev_src"]),
                                                       ("b_lay", bb["lay_src"]), ("b_lev", bb["lev_src"]))}

    plain = run(True)
    try:
        hiplib.ext_call(hip, "rte_hip_defer_sources", ["i"], 1)
        for interleaved in (True, False):
            got = run(interleaved)
            for k, ref in plain.items():
                assert np.array_equal(got[k], ref), (interleaved, k)
        # (2) the caller's own data in lay_source after the solve
        ncol = 900
        atm, args, emis = inputs(ncol, 8)
        b = {}
        go.gas_optics_lw(ncol, nlay, *args, atm.top_at_1, buffers=b)
        solve(ncol, atm, b, emis)
        torch.cuda.synchronize()
        mine = torch.full_like(b["lay_src"], 3.25)
        b["lay_src"].copy_(mine)        # not through the library
        b["lev_src"].fill_(3.25)
        r2 = solve(ncol, atm, b, emis)  # must solve from 3.25 everywhere, not from "fractions"
        torch.cuda.synchronize()
        hiplib.ext_call(hip, "rte_hip_defer_sources", ["i"], 0)
        r_plain = solve(ncol, atm, b, emis)
        torch.cuda.synchronize()
        assert torch.equal(r2["flux_up"], r_plain["flux_up"]) and torch.equal(r2["flux_dn"], r_plain["flux_dn"])
        assert torch.equal(b["lay_src"], mine)  # and nothing expanded into the caller's data
    finally:
        hiplib.ext_call(hip, "rte_hip_defer_sources", ["i"], 0)


@pytest.mark.gpu
@pytest.mark.parametrize("nlay,top_at_1", [(60, False), (60, True), (55, True), (57, False), (59, True)])
def test_sw_two_stream_on_segments_of_two_lengths(hip, oracle_c, nlay, top_at_1):
    """rte_sw_solver_2stream at 57 ... 60 layers runs on segments of seven and eight layers (sw_2stream_seg_mixed_kernel: no wave
    computes neutral slots; 55 layers: the 8 x 8 kernel): against the C oracle -- night columns, a diffuse boundary condition, ncol not a
    multiple of 64, several g-point groups -- against the 8 + 8 kernel (rte_hip_sw_mixed_segments(0)), and with g == NULL
    against an array of zeros (bit-identical)."""
    import torch

    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    rng = np.random.default_rng(100 + nlay)
    F = lambda *sh: np.asfortranarray(rng.random(sh))
    ncol, ngpt = 331, 48
    tau, ssa, g = F(ncol, nlay, ngpt) * 3.0, F(ncol, nlay, ngpt) * 0.999, F(ncol, nlay, ngpt) * 0.9 - 0.1
    tau[:, ::7, :] *= 1e-4  # optically thin layers among them
    mu0 = np.asfortranarray(np.repeat((rng.random(ncol) * 1.2 - 0.2)[:, None], nlay, axis=1))  # some <= 0
    adir, adif, idir, idif = F(ncol, ngpt), F(ncol, ngpt), F(ncol, ngpt) * 100, F(ncol, ngpt) * 10
    ref = frontend.rte_sw(oracle_c, frontend.NumpyArrays(), ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, idir, adir, adif, inc_flux_dif=idif)
    dev = [A(x) for x in (tau, ssa, g, mu0, idir, adir, adif)]
    out = frontend.rte_sw(hip, xp, ncol, nlay, ngpt, top_at_1, *dev, inc_flux_dif=A(idif), buffers={})
    for k in ("flux_up", "flux_dn", "flux_dir"):
        assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-12, (k, nlay, top_at_1)
    hiplib.ext_call(hip, "rte_hip_sw_mixed_segments", ["i"], 0)
    try:
        old = frontend.rte_sw(hip, xp, ncol, nlay, ngpt, top_at_1, *dev, inc_flux_dif=A(idif), buffers={})
    finally:
        hiplib.ext_call(hip, "rte_hip_sw_mixed_segments", ["i"], 1)
    for k in ("flux_up", "flux_dn", "flux_dir"):
        assert cases.rel_err(xp.to_numpy(out[k]), xp.to_numpy(old[k])) <= 1e-12, (k, nlay, top_at_1)
    # g == NULL (clear-sky SW): the instance that reads nothing for g, against an array of zeros
    zeros = torch.zeros_like(dev[2])
    z = frontend.rte_sw(hip, xp, ncol, nlay, ngpt, top_at_1, dev[0], dev[1], zeros, *dev[3:], buffers={})
    n = frontend.rte_sw(hip, xp, ncol, nlay, ngpt, top_at_1, dev[0], dev[1], None, *dev[3:], buffers={})
    for k in ("flux_up", "flux_dn", "flux_dir"):
        assert torch.equal(z[k], n[k]), k
    assert float(n["flux_dn"].max()) > 0


@pytest.mark.parametrize("nlay,top_at_1,nang,jac", [(60, False, 1, False), (60, True, 2, True), (57, True, 1, True), (58, False, 3, False), (59, True, 1, False)])
def test_lw_noscat_on_segments_of_two_lengths(hip, oracle_c, nlay, top_at_1, nang, jac):
    """rte_lw_solver_noscat (broadband, no rescaling) at 57 ... 60 layers runs on segments of seven and eight layers
    (lw_noscat_seg_mixed_kernel: no wave requests rows for neutral slots): against the C oracle -- several angles, Jacobian, an
    incident flux, ncol not a multiple of 64 -- and against the 8 x 8 kernel (rte_hip_lw_mixed_segments(0))."""
    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    rng = np.random.default_rng(200 + nlay)
    F = lambda *sh: np.asfortranarray(rng.random(sh))
    ncol, ngpt = 331, 48
    tau = F(ncol, nlay, ngpt) * 2.0
    tau[:, ::5, :] *= 1e-5  # optically thin layers among them (the Clough source's series branch)
    lay, lev = F(ncol, nlay, ngpt) * 10 + 1, F(ncol, nlay + 1, ngpt) * 10 + 1
    emis, sfc, inc, sj = F(ncol, ngpt) * 0.2 + 0.8, F(ncol, ngpt) * 10, F(ncol, ngpt), F(ncol, ngpt)
    kw = dict(inc_flux=inc, n_gauss_angles=nang)
    if jac:
        kw.update(sfc_src_jac=sj, do_jacobians=True)
    ref = frontend.rte_lw(oracle_c, frontend.NumpyArrays(), ncol, nlay, ngpt, top_at_1, tau, lay, lev, emis, sfc, **kw)
    kwd = dict(inc_flux=A(inc), n_gauss_angles=nang)
    if jac:
        kwd.update(sfc_src_jac=A(sj), do_jacobians=True)
    dev = [A(x) for x in (tau, lay, lev, emis, sfc)]
    out = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, top_at_1, *dev, buffers={}, **kwd)
    keys = ("flux_up", "flux_dn") + (("flux_up_jac",) if jac else ())
    for k in keys:
        assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-12, (k, nlay, top_at_1)
    hiplib.ext_call(hip, "rte_hip_lw_mixed_segments", ["i"], 0)
    try:
        old = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, top_at_1, *dev, buffers={}, **kwd)
    finally:
        hiplib.ext_call(hip, "rte_hip_lw_mixed_segments", ["i"], 1)
    for k in keys:
        assert cases.rel_err(xp.to_numpy(out[k]), xp.to_numpy(old[k])) <= 1e-12, (k, nlay, top_at_1)


@pytest.mark.parametrize("ncol", [1, 63, 65, 129])
def test_segments_of_two_lengths_with_few_columns(hip, oracle_c, ncol):
    """The 60-layer solvers (segments of seven and eight layers) on calls of one column, of one short of a wave, one more than a
    wave, and of two waves and one: the lanes past the last column compute on a clamped column and store nothing."""
    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    rng = np.random.default_rng(300 + ncol)
    F = lambda *sh: np.asfortranarray(rng.random(sh))
    nlay, ngpt = 60, 32
    tau, ssa, g = F(ncol, nlay, ngpt) * 3.0, F(ncol, nlay, ngpt) * 0.999, F(ncol, nlay, ngpt) * 0.9 - 0.1
    mu0 = np.asfortranarray(np.repeat((rng.random(ncol) * 1.2 - 0.2)[:, None], nlay, axis=1))
    adir, adif, idir = F(ncol, ngpt), F(ncol, ngpt), F(ncol, ngpt) * 100
    for top_at_1 in (False, True):
        ref = frontend.rte_sw(oracle_c, frontend.NumpyArrays(), ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, idir, adir, adif)
        out = frontend.rte_sw(hip, xp, ncol, nlay, ngpt, top_at_1, A(tau), A(ssa), A(g), A(mu0), A(idir), A(adir), A(adif), buffers={})
        for k in ("flux_up", "flux_dn", "flux_dir"):
            assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-12, (k, ncol, top_at_1)
        lay, lev = F(ncol, nlay, ngpt) * 10 + 1, F(ncol, nlay + 1, ngpt) * 10 + 1
        emis, sfc = F(ncol, ngpt) * 0.2 + 0.8, F(ncol, ngpt) * 10
        ref = frontend.rte_lw(oracle_c, frontend.NumpyArrays(), ncol, nlay, ngpt, top_at_1, tau, lay, lev, emis, sfc)
        out = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, top_at_1, A(tau), A(lay), A(lev), A(emis), A(sfc), buffers={})
        for k in ("flux_up", "flux_dn"):
            assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-12, (k, ncol, top_at_1)


@pytest.mark.parametrize("nlay,top_at_1", [(72, False), (67, True)])
def test_sw_two_stream_cosine_that_varies_with_the_layer(hip, oracle_c, nlay, top_at_1):
    """mu0 is an (ncol, nlay) array of the interface: at 65 ... 72 layers (nine per wave: one parked value per layer, the reciprocal
    formed per g-point) against the C oracle with mu0 per column, with mu0 that changes from layer to layer (spherical geometry)
    and with a single deviating element, night columns included; with and without the g array.  (Round 6 measured an instance
    for a layer-independent cosine picked by a device flag: 13.9 against 12.9 ms at 1e5 x 72 x 224 -- dropped, the test stays.)"""
    import torch

    xp = frontend.TorchArrays("cuda:0")
    A = xp.asarray
    rng = np.random.default_rng(400 + nlay)
    F = lambda *sh: np.asfortranarray(rng.random(sh))
    ncol, ngpt = 203, 40
    tau, ssa, g = F(ncol, nlay, ngpt) * 3.0, F(ncol, nlay, ngpt) * 0.999, F(ncol, nlay, ngpt) * 0.9 - 0.1
    adir, adif, idir, idif = F(ncol, ngpt), F(ncol, ngpt), F(ncol, ngpt) * 100, F(ncol, ngpt) * 10
    per_col = np.asfortranarray(np.repeat((rng.random(ncol) * 1.2 - 0.2)[:, None], nlay, axis=1))
    varying = np.asfortranarray(np.clip(per_col + (rng.random((ncol, nlay)) - 0.5) * 0.05, -0.3, 1.0))
    one_off = per_col.copy(order="F"); one_off[ncol // 2, nlay // 3] += 0.01  # a single element decides the flag
    for mu0 in (per_col, varying, one_off):
        ref = frontend.rte_sw(oracle_c, frontend.NumpyArrays(), ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, idir, adir, adif, inc_flux_dif=idif)
        out = frontend.rte_sw(hip, xp, ncol, nlay, ngpt, top_at_1, A(tau), A(ssa), A(g), A(mu0), A(idir), A(adir), A(adif), inc_flux_dif=A(idif), buffers={})
        for k in ("flux_up", "flux_dn", "flux_dir"):
            assert cases.rel_err(xp.to_numpy(out[k]), ref[k]) <= 1e-12, (k, nlay, top_at_1)
        zeros = torch.zeros((ngpt, nlay, ncol), dtype=torch.float64, device="cuda")
        z = frontend.rte_sw(hip, xp, ncol, nlay, ngpt, top_at_1, A(tau), A(ssa), zeros, A(mu0), A(idir), A(adir), A(adif), buffers={})
        n = frontend.rte_sw(hip, xp, ncol, nlay, ngpt, top_at_1, A(tau), A(ssa), None, A(mu0), A(idir), A(adir), A(adif), buffers={})
        for k in ("flux_up", "flux_dn", "flux_dir"):
            assert torch.equal(z[k], n[k]), k
