"""The reference's UNCHANGED Fortran frontend, built in extern mode (kernel interface modules
rte/kernels/api/*.F90, rrtmgp/kernels/api/*.F90 + the whole frontend, compiled by oracle/build_extern.sh with
flang) and linked against librte_rrtmgp_hip.so: SURVEY.md section 8f-4.

The reference's three data-free unit-test programs (tests/rte_lw_solver_unit_tests.F90, rte_sw_solver_unit_tests.F90,
rte_optic_prop_unit_tests.F90) are built that way into oracle/_ref/bin/ (binaries only; they travel to the GPU box,
the reference's sources do not).  Running them drives the real ``rte_lw`` / ``rte_sw`` / ``ty_optical_props`` classes
-- host arrays, decoy arguments, subsetting, increments, delta scaling, Jacobians, multi-angle quadrature -- through
the library's host-pointer staging path, and they check themselves (gray radiative equilibrium, invariances); a failed
check ends in ``error stop`` (non-zero exit status).
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "bin")
PROGRAMS = {
    "rte_lw_solver_unit_tests": ["RTE LW solver unit tests done", "Jacobian accurate to within", "Specified transport angle"],
    "rte_sw_solver_unit_tests": ["RTE SW solver unit tests done", "Linear in TOA flux"],
    "rte_optic_prop_unit_tests": ["Optical properties unit testing finished", "Delta scaling"],
}


def _run(path):
    # flang keeps automatic arrays on the stack
    return subprocess.run(f"ulimit -s unlimited 2>/dev/null; exec '{path}'", shell=True, capture_output=True, text=True,
                          timeout=600, cwd=ROOT)


def test_extern_symbol_check_recorded():
    """oracle/build_extern.sh (run by __graft_entry__.build() where /root/reference exists) refuses to finish unless
    every kernel symbol the extern-mode frontend references is exported by the HIP library; it records the list."""
    rec = os.path.join(ROOT, "oracle", "_ref", "extern_symbols_ok.txt")
    if not os.path.exists(rec):
        pytest.skip("oracle/_ref/extern_symbols_ok.txt absent: the reference tree was not available to build it")
    lines = open(rec).read().split()
    assert int(lines[0]) == len(lines) - 1 >= 30
    for needed in ("rrtmgp_interpolation", "rrtmgp_compute_tau_absorption", "rrtmgp_compute_Planck_source",
                   "rte_lw_solver_noscat", "rte_sw_solver_2stream", "zero_array_3D"):
        assert needed in lines


@pytest.mark.parametrize("prog", list(PROGRAMS))
def test_cpu_reference_build_of_the_programs_passes(prog):
    """Baseline: the same objects linked against the reference's own CPU kernels print their success messages."""
    path = os.path.join(BIN, prog + "_cpuref")
    if not os.path.exists(path):
        pytest.skip("reference CPU build of the unit-test programs absent")
    r = _run(path)
    assert r.returncode == 0, r.stdout + r.stderr
    for msg in PROGRAMS[prog]:
        assert msg in r.stdout, (msg, r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("prog", list(PROGRAMS))
def test_reference_unit_test_programs_on_the_hip_library(prog):
    path = os.path.join(BIN, prog)
    assert os.path.exists(path), f"{path} missing: run oracle/build_extern.sh where /root/reference exists"
    r = _run(path)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for msg in PROGRAMS[prog]:
        assert msg in r.stdout, (msg, r.stdout)
    # the same lines, in the same order, as the CPU reference build prints (only the Jacobian accuracy figure may differ)
    ref = os.path.join(BIN, prog + "_cpuref")
    if os.path.exists(ref):
        rr = _run(ref)

        def strip(s):
            return [ln.strip() for ln in s.splitlines() if ln.strip() and "accurate to within" not in ln]

        assert strip(r.stdout) == strip(rr.stdout)


# ---- the gas-optics frontend end to end ---------------------------------------------------
# oracle/ref_frontend_driver.F90 (ours): raw table -> the reference's k%load -> k%gas_optics -> rte_lw / rte_sw per block of
# columns with ty_fluxes_broadband, on host arrays.  ONE object file, linked once against librte_rrtmgp_hip.so (+ shim)
# and once against the reference's CPU kernels (oracle/build_extern.sh).
import sys  # noqa: E402

import numpy as np  # noqa: E402

sys.path.insert(0, ROOT)
import stream_io  # noqa: E402
from rte_rrtmgp_amd import frontend, kdist_load, synth  # noqa: E402

GASES = list(synth.GAS_NAMES)


def _frontend_case(tmp_path, kind, ncol, nlay, block, top_at_1, col_dry, tlev, ngpt=64, nbnd=4, seed=3, checks=True, nrep=1,
                   n_gauss=1, variant=0, **raw_kw):
    raw = kdist_load.synth_raw(kind, ngpt=ngpt, nbnd=nbnd, **raw_kw)
    kd = kdist_load.init_from_raw(raw, GASES)
    kd.scalars.pop("gas_names")
    atm = synth.make_atmosphere(ncol, nlay, seed=seed, kdist=kd, ngas=kd.ngas, top_at_1=top_at_1)
    kf, af = str(tmp_path / "k.bin"), str(tmp_path / "a.bin")
    stream_io.write_kdist_stream(kf, raw, kind == "lw")
    stream_io.write_atmosphere_stream(af, atm, kind == "lw", block=block, use_col_dry=col_dry, use_tlev=tlev, checks=checks,
                                      nrep=nrep, n_gauss=n_gauss, variant=variant)
    return raw, kd, atm, kf, af


def _have(binary):
    return os.path.exists(os.path.join(BIN, binary))


@pytest.mark.gpu
def test_fortran_binding_of_the_factored_sources():
    """INTEGRATION.md section 4a is compiled, not only printed: shim/mo_rte_hip_factored.F90 (the interface block a maintainer
    would add) beside the reference's own kernel interface modules, and oracle/factored_binding_driver.F90 calling the
    factored pair and the reference pair on the same host arrays -- fluxes and expanded sources bit-identical."""
    path = os.path.join(BIN, "factored_binding_driver")
    assert os.path.exists(path), f"{path} missing: run oracle/build_extern.sh where /root/reference exists"
    for env in ({}, {"RTE_HIP_HOST_MIRROR": "1"}):
        r = subprocess.run(f"ulimit -s unlimited 2>/dev/null; exec '{path}'", shell=True, capture_output=True, text=True, timeout=600,
                           cwd=ROOT, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "factored binding: PASS" in r.stdout, r.stdout


@pytest.mark.parametrize("kind,top_at_1", [("lw", False), ("lw", True), ("sw", False), ("sw", True)])
def test_python_mirror_of_the_frontend_matches_the_reference_frontend(kind, top_at_1, tmp_path):
    """rte-rrtmgp_amd/frontend.py (the call sequence bench.py and the GPU tests drive) on the C oracle against the
    reference's own Fortran frontend on the reference's CPU kernels: same fluxes, bit for bit."""
    if not _have("ref_frontend_driver_cpuref"):
        pytest.skip("oracle/_ref/bin/ref_frontend_driver_cpuref absent (needs /root/reference + flang)")
    from oracle import oracle as O

    ncol, nlay = 48, 20
    raw, kd, atm, kf, af = _frontend_case(tmp_path, kind, ncol, nlay, 16, top_at_1, True, True)
    ref, _ = stream_io.run_frontend_driver("ref_frontend_driver_cpuref", kf, af, str(tmp_path / "o.bin"), GASES, ncol, nlay, kind == "lw")
    c, xp = O.load_c(), frontend.NumpyArrays()
    A = xp.asarray
    go = frontend.GasOptics(c, kd, xp)
    if kind == "lw":
        b = go.gas_optics_lw(ncol, nlay, A(atm.play), A(atm.plev), A(atm.tlay), A(atm.tsfc), A(atm.col_gas), A(atm.tlev), atm.top_at_1)
        r = frontend.rte_lw(c, xp, ncol, nlay, kd.ngpt, atm.top_at_1, b["tau"], b["lay_src"], b["lev_src"],
                            xp.full((ncol, kd.ngpt), 0.98), b["sfc_src"])
        pairs = {"flux_up": "flux_up", "flux_dn": "flux_dn"}
    else:
        b = go.gas_optics_sw(ncol, nlay, A(atm.play), A(atm.plev), A(atm.tlay), A(atm.col_gas), A(atm.col_dry))
        r = frontend.rte_sw(c, xp, ncol, nlay, kd.ngpt, atm.top_at_1, b["tau"], b["ssa"], b["g"], xp.full((ncol, nlay), 0.86),
                            b["toa_src"], xp.full((ncol, kd.ngpt), 0.06), xp.full((ncol, kd.ngpt), 0.06))
        pairs = {"flux_up": "flux_up", "flux_dn": "flux_dn", "flux_dn_dir": "flux_dir"}
    for k, kk in pairs.items():
        assert np.max(np.abs(ref[k] - r[kk])) <= 1e-13 * np.max(np.abs(ref[k])), k


F4_CASES = [
    # kind, top_at_1, col_dry given, tlev given, block size
    ("lw", False, True, True, 8), ("lw", True, False, False, 8), ("lw", False, False, True, 512), ("lw", True, True, False, 512),
    ("sw", False, True, True, 8), ("sw", True, False, True, 8), ("sw", False, False, True, 512), ("sw", True, True, True, 512),
]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,top_at_1,col_dry,tlev,block", F4_CASES)
@pytest.mark.parametrize("mirror", [False, True], ids=["staged", "host-mirror"])
def test_reference_gas_optics_frontend_on_the_hip_library(kind, top_at_1, col_dry, tlev, block, mirror, tmp_path):
    """Row f4: load -> gas_optics (interpolation, zero_array, compute_tau_absorption [, tau_rayleigh], Planck source,
    col_dry through the Fortran shim) -> rte_lw / rte_sw of the UNCHANGED frontend on librte_rrtmgp_hip.so, against the same
    program on the reference's CPU kernels: 512 columns x 60 layers, g256 / g224-shaped tables, 8-column blocks (the
    reference's usage; small-problem kernels) and one 512-column block (production kernels), both vertical orientations,
    with and without col_dry / tlev.  Tolerance 1e-10 relative to the largest flux (contract: 1e-6)."""
    assert _have("ref_frontend_driver"), "oracle/_ref/bin/ref_frontend_driver missing: run oracle/build_extern.sh"
    if not _have("ref_frontend_driver_cpuref"):
        pytest.skip("reference CPU build of the driver absent")
    ncol, nlay = 512, 60
    ngpt, nbnd = (256, 16) if kind == "lw" else (224, 14)
    # host-mirror mode needs the frontend's value checks off (rte_config_checks, rte/frontend/mo_rte_config.F90:25-49): they
    # scan tau on the HOST (optical_props%validate, mo_rte_lw.F90:330), which the mode leaves on the device
    raw, kd, atm, kf, af = _frontend_case(tmp_path, kind, ncol, nlay, block, top_at_1, col_dry, tlev, ngpt=ngpt, nbnd=nbnd,
                                          nminor_lower=4 * nbnd, nminor_upper=2 * nbnd + 3, seed=11, checks=not mirror)
    ref, _ = stream_io.run_frontend_driver("ref_frontend_driver_cpuref", kf, af, str(tmp_path / "ref.bin"), GASES, ncol, nlay, kind == "lw")
    env = {"RTE_HIP_HOST_MIRROR": "1"} if mirror else {"RTE_HIP_HOST_MIRROR": "0"}
    out, log = stream_io.run_frontend_driver("ref_frontend_driver", kf, af, str(tmp_path / "hip.bin"), GASES, ncol, nlay, kind == "lw", env=env)
    worst = 0.0
    for k in ref:
        assert np.all(np.isfinite(out[k])), k
        err = float(np.max(np.abs(out[k] - ref[k])) / np.max(np.abs(ref[k])))
        worst = max(worst, err)
        assert err <= 1e-10, (k, err)
    print(f"f4 {kind} top_at_1={top_at_1} col_dry={col_dry} tlev={tlev} block={block} mirror={mirror}: worst {worst:.2e}")


def test_openmp_build_of_the_driver_matches_the_serial_one_on_the_reference_kernels(tmp_path):
    """The same driver source built with -fopenmp (blocks dealt to host threads, one set of frontend objects per thread):
    same fluxes as the serial program -- the reference frontend is re-entrant for calls on distinct buffers."""
    if not (_have("ref_frontend_driver_cpuref") and _have("ref_frontend_driver_omp_cpuref")):
        pytest.skip("reference CPU builds of the driver absent")
    ncol, nlay = 96, 20
    raw, kd, atm, kf, af = _frontend_case(tmp_path, "lw", ncol, nlay, 8, False, True, True)
    ref, _ = stream_io.run_frontend_driver("ref_frontend_driver_cpuref", kf, af, str(tmp_path / "o1.bin"), GASES, ncol, nlay, True)
    omp, log = stream_io.run_frontend_driver("ref_frontend_driver_omp_cpuref", kf, af, str(tmp_path / "o2.bin"), GASES, ncol, nlay, True,
                                             env={"OMP_NUM_THREADS": "3"})
    assert "3 host threads" in log
    for k in ref:
        assert np.array_equal(ref[k], omp[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["lw", "sw"])
def test_openmp_frontend_threads_on_their_own_contexts(kind, tmp_path):
    """Four host threads of the OpenMP build, each driving the unchanged frontend on its own blocks; RTE_HIP_THREAD_CONTEXTS=1
    gives every thread a context of its own (stream, arena, mirrors), host-mirror mode on: fluxes equal the serial staged run
    of the same block size bit for bit."""
    assert _have("ref_frontend_driver") and _have("ref_frontend_driver_omp")
    ncol, nlay, block = 4096, 60, 512
    ngpt, nbnd = (256, 16) if kind == "lw" else (224, 14)
    raw, kd, atm, kf, af = _frontend_case(tmp_path, kind, ncol, nlay, block, False, True, True, ngpt=ngpt, nbnd=nbnd,
                                          nminor_lower=4 * nbnd, nminor_upper=2 * nbnd + 3, seed=12, checks=False, nrep=2)
    ser, _ = stream_io.run_frontend_driver("ref_frontend_driver", kf, af, str(tmp_path / "s.bin"), GASES, ncol, nlay, kind == "lw",
                                           env={"RTE_HIP_HOST_MIRROR": "0"})
    omp, log = stream_io.run_frontend_driver("ref_frontend_driver_omp", kf, af, str(tmp_path / "p.bin"), GASES, ncol, nlay, kind == "lw",
                                             env={"RTE_HIP_HOST_MIRROR": "1", "RTE_HIP_THREAD_CONTEXTS": "1", "OMP_NUM_THREADS": "4"})
    assert "4 host threads" in log
    for k in ser:
        assert np.array_equal(ser[k], omp[k]), (k, float(np.max(np.abs(ser[k] - omp[k]))))


# ---- the invariances of the reference's tests/check_equivalence.F90 (which needs rrtmgp-data) on synthetic streams:
#      oracle/ref_equivalence_driver.F90, through the reference's unchanged frontend classes
N_EQUIV_CHECKS = {"lw": 18, "sw": 18}


def _equivalence(tmp_path, binary, kind, ncol, nlay, top_at_1, env=None):
    raw, kd, atm, kf, af = _frontend_case(tmp_path, kind, ncol, nlay, ncol, top_at_1, False, True)
    rc, checks, log = stream_io.run_equivalence_driver(binary, kf, af, GASES, env=env)
    return rc, checks, log


@pytest.mark.parametrize("kind,top_at_1", [("lw", False), ("sw", True)])
def test_equivalence_driver_on_the_reference_kernels(kind, top_at_1, tmp_path):
    """The driver itself, on the reference's own CPU kernels: every invariance holds within the reference's tolerances
    (2 ... 30 spacings) -- the baseline for the same program on the HIP library."""
    if not _have("ref_equivalence_driver_cpuref"):
        pytest.skip("oracle/_ref/bin/ref_equivalence_driver_cpuref absent (needs /root/reference + flang)")
    rc, checks, log = _equivalence(tmp_path, "ref_equivalence_driver_cpuref", kind, 24, 20, top_at_1)
    assert rc == 0 and "ref_equivalence_driver ok" in log, log[-3000:]
    assert len(checks) == N_EQUIV_CHECKS[kind] and all(ok for _, _, ok in checks.values()), checks


@pytest.mark.gpu
@pytest.mark.parametrize("kind,top_at_1,ncol", [("lw", False, 96), ("lw", True, 40), ("sw", False, 96), ("sw", True, 40),
                                                 ("lw", False, 1024), ("sw", True, 1024)])  # 1024: the production kernels
def test_equivalence_invariances_on_the_hip_library(kind, top_at_1, ncol, tmp_path):
    """check_equivalence's invariances through the reference's frontend ON THE HIP LIBRARY, with the reference's own
    tolerances: net fluxes, vertical flip, column subsets through get_subset (extract_subset kernels), halving + self-increment,
    transparent 1scl / 2str / nstr increments, Jacobian; SW: flip, TSI scaling, increments."""
    if not _have("ref_equivalence_driver"):
        pytest.skip("oracle/_ref/bin/ref_equivalence_driver absent (needs /root/reference + flang at build time)")
    rc, checks, log = _equivalence(tmp_path, "ref_equivalence_driver", kind, ncol, 24, top_at_1)
    assert rc == 0 and "ref_equivalence_driver ok" in log, log[-3000:]
    assert len(checks) == N_EQUIV_CHECKS[kind] and all(ok for _, _, ok in checks.values()), checks


# ---- all-sky (BASELINE configs[3]) through the reference's frontend: ty_cloud_optics_rrtmgp%load /
#      %cloud_optics (rrtmgp_compute_cld_from_table) -> gas_optics -> delta_scale -> increment (by band) -> rte_lw / rte_sw
def _allsky_case(tmp_path, kind, ncol, nlay, block, top_at_1, checks=True, **kw):
    raw, kd, atm, kf, af = _frontend_case(tmp_path, kind, ncol, nlay, block, top_at_1, True, True, checks=checks, **kw)
    tb = synth.make_cloud_optics(kd.nbnd)
    clouds = synth.make_cloud_field(atm, tb)
    cf = str(tmp_path / "c.bin")
    stream_io.write_cloud_stream(cf, tb, clouds)
    return raw, kd, atm, tb, clouds, kf, af, cf


@pytest.mark.parametrize("kind,top_at_1", [("lw", False), ("sw", True)])
def test_python_allsky_mirror_matches_the_reference_allsky_frontend(kind, top_at_1, tmp_path):
    """frontend.allsky_lw / allsky_sw (unfused, on the C oracle) against the reference's own cloud-optics class, delta scaling
    and by-band increments driven by oracle/ref_frontend_driver.F90 on the reference's CPU kernels."""
    if not _have("ref_frontend_driver_cpuref"):
        pytest.skip("oracle/_ref/bin/ref_frontend_driver_cpuref absent (needs /root/reference + flang)")
    from oracle import oracle as O

    ncol, nlay = 48, 24
    raw, kd, atm, tb, clouds, kf, af, cf = _allsky_case(tmp_path, kind, ncol, nlay, 16, top_at_1)
    assert clouds["lwp"].max() > 0 and clouds["iwp"].max() > 0
    ref, _ = stream_io.run_frontend_driver("ref_frontend_driver_cpuref", kf, af, str(tmp_path / "o.bin"), GASES, ncol, nlay, kind == "lw",
                                           cloud_file=cf)
    clr, _ = stream_io.run_frontend_driver("ref_frontend_driver_cpuref", kf, af, str(tmp_path / "o2.bin"), GASES, ncol, nlay, kind == "lw")
    assert np.max(np.abs(ref["flux_dn"] - clr["flux_dn"])) > 1.0  # the clouds matter
    c, xp = O.load_c(), frontend.NumpyArrays()
    A = xp.asarray
    go, co = frontend.GasOptics(c, kd, xp), frontend.CloudOptics(c, tb, xp)
    a = {k: A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas", "col_dry")}
    a["top_at_1"] = atm.top_at_1
    cl = {k: A(v) for k, v in clouds.items()}
    if kind == "lw":
        _, _, r = frontend.allsky_lw(c, xp, go, co, ncol, nlay, a, cl, xp.full((ncol, kd.ngpt), 0.98), fuse=False)
        pairs = {"flux_up": "flux_up", "flux_dn": "flux_dn"}
    else:
        _, _, r = frontend.allsky_sw(c, xp, go, co, ncol, nlay, a, cl, xp.full((ncol, nlay), 0.86), xp.full((ncol, kd.ngpt), 0.06), fuse=False)
        pairs = {"flux_up": "flux_up", "flux_dn": "flux_dn", "flux_dn_dir": "flux_dir"}
    for k, kk in pairs.items():
        assert np.max(np.abs(ref[k] - r[kk])) <= 1e-13 * np.max(np.abs(ref[k])), k


@pytest.mark.gpu
@pytest.mark.parametrize("kind,top_at_1,block", [("lw", False, 8), ("lw", True, 512), ("sw", False, 512), ("sw", True, 8)])
@pytest.mark.parametrize("mirror", [False, True], ids=["staged", "host-mirror"])
def test_reference_allsky_frontend_on_the_hip_library(kind, top_at_1, block, mirror, tmp_path):
    """The all-sky example's block loop through the UNCHANGED frontend on librte_rrtmgp_hip.so against the same program on
    the reference's CPU kernels: 512 columns x 72 layers, g256 / g224-shaped tables, clouds in two columns out of three
    (rrtmgp_compute_cld_from_table, rte_delta_scale_2str_k, rte_inc_*_bybnd behind the reference's classes)."""
    assert _have("ref_frontend_driver"), "oracle/_ref/bin/ref_frontend_driver missing: run oracle/build_extern.sh"
    if not _have("ref_frontend_driver_cpuref"):
        pytest.skip("reference CPU build of the driver absent")
    ncol, nlay = 512, 72
    ngpt, nbnd = (256, 16) if kind == "lw" else (224, 14)
    raw, kd, atm, tb, clouds, kf, af, cf = _allsky_case(tmp_path, kind, ncol, nlay, block, top_at_1, checks=not mirror, ngpt=ngpt,
                                                       nbnd=nbnd, nminor_lower=4 * nbnd, nminor_upper=2 * nbnd + 3, seed=13)
    ref, _ = stream_io.run_frontend_driver("ref_frontend_driver_cpuref", kf, af, str(tmp_path / "ref.bin"), GASES, ncol, nlay, kind == "lw",
                                           cloud_file=cf)
    env = {"RTE_HIP_HOST_MIRROR": "1" if mirror else "0"}
    out, _ = stream_io.run_frontend_driver("ref_frontend_driver", kf, af, str(tmp_path / "hip.bin"), GASES, ncol, nlay, kind == "lw", env=env,
                                           cloud_file=cf)
    for k in ref:
        assert np.all(np.isfinite(out[k])), k
        err = float(np.max(np.abs(out[k] - ref[k])) / np.max(np.abs(ref[k])))
        assert err <= 1e-10, (k, err)


# ---- the configurations of the reference's tests/check_variants.F90 through the unchanged frontend (row f2)
VARIANTS = [
    # kind, variant, n_gauss, clouds, what
    ("lw", 0, 3, False, "three quadrature angles"),
    ("lw", 1, 1, False, "optimal transport angles (compute_optimal_angles -> lw_Ds)"),
    ("lw", 2, 2, False, "fluxes by band, two angles (spectral output + rte_sum_byband)"),
    ("sw", 2, 1, False, "fluxes by band"),
    ("lw", 3, 1, True, "two-stream clouds: Tang rescaling"),
    ("lw", 4, 1, True, "two-stream clouds: lw_solver_2stream"),
    ("lw", 5, 1, False, "incident diffuse flux at the top"),
    ("sw", 5, 1, False, "incident diffuse flux at the top"),
]


def _variant_run(tmp_path, binary, kind, variant, n_gauss, clouds, ncol, nlay, block, env=None, **kw):
    raw, kd, atm, kf, af = _frontend_case(tmp_path, kind, ncol, nlay, block, False, True, True, n_gauss=n_gauss, variant=variant, **kw)
    cf = None
    if clouds:
        tb = synth.make_cloud_optics(kd.nbnd)
        cf = str(tmp_path / "c.bin")
        stream_io.write_cloud_stream(cf, tb, synth.make_cloud_field(atm, tb))
    out, _ = stream_io.run_frontend_driver(binary, kf, af, str(tmp_path / (binary + ".bin")), GASES, ncol, nlay, kind == "lw", env=env,
                                           cloud_file=cf)
    return out


def test_frontend_variants_run_on_the_reference_kernels(tmp_path):
    """The driver's variants on the reference's CPU kernels: each runs and differs from the default calculation (so the
    GPU comparison below is not a comparison of two default runs)."""
    if not _have("ref_frontend_driver_cpuref"):
        pytest.skip("oracle/_ref/bin/ref_frontend_driver_cpuref absent (needs /root/reference + flang)")
    base = {}
    for kind in ("lw", "sw"):
        base[kind] = _variant_run(tmp_path, "ref_frontend_driver_cpuref", kind, 0, 1, False, 24, 20, 8)
    for kind, variant, n_gauss, clouds, what in VARIANTS:
        out = _variant_run(tmp_path, "ref_frontend_driver_cpuref", kind, variant, n_gauss, clouds, 24, 20, 8)
        d = float(np.max(np.abs(out["flux_up"] - base[kind]["flux_up"])) / np.max(np.abs(base[kind]["flux_up"])))
        if variant == 2 and n_gauss == 1:  # by-band fluxes summed over the bands ARE the broadband fluxes
            assert d <= 1e-13, (what, d)
        else:
            assert d > 1e-6, (what, d)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,variant,n_gauss,clouds,what", VARIANTS, ids=[v[4].split(":")[0].split("(")[0].strip().replace(" ", "-") + "-" + v[0] for v in VARIANTS])
def test_frontend_variants_on_the_hip_library(kind, variant, n_gauss, clouds, what, tmp_path):
    """check_variants' configurations through the reference's frontend on librte_rrtmgp_hip.so against the CPU build:
    512 columns x 60 layers in one block (production kernels) with g256 / g224-shaped tables."""
    assert _have("ref_frontend_driver"), "oracle/_ref/bin/ref_frontend_driver missing: run oracle/build_extern.sh"
    if not _have("ref_frontend_driver_cpuref"):
        pytest.skip("reference CPU build of the driver absent")
    ngpt, nbnd = (256, 16) if kind == "lw" else (224, 14)
    kw = dict(ngpt=ngpt, nbnd=nbnd, nminor_lower=4 * nbnd, nminor_upper=2 * nbnd + 3, seed=17)
    ref = _variant_run(tmp_path, "ref_frontend_driver_cpuref", kind, variant, n_gauss, clouds, 512, 60, 512, **kw)
    out = _variant_run(tmp_path, "ref_frontend_driver", kind, variant, n_gauss, clouds, 512, 60, 512, env={"RTE_HIP_HOST_MIRROR": "0"}, **kw)
    for k in ref:
        assert np.all(np.isfinite(out[k])), (what, k)
        err = float(np.max(np.abs(out[k] - ref[k])) / np.max(np.abs(ref[k])))
        assert err <= 1e-10, (what, k, err)


# ---- single precision (the reference's RTE_ENABLE_SP): the frontend compiled with wp = single on librte_rrtmgp_hip_sp.so
#      (oracle/build_extern_sp.sh)
@pytest.mark.gpu
@pytest.mark.parametrize("prog", list(PROGRAMS))
def test_reference_unit_test_programs_in_single_precision_on_the_hip_library(prog):
    """The reference's three data-free unit-test programs, -DRTE_USE_SP, on the single-precision HIP library: they apply
    their own (spacing-based) tolerances in single precision and must print the same lines as on the reference's SP kernels."""
    path = os.path.join(BIN, prog + "_sp")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/bin/*_sp absent: run oracle/build_extern_sp.sh where /root/reference exists")
    r = _run(path)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for msg in PROGRAMS[prog]:
        assert msg in r.stdout, (msg, r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,top_at_1,block", [("lw", False, 8), ("lw", True, 512), ("sw", False, 512), ("sw", True, 8)])
def test_reference_gas_optics_frontend_in_single_precision(kind, top_at_1, block, tmp_path):
    """Row f4 in single precision: load -> gas_optics -> rte_lw / rte_sw with wp = single on librte_rrtmgp_hip_sp.so.  Two
    single-precision implementations differ from each other by as much as each differs from the truth (SW upward fluxes:
    a few 1e-2 W/m2), so the yardstick is the DOUBLE-precision run of the same program on the reference's CPU kernels: the HIP
    SP fluxes must be within the reference's own acceptance threshold for SP results (3.5e-1 W/m2 absolute,
    examples/compare-to-reference.py) and not further from the truth than 3x the reference's SP kernels are."""
    need = ("ref_frontend_driver_sp", "ref_frontend_driver_sp_cpuref", "ref_frontend_driver_cpuref")
    if not all(_have(b) for b in need):
        pytest.skip("oracle/_ref/bin/ref_frontend_driver_sp[_cpuref] absent: run oracle/build_extern_sp.sh")
    ncol, nlay = 512, 60
    ngpt, nbnd = (256, 16) if kind == "lw" else (224, 14)
    raw, kd, atm, kf, af = _frontend_case(tmp_path, kind, ncol, nlay, block, top_at_1, True, True, ngpt=ngpt, nbnd=nbnd,
                                          nminor_lower=4 * nbnd, nminor_upper=2 * nbnd + 3, seed=19, checks=True)

    def run(binary, **kw):
        return stream_io.run_frontend_driver(binary, kf, af, str(tmp_path / (binary + ".bin")), GASES, ncol, nlay, kind == "lw", **kw)[0]

    truth, cpu_sp = run("ref_frontend_driver_cpuref"), run("ref_frontend_driver_sp_cpuref")
    out = run("ref_frontend_driver_sp", env={"RTE_HIP_HOST_MIRROR": "0"})
    for k in truth:
        assert np.all(np.isfinite(out[k])), k
        e_hip, e_cpu = float(np.max(np.abs(out[k] - truth[k]))), float(np.max(np.abs(cpu_sp[k] - truth[k])))
        assert e_hip <= 3.5e-1, (k, e_hip)
        assert e_hip <= 3.0 * e_cpu + 1e-3, (k, e_hip, e_cpu)


@pytest.mark.gpu
def test_host_addresses_of_openmp_mapped_arrays_are_resolved():
    """A host program with OpenMP target offload (what the reference frontend is when built with -fopenmp
    --offload-arch=gfx950) keeps its arrays on the device in `target data` regions and calls the kernel symbols with the HOST
    addresses of the mapped arrays.  The library resolves them with omp_get_mapped_ptr and works on the device copies in
    place: oracle/omp_mapped_check.c (ours) verifies the result ON the device before anything is copied back, that no byte
    was staged for the mapped call, and that the staged call on unmapped arrays gives the same bits."""
    path = os.path.join(BIN, "omp_mapped_check")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/bin/omp_mapped_check absent (oracle/build_extern_offload.sh needs the build container)")
    r = subprocess.run([path], capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(os.environ, OMP_TARGET_OFFLOAD="MANDATORY"))
    assert r.returncode == 0 and "omp_mapped_check ok" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    assert "mapped call: 0 bytes staged to the device, 0 back" in r.stdout


@pytest.mark.gpu
def test_environment_opt_ins_of_an_unchanged_binary(tmp_path):
    """RTE_HIP_DEFER_ZERO / RTE_HIP_SHARE_GEOMETRY: the opt-in modes for a device-pointer program that cannot call an
    extension.  A child process that only uses the reference ABI (device arrays, zero_array -> compute_tau_absorption ->
    compute_Planck_source -> rte_lw) runs with and without the variables: same fluxes bit for bit; with them no fill kernel is
    launched (the zero fill is folded into the tau kernel) and the tau geometry comes from the interpolation call's masks."""
    import json
    import sys

    script = tmp_path / "child.py"
    script.write_text('''
import ctypes, json, sys
sys.path.insert(0, %r)
import numpy as np, torch
import rte_rrtmgp_amd
from rte_rrtmgp_amd import frontend, hiplib, synth
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0"); A = xp.asarray
kd = synth.make_kdist("lw"); ncol, nlay = 2048, 60
atm = synth.make_atmosphere(ncol, nlay, seed=12, kdist=kd)
go = frontend.GasOptics(lib, kd, xp)
args = [A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tsfc", "col_gas", "tlev")]
hiplib.ext_call(lib, "rte_hip_profile_reset", []); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 1)
b = go.gas_optics_lw(ncol, nlay, *args, atm.top_at_1)
r = frontend.rte_lw(lib, xp, ncol, nlay, kd.ngpt, atm.top_at_1, b["tau"], b["lay_src"], b["lev_src"], xp.full((ncol, kd.ngpt), 0.98), b["sfc_src"])
torch.cuda.synchronize(); hiplib.ext_call(lib, "rte_hip_profile_enable", ["i"], 0)
names = []
for i in range(hiplib.ext_call(lib, "rte_hip_profile_count", [])):
    buf = ctypes.create_string_buffer(128); cnt, ms = ctypes.c_longlong(0), ctypes.c_double(0)
    lib.raw("rte_hip_profile_get")(ctypes.c_int(i), buf, ctypes.c_int(128), ctypes.byref(cnt), ctypes.byref(ms))
    names.append(buf.value.decode())
up = xp.to_numpy(r["flux_up"])
print(json.dumps({"kernels": names, "geom_source": hiplib.ext_call(lib, "rte_hip_stat", ["i"], 2), "sum": float(up.sum()), "hex": up.tobytes().hex()[:4096]}))
''' % ROOT)
    out = {}
    for label, env in (("plain", {}), ("opt-in", {"RTE_HIP_DEFER_ZERO": "1", "RTE_HIP_SHARE_GEOMETRY": "1"})):
        r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        out[label] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["plain"]["hex"] == out["opt-in"]["hex"] and out["plain"]["sum"] == out["opt-in"]["sum"]
    assert "fill_kernel" in out["plain"]["kernels"] and "fill_kernel" not in out["opt-in"]["kernels"]
    assert "tau_is_zero_kernel" in out["plain"]["kernels"] and "tau_is_zero_kernel" not in out["opt-in"]["kernels"]
    assert out["plain"]["geom_source"] == 2 and out["opt-in"]["geom_source"] == 1   # derived by the geometry kernel / taken from the interpolation call
