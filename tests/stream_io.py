"""Flat record streams exchanged with the Fortran test drivers (oracle/mo_raw_stream.F90): a raw k-distribution table
for the reference's own ``ty_gas_optics_rrtmgp%load`` and an atmosphere for oracle/ref_frontend_driver.F90.
Every record: tag (32 chars), rank (int32), dims (rank x int32), payload (float64 / int32, Fortran order), or for
string tables n x 32 chars.  Test infrastructure."""
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "bin")
last_stderr = ""


def _rec(f, tag, arr=None, strings=None, scalar=None, kind=None):
    f.write(tag.ljust(32).encode()[:32])
    if strings is not None:
        f.write(struct.pack("<ii", 1, len(strings)))
        for s in strings:
            f.write(s.ljust(32).encode()[:32])
    elif scalar is not None:
        f.write(struct.pack("<i", 0))
        f.write(struct.pack("<d", scalar) if kind == "r" else struct.pack("<i", scalar))
    else:
        a = np.asfortranarray(arr)
        f.write(struct.pack("<i", a.ndim))
        f.write(struct.pack("<" + "i" * a.ndim, *a.shape))
        if kind == "r":
            f.write(np.asfortranarray(a, dtype="<f8").tobytes(order="F"))
        else:
            f.write(np.asfortranarray(a, dtype="<i4").tobytes(order="F"))


def write_kdist_stream(path, raw, is_lw):
    """Raw table (rte-rrtmgp_amd/kdist_load.py naming) in the order oracle/mo_raw_stream.F90::load_kdist_stream reads it."""
    rec = _rec
    with open(path, "wb") as f:
        rec(f, "gas_names", strings=raw["gas_names"])
        rec(f, "key_species", raw["key_species"], kind="i")
        rec(f, "bnd_limits_gpt", raw["bnd_limits_gpt"], kind="i")
        rec(f, "bnd_limits_wavenumber", raw["bnd_limits_wavenumber"], kind="r")
        rec(f, "press_ref", raw["press_ref"], kind="r")
        rec(f, "temp_ref", raw["temp_ref"], kind="r")
        for k in ("press_ref_trop", "absorption_coefficient_ref_P", "absorption_coefficient_ref_T"):
            rec(f, k, scalar=float(raw[k]), kind="r")
        rec(f, "vmr_ref", raw["vmr_ref"], kind="r")
        rec(f, "kmajor", raw["kmajor"], kind="r")
        rec(f, "kminor_lower", raw["kminor_lower"], kind="r")
        rec(f, "kminor_upper", raw["kminor_upper"], kind="r")
        for k in ("gas_minor", "identifier_minor", "minor_gases_lower", "minor_gases_upper"):
            rec(f, k, strings=raw[k])
        rec(f, "minor_limits_gpt_lower", raw["minor_limits_gpt_lower"], kind="i")
        rec(f, "minor_limits_gpt_upper", raw["minor_limits_gpt_upper"], kind="i")
        rec(f, "sd_lower", np.asarray(raw["minor_scales_with_density_lower"]).astype(np.int32), kind="i")
        rec(f, "sd_upper", np.asarray(raw["minor_scales_with_density_upper"]).astype(np.int32), kind="i")
        rec(f, "scaling_gas_lower", strings=raw["scaling_gas_lower"])
        rec(f, "scaling_gas_upper", strings=raw["scaling_gas_upper"])
        rec(f, "sc_lower", np.asarray(raw["scale_by_complement_lower"]).astype(np.int32), kind="i")
        rec(f, "sc_upper", np.asarray(raw["scale_by_complement_upper"]).astype(np.int32), kind="i")
        rec(f, "kminor_start_lower", raw["kminor_start_lower"], kind="i")
        rec(f, "kminor_start_upper", raw["kminor_start_upper"], kind="i")
        rec(f, "is_lw", scalar=1 if is_lw else 0, kind="i")
        if is_lw:
            rec(f, "totplnk", raw["totplnk"], kind="r")
            rec(f, "plank_fraction", raw["plank_fraction"], kind="r")
            rec(f, "optimal_angle_fit", raw["optimal_angle_fit"], kind="r")
        else:
            rec(f, "rayl_lower", raw["rayl_lower"], kind="r")
            rec(f, "rayl_upper", raw["rayl_upper"], kind="r")
            for k in ("solar_source_quiet", "solar_source_facular", "solar_source_sunspot"):
                rec(f, k, raw[k], kind="r")
            for k in ("tsi_default", "mg_default", "sb_default"):
                rec(f, k, scalar=float(raw[k]), kind="r")


def write_atmosphere_stream(path, atm, is_lw, block, use_col_dry=True, use_tlev=True, checks=False, nrep=1, n_gauss=1,
                            sfc_emis=None, mu0=None, sfc_alb=None, variant=0):
    """Atmosphere (rte-rrtmgp_amd/synth.py::Atmosphere) + options for oracle/ref_frontend_driver.F90."""
    ncol, nlay = atm.play.shape
    with open(path, "wb") as f:
        _rec(f, "opts", np.array([ncol, nlay, block, int(use_col_dry), int(use_tlev), int(checks), nrep, n_gauss, variant], np.int32), kind="i")
        for tag, a in (("p_lay", atm.play), ("p_lev", atm.plev), ("t_lay", atm.tlay), ("t_lev", atm.tlev), ("vmr", atm.vmr),
                       ("col_dry", atm.col_dry)):
            _rec(f, tag, a, kind="r")
        if is_lw:
            _rec(f, "t_sfc", atm.tsfc, kind="r")
            _rec(f, "sfc_emis", sfc_emis if sfc_emis is not None else np.full(ncol, 0.98), kind="r")
        else:
            _rec(f, "mu0", mu0 if mu0 is not None else np.full(ncol, 0.86), kind="r")
            _rec(f, "sfc_alb", sfc_alb if sfc_alb is not None else np.full(ncol, 0.06), kind="r")


def write_cloud_stream(path, tb, clouds, nrough=2, rough=1):
    """Cloud look-up tables (rte-rrtmgp_amd/synth.py::make_cloud_optics, by band) + cloud field (make_cloud_field) for the
    all-sky flow of oracle/ref_frontend_driver.F90.  The reference's tables carry an ice-roughness axis: the synthetic ice
    tables become roughness ``rough`` (1-based) of ``nrough``, the others are perturbed copies that must not be used."""
    with open(path, "wb") as f:
        for k in ("radliq_lwr", "radliq_upr", "diamice_lwr", "diamice_upr"):
            _rec(f, k, scalar=float(tb[k]), kind="r")
        for k in ("extliq", "ssaliq", "asyliq"):
            _rec(f, k, tb[k], kind="r")
        for k in ("extice", "ssaice", "asyice"):
            a = np.stack([tb[k] if r == rough - 1 else 0.5 * tb[k] for r in range(nrough)], axis=2)
            _rec(f, k, a, kind="r")
        _rec(f, "ice_roughness", scalar=int(rough), kind="i")
        for k in ("lwp", "iwp", "rel", "dei"):
            _rec(f, k, clouds[k], kind="r")


def run_frontend_driver(binary, kfile, afile, ofile, gases, ncol, nlay, is_lw, env=None, timeout=1800, cloud_file=None):
    """Run oracle/_ref/bin/<binary>; returns (fluxes dict of (ncol, nlay+1) arrays, stdout)."""
    path = os.path.join(BIN, binary)
    env = dict(env or {})
    if "_omp" in binary:  # ... and OpenMP worker threads have stacks of their own (virtual reservation only)
        env.setdefault("OMP_STACKSIZE", "24G")
    # flang keeps automatic arrays on the stack
    # REF_DRIVER_PREFIX: a command to run the program under (e.g. "rocprofv3 --hip-trace --stats -d gpurun_out/x --": experiments)
    prefix = os.environ.get("REF_DRIVER_PREFIX", "")
    r = subprocess.run(f"ulimit -s unlimited 2>/dev/null; exec {prefix} '{path}' '{kfile}' '{afile}' '{ofile}' '{','.join(gases)}'"
                       + (f" '{cloud_file}'" if cloud_file else ""),
                       shell=True, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=dict(os.environ, **(env or {})))
    global last_stderr
    last_stderr = r.stderr
    assert r.returncode == 0 and "ref_frontend_driver ok" in r.stdout, (binary, r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    raw = np.fromfile(ofile, dtype="<f8")
    names = ["flux_up", "flux_dn"] + ([] if is_lw else ["flux_dn_dir"])
    n = ncol * (nlay + 1)
    assert raw.size == n * len(names), (raw.size, n, names)
    return {k: raw[i * n:(i + 1) * n].reshape((ncol, nlay + 1), order="F") for i, k in enumerate(names)}, r.stdout


def run_equivalence_driver(binary, kfile, afile, gases, env=None, timeout=1800):
    """Run oracle/_ref/bin/<binary> (ref_equivalence_driver[_cpuref]); returns (returncode, {check name: (worst deviation in
    spacings, limit, ok)}, stdout)."""
    path = os.path.join(BIN, binary)
    r = subprocess.run(f"ulimit -s unlimited 2>/dev/null; exec '{path}' '{kfile}' '{afile}' '{','.join(gases)}'",
                       shell=True, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=dict(os.environ, **(env or {})))
    checks = {}
    for ln in r.stdout.splitlines():
        ln = ln.strip()
        if ln.startswith("check ") and "spacings" in ln:
            name, rest = ln[6:].rsplit(":", 1)
            w = rest.split()
            checks[name] = (float(w[0]), float(w[3].rstrip(")")), ln.endswith("ok"))
    return r.returncode, checks, r.stdout + r.stderr


def measure_frontend_driver(kind="lw", ncol=98304, block=8192, modes=("mirror",), nlay=60, nrep=3, seed=42, env_extra=None, threads=1):
    """Columns/s of the reference's UNCHANGED Fortran frontend (oracle/_ref/bin/ref_frontend_driver: k%load -> k%gas_optics
    -> rte_lw / rte_sw per block, pageable host arrays) on the HIP library in the given modes ("mirror": host-mirror mode,
    "staged": every array staged both ways), and with "cpuref" the same program on the reference's CPU kernels (one core,
    bounded sample).  threads > 1: the OpenMP build of the driver with that many host threads, each on its own library
    context (RTE_HIP_THREAD_CONTEXTS=1).  Fluxes of the HIP modes must agree bit for bit.  Returns {mode: {"columns_per_s", "report"}}."""
    import shutil
    import sys
    import tempfile

    sys.path.insert(0, ROOT)
    from rte_rrtmgp_amd import kdist_load, synth

    ngpt, nbnd = (256, 16) if kind == "lw" else (224, 14)
    gases = list(synth.GAS_NAMES)
    raw = kdist_load.synth_raw(kind, ngpt=ngpt, nbnd=nbnd, nminor_lower=4 * nbnd, nminor_upper=2 * nbnd + 3)
    kd = kdist_load.init_from_raw(raw, gases)
    kd.scalars.pop("gas_names")
    d = tempfile.mkdtemp(prefix="rte_f4_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    out = {}
    try:
        kf, af, of = (os.path.join(d, n) for n in ("k.bin", "a.bin", "o.bin"))
        write_kdist_stream(kf, raw, kind == "lw")
        ref = None
        for mode in modes:
            if mode == "cpuref":
                n = min(ncol, 2048)
                atm = synth.make_atmosphere(n, nlay, seed=seed, kdist=kd, ngas=kd.ngas)
                write_atmosphere_stream(af, atm, kind == "lw", block=32, checks=False, nrep=1)
                fl, log = run_frontend_driver("ref_frontend_driver_cpuref", kf, af, of, gases, n, nlay, kind == "lw")
            else:
                atm = synth.make_atmosphere(ncol, nlay, seed=seed, kdist=kd, ngas=kd.ngas)
                write_atmosphere_stream(af, atm, kind == "lw", block=block, checks=False, nrep=nrep)
                env = {"RTE_HIP_HOST_MIRROR": "1" if mode == "mirror" else "0", "RTE_HIP_STAGING_REPORT": "1"}
                if threads > 1:
                    env.update({"OMP_NUM_THREADS": str(threads), "RTE_HIP_THREAD_CONTEXTS": "1"})
                env.update(env_extra or {})
                fl, log = run_frontend_driver("ref_frontend_driver_omp" if threads > 1 else "ref_frontend_driver", kf, af, of, gases,
                                              ncol, nlay, kind == "lw", env=env)
                if ref is None:
                    ref = fl
                for k in fl:  # same block size -> same kernels and reduction order: the modes must agree bit for bit
                    assert np.array_equal(fl[k], ref[k]), (mode, k, float(np.max(np.abs(fl[k] - ref[k]))))
            best = float([ln for ln in log.splitlines() if "best columns/s" in ln][0].split(":")[1])
            rep = [ln.strip() for ln in last_stderr.splitlines() if "staging report" in ln]
            detail = [ln.rstrip() for ln in last_stderr.splitlines() if ln.startswith("    ")]  # per-entry wall clock
            detail += [ln.rstrip() for ln in log.splitlines() if ln.startswith("  thread 0:")]     # REF_DRIVER_TIMING=1
            passes = [ln.split(":", 1)[1].strip() for ln in log.splitlines() if ln.startswith("pass")]
            rates = []
            for p_ in passes:  # "0.0611 s,    1609088.3 columns/s (8 host threads)"
                try:
                    rates.append(float(p_.split(",")[1].split()[0]))
                except (IndexError, ValueError):
                    pass
            out[mode] = {"columns_per_s": best, "report": rep[0] if rep else None, "reports": rep, "detail": detail,
                         "passes": passes, "pass_rates": rates}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out
