"""Test infrastructure: the CPU kernels (the reference's own Fortran kernels from oracle/_ref if present, else the C
restatement) run over MANY columns on all usable cores -- one single-threaded process per core, each taking a
contiguous column range in blocks of 32 columns (the reference's usage pattern) and keeping only the broadband fluxes.
Lets the GPU tests compare full-size configurations (1e5 columns) elementwise with the checker instead of with a
small run of the library itself.  The reference kernels do ~4.5 k columns/s/core (LW), so 1e5 columns take seconds.

    fluxes = ref_pool.run("lw" | "sw" | "allsky", atm, nlay)      # dict of (ncol, nlay+1) arrays
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

ATM_FIELDS = ("play", "plev", "tlay", "tlev", "tsfc", "col_gas", "col_dry")
BLOCK = 32


def usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def _chain(workload, lib, xp, nlay):
    """Returns f(atm_slice_dict, ncol) -> dict of broadband flux arrays for one block."""
    from rte_rrtmgp_amd import frontend, synth

    if workload == "allsky":
        kdl, kds = synth.make_kdist("lw"), synth.make_kdist("sw")
        gol, gos = frontend.GasOptics(lib, kdl, xp), frontend.GasOptics(lib, kds, xp)
        tbl, tbs = synth.make_cloud_optics(kdl.nbnd), synth.make_cloud_optics(kds.nbnd)
        col, cos_ = frontend.CloudOptics(lib, tbl, xp), frontend.CloudOptics(lib, tbs, xp)

        def run(a, n, clouds):
            a = dict(a, top_at_1=False)
            rl = frontend.allsky_lw(lib, xp, gol, col, n, nlay, a, clouds, xp.full((n, kdl.ngpt), 0.98))
            rs = frontend.allsky_sw(lib, xp, gos, cos_, n, nlay, a, clouds, xp.full((n, nlay), 0.86), xp.full((n, kds.ngpt), 0.06))
            fl, fs = rl[2], rs[2]
            return {"lw_up": fl["flux_up"], "lw_dn": fl["flux_dn"], "sw_up": fs["flux_up"], "sw_dn": fs["flux_dn"], "sw_dir": fs["flux_dir"]}

        return run
    kd = synth.make_kdist(workload)
    go = frontend.GasOptics(lib, kd, xp)

    def run_lw(a, n, clouds=None):
        b = go.gas_optics_lw(n, nlay, a["play"], a["plev"], a["tlay"], a["tsfc"], a["col_gas"], a["tlev"], False)
        r = frontend.rte_lw(lib, xp, n, nlay, kd.ngpt, False, b["tau"], b["lay_src"], b["lev_src"], xp.full((n, kd.ngpt), 0.98), b["sfc_src"])
        return {"up": r["flux_up"], "dn": r["flux_dn"]}

    def run_sw(a, n, clouds=None):
        b = go.gas_optics_sw(n, nlay, a["play"], a["plev"], a["tlay"], a["col_gas"], a["col_dry"])
        r = frontend.rte_sw(lib, xp, n, nlay, kd.ngpt, False, b["tau"], b["ssa"], b["g"], xp.full((n, nlay), 0.86), b["toa_src"],
                            xp.full((n, kd.ngpt), 0.06), xp.full((n, kd.ngpt), 0.06))
        return {"up": r["flux_up"], "dn": r["flux_dn"], "dir": r["flux_dir"]}

    return run_lw if workload == "lw" else run_sw


def _worker(workload, d, c0, c1, nlay):
    import threading

    threading.stack_size(1 << 30)  # flang keeps automatic arrays on the stack

    def body():
        from oracle import oracle as O
        from rte_rrtmgp_amd import frontend

        try:
            lib = O.load_ref()
        except Exception:
            lib = None
        checker = "reference" if lib is not None else "port"
        if lib is None:
            lib = O.load_c()
        # where the reference build exists (oracle/_ref/librefkernels.so travels to the GPU box) it MUST be the checker
        if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "librefkernels.so")):
            assert checker == "reference", "oracle/_ref/librefkernels.so is present but could not be loaded"
        with open(os.path.join(d, f"checker_{c0}.txt"), "w") as f:
            f.write(checker)
        xp = frontend.NumpyArrays()
        atm = {k: np.load(os.path.join(d, k + ".npy"), mmap_mode="r") for k in ATM_FIELDS}
        clouds = None
        if workload == "allsky":
            clouds = {k: np.load(os.path.join(d, "cld_" + k + ".npy"), mmap_mode="r") for k in ("lwp", "iwp", "rel", "dei")}
        run = _chain(workload, lib, xp, nlay)
        out = None
        for b0 in range(c0, c1, BLOCK):
            b1 = min(c1, b0 + BLOCK)
            a = {k: np.asfortranarray(v[b0:b1]) for k, v in atm.items()}
            cl = {k: np.asfortranarray(v[b0:b1]) for k, v in clouds.items()} if clouds else None
            r = run(a, b1 - b0, cl)
            if out is None:
                out = {k: np.empty((c1 - c0, nlay + 1)) for k in r}
            for k, v in r.items():
                out[k][b0 - c0:b1 - c0] = v
        np.savez(os.path.join(d, f"out_{c0}.npz"), **out)

    t = threading.Thread(target=body)
    t.start()
    t.join()


last_checker = None


def run(workload, atm, nlay, clouds=None, cores=None):
    """Broadband fluxes of `workload` for every column of `atm` (rte-rrtmgp_amd/synth.py::Atmosphere, surface at index 1)
    from the CPU kernels; `clouds`: dict lwp/iwp/rel/dei for the all-sky chain."""
    assert not atm.top_at_1
    ncol = atm.play.shape[0]
    cores = cores or usable_cores()
    d = tempfile.mkdtemp(prefix="rte_refpool_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for k in ATM_FIELDS:
            np.save(os.path.join(d, k + ".npy"), np.ascontiguousarray(getattr(atm, k)))
        if clouds:
            for k, v in clouds.items():
                np.save(os.path.join(d, "cld_" + k + ".npy"), np.ascontiguousarray(v))
        per = -(-ncol // cores)
        per = -(-per // BLOCK) * BLOCK
        ranges = [(c0, min(ncol, c0 + per)) for c0 in range(0, ncol, per)]
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", workload, d, str(c0), str(c1), str(nlay)],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for c0, c1 in ranges]
        for p, rg in zip(procs, ranges):
            so, se = p.communicate(timeout=3600)
            assert p.returncode == 0, (rg, so[-2000:], se[-2000:])
        parts = [np.load(os.path.join(d, f"out_{c0}.npz")) for c0, _ in ranges]
        kinds = {open(os.path.join(d, f"checker_{c0}.txt")).read() for c0, _ in ranges}
        assert len(kinds) == 1, kinds
        global last_checker
        last_checker = kinds.pop()   # "reference" (the reference's own Fortran kernels) or "port" (the C restatement)
        return {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0].files}
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--worker":
    _worker(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]))
