#!/usr/bin/env python
"""Generate tests/golden/allsky_72.npz from the REFERENCE kernels: the all-sky chain of BASELINE configs[3]
(examples/all-sky/rrtmgp_allsky.F90:336-404) at its real shape -- 72 layers, the g256-shaped LW and g224-shaped SW
synthetic k-distributions, cloud optics from tables -- for 24 seeded columns.

Run in the build container (needs /root/reference and flang):  python tests/golden/make_allsky_golden.py
The reference's own `default` Fortran kernels (oracle/_ref/librefkernels.so, built in place by oracle/build_ref.sh) are
driven through the host mirror of the frontend's call sequence (rte-rrtmgp_amd/frontend.py).  Stored: broadband fluxes in
full, cloud and gas optical properties as strided samples; inputs are regenerated from the seeds in the test
(tests/test_allsky.py::test_hip_matches_allsky_golden), and a digest of them is stored.  Data only.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as O  # noqa: E402
from rte_rrtmgp_amd import frontend, synth  # noqa: E402

NCOL, NLAY, SEED = 24, 72, 314


def setup(kind, ncol=NCOL):
    kd = synth.make_kdist(kind)
    atm = synth.make_atmosphere(ncol, NLAY, seed=SEED, kdist=kd)
    tb = synth.make_cloud_optics(kd.nbnd)
    cl = synth.make_cloud_field(atm, tb)
    return kd, atm, tb, cl


def digest(kd, atm, tb, cl):
    h = hashlib.sha256()
    for d in (kd.arrays, {k: getattr(atm, k) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas", "col_dry")},
              {k: v for k, v in tb.items() if hasattr(v, "shape")}, cl):
        for k in sorted(d):
            h.update(k.encode())
            h.update(np.ascontiguousarray(d[k]).tobytes())
    return h.hexdigest()


def run(lib, xp, kind, kd, atm, tb, cl, ncol):
    A = xp.asarray
    go, co = frontend.GasOptics(lib, kd, xp), frontend.CloudOptics(lib, tb, xp)
    a = {k: A(getattr(atm, k)) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas", "col_dry")}
    a["top_at_1"] = atm.top_at_1
    c = {k: A(v) for k, v in cl.items()}
    if kind == "lw":
        gb, cb, rb = frontend.allsky_lw(lib, xp, go, co, ncol, NLAY, a, c, xp.full((ncol, kd.ngpt), 0.98))
        keys = [("cld_tau", cb), ("tau", gb), ("flux_up", rb), ("flux_dn", rb)]
    else:
        mu0, alb = xp.full((ncol, NLAY), 0.86), xp.full((ncol, kd.ngpt), 0.06)
        gb, cb, rb = frontend.allsky_sw(lib, xp, go, co, ncol, NLAY, a, c, mu0, alb)
        keys = [("cld_tau", cb), ("cld_ssa", cb), ("cld_g", cb), ("tau", gb), ("ssa", gb), ("g", gb), ("flux_up", rb),
                ("flux_dn", rb), ("flux_dir", rb)]
    xp.sync()
    return {k: np.array(xp.to_numpy(d[k])) for k, d in keys}


SAMPLE = 997  # prime stride through the (column-fastest) 3-D arrays


def main():
    ref = O.load_ref()
    if ref is None:
        raise SystemExit("reference build unavailable (needs /root/reference + flang)")
    store = {}
    for kind in ("lw", "sw"):
        kd, atm, tb, cl = setup(kind)
        out = O.big_stack(run, ref, frontend.NumpyArrays(), kind, kd, atm, tb, cl, NCOL)
        store[f"{kind}.__digest__"] = np.array(digest(kd, atm, tb, cl))
        for k, v in out.items():
            if v.ndim == 2:
                store[f"{kind}.{k}|full"] = v
            else:
                store[f"{kind}.{k}|sample"] = v.ravel(order="F")[::SAMPLE].copy()
    path = os.path.join(HERE, "allsky_72.npz")
    np.savez_compressed(path, **store)
    print("allsky_72.npz", os.path.getsize(path) // 1024, "KiB", sorted(store))


if __name__ == "__main__":
    main()
