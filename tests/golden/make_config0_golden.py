#!/usr/bin/env python
"""BASELINE configs[0] at its own size -- RFMIP-like clear-sky LW, 100 columns x 60 layers x 256 g-points -- from the
REFERENCE kernels themselves (oracle/_ref/librefkernels.so, built in place by oracle/build_ref.sh; binary only, never
committed), driven through the kernel C ABI in blocks of 8 columns, the block size of the reference's RFMIP driver
(examples/rfmip-clear-sky/rrtmgp_rfmip_lw.F90:88,247-281).  Stores the broadband fluxes in full (100 x 61 x 2 doubles)
and a SHA-256 of the inputs.  Run in the build container:  python tests/golden/make_config0_golden.py
A fixture is data only: no reference source text is stored anywhere."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as O  # noqa: E402
from rte_rrtmgp_amd import frontend, synth  # noqa: E402

NCOL, NLAY, SEED, BLOCK = 100, 60, 2024, 8
FIELDS = ("play", "plev", "tlay", "tlev", "tsfc", "col_gas")


def inputs():
    kd = synth.make_kdist("lw")
    atm = synth.make_atmosphere(NCOL, NLAY, seed=SEED, kdist=kd, climate="sites")  # 100 site-like profiles
    return kd, atm


def digest(kd, atm):
    h = hashlib.sha256()
    for k in sorted(kd.arrays):
        h.update(np.ascontiguousarray(kd.arrays[k]).tobytes())
    for k in FIELDS:
        h.update(np.ascontiguousarray(getattr(atm, k)).tobytes())
    return h.hexdigest()


def run(lib, xp, kd, atm, block):
    go = frontend.GasOptics(lib, kd, xp)
    up, dn = np.empty((NCOL, NLAY + 1)), np.empty((NCOL, NLAY + 1))
    for c0 in range(0, NCOL, block):
        c1 = min(NCOL, c0 + block)
        n = c1 - c0
        a = {k: xp.asarray(np.asfortranarray(getattr(atm, k)[c0:c1])) for k in FIELDS}
        b = go.gas_optics_lw(n, NLAY, a["play"], a["plev"], a["tlay"], a["tsfc"], a["col_gas"], a["tlev"], atm.top_at_1)
        r = frontend.rte_lw(lib, xp, n, NLAY, kd.ngpt, atm.top_at_1, b["tau"], b["lay_src"], b["lev_src"], xp.full((n, kd.ngpt), 0.98), b["sfc_src"])
        xp.sync()
        up[c0:c1], dn[c0:c1] = xp.to_numpy(r["flux_up"]), xp.to_numpy(r["flux_dn"])
    return up, dn


def main():
    ref = O.load_ref()
    if ref is None:
        raise SystemExit("reference build unavailable (needs /root/reference + flang)")
    kd, atm = inputs()
    up, dn = O.big_stack(run, ref, frontend.NumpyArrays(), kd, atm, BLOCK)
    path = os.path.join(HERE, "config0_lw.npz")
    np.savez_compressed(path, flux_up=up, flux_dn=dn, __digest__=np.array(digest(kd, atm)))
    print("config0_lw:", up.shape, "->", os.path.getsize(path) // 1024, "KiB; OLR mean", up[:, -1].mean())


if __name__ == "__main__":
    main()
