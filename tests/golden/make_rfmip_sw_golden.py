#!/usr/bin/env python
"""The boundary conditions of the reference's RFMIP shortwave driver -- total-solar-irradiance renormalisation of toa_flux,
spectrally constant surface albedo per band, cosine of the solar zenith angle with night columns set to 1, fluxes of night
columns zeroed (examples/rfmip-clear-sky/rrtmgp_rfmip_sw.F90:269-318, :331-337) -- recorded from the reference's OWN statements:
oracle/build_rfmip_sw_glue.sh cuts them out of the reference file where it lies and compiles them inside a C-callable frame
(oracle/_ref/librfmipswglue.so; the driver program itself needs netCDF and cannot be linked here).  Stored in
tests/golden/rfmip_sw_glue.npz: seeded inputs and the block's outputs.  Run in the build container (needs /root/reference):
    sh oracle/build_rfmip_sw_glue.sh && python tests/golden/make_rfmip_sw_golden.py          A fixture is data only."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BLOCK, NGPT, NBND, NLAY, NBLOCKS, B, SEED = 37, 24, 3, 5, 3, 2, 77


def inputs():
    rng = np.random.default_rng(SEED)
    F = np.asfortranarray
    sza = F(rng.uniform(0.0, 120.0, (BLOCK, NBLOCKS)))
    return dict(toa_flux=F(rng.uniform(0.1, 9.0, (BLOCK, NGPT))), tsi=F(rng.uniform(1300.0, 1420.0, (BLOCK, NBLOCKS))),
                albedo=F(rng.uniform(0.0, 1.0, (BLOCK, NBLOCKS))), sza=sza, usecol=F((sza < 90.0).astype(np.int32)),
                flux_up=F(rng.uniform(0.0, 400.0, (BLOCK, NLAY + 1, NBLOCKS))), flux_dn=F(rng.uniform(0.0, 400.0, (BLOCK, NLAY + 1, NBLOCKS))))


def main():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "librfmipswglue.so"))
    d = inputs()
    toa, fu, fd = d["toa_flux"].copy(order="F"), d["flux_up"].copy(order="F"), d["flux_dn"].copy(order="F")
    def_tsi, alb_spec, mu0 = np.zeros(BLOCK), np.zeros((NBND, BLOCK), order="F"), np.zeros(BLOCK)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    I = ctypes.c_int
    lib.ref_rfmip_sw_boundary(I(BLOCK), I(NGPT), I(NBND), I(NLAY), I(NBLOCKS), I(B), P(toa), P(d["tsi"]), P(d["albedo"]), P(d["sza"]),
                              P(d["usecol"]), P(def_tsi), P(alb_spec), P(mu0), P(fu), P(fd))
    assert (d["usecol"][:, B - 1] == 0).any() and (d["usecol"][:, B - 1] != 0).any()
    np.savez_compressed(os.path.join(HERE, "rfmip_sw_glue.npz"), block=B, **{"in_" + k: v for k, v in d.items()},
                        out_toa_flux=toa, out_def_tsi=def_tsi, out_sfc_alb_spec=alb_spec, out_mu0=mu0, out_flux_up=fu, out_flux_dn=fd)
    print("wrote rfmip_sw_glue.npz: toa_flux", toa.shape, "night columns in the block:", int((d["usecol"][:, B - 1] == 0).sum()))


if __name__ == "__main__":
    main()
