#!/usr/bin/env python
"""What the reference's Fortran FRONTEND computes between its kernel calls, recorded from a run of
the reference's own frontend on its own CPU kernels: oracle/_ref/bin/ref_frontend_driver_glue = oracle/ref_frontend_driver.F90
(k%load -> k%gas_optics -> k%compute_optimal_angles -> rte_lw(lw_Ds=) per block of columns) with oracle/glue_recorder.c in
front of three kernel symbols.  Stored in tests/golden/glue_frontend.npz: the inputs the frontend was given and, as they
arrived at the kernels,
  col_gas        vmr x col_dry per gas, dry air in slot 0        rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:594-609
  tlev           level temperatures interpolated by the frontend (the caller gave none)           :893-912
  Ds             secants from compute_optimal_angles, through rte_lw(lw_Ds=)                      :1536-1561
  sfc_emis_gpt   expand_and_transpose of a band-dependent emissivity     rte/frontend/mo_rte_lw.F90:478-501
and tau (the optical depths compute_optimal_angles read).  Run in the build container (needs /root/reference for
oracle/build_extern.sh):  python tests/golden/make_glue_golden.py        A fixture is data only."""
import os
import struct
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import stream_io  # noqa: E402
from rte_rrtmgp_amd import kdist_load, synth  # noqa: E402

NCOL, NLAY, BLOCK, SEED = 48, 14, 16, 31
NGPT, NBND = 64, 4


def read_records(path):
    out = []
    with open(path, "rb") as f:
        while True:
            tag = f.read(32)
            if len(tag) < 32:
                break
            kind, rank = struct.unpack("<ii", f.read(8))
            dims = struct.unpack("<" + "i" * rank, f.read(4 * rank))
            n = int(np.prod(dims))
            a = np.frombuffer(f.read(8 * n), dtype="<f8") if kind == 1 else np.frombuffer(f.read(4 * n), dtype="<i4")
            out.append((tag.decode().strip(), a.reshape(dims, order="F")))
    return out


def main():
    gases = list(synth.GAS_NAMES)
    raw = kdist_load.synth_raw("lw", ngpt=NGPT, nbnd=NBND, nminor_lower=2 * NBND, nminor_upper=NBND + 1)
    kd = kdist_load.init_from_raw(raw, gases)
    kd.scalars.pop("gas_names")
    atm = synth.make_atmosphere(NCOL, NLAY, seed=SEED, kdist=kd, ngas=kd.ngas)
    rng = np.random.default_rng(SEED)
    sfc_emis = np.asfortranarray(rng.uniform(0.85, 0.99, NCOL))
    d = tempfile.mkdtemp(prefix="rte_glue_")
    kf, af, of, rf = (os.path.join(d, n) for n in ("k.bin", "a.bin", "o.bin", "rec.bin"))
    stream_io.write_kdist_stream(kf, raw, True)
    # col_dry given (the frontend multiplies it with the mixing ratios), no level temperatures (the frontend interpolates them),
    # variant 1 = optimal transport angles
    stream_io.write_atmosphere_stream(af, atm, True, block=BLOCK, use_col_dry=True, use_tlev=False, checks=True, nrep=1,
                                      sfc_emis=sfc_emis, variant=1)
    stream_io.run_frontend_driver("ref_frontend_driver_glue", kf, af, of, gases, NCOL, NLAY, True,
                                  env={"RTE_ABI_RECORD": rf, "REF_DRIVER_BAND_EMIS": "1"})
    recs = read_records(rf)
    cat = {}
    for tag, a in recs:
        cat.setdefault(tag, []).append(a)
    nblk = NCOL // BLOCK
    assert all(len(cat[t]) == nblk for t in ("col_gas", "tlev", "Ds", "sfc_emis_gpt", "tau")), {t: len(v) for t, v in cat.items()}
    out = {t: np.asfortranarray(np.concatenate(cat[t], axis=0)) for t in ("col_gas", "tlev", "Ds", "sfc_emis_gpt", "tau")}
    band_factor = 1.0 - 0.00390625 * np.arange(1, NBND + 1)   # oracle/ref_frontend_driver.F90 (REF_DRIVER_BAND_EMIS)
    np.savez_compressed(
        os.path.join(HERE, "glue_frontend.npz"),
        play=atm.play, plev=atm.plev, tlay=atm.tlay, vmr=atm.vmr, col_dry=atm.col_dry,
        sfc_emis_bnd=np.asfortranarray(band_factor[:, None] * sfc_emis[None, :]),   # (nbnd, ncol): what rte_lw was given
        band_lims_gpt=np.asfortranarray(kd.band_lims_gpt), optimal_angle_fit=np.asfortranarray(kd.optimal_angle_fit),
        **{"ref_" + k: v for k, v in out.items()},
        meta=np.array([NCOL, NLAY, BLOCK, SEED, NGPT, NBND, kd.ngas]))
    print({k: v.shape for k, v in out.items()}, "->", os.path.join(HERE, "glue_frontend.npz"),
          os.path.getsize(os.path.join(HERE, "glue_frontend.npz")), "bytes")


if __name__ == "__main__":
    main()
