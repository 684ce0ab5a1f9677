#!/usr/bin/env python
"""Convert an RRTMGP coefficient file (netCDF, rrtmgp-data) to this project's kernel-side .npz k-distribution.

    python tools/netcdf_to_npz.py rrtmgp-gas-lw-g256.nc lw_g256.npz [--gases h2o,co2,o3,n2o,co,ch4,o2,n2]

Only the file I/O lives here: the variables are read under the names the reference's loader uses
(rrtmgp/data-loading-examples/mo_optics_utils_rrtmgp.F90:102-182), turned to the Fortran orientation of its reader
(netCDF's C order reversed), and handed to rte-rrtmgp_amd/kdist_load.init_from_raw, which performs the load-time
reductions (validated against the reference's own load: tests/test_kdist_load.py).
UNTESTED in the build environment: neither netCDF4 nor the data files are available offline.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rte_rrtmgp_amd import kdist_io, kdist_load  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--gases", default=None, help="comma-separated gases the host model provides (default: all in the file)")
    a = ap.parse_args()
    import netCDF4  # noqa: PLC0415  (not installed in the build image)

    nc = netCDF4.Dataset(a.src)
    nc.set_auto_mask(False)

    def arr(name):  # netCDF (C order) -> Fortran orientation of the reference's read_field
        return np.asfortranarray(np.transpose(np.asarray(nc.variables[name][...])))

    def strings(name):
        raw = nc.variables[name][...]
        return [b"".join(row).decode().strip() for row in np.asarray(raw)]

    raw = {"gas_names": strings("gas_names")}
    for n in ("key_species", "bnd_limits_gpt", "minor_limits_gpt_lower", "minor_limits_gpt_upper", "kminor_start_lower", "kminor_start_upper"):
        raw[n] = arr(n).astype(np.int32)
    for n in ("bnd_limits_wavenumber", "press_ref", "temp_ref", "vmr_ref", "kmajor", "kminor_lower", "kminor_upper"):
        raw[n] = arr(n)
    for n in ("press_ref_trop", "absorption_coefficient_ref_P", "absorption_coefficient_ref_T"):
        raw[n] = float(nc.variables[n][...])
    for n in ("gas_minor", "identifier_minor", "minor_gases_lower", "minor_gases_upper", "scaling_gas_lower", "scaling_gas_upper"):
        raw[n] = strings(n)
    for n in ("minor_scales_with_density_lower", "minor_scales_with_density_upper", "scale_by_complement_lower", "scale_by_complement_upper"):
        raw[n] = arr(n).astype(bool)
    if "totplnk" in nc.variables:
        for n in ("totplnk", "plank_fraction", "optimal_angle_fit"):
            raw[n] = arr(n)
    else:
        for n in ("rayl_lower", "rayl_upper", "solar_source_quiet", "solar_source_facular", "solar_source_sunspot"):
            raw[n] = arr(n)
        for n in ("tsi_default", "mg_default", "sb_default"):
            raw[n] = float(nc.variables[n][...])
    gases = a.gases.split(",") if a.gases else raw["gas_names"]
    kd = kdist_load.init_from_raw(raw, gases)
    names = kd.scalars.pop("gas_names")
    kdist_io.save_kdist(a.dst, kd)
    print(f"{a.dst}: {kd.kind}, {kd.ngpt} g-points in {kd.nbnd} bands, gases {names}")


if __name__ == "__main__":
    main()
