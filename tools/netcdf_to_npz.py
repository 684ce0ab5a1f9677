#!/usr/bin/env python
"""Convert an RRTMGP coefficient file (netCDF, rrtmgp-data) to this project's kernel-side .npz k-distribution.

    python tools/netcdf_to_npz.py rrtmgp-gas-lw-g256.nc lw_g256.npz [--gases h2o,co2,o3,n2o,co,ch4,o2,n2]

Only the file I/O lives here: the variables are read under the names the reference's loader uses
(rrtmgp/data-loading-examples/mo_optics_utils_rrtmgp.F90:102-182), turned to the Fortran orientation of its reader
(netCDF's C order reversed), and handed to rte-rrtmgp_amd/kdist_load.init_from_raw, which performs the load-time
reductions (validated against the reference's own load: tests/test_kdist_load.py).
The real files are netCDF-4 (HDF5).  They are read with netCDF4 where it is installed, otherwise through the HDF5 C library
itself (rte-rrtmgp_amd/hdf5_reader.py: a ctypes binding, no Python package needed -- any libhdf5 >= 1.8 on the machine);
netCDF-3 classic / 64-bit-offset files go through scipy's reader.  The data are not offline, so what IS tested
(tests/test_netcdf_converter.py) is this code path on synthetic files with the reference's variable names and dimension order:
one in the classic format, and netCDF-4-style HDF5 files written by the HDF5 library the way netCDF-C lays them out (chunked,
shuffled, deflated, dimension-scale datasets; old and latest file-format generation).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rte_rrtmgp_amd import hdf5_reader, kdist_io, kdist_load  # noqa: E402


def open_dataset(path):
    """netCDF4 where it is installed; otherwise an HDF5 file (netCDF-4: what rrtmgp-data ships) goes through the HDF5 C library
    (hdf5_reader) and a netCDF-3 classic / 64-bit-offset file through scipy's reader.  Returns (variables mapping, close function)."""
    try:
        import netCDF4  # noqa: PLC0415  (not installed in the build image)

        nc = netCDF4.Dataset(path)
        nc.set_auto_mask(False)
        return nc.variables, nc.close
    except ImportError:
        pass
    if hdf5_reader.is_hdf5(path):
        variables, _, close = hdf5_reader.open_netcdf4(path)  # raises HDF5Unavailable with the places it looked in
        return variables, close
    from scipy.io import netcdf_file  # noqa: PLC0415

    nc = netcdf_file(path, "r", mmap=False)
    return nc.variables, nc.close


def read_raw(path):
    """The RAW contents of a coefficient file under the names the reference's loader uses
    (rrtmgp/data-loading-examples/mo_optics_utils_rrtmgp.F90:102-182), in the Fortran orientation of its reader."""
    variables, close = open_dataset(path)

    def data(name):
        return np.array(variables[name][...])

    def arr(name):  # netCDF (C order) -> Fortran orientation of the reference's read_field
        return np.asfortranarray(np.transpose(data(name)))

    def strings(name):
        return [b"".join(bytes(c) if not isinstance(c, bytes) else c for c in row).decode().strip() for row in data(name)]

    raw = {"gas_names": strings("gas_names")}
    for n in ("key_species", "bnd_limits_gpt", "minor_limits_gpt_lower", "minor_limits_gpt_upper", "kminor_start_lower", "kminor_start_upper"):
        raw[n] = arr(n).astype(np.int32)
    for n in ("bnd_limits_wavenumber", "press_ref", "temp_ref", "vmr_ref", "kmajor", "kminor_lower", "kminor_upper"):
        raw[n] = arr(n).astype(np.float64)
    for n in ("press_ref_trop", "absorption_coefficient_ref_P", "absorption_coefficient_ref_T"):
        raw[n] = float(data(n))
    for n in ("gas_minor", "identifier_minor", "minor_gases_lower", "minor_gases_upper", "scaling_gas_lower", "scaling_gas_upper"):
        raw[n] = strings(n)
    for n in ("minor_scales_with_density_lower", "minor_scales_with_density_upper", "scale_by_complement_lower", "scale_by_complement_upper"):
        raw[n] = arr(n).astype(bool)
    if "totplnk" in variables:
        for n in ("totplnk", "plank_fraction", "optimal_angle_fit"):
            raw[n] = arr(n).astype(np.float64)
    else:
        for n in ("rayl_lower", "rayl_upper", "solar_source_quiet", "solar_source_facular", "solar_source_sunspot"):
            raw[n] = arr(n).astype(np.float64)
        for n in ("tsi_default", "mg_default", "sb_default"):
            raw[n] = float(data(n))
    close()
    return raw


def convert(src, dst, gases=None):
    raw = read_raw(src)
    kd = kdist_load.init_from_raw(raw, gases if gases else raw["gas_names"])
    names = kd.scalars.pop("gas_names")
    kdist_io.save_kdist(dst, kd)
    return kd, names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--gases", default=None, help="comma-separated gases the host model provides (default: all in the file)")
    a = ap.parse_args()
    kd, names = convert(a.src, a.dst, a.gases.split(",") if a.gases else None)
    print(f"{a.dst}: {kd.kind}, {kd.ngpt} g-points in {kd.nbnd} bands, gases {names}")


if __name__ == "__main__":
    main()
