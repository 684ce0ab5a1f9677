"""Test helper: a netCDF-4-style HDF5 file written THROUGH THE HDF5 LIBRARY (the one rte-rrtmgp_amd/hdf5_reader.py binds), laid out
as netCDF-C lays its files out: every variable a dataset of the root group -- chunked, byte-shuffled and deflated like the
rrtmgp-data files -- NC_CHAR arrays as fixed-length strings of size 1, NC_DOUBLE / NC_INT as little-endian IEEE / 32-bit integers,
scalars on a scalar dataspace, and one dataset per dimension carrying the CLASS / NAME attributes of a dimension scale without a
coordinate variable ("This is a netCDF dimension but not a netCDF variable.").  The image has neither netCDF4 nor h5py; the bytes
of the file are the library's own, in the old (symbol-table groups, version-0 superblock) or the latest file-format generation."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rte_rrtmgp_amd import hdf5_reader  # noqa: E402

H5S_SCALAR = 0


class Writer:
    def __init__(self, path, latest=False, deflate=4):
        self.L = L = hdf5_reader.hdf5()
        self.deflate = deflate
        fapl = 0
        if latest:  # new-style groups (link messages / fractal heaps), version-2+ superblock, version-2 object headers
            fapl = L.H5Pcreate(L.const("H5P_CLS_FILE_ACCESS_ID_g"))
            hi = {8: 1, 10: 2, 12: 3, 14: 4}.get(L.version[1], 2)  # H5F_LIBVER_LATEST of this library generation
            assert L.H5Pset_libver_bounds(fapl, hi, hi) >= 0
        self.fid = L.H5Fcreate(os.fsencode(path), hdf5_reader.H5F_ACC_TRUNC, 0, fapl)
        assert self.fid >= 0, path
        if fapl:
            L.H5Pclose(fapl)
        self.dims = {}  # size -> dataset name

    def _str_attr(self, obj, name, value):
        L = self.L
        t = L.H5Tcopy(L.const("H5T_C_S1_g"))
        L.H5Tset_size(t, len(value) + 1)
        s = L.H5Screate(H5S_SCALAR)
        a = L.H5Acreate2(obj, name.encode(), t, s, 0, 0)
        assert a >= 0
        buf = ctypes.create_string_buffer(value, len(value) + 1)
        assert L.H5Awrite(a, t, buf) >= 0
        L.H5Aclose(a); L.H5Sclose(s); L.H5Tclose(t)

    def dim(self, n):
        """A dimension of size n without a coordinate variable, as netCDF-C writes it."""
        if n in self.dims:
            return self.dims[n]
        L = self.L
        name = f"dim_{n:04d}"
        ext = (ctypes.c_uint64 * 1)(n)
        s = L.H5Screate_simple(1, ext, None)
        d = L.H5Dcreate2(self.fid, name.encode(), L.const("H5T_IEEE_F32BE_g"), s, 0, 0, 0)
        assert d >= 0
        self._str_attr(d, "CLASS", b"DIMENSION_SCALE")
        self._str_attr(d, "NAME", b"This is a netCDF dimension but not a netCDF variable.%10d" % n)
        L.H5Dclose(d); L.H5Sclose(s)
        self.dims[n] = name
        return name

    def var(self, name, a, chunk=None):
        """a: numpy array in netCDF (C) dimension order; dtype f8, i4 or S1; 0-d for a scalar variable."""
        L = self.L
        a = np.asarray(a, order="C")  # (ascontiguousarray would make a 0-d array 1-d)
        if a.dtype == np.float64:
            ft, mt = L.const("H5T_IEEE_F64LE_g"), L.const("H5T_NATIVE_DOUBLE_g")
        elif a.dtype == np.int32:
            ft, mt = L.const("H5T_STD_I32LE_g"), L.const("H5T_NATIVE_INT_g")
        elif a.dtype == np.dtype("S1"):
            ft = mt = L.H5Tcopy(L.const("H5T_C_S1_g"))
        else:
            raise TypeError(a.dtype)
        dcpl = 0
        if a.ndim == 0:
            s = L.H5Screate(H5S_SCALAR)
        else:
            for n in a.shape:
                self.dim(n)
            ext = (ctypes.c_uint64 * a.ndim)(*a.shape)
            s = L.H5Screate_simple(a.ndim, ext, None)
            if a.size:
                dcpl = L.H5Pcreate(L.const("H5P_CLS_DATASET_CREATE_ID_g"))
                ch = tuple(chunk) if chunk else tuple(max(1, min(n, 16 if a.ndim > 1 else 64)) for n in a.shape)
                assert L.H5Pset_chunk(dcpl, a.ndim, (ctypes.c_uint64 * a.ndim)(*ch)) >= 0
                if self.deflate and a.dtype != np.dtype("S1"):
                    assert L.H5Pset_shuffle(dcpl) >= 0 and L.H5Pset_deflate(dcpl, self.deflate) >= 0
        d = L.H5Dcreate2(self.fid, name.encode(), ft, s, 0, dcpl, 0)
        assert d >= 0, name
        if a.size:
            assert L.H5Dwrite(d, mt, 0, 0, 0, a.ctypes.data_as(ctypes.c_void_p)) >= 0, name
        L.H5Dclose(d); L.H5Sclose(s)
        if dcpl:
            L.H5Pclose(dcpl)
        if a.dtype == np.dtype("S1"):
            L.H5Tclose(ft)

    def close(self):
        self.L.H5Fclose(self.fid)


def write_coefficient_file(path, raw, is_lw, latest=False):
    """`raw` (kdist_load.synth_raw: the reference loader's names, Fortran orientation) as a coefficient file in netCDF's C order."""
    w = Writer(path, latest=latest)

    def var(name, a, dtype):
        w.var(name, np.ascontiguousarray(np.transpose(np.asarray(a))).astype(dtype))

    def scalar(name, x):
        w.var(name, np.array(float(x)))

    def strings(name, lst, width=32):
        a = np.array([list(s.ljust(width)) for s in lst], dtype="S1") if lst else np.zeros((0, width), dtype="S1")
        w.var(name, a)

    strings("gas_names", raw["gas_names"])
    for n in ("key_species", "bnd_limits_gpt", "minor_limits_gpt_lower", "minor_limits_gpt_upper", "kminor_start_lower", "kminor_start_upper"):
        var(n, raw[n], np.int32)
    for n in ("bnd_limits_wavenumber", "press_ref", "temp_ref", "vmr_ref", "kmajor", "kminor_lower", "kminor_upper"):
        var(n, raw[n], np.float64)
    for n in ("press_ref_trop", "absorption_coefficient_ref_P", "absorption_coefficient_ref_T"):
        scalar(n, raw[n])
    for n in ("gas_minor", "identifier_minor", "minor_gases_lower", "minor_gases_upper", "scaling_gas_lower", "scaling_gas_upper"):
        strings(n, raw[n])
    for n in ("minor_scales_with_density_lower", "minor_scales_with_density_upper", "scale_by_complement_lower", "scale_by_complement_upper"):
        var(n, np.asarray(raw[n]).astype(np.int32), np.int32)
    if is_lw:
        for n in ("totplnk", "plank_fraction", "optimal_angle_fit"):
            var(n, raw[n], np.float64)
    else:
        for n in ("rayl_lower", "rayl_upper", "solar_source_quiet", "solar_source_facular", "solar_source_sunspot"):
            var(n, raw[n], np.float64)
        for n in ("tsi_default", "mg_default", "sb_default"):
            scalar(n, raw[n])
    w.close()
