"""netCDF-4 coefficient files read through the HDF5 C library, bound with ctypes (no h5py / netCDF4 needed).

The rrtmgp-data files the reference loads (rrtmgp/data-loading-examples/mo_optics_utils_rrtmgp.F90:102-182 through
netCDF-Fortran) are netCDF-4, i.e. HDF5 files with one dataset per variable in the root group and one "dimension scale"
dataset per dimension.  Any HDF5 library >= 1.8 reads them whatever file-format generation, chunking or filters they were
written with; this module is the thinnest possible binding of that library: enumerate the root group's datasets, tell
variables from pure dimensions, read a dataset into a numpy array in the file's (C) dimension order.

    variables, dimensions, close = open_netcdf4("rrtmgp-gas-lw-g256.nc")
    kmajor = variables["kmajor"][...]      # (temperature, pressure_interp, mixing_fraction, gpt) as netCDF orders it

The mapping has the part of netCDF4.Dataset.variables' interface that tools/netcdf_to_npz.py uses (`name in variables`,
`variables[name][...]`, `.shape`, `.dtype`).  Library search order: $RTE_HDF5_LIBRARY, the loader's default path
(`ctypes.util.find_library`), then the usual distribution and conda locations.
Host-side file I/O only: nothing here is on the kernel path.
"""
import ctypes
import ctypes.util
import glob
import os
import sys

import numpy as np

_PURE_DIM = b"This is a netCDF dimension but not a netCDF variable."

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8
H5_INDEX_NAME, H5_ITER_INC = 0, 0


class HDF5Unavailable(ImportError):
    pass


def _candidates():
    env = os.environ.get("RTE_HDF5_LIBRARY")
    if env:
        yield env
    for n in ("hdf5", "hdf5_serial"):
        p = ctypes.util.find_library(n)
        if p:
            yield p
    pats = ["/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
            "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*"]
    for root in (os.environ.get("CONDA_PREFIX"), sys.prefix, "/opt/conda", os.environ.get("HDF5_ROOT"), os.environ.get("HDF5_DIR")):
        if root:
            pats.append(os.path.join(root, "lib", "libhdf5.so*"))
    for pat in pats:
        for p in sorted(glob.glob(pat), key=len):
            yield p


class _Lib:
    """The functions used, with their C signatures (hid_t is 64 bits wide from HDF5 1.10 on, an int before)."""

    def __init__(self):
        lib, tried = None, []
        for p in _candidates():
            try:
                lib = ctypes.CDLL(p)
                self.path = p
                break
            except OSError as e:
                tried.append(f"{p}: {e}")
        if lib is None:
            raise HDF5Unavailable("no HDF5 C library found (set RTE_HDF5_LIBRARY=/path/to/libhdf5.so); tried: " + "; ".join(tried[:6]))
        self.lib = lib
        lib.H5open.restype = ctypes.c_int
        if lib.H5open() < 0:
            raise HDF5Unavailable(f"{self.path}: H5open failed")
        a, b, c = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
        lib.H5get_libversion(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        self.version = (a.value, b.value, c.value)
        if self.version < (1, 8, 0):
            raise HDF5Unavailable(f"{self.path}: HDF5 {self.version} is older than 1.8")
        hid = self.hid_t = ctypes.c_int64 if self.version >= (1, 10, 0) else ctypes.c_int
        hsz, cp, vp, ci, cu, st = ctypes.c_uint64, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_size_t
        sigs = {
            "H5Eset_auto2": (ci, [hid, vp, vp]),
            "H5Fopen": (hid, [cp, cu, hid]), "H5Fcreate": (hid, [cp, cu, hid, hid]), "H5Fclose": (ci, [hid]),
            "H5Gget_info": (ci, [hid, vp]),
            "H5Lget_name_by_idx": (ctypes.c_ssize_t, [hid, cp, ci, ci, hsz, cp, st, hid]),
            "H5Dopen2": (hid, [hid, cp, hid]), "H5Dclose": (ci, [hid]), "H5Dget_space": (hid, [hid]), "H5Dget_type": (hid, [hid]),
            "H5Dread": (ci, [hid, hid, hid, hid, hid, vp]), "H5Dwrite": (ci, [hid, hid, hid, hid, hid, vp]),
            "H5Dcreate2": (hid, [hid, cp, hid, hid, hid, hid, hid]),
            "H5Sclose": (ci, [hid]), "H5Sget_simple_extent_ndims": (ci, [hid]),
            "H5Sget_simple_extent_dims": (ci, [hid, ctypes.POINTER(hsz), ctypes.POINTER(hsz)]),
            "H5Screate_simple": (hid, [ci, ctypes.POINTER(hsz), ctypes.POINTER(hsz)]), "H5Screate": (hid, [ci]),
            "H5Tclose": (ci, [hid]), "H5Tget_class": (ci, [hid]), "H5Tget_size": (st, [hid]), "H5Tget_sign": (ci, [hid]),
            "H5Tcopy": (hid, [hid]), "H5Tset_size": (ci, [hid, st]), "H5Tget_super": (hid, [hid]),
            "H5Aexists": (ci, [hid, cp]), "H5Aopen": (hid, [hid, cp, hid]), "H5Aget_type": (hid, [hid]), "H5Aget_space": (hid, [hid]),
            "H5Aread": (ci, [hid, hid, vp]), "H5Aclose": (ci, [hid]),
            "H5Acreate2": (hid, [hid, cp, hid, hid, hid, hid]), "H5Awrite": (ci, [hid, hid, vp]),
            "H5Pcreate": (hid, [hid]), "H5Pclose": (ci, [hid]), "H5Pset_chunk": (ci, [hid, ci, ctypes.POINTER(hsz)]),
            "H5Pset_deflate": (ci, [hid, cu]), "H5Pset_shuffle": (ci, [hid]), "H5Pset_libver_bounds": (ci, [hid, ci, ci]),
        }
        for name, (res, args) in sigs.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
            setattr(self, name, f)
        self.H5Eset_auto2(0, None, None)  # failures come back as negative return values; no stack traces on stderr

    def const(self, name):
        """A library global holding a type / property-class identifier (valid after H5open)."""
        return self.hid_t.in_dll(self.lib, name).value


_lib_singleton = None


def hdf5():
    global _lib_singleton
    if _lib_singleton is None:
        _lib_singleton = _Lib()
    return _lib_singleton


def available():
    try:
        hdf5()
        return True
    except HDF5Unavailable:
        return False


def is_hdf5(path):
    with open(path, "rb") as f:
        return f.read(8) == b"\x89HDF\r\n\x1a\n"


class Variable:
    """One dataset of the root group; `v[...]` (or `v[:]`, any index applied after the full read) gives its values."""

    def __init__(self, owner, name):
        self._o, self.name = owner, name
        L = owner.L
        d = L.H5Dopen2(owner.fid, name.encode(), 0)
        if d < 0:
            raise KeyError(name)
        try:
            s, t = L.H5Dget_space(d), L.H5Dget_type(d)
            nd = L.H5Sget_simple_extent_ndims(s)
            dims = (ctypes.c_uint64 * max(nd, 1))()
            if nd > 0:
                L.H5Sget_simple_extent_dims(s, dims, None)
            self.shape = tuple(int(dims[i]) for i in range(nd))
            self.dtype, self._mem = self._types(t)
            L.H5Tclose(t)
            L.H5Sclose(s)
        finally:
            L.H5Dclose(d)

    def _types(self, t):
        """numpy dtype of the values and the name of the library's matching native memory type."""
        L = self._o.L
        cls, size = L.H5Tget_class(t), int(L.H5Tget_size(t))
        if cls == H5T_ENUM:  # NC_BOOL-like enumerations: read as their base integer
            base = L.H5Tget_super(t)
            try:
                return self._types(base)
            finally:
                L.H5Tclose(base)
        if cls == H5T_FLOAT and size in (4, 8):
            return (np.dtype("f8"), "H5T_NATIVE_DOUBLE_g") if size == 8 else (np.dtype("f4"), "H5T_NATIVE_FLOAT_g")
        if cls == H5T_INTEGER and size in (1, 2, 4, 8):
            signed = L.H5Tget_sign(t) != 0
            names = {1: ("SCHAR", "UCHAR"), 2: ("SHORT", "USHORT"), 4: ("INT", "UINT"), 8: ("LLONG", "ULLONG")}[size]
            return np.dtype(("i" if signed else "u") + str(size)), f"H5T_NATIVE_{names[0 if signed else 1]}_g"
        if cls == H5T_STRING and size >= 1:  # NC_CHAR arrays are fixed-length strings of size 1
            return np.dtype(f"S{size}"), None
        raise TypeError(f"{self.name}: HDF5 type class {cls} of {size} bytes is not one a coefficient file holds")

    @property
    def ndim(self):
        return len(self.shape)

    def read(self):
        L = self._o.L
        out = np.empty(self.shape, dtype=self.dtype)
        d = L.H5Dopen2(self._o.fid, self.name.encode(), 0)
        try:
            if self._mem is None:
                mt = L.H5Tcopy(L.const("H5T_C_S1_g"))
                L.H5Tset_size(mt, self.dtype.itemsize)
            else:
                mt = L.const(self._mem)
            rc = L.H5Dread(d, mt, 0, 0, 0, out.ctypes.data_as(ctypes.c_void_p)) if out.size else 0
            if self._mem is None:
                L.H5Tclose(mt)
            if rc < 0:
                raise OSError(f"H5Dread failed on {self.name} (a filter the library was built without?)")
        finally:
            L.H5Dclose(d)
        return out

    def __getitem__(self, idx):
        return self.read()[idx]

    def __array__(self, dtype=None, copy=None):
        a = self.read()
        return a.astype(dtype) if dtype is not None else a


class _File:
    def __init__(self, path):
        self.L = hdf5()
        self.fid = self.L.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, 0)
        if self.fid < 0:
            raise OSError(f"{path}: not an HDF5 file the library ({self.L.path}) can open")

    def names(self):
        L = self.L
        info = ctypes.create_string_buffer(64)  # H5G_info_t { storage_type; hsize_t nlinks; int64 max_corder; mounted }
        if L.H5Gget_info(self.fid, info) < 0:
            raise OSError("H5Gget_info failed on the root group")
        n = int.from_bytes(info.raw[8:16], sys.byteorder)
        out = []
        for i in range(n):
            ln = L.H5Lget_name_by_idx(self.fid, b".", H5_INDEX_NAME, H5_ITER_INC, i, None, 0, 0)
            buf = ctypes.create_string_buffer(ln + 1)
            L.H5Lget_name_by_idx(self.fid, b".", H5_INDEX_NAME, H5_ITER_INC, i, buf, ln + 1, 0)
            out.append(buf.value.decode())
        return out

    def pure_dimension(self, name):
        """netCDF-4 writes a dimension without a coordinate variable as a dataset whose NAME attribute says so."""
        L = self.L
        d = L.H5Dopen2(self.fid, name.encode(), 0)
        if d < 0:
            return None  # a group or a named type
        try:
            if L.H5Aexists(d, b"NAME") <= 0:
                return False
            a = L.H5Aopen(d, b"NAME", 0)
            t = L.H5Aget_type(a)
            n = int(L.H5Tget_size(t))
            ok = L.H5Tget_class(t) == H5T_STRING and n < 4096
            val = b""
            if ok:
                mt = L.H5Tcopy(L.const("H5T_C_S1_g"))
                L.H5Tset_size(mt, n)
                buf = ctypes.create_string_buffer(n + 1)
                if L.H5Aread(a, mt, buf) >= 0:
                    val = buf.raw[:n]
                L.H5Tclose(mt)
            L.H5Tclose(t)
            L.H5Aclose(a)
            return val.startswith(_PURE_DIM)
        finally:
            L.H5Dclose(d)

    def close(self):
        if self.fid >= 0:
            self.L.H5Fclose(self.fid)
            self.fid = -1


def open_netcdf4(path):
    """(variables, dimensions, close): the root group's variables by name, the sizes of the dimensions that have no coordinate
    variable, and the function that closes the file."""
    f = _File(path)
    variables, dimensions = {}, {}
    for name in f.names():
        kind = f.pure_dimension(name)
        if kind is None:
            continue
        v = Variable(f, name)
        if kind:
            dimensions[name] = v.shape[0] if v.shape else 1
        else:
            variables[name] = v
    return variables, dimensions, f.close
