"""The netCDF layer of the k-distribution reader: tools/netcdf_to_npz.py on a synthetic coefficient file
with the reference's variable names and (C-order) dimension order -- rrtmgp/data-loading-examples/mo_optics_utils_rrtmgp.F90:102-182
reads e.g. kmajor(gpt, mixing_fraction, pressure_interp, temperature) in Fortran order -- written in the netCDF-3 classic format
(scipy; the image has no netCDF4, and the real rrtmgp-data files are netCDF-4).  The converter must hand init_from_raw exactly
the raw table the file was made from, and the .npz it writes must hold the arrays of the direct path."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rte_rrtmgp_amd import kdist_io, kdist_load  # noqa: E402

scipy_io = pytest.importorskip("scipy.io")


def _write_nc(path, raw, is_lw):
    f = scipy_io.netcdf_file(path, "w")
    dims = {}

    def dim(n):
        name = f"d{n}"
        if name not in dims:
            f.createDimension(name, n)
            dims[name] = n
        return name

    def var(name, a, typ):
        a = np.asarray(a)
        c = np.ascontiguousarray(np.transpose(a))  # Fortran orientation -> netCDF C order (dimensions reversed)
        v = f.createVariable(name, typ, tuple(dim(n) for n in c.shape))
        v[...] = c

    def scalar(name, x):
        v = f.createVariable(name, "d", ())
        v.data[()] = float(x)  # (assignValue of a 0-d variable fails with numpy 2)

    def strings(name, lst, width=32):
        a = np.array([list(s.ljust(width)) for s in lst], dtype="S1") if lst else np.zeros((0, width), dtype="S1")
        v = f.createVariable(name, "c", (dim(max(len(lst), 1)) if len(lst) else dim(1), dim(width)))
        if len(lst):
            v[...] = a

    strings("gas_names", raw["gas_names"])
    for n in ("key_species", "bnd_limits_gpt", "minor_limits_gpt_lower", "minor_limits_gpt_upper", "kminor_start_lower", "kminor_start_upper"):
        var(n, np.asarray(raw[n], dtype=np.int32), "i")
    for n in ("bnd_limits_wavenumber", "press_ref", "temp_ref", "vmr_ref", "kmajor", "kminor_lower", "kminor_upper"):
        var(n, raw[n], "d")
    for n in ("press_ref_trop", "absorption_coefficient_ref_P", "absorption_coefficient_ref_T"):
        scalar(n, raw[n])
    for n in ("gas_minor", "identifier_minor", "minor_gases_lower", "minor_gases_upper", "scaling_gas_lower", "scaling_gas_upper"):
        strings(n, raw[n])
    for n in ("minor_scales_with_density_lower", "minor_scales_with_density_upper", "scale_by_complement_lower", "scale_by_complement_upper"):
        var(n, np.asarray(raw[n]).astype(np.int32), "i")
    if is_lw:
        for n in ("totplnk", "plank_fraction", "optimal_angle_fit"):
            var(n, raw[n], "d")
    else:
        for n in ("rayl_lower", "rayl_upper", "solar_source_quiet", "solar_source_facular", "solar_source_sunspot"):
            var(n, raw[n], "d")
        for n in ("tsi_default", "mg_default", "sb_default"):
            scalar(n, raw[n])
    f.close()


@pytest.mark.parametrize("kind", ["lw", "sw"])
def test_converter_reads_a_file_with_the_reference_variable_layout(kind, tmp_path):
    import netcdf_to_npz as conv

    raw = kdist_load.synth_raw(kind)
    nc, npz = str(tmp_path / "k.nc"), str(tmp_path / "k.npz")
    _write_nc(nc, raw, kind == "lw")
    back = conv.read_raw(nc)
    for k, v in raw.items():
        if isinstance(v, list):
            assert back[k] == [s.strip() for s in v], k
        elif isinstance(v, float):
            assert back[k] == v, k
        else:
            assert back[k].shape == np.asarray(v).shape and np.array_equal(back[k], np.asarray(v)), k
    gases = ["h2o", "co2", "o3", "n2o", "co", "ch4", "o2"]
    kd, names = conv.convert(nc, npz, gases)
    assert names == gases
    direct = kdist_load.init_from_raw(raw, gases)
    loaded = kdist_io.load_kdist(npz)
    for k, v in direct.arrays.items():
        assert np.array_equal(loaded.arrays[k], v), k


def _same_raw(back, raw):
    for k, v in raw.items():
        if isinstance(v, list):
            assert back[k] == [s.strip() for s in v], k
        elif isinstance(v, float):
            assert back[k] == v, k
        else:
            assert back[k].shape == np.asarray(v).shape and np.array_equal(back[k], np.asarray(v)), k


@pytest.mark.parametrize("latest", [False, True], ids=["earliest-format", "latest-format"])
@pytest.mark.parametrize("kind", ["lw", "sw"])
def test_converter_reads_a_netcdf4_style_hdf5_file(kind, latest, tmp_path):
    """The rrtmgp-data files are netCDF-4 = HDF5 (missing 4).  Without netCDF4 / h5py in the image the file is written
    by the HDF5 library itself in netCDF-C's layout (tests/h5_netcdf4_writer.py: chunked + shuffle + deflate datasets, size-1
    strings, scalar dataspaces, dimension-scale datasets) and read back through tools/netcdf_to_npz.py -> hdf5_reader: same raw
    table, same .npz as the direct path."""
    from rte_rrtmgp_amd import hdf5_reader

    if not hdf5_reader.available():
        pytest.skip("no HDF5 C library on this machine")
    import h5_netcdf4_writer as h5w
    import netcdf_to_npz as conv

    raw = kdist_load.synth_raw(kind)
    nc, npz = str(tmp_path / "k4.nc"), str(tmp_path / "k4.npz")
    h5w.write_coefficient_file(nc, raw, kind == "lw", latest=latest)
    assert hdf5_reader.is_hdf5(nc)
    variables, dims, close = hdf5_reader.open_netcdf4(nc)
    assert "kmajor" in variables and not any(n.startswith("dim_") for n in variables)  # pure dimensions are not variables
    km = np.asarray(raw["kmajor"])
    assert variables["kmajor"].shape == km.shape[::-1] and variables["kmajor"].dtype == np.float64
    assert set(km.shape) <= set(dims.values())
    assert variables["gas_names"].dtype == np.dtype("S1") and variables["press_ref_trop"].shape == ()
    close()
    _same_raw(conv.read_raw(nc), raw)
    gases = ["h2o", "co2", "o3", "n2o", "co", "ch4", "o2"]
    kd, names = conv.convert(nc, npz, gases)
    assert names == gases
    direct = kdist_load.init_from_raw(raw, gases)
    loaded = kdist_io.load_kdist(npz)
    for k, v in direct.arrays.items():
        assert np.array_equal(loaded.arrays[k], v), k


def test_hdf5_reader_types_and_filters(tmp_path):
    """Every storage form a coefficient file uses, plus the corners: an empty dataset, one chunk smaller than the array in every
    dimension (edge chunks are partial), 1-D char data, a scalar integer."""
    from rte_rrtmgp_amd import hdf5_reader

    if not hdf5_reader.available():
        pytest.skip("no HDF5 C library on this machine")
    import h5_netcdf4_writer as h5w

    rng = np.random.default_rng(5)
    a = rng.standard_normal((7, 5, 3, 11))
    b = rng.integers(-2**31, 2**31 - 1, size=(13, 2), dtype=np.int32)
    c = np.array(list("h2o_and_friends".ljust(32)), dtype="S1")
    for latest in (False, True):
        p = str(tmp_path / f"t{int(latest)}.h5")
        w = h5w.Writer(p, latest=latest)
        w.var("a", a, chunk=(3, 2, 2, 4)); w.var("b", b); w.var("c", c); w.var("e", np.zeros((0, 32), dtype="S1"))
        w.var("s", np.array(2.5)); w.var("i", np.array(7, dtype=np.int32))
        w.close()
        v, dims, close = hdf5_reader.open_netcdf4(p)
        assert np.array_equal(v["a"][...], a) and np.array_equal(v["b"][...], b) and np.array_equal(v["c"][...], c)
        assert v["e"][...].shape == (0, 32) and float(v["s"][...]) == 2.5 and int(v["i"][...]) == 7
        assert sorted(dims.values()) == sorted({7, 5, 3, 11, 13, 2, 32, 0})
        close()
    with pytest.raises(OSError):
        hdf5_reader.open_netcdf4(__file__)
