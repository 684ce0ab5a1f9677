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
