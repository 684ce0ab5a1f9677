"""Flat on-disk format for a k-distribution as the kernels consume it.

The reference ships its k-distributions as netCDF files and, at load time, reshapes them
(``init_abs_coeffs``, rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:1151-1381: key species -> flavors,
``reduce_minor_arrays`` :1790-1907, ``create_gpoint_flavor`` :1930-1946, ...).  Neither netCDF nor the
data files exist offline, so this module defines the format on the *kernel side* of that step: one
``.npz`` holding exactly the arrays and scalars ``ty_gas_optics_rrtmgp`` hands to the kernels, in the
kernels' own layout (column-major, 1-based index values, ``Bool`` as one byte).  A converter that
performs the load-time reductions on a netCDF file can be run wherever netCDF exists and only has to
produce these names.

File layout (``numpy.savez``; every array Fortran-ordered, float64 / int32 / bool):
    __meta__            json: {"format": "rte-rrtmgp-kdist", "version": 1, "kind": "lw"|"sw", dims..., scalars...}
    <name>              one entry per array of ``frontend.GasOptics.LUT_NAMES`` present in the table
"""
from __future__ import annotations

import json
from typing import Dict

import numpy as np

from .synth import F, KDist

FORMAT, VERSION = "rte-rrtmgp-kdist", 1
_DIMS = ("ngas", "nflav", "neta", "npres", "ntemp", "nbnd", "ngpt")
# what every table must carry (mo_gas_optics_rrtmgp.F90:60-154), and what only one kind has
REQUIRED = ["flavor", "press_ref_log", "temp_ref", "vmr_ref", "gpoint_flavor", "band_lims_gpt", "kmajor",
            "kminor_lower", "kminor_upper", "minor_limits_gpt_lower", "minor_limits_gpt_upper",
            "minor_scales_with_density_lower", "minor_scales_with_density_upper", "scale_by_complement_lower",
            "scale_by_complement_upper", "idx_minor_lower", "idx_minor_upper", "idx_minor_scaling_lower",
            "idx_minor_scaling_upper", "kminor_start_lower", "kminor_start_upper"]
REQUIRED_KIND = {"lw": ["planck_frac", "totplnk"], "sw": ["krayl", "solar_source"]}


def save_kdist(path: str, kd: KDist) -> None:
    """Write ``kd`` to ``path`` (.npz)."""
    meta = {"format": FORMAT, "version": VERSION, "kind": kd.kind}
    meta.update({d: int(getattr(kd, d)) for d in _DIMS})
    meta["scalars"] = {k: (float(v) if isinstance(v, (float, np.floating)) else int(v)) for k, v in kd.scalars.items()}
    arrays = {k: np.asfortranarray(v) for k, v in kd.arrays.items()}
    np.savez(path, __meta__=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)


def validate(kd: KDist) -> None:
    """Shape / range checks of what the kernels rely on (they have no error channel)."""
    a = kd.arrays
    missing = [n for n in REQUIRED + REQUIRED_KIND[kd.kind] if n not in a]
    if missing:
        raise ValueError(f"k-distribution lacks {missing}")
    if a["kmajor"].shape != (kd.ntemp, kd.neta, kd.npres + 1, kd.ngpt):
        raise ValueError(f"kmajor has shape {a['kmajor'].shape}, expected (ntemp, neta, npres+1, ngpt)")
    if a["gpoint_flavor"].shape != (2, kd.ngpt) or a["band_lims_gpt"].shape != (2, kd.nbnd):
        raise ValueError("gpoint_flavor must be (2, ngpt) and band_lims_gpt (2, nbnd)")
    bl = a["band_lims_gpt"]
    if bl[0, 0] != 1 or bl[1, -1] != kd.ngpt or np.any(bl[0, 1:] != bl[1, :-1] + 1):
        raise ValueError("band_lims_gpt must tile 1..ngpt without gaps (1-based, inclusive)")
    if a["gpoint_flavor"].min() < 1 or a["gpoint_flavor"].max() > kd.nflav:
        raise ValueError("gpoint_flavor holds 1-based flavor indices")
    for reg in ("lower", "upper"):
        lim, ks, km = a[f"minor_limits_gpt_{reg}"], a[f"kminor_start_{reg}"], a[f"kminor_{reg}"]
        n = a[f"idx_minor_{reg}"].shape[0]
        if lim.shape != (2, n) or ks.shape != (n,):
            raise ValueError(f"minor_limits_gpt_{reg} / kminor_start_{reg} do not match idx_minor_{reg}")
        if n and (ks.min() < 1 or np.any(ks - 1 + (lim[1] - lim[0] + 1) > km.shape[2])):
            raise ValueError(f"kminor_start_{reg} + interval width runs past kminor_{reg}")
    if kd.kind == "lw" and a["totplnk"].shape[1] != kd.nbnd:
        raise ValueError("totplnk must be (nPlanckTemp, nbnd)")


def load_kdist(path: str, check: bool = True) -> KDist:
    """Read a table written by :func:`save_kdist` (or by a converter that follows the format)."""
    with np.load(path, allow_pickle=False) as z:
        meta = json.loads(bytes(z["__meta__"]).decode())
        if meta.get("format") != FORMAT or meta.get("version") != VERSION:
            raise ValueError(f"{path}: not a {FORMAT} v{VERSION} file")
        arrays: Dict[str, np.ndarray] = {k: F(z[k]) for k in z.files if k != "__meta__"}
    kd = KDist(kind=meta["kind"], arrays=arrays, scalars=dict(meta["scalars"]), **{d: meta[d] for d in _DIMS})
    if check:
        validate(kd)
    return kd
