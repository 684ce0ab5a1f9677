"""Load-time reductions of a k-distribution: from the RAW contents of an RRTMGP coefficient file to the
arrays the kernels consume.

The reference does this in ``ty_gas_optics_rrtmgp%load`` -> ``init_abs_coeffs``
(rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:938-1145, :1151-1381): keep the gases the host model provides, drop the
minor-absorber intervals of missing gases and compact their coefficient table (``reduce_minor_arrays`` :1790-1907),
map key-species pairs to flavors (``create_key_species_reduce`` :1752-1786, ``create_flavor`` :1598-1632,
``create_gpoint_flavor`` :1930-1946), look up gas indices of the minor absorbers and their scaling gases
(``create_idx_minor`` :1637-1657, ``create_idx_minor_scaling`` :1663-1677), transpose the coefficient tables to
temperature-fastest order (:1301-1318, :1016-1017) and derive the grid scalars (:1325-1365).  This module restates
those steps in numpy on a plain dict of raw arrays named as in the file
(rrtmgp/data-loading-examples/mo_optics_utils_rrtmgp.F90:102-182); ``tools/netcdf_to_npz.py`` fills that dict from a
netCDF file where netCDF exists, ``synth_raw`` makes a synthetic one with the real names, shapes and string tables.

Raw arrays use the FORTRAN orientation of the reference's reader (first index fastest), e.g. ``kmajor(gpt, eta,
pressure+1, temperature)``; string tables are lists of str.  Validated against the reference's own ``load`` compiled
with flang (oracle/build_load_check.sh, tests/test_kdist_load.py).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

from .synth import F, KDist


def _loc(name: str, names: Sequence[str]) -> int:
    """string_loc_in_array: 1-based position, -1 if absent (case-insensitive, blanks trimmed)."""
    key = name.strip().lower()
    for i, n in enumerate(names):
        if n.strip().lower() == key:
            return i + 1
    return -1


def create_key_species_reduce(gas_names, gas_names_red, key_species):
    """:1752-1786 -- key_species (2, 2, nbnd) holds 1-based indices into gas_names (0 = none)."""
    red = np.zeros_like(key_species)
    present = np.ones(len(gas_names), dtype=bool)
    for idx in np.ndindex(*key_species.shape):
        k = int(key_species[idx])
        if k != 0:
            red[idx] = _loc(gas_names[k - 1], gas_names_red)
            if red[idx] == -1:
                present[k - 1] = False
    return red, present


def _rewrite_pair(pair):
    """:1568-1576 -- (0,0) becomes (2,2)."""
    return (2, 2) if tuple(pair) == (0, 0) else tuple(int(x) for x in pair)


def create_flavor(key_species):
    """:1598-1632 -- unique key-species pairs in order of first appearance (band-major, lower before upper)."""
    flav: List[tuple] = []
    for ibnd in range(key_species.shape[2]):
        for iatm in range(key_species.shape[1]):
            p = _rewrite_pair(key_species[:, iatm, ibnd])
            if p not in flav:
                flav.append(p)
    return F(np.array(flav, dtype=np.int32).T.reshape(2, len(flav)), np.int32)


def create_gpoint_flavor(key_species, gpt2band, flavor):
    """:1930-1946 -- (2, ngpt) 1-based flavor index per g-point and regime (-1 if the pair is not a flavor)."""
    pairs = [tuple(int(x) for x in flavor[:, i]) for i in range(flavor.shape[1])]
    out = np.empty((2, len(gpt2band)), dtype=np.int32)
    for g, b in enumerate(gpt2band):
        for iatm in range(2):
            p = _rewrite_pair(key_species[:, iatm, b - 1])
            out[iatm, g] = pairs.index(p) + 1 if p in pairs else -1
    return F(out, np.int32)


def reduce_minor_arrays(available, gas_minor, identifier_minor, kminor, minor_gases, limits, scales_with_density,
                        scaling_gas, scale_by_complement, kminor_start):
    """:1790-1907 -- keep the intervals whose gas is available; compact kminor; transpose it to (ntemp, neta, nk).
    ``kminor`` raw: (ncontributors, neta, ntemp)."""
    nm = len(minor_gases)
    present = np.array([_loc(gas_minor[_loc(minor_gases[i], identifier_minor) - 1], available) > 0 for i in range(nm)], dtype=bool)
    ng = (limits[1] - limits[0] + 1).astype(int)
    tot_g = int(ng[present].sum())
    keep = np.nonzero(present)[0]
    if present.all():
        k_red_t = np.array(kminor, copy=True)
        start_red = np.array(kminor_start, dtype=np.int32)
    else:
        k_red_t = np.zeros((tot_g,) + kminor.shape[1:], dtype=kminor.dtype)
        start_red = np.zeros(keep.size, dtype=np.int32)
        n_elim, icnt = 0, 0
        for i in range(nm):
            if present[i]:
                start_red[icnt] = kminor_start[i] - n_elim
                k_red_t[start_red[icnt] - 1:start_red[icnt] - 1 + ng[i]] = kminor[kminor_start[i] - 1:kminor_start[i] - 1 + ng[i]]
                icnt += 1
            else:
                n_elim += ng[i]
    return dict(kminor=F(np.transpose(k_red_t, (2, 1, 0))), minor_gases=[minor_gases[i] for i in keep],
                limits=F(limits[:, keep], np.int32), scales_with_density=F(np.asarray(scales_with_density, dtype=bool)[keep], np.bool_),
                scaling_gas=[scaling_gas[i] for i in keep], scale_by_complement=F(np.asarray(scale_by_complement, dtype=bool)[keep], np.bool_),
                kminor_start=F(start_red, np.int32))


def init_from_raw(raw: Dict[str, object], available_gases: Sequence[str]) -> KDist:
    """``ty_gas_optics_rrtmgp%load`` on a raw table: returns the kernel-side k-distribution (what kdist_io stores)."""
    gas_names = list(raw["gas_names"])
    is_lw = "totplnk" in raw
    present = [_loc(g, available_gases) > 0 for g in gas_names]
    names_red = [g for g, p in zip(gas_names, present) if p]                      # this%gas_names :1224
    ngas = len(names_red)
    vmr_ref = np.asarray(raw["vmr_ref"])                                          # (2, nextabsorbers, ntemp)
    vmr_red = np.empty((vmr_ref.shape[0], ngas + 1, vmr_ref.shape[2]))
    vmr_red[:, 0, :] = vmr_ref[:, 0, :]                                           # :1230
    for i, g in enumerate(names_red):
        vmr_red[:, i + 1, :] = vmr_ref[:, _loc(g, gas_names), :]                  # vmr_ref(:, idx+1, :), 1-based :1233
    A: Dict[str, np.ndarray] = {"vmr_ref": F(vmr_red)}
    red = {}
    for reg in ("lower", "upper"):
        red[reg] = reduce_minor_arrays(available_gases, raw["gas_minor"], raw["identifier_minor"], np.asarray(raw[f"kminor_{reg}"]),
                                       list(raw[f"minor_gases_{reg}"]), np.asarray(raw[f"minor_limits_gpt_{reg}"]),
                                       raw[f"minor_scales_with_density_{reg}"], list(raw[f"scaling_gas_{reg}"]),
                                       raw[f"scale_by_complement_{reg}"], np.asarray(raw[f"kminor_start_{reg}"]))
        r = red[reg]
        A[f"kminor_{reg}"], A[f"minor_limits_gpt_{reg}"] = r["kminor"], r["limits"]
        A[f"minor_scales_with_density_{reg}"], A[f"scale_by_complement_{reg}"] = r["scales_with_density"], r["scale_by_complement"]
        A[f"kminor_start_{reg}"] = r["kminor_start"]
        # create_idx_minor :1637-1657, create_idx_minor_scaling :1663-1677
        A[f"idx_minor_{reg}"] = F([_loc(raw["gas_minor"][_loc(m, raw["identifier_minor"]) - 1], names_red) for m in r["minor_gases"]], np.int32)
        A[f"idx_minor_scaling_{reg}"] = F([_loc(s, names_red) for s in r["scaling_gas"]], np.int32)
    kmajor = np.asarray(raw["kmajor"])                                             # (gpt, eta, pres+1, temp)
    A["kmajor"] = F(np.transpose(kmajor, (3, 1, 2, 0)))                            # :1301-1304
    press_ref, temp_ref = np.asarray(raw["press_ref"], dtype=float), np.asarray(raw["temp_ref"], dtype=float)
    A["press_ref"], A["temp_ref"], A["press_ref_log"] = F(press_ref), F(temp_ref), F(np.log(press_ref))
    if "rayl_lower" in raw:                                                        # :1314-1318 krayl(temp, eta, gpt, 2)
        A["krayl"] = F(np.stack([np.transpose(np.asarray(raw["rayl_lower"]), (2, 1, 0)),
                                 np.transpose(np.asarray(raw["rayl_upper"]), (2, 1, 0))], axis=-1))
    key_species = np.asarray(raw["key_species"], dtype=np.int32)                   # (2, 2, nbnd): (pair, lower/upper, band)
    ks_red, ks_present = create_key_species_reduce(gas_names, names_red, key_species)
    missing = [g for g, ok in zip(gas_names, ks_present) if not ok]
    if missing:                                                                    # check_key_species_present_init :1383-1397
        raise ValueError("gas_optics: required gases " + " ".join(missing) + " are not provided")
    band2gpt = np.asarray(raw["bnd_limits_gpt"], dtype=np.int32)                   # (2, nbnd)
    ngpt, nbnd = int(band2gpt.max()), band2gpt.shape[1]
    gpt2band = np.zeros(ngpt, dtype=np.int32)
    for b in range(nbnd):
        gpt2band[band2gpt[0, b] - 1:band2gpt[1, b]] = b + 1
    A["band_lims_gpt"], A["gpoint_bands"] = F(band2gpt, np.int32), F(gpt2band, np.int32)
    A["flavor"] = create_flavor(ks_red)
    A["gpoint_flavor"] = create_gpoint_flavor(ks_red, gpt2band, A["flavor"])
    S = {"temp_ref_min": float(temp_ref[0]), "temp_ref_max": float(temp_ref[-1]), "press_ref_min": float(press_ref[-1]),
         "press_ref_max": float(press_ref[0]),
         "press_ref_log_delta": float((np.log(press_ref[-1]) - np.log(press_ref[0])) / (press_ref.size - 1)),     # :1363
         "temp_ref_delta": float((temp_ref[-1] - temp_ref[0]) / (temp_ref.size - 1)),
         "press_ref_trop_log": float(np.log(float(raw["press_ref_trop"]))), "idx_h2o": _loc("h2o", names_red)}
    if is_lw:
        A["totplnk"] = F(np.asarray(raw["totplnk"]))                                # (nPlanckTemp, nbnd)
        A["planck_frac"] = F(np.transpose(np.asarray(raw["plank_fraction"]), (3, 1, 2, 0)))   # :1016-1017
        A["optimal_angle_fit"] = F(np.asarray(raw["optimal_angle_fit"]))
        S["nPlanckTemp"] = int(A["totplnk"].shape[0])
        S["totplnk_delta"] = float((temp_ref[-1] - temp_ref[0]) / (A["totplnk"].shape[0] - 1))                    # :1025
    else:
        q, f, s = (np.asarray(raw[k], dtype=float) for k in ("solar_source_quiet", "solar_source_facular", "solar_source_sunspot"))
        mg, sb = float(raw["mg_default"]), float(raw["sb_default"])
        # load_ext :1131-1142 / set_solar_variability :776-791 with the file's default indices
        A["solar_source"] = F(q + (mg - 0.1495954) * f + (sb - 0.00066696) * s)
        for k, v in (("solar_source_quiet", q), ("solar_source_facular", f), ("solar_source_sunspot", s)):
            A[k] = F(v)
    kd = KDist("lw" if is_lw else "sw", ngas, int(A["flavor"].shape[1]), int(kmajor.shape[1]), int(press_ref.size),
               int(temp_ref.size), nbnd, ngpt, arrays=A, scalars=S)
    kd.scalars["gas_names"] = names_red  # not an ABI array; kept for drivers that map host gases to indices
    return kd


# --------------------------------------------------------------------------------------
# synthetic RAW table: the file's variable names, orientations and string tables
# --------------------------------------------------------------------------------------
FILE_GASES = ["h2o", "co2", "o3", "n2o", "co", "ch4", "o2", "n2", "ccl4", "cfc11"]


def synth_raw(kind: str = "lw", seed: int = 99, ngpt: int = 64, nbnd: int = 4, ntemp: int = 14, npres: int = 59, neta: int = 9,
              nminor_lower: int = 11, nminor_upper: int = 7) -> Dict[str, object]:
    """A seeded raw table: 10 file gases (two of them minor-only), key species per band and regime including a
    single-species pair and the (0,0) pair, minor absorbers with identifiers such as ``h2o_self`` / ``h2o_frgn``, scaling
    gases or none, whole-band minor intervals, 16-aligned bands."""
    rng = np.random.default_rng(seed + (0 if kind == "lw" else 5))
    gpb = ngpt // nbnd
    raw: Dict[str, object] = {"gas_names": list(FILE_GASES)}
    pairs = [(1, 2), (1, 0), (2, 3), (1, 4), (0, 0), (2, 0), (6, 1), (3, 0)]
    ks = np.zeros((2, 2, nbnd), dtype=np.int32)
    for b in range(nbnd):
        for iatm in range(2):
            ks[:, iatm, b] = pairs[int(rng.integers(0, len(pairs)))]
    raw["key_species"] = F(ks, np.int32)
    raw["bnd_limits_gpt"] = F(np.stack([1 + gpb * np.arange(nbnd), gpb * (1 + np.arange(nbnd))]), np.int32)
    edges = np.linspace(10.0, 3250.0, nbnd + 1) if kind == "lw" else np.linspace(820.0, 50000.0, nbnd + 1)
    raw["bnd_limits_wavenumber"] = F(np.stack([edges[:-1], edges[1:]]))
    raw["press_ref"] = F(np.exp(np.linspace(np.log(109663.31), np.log(1.005), npres)))
    raw["temp_ref"] = F(160.0 + 15.0 * np.arange(ntemp))
    raw["press_ref_trop"], raw["absorption_coefficient_ref_P"], raw["absorption_coefficient_ref_T"] = 9948.431564193395, 101325.0, 296.0
    base = np.array([1.0, 5e-3, 4e-4, 2e-6, 3e-7, 1e-7, 1.7e-6, 0.209, 0.781, 1e-10, 2e-10])
    raw["vmr_ref"] = F(base[None, :, None] * np.exp(rng.uniform(-0.7, 0.7, size=(2, len(FILE_GASES) + 1, ntemp))))
    raw["kmajor"] = F(np.exp(-57.0 + rng.uniform(-2.0, 6.0, size=(ngpt, neta, npres + 1, ntemp))))
    ident = ["h2o_self", "h2o_frgn", "co2", "o3", "n2o", "ch4", "o2", "n2", "ccl4", "cfc11"]
    gas_of = ["h2o", "h2o", "co2", "o3", "n2o", "ch4", "o2", "n2", "ccl4", "cfc11"]
    raw["identifier_minor"], raw["gas_minor"] = ident, gas_of
    for reg, nmin in (("lower", nminor_lower), ("upper", nminor_upper)):
        bands = np.sort(np.arange(nmin) % nbnd)
        raw[f"minor_limits_gpt_{reg}"] = F(np.stack([1 + gpb * bands, gpb * (1 + bands)]), np.int32)
        raw[f"kminor_start_{reg}"] = F(1 + gpb * np.arange(nmin), np.int32)
        raw[f"kminor_{reg}"] = F(np.exp(-60.0 + rng.uniform(-2.0, 2.0, size=(gpb * nmin, neta, ntemp))))
        raw[f"minor_gases_{reg}"] = [ident[int(i)] for i in rng.integers(0, len(ident), nmin)]
        raw[f"scaling_gas_{reg}"] = [("" if rng.random() < 0.4 else FILE_GASES[int(rng.integers(0, 8))]) for _ in range(nmin)]
        raw[f"minor_scales_with_density_{reg}"] = F(rng.random(nmin) < 0.6, np.bool_)
        raw[f"scale_by_complement_{reg}"] = F(rng.random(nmin) < 0.4, np.bool_)
    if kind == "lw":
        nPl = 196
        tpl = np.linspace(160.0, 355.0, nPl)
        raw["totplnk"] = F(np.outer(tpl ** 4, np.linspace(1.0, 2.0, nbnd)) * 1e-8)
        pf = rng.uniform(0.2, 1.0, size=(ngpt, neta, npres + 1, ntemp)).reshape(nbnd, gpb, neta, npres + 1, ntemp)
        raw["plank_fraction"] = F((pf / pf.sum(axis=1, keepdims=True)).reshape(ngpt, neta, npres + 1, ntemp))
        raw["optimal_angle_fit"] = F(np.stack([rng.uniform(0.05, 0.35, nbnd), rng.uniform(1.5, 1.75, nbnd)]))
    else:
        raw["rayl_lower"] = F(np.exp(-62.0 + rng.uniform(0.0, 3.0, size=(ngpt, neta, ntemp))))
        raw["rayl_upper"] = F(np.exp(-62.0 + rng.uniform(0.0, 3.0, size=(ngpt, neta, ntemp))))
        for k in ("solar_source_quiet", "solar_source_facular", "solar_source_sunspot"):
            raw[k] = F(rng.uniform(0.1, 9.0, ngpt))
        raw["tsi_default"], raw["mg_default"], raw["sb_default"] = 1360.85, 0.1567, 902.7e-6
    return raw
