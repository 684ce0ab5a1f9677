"""Seeded synthetic inputs for the RRTMGP/RTE hot path (numpy, host side).

The real k-distribution files (rrtmgp-data v1.9.1, fetched by the reference at
build time, reference rrtmgp/CMakeLists.txt:13-24) are not available offline, so
tests and the benchmark use a *synthetic* k-distribution that has exactly the
shapes, index conventions and consistency rules the kernels rely on:

  * array shapes/orders as delivered by ``ty_gas_optics_rrtmgp%load`` AFTER its
    load-time transposes (temperature fastest): reference
    rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:1301-1318,1016-1017;
  * ``flavor``/``gpoint_flavor`` constant within a band (:1598-1632,1930-1946);
  * ``kminor_start``/``minor_limits_gpt`` concatenated g-point intervals (:1885-1897);
  * ``press_ref`` descending and log-uniform, ``temp_ref`` ascending uniform (:1356-1365).

and a synthetic atmosphere in the spirit of the reference all-sky example's
RCEMIP-like profile (reference examples/all-sky/rrtmgp_allsky.F90:496-587) with a
seeded per-column perturbation, so neighbouring columns are similar but not equal.

All arrays are numpy, Fortran (column-major) order, dtype float64/int32/bool,
i.e. exactly what crosses the C ABI of include/rte_rrtmgp_kernels.h.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict

import numpy as np

# physical constants, reference rte/kernels/mo_gas_optics_constants.F90:17,32-35
AVOGAD = 6.02214076e23
M_DRY = 0.028964
M_H2O = 0.018016
GRAV = 9.80665

GAS_NAMES = ["h2o", "co2", "o3", "n2o", "co", "ch4", "o2", "n2"]  # indices 1..8; 0 = dry air


def F(a, dtype=None):
    """Fortran-ordered array of the ABI dtype."""
    return np.asfortranarray(a, dtype=dtype)


@dataclass
class KDist:
    """Synthetic k-distribution: the fields of ``ty_gas_optics_rrtmgp`` the kernels receive
    (reference rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:60-154)."""

    kind: str  # "lw" | "sw"
    ngas: int
    nflav: int
    neta: int
    npres: int
    ntemp: int
    nbnd: int
    ngpt: int
    arrays: Dict[str, np.ndarray] = field(default_factory=dict)
    scalars: Dict[str, float] = field(default_factory=dict)

    def __getattr__(self, name):
        d = object.__getattribute__(self, "arrays")
        if name in d:
            return d[name]
        s = object.__getattribute__(self, "scalars")
        if name in s:
            return s[name]
        raise AttributeError(name)


def _planck_band_integrals(temps: np.ndarray, edges_cm: np.ndarray) -> np.ndarray:
    """Band-integrated Planck radiance [W m-2 sr-1] on a (T, band) grid (trapezoid rule)."""
    h, c, kb = 6.62607015e-34, 2.99792458e8, 1.380649e-23
    out = np.zeros((temps.size, edges_cm.size - 1))
    for b in range(edges_cm.size - 1):
        nu = np.linspace(edges_cm[b], edges_cm[b + 1], 400) * 100.0  # m-1
        x = h * c * nu[None, :] / (kb * temps[:, None])
        bnu = 2.0 * h * c * c * nu[None, :] ** 3 / np.expm1(x)
        out[:, b] = np.trapezoid(bnu, nu, axis=1)
    return out


def make_kdist(kind: str = "lw", seed: int = 1234, ngpt: int | None = None, nbnd: int | None = None,
               ntemp: int = 14, npres: int = 59, neta: int = 9, nflav: int = 10, ngas: int = 8,
               nminor_lower: int | None = None, nminor_upper: int | None = None, minor_distribution: str = "even") -> KDist:
    """Seeded synthetic k-distribution with the g256 (LW) / g224 (SW) shapes by default.
    ``minor_distribution``: "even" -- every band gets the same number of minor-absorber intervals per regime (4 lower, 2-3
    upper); "ragged" -- the same totals spread unevenly, 1 ... 8 intervals per band in the lower atmosphere and 0 ... 6 in
    the upper, as in real coefficient files (what ``reduce_minor_arrays`` leaves is whatever the file holds,
    rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:1790-1907)."""
    rng = np.random.default_rng(seed + (0 if kind == "lw" else 7))
    if ngpt is None:
        ngpt = 256 if kind == "lw" else 224
    if nbnd is None:
        nbnd = 16 if kind == "lw" else 14
    assert ngpt % nbnd == 0
    gpb = ngpt // nbnd
    if nminor_lower is None:
        nminor_lower = 4 * nbnd
    if nminor_upper is None:
        nminor_upper = (9 * nbnd) // 4
    kd = KDist(kind, ngas, nflav, neta, npres, ntemp, nbnd, ngpt)
    A, S = kd.arrays, kd.scalars

    # ---- reference grids (mo_gas_optics_rrtmgp.F90:1325-1365)
    press_ref = np.exp(np.linspace(np.log(109663.31), np.log(1.005), npres))
    temp_ref = 160.0 + 15.0 * np.arange(ntemp) if ntemp == 14 else np.linspace(160.0, 355.0, ntemp)
    A["press_ref"] = F(press_ref)
    A["press_ref_log"] = F(np.log(press_ref))
    A["temp_ref"] = F(temp_ref)
    S["press_ref_log_delta"] = float((np.log(press_ref.min()) - np.log(press_ref.max())) / (npres - 1))
    S["temp_ref_min"] = float(temp_ref[0])
    S["temp_ref_max"] = float(temp_ref[-1])
    S["temp_ref_delta"] = float((temp_ref[-1] - temp_ref[0]) / (ntemp - 1))
    S["press_ref_min"] = float(press_ref.min())
    S["press_ref_max"] = float(press_ref.max())
    S["press_ref_trop_log"] = float(np.log(9948.431564193395))

    # ---- reference volume mixing ratios vmr_ref(2, 0:ngas, ntemp) > 0
    base = np.array([1.0, 5e-3, 4e-4, 2e-6, 3e-7, 1e-7, 1.7e-6, 0.209, 0.781])[: ngas + 1]
    vmr_ref = base[None, :, None] * np.exp(rng.uniform(-0.7, 0.7, size=(2, ngas + 1, ntemp)))
    A["vmr_ref"] = F(vmr_ref)

    # ---- flavors: unique pairs over 0..ngas (never (0,0): mo_gas_optics_rrtmgp.F90:1568-1576)
    pairs = [(1, 0), (1, 2), (1, 3), (2, 0), (2, 2), (1, 4), (3, 0), (1, 6), (2, 4), (7, 0), (6, 0), (3, 2)]
    assert nflav <= len(pairs)
    flavor = np.array(pairs[:nflav], dtype=np.int32).T  # (2, nflav)
    A["flavor"] = F(flavor, np.int32)
    band_lims = np.stack([1 + gpb * np.arange(nbnd), gpb * (1 + np.arange(nbnd))]).astype(np.int32)
    A["band_lims_gpt"] = F(band_lims, np.int32)
    A["gpoint_bands"] = F(np.repeat(np.arange(1, nbnd + 1), gpb), np.int32)
    band_flav = rng.integers(1, nflav + 1, size=(2, nbnd))
    A["gpoint_flavor"] = F(np.repeat(band_flav, gpb, axis=1), np.int32)  # (2, ngpt)

    # ---- major absorption coefficients kmajor(ntemp, neta, npres+1, ngpt)
    # increasing with g inside a band (sorted k-distribution), smooth in T, eta, p
    gfrac = (np.arange(ngpt) % gpb + 0.5) / gpb
    band_off = np.repeat(rng.uniform(-1.5, 1.5, nbnd), gpb)
    logk = (-57.0 + band_off[None, None, None, :] + 9.0 * gfrac[None, None, None, :] ** 1.5
            + 0.06 * np.arange(ntemp)[:, None, None, None]
            + 0.15 * np.arange(neta)[None, :, None, None]
            - 0.02 * np.arange(npres + 1)[None, None, :, None]
            + rng.uniform(-0.25, 0.25, size=(ntemp, neta, npres + 1, ngpt)))
    A["kmajor"] = F(np.exp(logk))

    # ---- minor absorbers, per regime
    for reg, nmin in (("lower", nminor_lower), ("upper", nminor_upper)):
        if minor_distribution == "ragged":
            lo_cnt, hi_cnt = (1, 8) if reg == "lower" else (0, 6)
            assert lo_cnt * nbnd <= nmin <= hi_cnt * nbnd
            rr = np.random.default_rng(seed + (91 if reg == "lower" else 92))
            counts = np.full(nbnd, lo_cnt)
            while counts.sum() < nmin:
                b = int(rr.integers(0, nbnd))
                if counts[b] < hi_cnt:
                    counts[b] += 1
            bands = np.repeat(np.arange(nbnd), counts)
        else:
            bands = np.sort(np.arange(nmin) % nbnd)  # intervals are whole bands, ascending
        lims = np.stack([band_lims[0, bands], band_lims[1, bands]]).astype(np.int32)
        A[f"minor_limits_gpt_{reg}"] = F(lims, np.int32)
        A[f"kminor_start_{reg}"] = F(1 + gpb * np.arange(nmin), np.int32)
        nk = gpb * nmin
        logkm = (-60.0 + rng.uniform(-2.0, 2.0, size=(1, 1, nk))
                 + 0.05 * np.arange(ntemp)[:, None, None] + 0.1 * np.arange(neta)[None, :, None]
                 + rng.uniform(-0.2, 0.2, size=(ntemp, neta, nk)))
        A[f"kminor_{reg}"] = F(np.exp(logkm))
        A[f"idx_minor_{reg}"] = F(rng.integers(1, ngas + 1, nmin), np.int32)
        scal = rng.integers(1, ngas + 1, nmin)
        scal[rng.random(nmin) < 0.4] = -1  # "no scaling gas" (string_loc_in_array -> -1)
        A[f"idx_minor_scaling_{reg}"] = F(scal, np.int32)
        A[f"minor_scales_with_density_{reg}"] = F(rng.random(nmin) < 0.6, np.bool_)
        A[f"scale_by_complement_{reg}"] = F(rng.random(nmin) < 0.4, np.bool_)
    S["idx_h2o"] = 1

    if kind == "lw":
        # ---- Planck tables
        nPlanckTemp = 196
        tpl = np.linspace(temp_ref[0], temp_ref[-1], nPlanckTemp)
        edges = np.array([10, 250, 500, 630, 700, 820, 980, 1080, 1180, 1390, 1480, 1800, 2080, 2250,
                          2380, 2600, 3250.0])
        if nbnd != 16:
            edges = np.linspace(10.0, 3250.0, nbnd + 1)
        A["totplnk"] = F(_planck_band_integrals(tpl, edges))  # (nPlanckTemp, nbnd)
        S["totplnk_delta"] = float((temp_ref[-1] - temp_ref[0]) / (nPlanckTemp - 1))
        S["nPlanckTemp"] = nPlanckTemp
        pf = rng.uniform(0.2, 1.0, size=(ntemp, neta, npres + 1, ngpt)) * (0.3 + gfrac)[None, None, None, :]
        pf = pf.reshape(ntemp, neta, npres + 1, nbnd, gpb)
        pf /= pf.sum(axis=-1, keepdims=True)
        A["planck_frac"] = F(pf.reshape(ntemp, neta, npres + 1, ngpt))
        fit = np.stack([rng.uniform(0.05, 0.35, nbnd), rng.uniform(1.5, 1.75, nbnd)])
        A["optimal_angle_fit"] = F(fit)  # (2, nbnd): D = fit1*exp(-tau)+fit2 >= 1
    else:
        A["krayl"] = F(np.exp(-62.0 + 3.0 * gfrac[None, None, :, None]
                              + rng.uniform(-0.3, 0.3, size=(ntemp, neta, ngpt, 2))))
        ss = rng.uniform(0.5, 1.5, ngpt)
        A["solar_source"] = F(1360.8 * ss / ss.sum())
    return kd


@dataclass
class Atmosphere:
    """Profile inputs in the layout the Fortran frontend hands to the kernels."""

    ncol: int
    nlay: int
    top_at_1: bool
    play: np.ndarray  # (ncol, nlay)
    plev: np.ndarray  # (ncol, nlay+1)
    tlay: np.ndarray
    tlev: np.ndarray
    tsfc: np.ndarray  # (ncol)
    vmr: np.ndarray  # (ncol, nlay, ngas)  gas 1..ngas
    col_dry: np.ndarray  # (ncol, nlay)
    col_gas: np.ndarray  # (ncol, nlay, 0:ngas)


def col_dry_from_plev(vmr_h2o: np.ndarray, plev: np.ndarray) -> np.ndarray:
    """Host restatement of ``get_layer_number`` (reference rte/kernels/mo_gas_optics_utils.F90:127-152)."""
    delta_plev = np.abs(plev[:, :-1] - plev[:, 1:])
    fact = 1.0 / (1.0 + vmr_h2o)
    m_air = (M_DRY + M_H2O * vmr_h2o) * fact
    return 10.0 * delta_plev * AVOGAD * fact / (1000.0 * m_air * 100.0 * GRAV)


def make_atmosphere(ncol: int, nlay: int = 60, seed: int = 42, top_at_1: bool = False,
                    ngas: int = 8, kdist: KDist | None = None, climate: str = "rce") -> Atmosphere:
    """RCEMIP-flavoured columns: surface ~300 K, 6.7 K/km lapse rate to a ~15 km tropopause,
    warming stratosphere, exponential pressure up to ~70 km; each column gets a seeded, vertically
    smooth temperature/humidity/surface-pressure perturbation.
    ``climate="sites"``: columns spread like the RFMIP sites instead -- polar to tropical surfaces (235 ... 305 K),
    sea level to high plateaus (650 ... 1030 hPa), tropopause height and humidity following the surface temperature."""
    rng = np.random.default_rng(seed)
    z_lev = np.linspace(0.0, 70.0, nlay + 1)  # km, index 0 = surface
    z_lay = 0.5 * (z_lev[:-1] + z_lev[1:])
    if climate == "sites":
        ps = rng.uniform(65000.0, 103000.0, ncol)
        sst = rng.uniform(235.0, 305.0, ncol)
    else:
        ps = 100000.0 * (1.0 + 0.03 * rng.standard_normal(ncol))
        sst = 300.0 + 4.0 * rng.standard_normal(ncol)

    def temperature(z, c):
        ztrop = (8.0 + 8.0 * (c["sst"] - 235.0) / 70.0 if climate == "sites" else 15.0) + 1.0 * c["a"]
        t_trop = c["sst"][:, None] - 6.7 * np.minimum(z[None, :], ztrop[:, None])
        strat = np.clip((z[None, :] - ztrop[:, None]), 0.0, None)
        t = t_trop + 2.2 * np.minimum(strat, 33.0) - 2.8 * np.clip(strat - 33.0, 0.0, None)
        wob = (c["b"][:, None] * np.sin(2 * np.pi * z[None, :] / 23.0 + c["ph"][:, None])
               + c["d"][:, None] * np.cos(2 * np.pi * z[None, :] / 9.0))
        return np.clip(t + 2.0 * wob, 165.0, 350.0)

    c = {"a": rng.standard_normal(ncol), "b": rng.standard_normal(ncol), "d": rng.standard_normal(ncol),
         "ph": rng.uniform(0, 2 * np.pi, ncol), "sst": sst}
    tlev = temperature(z_lev, c)
    tlay = temperature(z_lay, c)
    scale_h = 7.4 + 0.15 * rng.standard_normal(ncol)
    plev = ps[:, None] * np.exp(-z_lev[None, :] / scale_h[:, None])
    play = ps[:, None] * np.exp(-z_lay[None, :] / scale_h[:, None])
    if kdist is not None:
        plev = np.clip(plev, kdist.press_ref_min * 1.0001, kdist.press_ref_max * 0.9999)
        play = np.clip(play, kdist.press_ref_min * 1.0002, kdist.press_ref_max * 0.9998)
    vmr = np.zeros((ncol, nlay, ngas))
    q0 = 0.018 * np.exp(0.3 * rng.standard_normal(ncol))
    if climate == "sites":
        q0 = q0 * np.exp(0.065 * (sst - 300.0))  # saturation humidity falls ~6.5 % per kelvin
    h2o = q0[:, None] * np.exp(-z_lay[None, :] / 2.6) * np.exp(-(z_lay[None, :] / 11.0) ** 2)
    vmr[:, :, 0] = np.maximum(h2o, 3.0e-6)
    consts = {1: 348e-6, 3: 306e-9, 4: 0.12e-6, 5: 1650e-9, 6: 0.2095, 7: 0.7808}
    for ig, v in consts.items():
        if ig < ngas:
            vmr[:, :, ig] = v
    if ngas > 2:
        zz = z_lay[None, :]
        vmr[:, :, 2] = 8.0e-6 * (zz / 25.0) ** 3 * np.exp(3.0 * (1 - zz / 25.0)) + 2e-8
    col_dry = col_dry_from_plev(vmr[:, :, 0], plev)
    col_gas = np.empty((ncol, nlay, ngas + 1))
    col_gas[:, :, 0] = col_dry
    col_gas[:, :, 1:] = vmr * col_dry[:, :, None]
    tsfc = tlev[:, 0].copy()
    if top_at_1:
        play, plev, tlay, tlev = play[:, ::-1], plev[:, ::-1], tlay[:, ::-1], tlev[:, ::-1]
        vmr, col_dry, col_gas = vmr[:, ::-1], col_dry[:, ::-1], col_gas[:, ::-1]
    return Atmosphere(ncol, nlay, top_at_1, F(play), F(plev), F(tlay), F(tlev), F(tsfc), F(vmr),
                      F(col_dry), F(col_gas))


def make_cloud_optics(nbnd: int, seed: int = 77) -> dict:
    """Synthetic cloud-optics look-up tables with the shapes and ranges of the RRTMGP cloud files
    (rrtmgp/frontend/mo_cloud_optics_rrtmgp.F90:36-75; one ice roughness already selected): extinction
    [m2/g] falling with particle size, single-scattering albedo < 1, asymmetry 0.7 - 0.95."""
    rng = np.random.default_rng(seed)
    tb = {"liq_nsteps": 58, "liq_step_size": 0.3, "radliq_lwr": 2.5, "ice_nsteps": 94, "ice_step_size": 2.0,
          "diamice_lwr": 10.0}
    for ph, n, r0, dr in (("liq", 58, 2.5, 0.3), ("ice", 94, 10.0, 2.0)):
        r = r0 + dr * np.arange(n)
        ext = (1.5 / r)[:, None] * rng.uniform(0.8, 1.2, size=(1, nbnd)) * (1.0 + 0.05 * rng.standard_normal((n, nbnd)))
        tb[f"ext{ph}"] = F(np.abs(ext))
        tb[f"ssa{ph}"] = F(np.clip(rng.uniform(0.4, 0.999, size=(1, nbnd)) - 0.002 * r[:, None], 0.05, 0.9999))
        tb[f"asy{ph}"] = F(np.clip(rng.uniform(0.75, 0.9, size=(1, nbnd)) + 0.001 * r[:, None], 0.0, 0.97))
    tb["radliq_upr"] = tb["radliq_lwr"] + tb["liq_step_size"] * (tb["liq_nsteps"] - 1)
    tb["diamice_upr"] = tb["diamice_lwr"] + tb["ice_step_size"] * (tb["ice_nsteps"] - 1)
    return tb


def make_cloud_field(atm: "Atmosphere", tb: dict) -> dict:
    """The cloud field of examples/all-sky/rrtmgp_allsky.F90:645-662: clouds between 100 and 900 hPa in two
    columns out of three, liquid above 263 K, ice below 273 K, 10 g/m2 each, mid-range particle sizes."""
    ncol = atm.play.shape[0]
    icol1 = np.arange(1, ncol + 1)[:, None]
    mask = (atm.play > 100.0 * 100.0) & (atm.play < 900.0 * 100.0) & (icol1 % 3 != 0)
    lwp = np.where(mask & (atm.tlay > 263.0), 10.0, 0.0)
    iwp = np.where(mask & (atm.tlay < 273.0), 10.0, 0.0)
    rel_val = 0.5 * (tb["radliq_lwr"] + tb["radliq_upr"])
    dei_val = 0.5 * (tb["diamice_lwr"] + tb["diamice_upr"])
    return {"lwp": F(lwp), "iwp": F(iwp), "rel": F(np.where(lwp > 0, rel_val, 0.0)), "dei": F(np.where(iwp > 0, dei_val, 0.0))}
