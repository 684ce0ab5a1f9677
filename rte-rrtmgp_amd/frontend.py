"""Host-side mirror of the reference frontend's calls on the hot path, over the kernel C ABI.

This is NOT a re-implementation of the reference's Fortran classes; it is the thin sequence of
kernel calls (plus the few frontend "glue" loops that sit between them) that
``ty_gas_optics_rrtmgp%gas_optics`` , ``rte_lw`` and ``rte_sw`` perform, so that a device-resident
driver can chain the same ``bind(C)`` symbols the unchanged Fortran frontend would call:

  GasOptics.gas_optics_lw   <- gas_optics_int   rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:220-331
  GasOptics.gas_optics_sw   <- gas_optics_ext   :337-414
  GasOptics.compute_gas_taus<- compute_gas_taus :419-745   (interpolation, zero, tau_absorption
                                                            [, tau_rayleigh, combine])
  GasOptics.source          <- source           :840-928   (compute_Planck_source)
  rte_lw                    <- rte_lw           rte/frontend/mo_rte_lw.F90:79-473 (lw_solver_noscat /
                                                            lw_solver_2stream)
  rte_sw                    <- rte_sw_mu0_full  rte/frontend/mo_rte_sw.F90:103-394 (sw_solver_2stream /
                                                            sw_solver_noscat)

It works with any library exporting the ABI (``cabi.KernelLib``) and any array container through
an ``Arrays`` backend: ``NumpyArrays`` (host arrays; the HIP library stages them, the oracles use
them directly) or ``TorchArrays`` (device-resident tensors; the HIP library launches in place).
A Fortran array of shape (n1,n2,n3) (n1 fastest) is a torch tensor of shape (n3,n2,n1).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np

# Gauss-Jacobi-5 quadrature secants / weights for 1..4 angles, reference
# rte/frontend/mo_rte_lw.F90:146-160 (row n-1 holds the n-angle rule)
GAUSS_DS = [
    [1.0 / 0.6096748751],
    [1.0 / 0.2509907356, 1.0 / 0.7908473988],
    [1.0 / 0.1024922169, 1.0 / 0.4417960320, 1.0 / 0.8633751621],
    [1.0 / 0.0454586727, 1.0 / 0.2322334416, 1.0 / 0.5740198775, 1.0 / 0.9030775973],
]
GAUSS_WTS = [
    [1.0],
    [0.2300253764, 0.7699746236],
    [0.0437820218, 0.3875796738, 0.5686383044],
    [0.0092068785, 0.1285704278, 0.4323381850, 0.4298845087],
]


# --------------------------------------------------------------------------------------
# array backends
# --------------------------------------------------------------------------------------
class NumpyArrays:
    """Host arrays in Fortran order (what the unchanged Fortran frontend would pass)."""

    def __init__(self, precision: str = "dp"):
        self.ftype = np.float64 if precision == "dp" else np.float32
        self._dt = {"f": self.ftype, "i": np.int32, "b": np.bool_}

    def empty(self, shape, kind="f"):
        return np.empty(shape, dtype=self._dt[kind], order="F")

    def zeros(self, shape, kind="f"):
        return np.zeros(shape, dtype=self._dt[kind], order="F")

    def full(self, shape, value, kind="f"):
        return np.full(shape, value, dtype=self._dt[kind], order="F")

    def asarray(self, a):
        a = np.asarray(a)
        if a.dtype.kind == "f":
            a = a.astype(self.ftype, copy=False)  # no copy when already right: read-only shared (mmap) tables stay shared
        return np.asfortranarray(a)

    def to_numpy(self, a):
        return np.asarray(a)

    def sync(self):
        pass


class TorchArrays:
    """Device-resident tensors; Fortran shape (n1,..,nk) is stored as a C-contiguous tensor of
    shape (nk,..,n1), which is byte-identical to the column-major array."""

    def __init__(self, device="cuda:0", precision: str = "dp"):
        import torch

        self.torch = torch
        self.device = torch.device(device)
        self.ftype = np.float64 if precision == "dp" else np.float32
        self._dt = {"f": torch.float64 if precision == "dp" else torch.float32, "i": torch.int32,
                    "b": torch.bool}

    def empty(self, shape, kind="f"):
        return self.torch.empty(tuple(reversed(tuple(shape))), dtype=self._dt[kind], device=self.device)

    def zeros(self, shape, kind="f"):
        return self.torch.zeros(tuple(reversed(tuple(shape))), dtype=self._dt[kind], device=self.device)

    def full(self, shape, value, kind="f"):
        return self.torch.full(tuple(reversed(tuple(shape))), value, dtype=self._dt[kind],
                               device=self.device)

    def asarray(self, a):
        a = np.asarray(a)
        if a.dtype.kind == "f":
            a = a.astype(self.ftype)
        t = self.torch.from_numpy(np.ascontiguousarray(a.T))
        return t.to(self.device)

    def to_numpy(self, t):
        return t.detach().cpu().numpy().T

    def sync(self):
        if self.device.type == "cuda":
            self.torch.cuda.synchronize(self.device)


@dataclass
class InterpState:
    """Outputs of ``interpolation`` (the reference keeps them as locals of compute_gas_taus and
    hands them on to ``source``: mo_gas_optics_rrtmgp.F90:264-273,310-315)."""

    jtemp: object
    jpress: object
    tropo: object
    jeta: object
    col_mix: object
    fmajor: object
    fminor: object


# --------------------------------------------------------------------------------------
# gas optics
# --------------------------------------------------------------------------------------
class GasOptics:
    """The kernel-facing part of ``ty_gas_optics_rrtmgp`` for one k-distribution."""

    LUT_NAMES = ["flavor", "press_ref_log", "temp_ref", "vmr_ref", "gpoint_flavor", "band_lims_gpt",
                 "gpoint_bands", "kmajor", "kminor_lower", "kminor_upper", "minor_limits_gpt_lower",
                 "minor_limits_gpt_upper", "minor_scales_with_density_lower",
                 "minor_scales_with_density_upper", "scale_by_complement_lower",
                 "scale_by_complement_upper", "idx_minor_lower", "idx_minor_upper",
                 "idx_minor_scaling_lower", "idx_minor_scaling_upper", "kminor_start_lower",
                 "kminor_start_upper", "planck_frac", "totplnk", "krayl", "solar_source",
                 "optimal_angle_fit"]

    def __init__(self, lib, kdist, arrays):
        self.lib = lib
        self.kd = kdist
        self.xp = arrays
        self.t = {n: arrays.asarray(kdist.arrays[n]) for n in self.LUT_NAMES if n in kdist.arrays}
        # The HIP library caches small host-side plans keyed by the DEVICE addresses of the index tables (it
        # cannot cheaply look inside device memory).  Freshly uploaded tables may reuse the addresses of
        # released ones, so uploading tables invalidates those plans (no-op for other libraries).
        inval = getattr(lib, "raw", None)
        if inval is not None:
            try:
                fn = lib.raw("rte_hip_invalidate_plans")
            except Exception:  # oracle / reference libraries have no such entry
                fn = None
            if fn is not None:
                fn()
        self.ngas, self.nflav, self.neta = kdist.ngas, kdist.nflav, kdist.neta
        self.npres, self.ntemp = kdist.npres, kdist.ntemp
        self.nbnd, self.ngpt = kdist.nbnd, kdist.ngpt
        self.nminorlower = kdist.arrays["idx_minor_lower"].shape[0]
        self.nminorupper = kdist.arrays["idx_minor_upper"].shape[0]
        self.nminorklower = kdist.arrays["kminor_lower"].shape[2]
        self.nminorkupper = kdist.arrays["kminor_upper"].shape[2]
        self.is_lw = "totplnk" in kdist.arrays

    # -- interpolation (mo_gas_optics_rrtmgp.F90:615-635)
    def interpolation(self, ncol, nlay, play, tlay, col_gas, out: Optional[InterpState] = None):
        xp, kd, t = self.xp, self.kd, self.t
        if out is None:
            out = InterpState(
                jtemp=xp.empty((ncol, nlay), "i"), jpress=xp.empty((ncol, nlay), "i"),
                tropo=xp.empty((ncol, nlay), "b"), jeta=xp.empty((2, ncol, nlay, self.nflav), "i"),
                col_mix=xp.empty((2, ncol, nlay, self.nflav)),
                fmajor=xp.empty((2, 2, 2, ncol, nlay, self.nflav)),
                fminor=xp.empty((2, 2, ncol, nlay, self.nflav)))
        self.lib.rrtmgp_interpolation(
            ncol, nlay, self.ngas, self.nflav, self.neta, self.npres, self.ntemp, t["flavor"],
            t["press_ref_log"], t["temp_ref"], kd.press_ref_log_delta, kd.temp_ref_min,
            kd.temp_ref_delta, kd.press_ref_trop_log, t["vmr_ref"], play, tlay, col_gas, out.jtemp,
            out.fmajor, out.fminor, out.col_mix, out.tropo, out.jeta, out.jpress)
        return out

    # -- compute_tau_absorption (mo_gas_optics_rrtmgp.F90:638-665 / 680-707); tau is ACCUMULATED
    def compute_tau_absorption(self, ncol, nlay, st: InterpState, play, tlay, col_gas, tau, tau_bybnd=None):
        t = self.t
        if tau_bybnd is not None:
            # library extension: the band-wise increment by `tau_bybnd` (ncol, nlay, nbnd) in the same pass
            from .hiplib import ext_call

            ext_call(self.lib, "rte_hip_compute_tau_absorption_inc_bybnd", ["i"] * 14 + ["a"] * 29,
                     ncol, nlay, self.nbnd, self.ngpt, self.ngas, self.nflav, self.neta, self.npres,
                     self.ntemp, self.nminorlower, self.nminorklower, self.nminorupper, self.nminorkupper,
                     self.kd.idx_h2o, t["gpoint_flavor"], t["band_lims_gpt"], t["kmajor"], t["kminor_lower"],
                     t["kminor_upper"], t["minor_limits_gpt_lower"], t["minor_limits_gpt_upper"],
                     t["minor_scales_with_density_lower"], t["minor_scales_with_density_upper"],
                     t["scale_by_complement_lower"], t["scale_by_complement_upper"], t["idx_minor_lower"],
                     t["idx_minor_upper"], t["idx_minor_scaling_lower"], t["idx_minor_scaling_upper"],
                     t["kminor_start_lower"], t["kminor_start_upper"], st.tropo, st.col_mix, st.fmajor,
                     st.fminor, play, tlay, col_gas, st.jeta, st.jtemp, st.jpress, tau, tau_bybnd)
            return
        self.lib.rrtmgp_compute_tau_absorption(
            ncol, nlay, self.nbnd, self.ngpt, self.ngas, self.nflav, self.neta, self.npres,
            self.ntemp, self.nminorlower, self.nminorklower, self.nminorupper, self.nminorkupper,
            self.kd.idx_h2o, t["gpoint_flavor"], t["band_lims_gpt"], t["kmajor"], t["kminor_lower"],
            t["kminor_upper"], t["minor_limits_gpt_lower"], t["minor_limits_gpt_upper"],
            t["minor_scales_with_density_lower"], t["minor_scales_with_density_upper"],
            t["scale_by_complement_lower"], t["scale_by_complement_upper"], t["idx_minor_lower"],
            t["idx_minor_upper"], t["idx_minor_scaling_lower"], t["idx_minor_scaling_upper"],
            t["kminor_start_lower"], t["kminor_start_upper"], st.tropo, st.col_mix, st.fmajor,
            st.fminor, play, tlay, col_gas, st.jeta, st.jtemp, st.jpress, tau)

    def compute_tau_rayleigh(self, ncol, nlay, st: InterpState, col_dry, col_gas, tau_rayleigh):
        t = self.t
        self.lib.rrtmgp_compute_tau_rayleigh(
            ncol, nlay, self.nbnd, self.ngpt, self.ngas, self.nflav, self.neta, self.npres,
            self.ntemp, t["gpoint_flavor"], t["band_lims_gpt"], t["krayl"], self.kd.idx_h2o, col_dry,
            col_gas, st.fminor, st.jeta, st.tropo, st.jtemp, tau_rayleigh)

    # -- source (mo_gas_optics_rrtmgp.F90:840-928)
    def source(self, ncol, nlay, st: InterpState, tlay, tlev, tsfc, top_at_1, sfc_src, lay_src,
               lev_src, sfc_src_jac):
        t, kd = self.t, self.kd
        sfc_lay = nlay if top_at_1 else 1  # :920
        self.lib.rrtmgp_compute_Planck_source(
            ncol, nlay, self.nbnd, self.ngpt, self.nflav, self.neta, self.npres, self.ntemp,
            int(kd.nPlanckTemp), tlay, tlev, tsfc, sfc_lay, st.fmajor, st.jeta, st.tropo, st.jtemp,
            st.jpress, t["gpoint_bands"], t["band_lims_gpt"], t["planck_frac"], kd.temp_ref_min,
            kd.totplnk_delta, t["totplnk"], t["gpoint_flavor"], sfc_src, lay_src, lev_src, sfc_src_jac)

    def source_factored(self, ncol, nlay, st: InterpState, tlay, tlev, tsfc, top_at_1, sfc_src, pfrac, planck_lay, planck_lev,
                        sfc_src_jac):
        """``source`` with FACTORED output (library extension ``rte_hip_compute_Planck_source_factored``): the Planck fraction
        ``pfrac`` (ncol, nlay, ngpt) and the band's Planck function at the layer / level temperatures, ``planck_lay`` (ncol, nlay,
        nbnd) / ``planck_lev`` (ncol, nlay+1, nbnd), instead of their products lay_source / lev_source (:674, :695-705).
        ``rte_lw_factored`` consumes them; ``expand_factored_sources`` turns them into the reference's arrays."""
        from .hiplib import ext_call

        t, kd = self.t, self.kd
        sfc_lay = nlay if top_at_1 else 1  # :920
        rc = ext_call(self.lib, "rte_hip_compute_Planck_source_factored", "iiiiiiiiiaaaiaaaaaaaddaaaaaaa", ncol, nlay, self.nbnd,
                      self.ngpt, self.nflav, self.neta, self.npres, self.ntemp, int(kd.nPlanckTemp), tlay, tlev, tsfc, sfc_lay,
                      st.fmajor, st.jeta, st.tropo, st.jtemp, st.jpress, t["band_lims_gpt"], t["planck_frac"],
                      float(kd.temp_ref_min), float(kd.totplnk_delta), t["totplnk"], t["gpoint_flavor"], sfc_src, pfrac,
                      planck_lay, planck_lev, sfc_src_jac)
        assert rc == 0, rc

    def expand_factored_sources(self, ncol, nlay, pfrac, planck_lay, planck_lev, lay_src, lev_src):
        from .hiplib import ext_call

        rc = ext_call(self.lib, "rte_hip_expand_factored_sources", "iiiiaaaaaa", ncol, nlay, self.nbnd, self.ngpt,
                      self.t["band_lims_gpt"], pfrac, planck_lay, planck_lev, lay_src, lev_src)
        assert rc == 0, rc

    # -- gas_optics_int: LW, returns 1scl optical props + sources
    def gas_optics_lw(self, ncol, nlay, play, plev, tlay, tsfc, col_gas, tlev, top_at_1,
                      buffers: Optional[Dict[str, object]] = None, tau_bybnd=None, factored_sources: bool = False):
        xp = self.xp
        b = buffers if buffers is not None else {}

        def buf(name, shape, kind="f"):
            if name not in b:
                b[name] = xp.empty(shape, kind)
            return b[name]

        st = self.interpolation(ncol, nlay, play, tlay, col_gas, b.get("interp"))
        b["interp"] = st
        tau = buf("tau", (ncol, nlay, self.ngpt))
        self.lib.zero_array_3D(ncol, nlay, self.ngpt, tau)  # :679
        self.compute_tau_absorption(ncol, nlay, st, play, tlay, col_gas, tau, tau_bybnd=tau_bybnd)
        sfc_src = buf("sfc_src", (ncol, self.ngpt))
        sfc_src_jac = buf("sfc_src_jac", (ncol, self.ngpt))
        if factored_sources:  # library extension: the sources stay factored (see source_factored / rte_lw_factored)
            self.source_factored(ncol, nlay, st, tlay, tlev, tsfc, top_at_1, sfc_src, buf("pfrac", (ncol, nlay, self.ngpt)),
                                 buf("planck_lay", (ncol, nlay, self.nbnd)), buf("planck_lev", (ncol, nlay + 1, self.nbnd)),
                                 sfc_src_jac)
            return b
        lay_src = buf("lay_src", (ncol, nlay, self.ngpt))
        lev_src = buf("lev_src", (ncol, nlay + 1, self.ngpt))
        self.source(ncol, nlay, st, tlay, tlev, tsfc, top_at_1, sfc_src, lay_src, lev_src, sfc_src_jac)
        return b

    # -- gas_optics_ext: SW, returns 2str optical props + toa source
    def gas_optics_sw(self, ncol, nlay, play, plev, tlay, col_gas, col_dry,
                      buffers: Optional[Dict[str, object]] = None, glue=None, fuse_rayleigh: bool = False, clouds_bybnd=None,
                      implicit_g: bool = False):
        """``implicit_g`` (with ``fuse_rayleigh="all"``, no clouds): combine_abs_and_rayleigh's ``g = 0`` (:1983-2002) is not stored --
        ``b["g"]`` is None and ``rte_sw`` / the library's ``rte_sw_solver_2stream`` take a missing ``g`` as zero (library extension)."""
        xp = self.xp
        b = buffers if buffers is not None else {}

        def buf(name, shape, kind="f"):
            if name not in b:
                b[name] = xp.empty(shape, kind)
            return b[name]

        st = self.interpolation(ncol, nlay, play, tlay, col_gas, b.get("interp"))
        b["interp"] = st
        gl = glue or default_glue(self.lib, xp)
        fused = fuse_rayleigh and hasattr(gl, "tau_rayleigh_combine_2str")
        assert clouds_bybnd is None or fused, "the by-band cloud increment is part of the fused kernel"
        if fused and fuse_rayleigh == "all" and hasattr(gl, "gas_optics_sw_2str"):
            # absorption, Rayleigh, combine (and the by-band cloud increment) in ONE pass: tau_abs never goes to memory
            tau, ssa = (buf(n, (ncol, nlay, self.ngpt)) for n in ("tau", "ssa"))
            if implicit_g and clouds_bybnd is None:
                b["g"] = g = None
            else:
                g = buf("g", (ncol, nlay, self.ngpt))
            gl.gas_optics_sw_2str(self, ncol, nlay, st, play, tlay, col_gas, col_dry, tau, ssa, g, clouds_bybnd)
            toa = buf("toa_src", (ncol, self.ngpt))
            gl.broadcast_gpt(ncol, self.ngpt, self.t["solar_source"], toa)
            return b
        # fused: the absorption optical depth is computed straight into `tau`, which the fused kernel updates in place
        tau_abs = buf("tau" if fused else "tau_abs", (ncol, nlay, self.ngpt))
        self.lib.zero_array_3D(ncol, nlay, self.ngpt, tau_abs)  # :637
        self.compute_tau_absorption(ncol, nlay, st, play, tlay, col_gas, tau_abs)
        if fused:
            # compute_tau_rayleigh + combine_abs_and_rayleigh (:666-678) in one pass: tau_rayleigh never goes to memory
            ssa, g = (buf(n, (ncol, nlay, self.ngpt)) for n in ("ssa", "g"))
            t = self.t
            gl.tau_rayleigh_combine_2str(ncol, nlay, self.nbnd, self.ngpt, self.ngas, self.nflav, self.neta, self.ntemp,
                                         t["gpoint_flavor"], t["band_lims_gpt"], t["krayl"], self.kd.idx_h2o, col_dry,
                                         col_gas, st.fminor, st.jeta, st.tropo, st.jtemp, tau_abs, tau_abs, ssa, g,
                                         clouds_bybnd)
        else:
            tau_ray = buf("tau_rayleigh", (ncol, nlay, self.ngpt))
            self.compute_tau_rayleigh(ncol, nlay, st, col_dry, col_gas, tau_ray)
            tau, ssa, g = (buf(n, (ncol, nlay, self.ngpt)) for n in ("tau", "ssa", "g"))
            gl.combine_abs_and_rayleigh_2str(ncol, nlay, self.ngpt, tau_abs, tau_ray, tau, ssa, g)
        toa = buf("toa_src", (ncol, self.ngpt))
        gl.broadcast_gpt(ncol, self.ngpt, self.t["solar_source"], toa)
        return b


# --------------------------------------------------------------------------------------
# frontend glue loops (not behind the reference's C API)
# --------------------------------------------------------------------------------------
class NumpyGlue:
    """Host glue for host arrays (numpy restatements of the frontend's elementwise loops)."""

    # combine_abs_and_rayleigh, 2-stream branch: mo_gas_optics_rrtmgp.F90:1983-2002
    def combine_abs_and_rayleigh_2str(self, ncol, nlay, ngpt, tau_abs, tau_ray, tau, ssa, g):
        t = tau_abs + tau_ray
        tiny2 = 2.0 * np.finfo(t.dtype).tiny
        with np.errstate(divide="ignore", invalid="ignore"):
            ssa[...] = np.where(t > tiny2, tau_ray / t, 0.0)
        tau[...] = t
        g[...] = 0.0

    # toa_src(icol,igpt) = solar_source(igpt): mo_gas_optics_rrtmgp.F90:405-411
    def broadcast_gpt(self, ncol, ngpt, per_gpt, out):
        out[...] = np.asarray(per_gpt)[None, :]

    # cloud_optics: masks (mo_cloud_optics_rrtmgp.F90:334-341) and liquid + ice combination (:392-425)
    def cloud_masks(self, ncol, nlay, clwp, ciwp, liqmsk, icemsk):
        liqmsk[...] = clwp > 0
        icemsk[...] = ciwp > 0

    def cloud_combine(self, ncol, nlay, nspec, twostr, liq, ice, tau, ssa, g):
        (lt, lts, ltsg), (it, its, itsg) = liq, ice
        if not twostr:
            tau[...] = (lt - lts) + (it - its)
            return
        t, ts = lt + it, lts + its
        eps = np.finfo(t.dtype).eps
        g[...] = (ltsg + itsg) / np.maximum(eps, ts)
        ssa[...] = ts / np.maximum(eps, t)
        tau[...] = t


class HipGlue:
    """Device glue: extension entry points of the HIP library (names ``rte_hip_*``)."""

    def __init__(self, lib):
        self.lib = lib

    def combine_abs_and_rayleigh_2str(self, ncol, nlay, ngpt, tau_abs, tau_ray, tau, ssa, g):
        from .hiplib import ext_call

        ext_call(self.lib, "rte_hip_combine_abs_and_rayleigh_2str", ["i", "i", "i", "a", "a", "a", "a", "a"],
                 ncol, nlay, ngpt, tau_abs, tau_ray, tau, ssa, g)

    def broadcast_gpt(self, ncol, ngpt, per_gpt, out):
        from .hiplib import ext_call

        ext_call(self.lib, "rte_hip_broadcast_gpt", ["i", "i", "a", "a"], ncol, ngpt, per_gpt, out)

    # compute_tau_rayleigh fused with the 2-stream combine (csrc/tau_absorption.hip: rte_hip_tau_rayleigh_combine_2str)
    def tau_rayleigh_combine_2str(self, ncol, nlay, nbnd, ngpt, ngas, nflav, neta, ntemp, gpoint_flavor, band_lims_gpt, krayl,
                                  idx_h2o, col_dry, col_gas, fminor, jeta, tropo, jtemp, tau_abs, tau, ssa, g, clouds_bybnd=None):
        from .hiplib import ext_call

        ct, cs, cg = clouds_bybnd if clouds_bybnd is not None else (None, None, None)
        ext_call(self.lib, "rte_hip_tau_rayleigh_combine_2str", ["i"] * 8 + ["a", "a", "a", "i"] + ["a"] * 13,
                 ncol, nlay, nbnd, ngpt, ngas, nflav, neta, ntemp, gpoint_flavor, band_lims_gpt, krayl, idx_h2o,
                 col_dry, col_gas, fminor, jeta, tropo, jtemp, tau_abs, tau, ssa, g, ct, cs, cg)

    # the whole SW gas optics in one pass (csrc/tau_absorption.hip: rte_hip_gas_optics_sw_2str)
    def gas_optics_sw_2str(self, go, ncol, nlay, st, play, tlay, col_gas, col_dry, tau, ssa, g, clouds_bybnd=None):
        from .hiplib import ext_call

        t = go.t
        ct, cs, cg = clouds_bybnd if clouds_bybnd is not None else (None, None, None)
        ext_call(self.lib, "rte_hip_gas_optics_sw_2str", ["i"] * 14 + ["a"] * 36,
                 ncol, nlay, go.nbnd, go.ngpt, go.ngas, go.nflav, go.neta, go.npres, go.ntemp, go.nminorlower,
                 go.nminorklower, go.nminorupper, go.nminorkupper, go.kd.idx_h2o, t["gpoint_flavor"], t["band_lims_gpt"],
                 t["kmajor"], t["kminor_lower"], t["kminor_upper"], t["minor_limits_gpt_lower"], t["minor_limits_gpt_upper"],
                 t["minor_scales_with_density_lower"], t["minor_scales_with_density_upper"], t["scale_by_complement_lower"],
                 t["scale_by_complement_upper"], t["idx_minor_lower"], t["idx_minor_upper"], t["idx_minor_scaling_lower"],
                 t["idx_minor_scaling_upper"], t["kminor_start_lower"], t["kminor_start_upper"], st.tropo, st.col_mix,
                 st.fmajor, st.fminor, play, tlay, col_gas, st.jeta, st.jtemp, st.jpress, t["krayl"], col_dry, tau, ssa, g,
                 ct, cs, cg)

    # masks + both table look-ups + liquid/ice combination (+ delta scaling) in one pass (csrc/optical_props.hip)
    def cloud_optics_fused(self, ncol, nlay, nbnd, twostr, delta_scale, clwp, ciwp, reliq, deice, tb, t, tau, ssa, g):
        from .hiplib import ext_call

        ext_call(self.lib, "rte_hip_cloud_optics_fused", ["i"] * 5 + ["a"] * 4 + ["i", "d", "d", "a", "a", "a"] * 2 + ["a"] * 3,
                 ncol, nlay, nbnd, 1 if twostr else 0, 1 if delta_scale else 0, clwp, ciwp, reliq, deice,
                 int(tb["liq_nsteps"]), float(tb["liq_step_size"]), float(tb["radliq_lwr"]), t["extliq"], t["ssaliq"], t["asyliq"],
                 int(tb["ice_nsteps"]), float(tb["ice_step_size"]), float(tb["diamice_lwr"]), t["extice"], t["ssaice"], t["asyice"],
                 tau, ssa if twostr else tau, g if twostr else tau)

    def cloud_masks(self, ncol, nlay, clwp, ciwp, liqmsk, icemsk):
        from .hiplib import ext_call

        ext_call(self.lib, "rte_hip_cloud_masks", ["i", "i", "a", "a", "a", "a"], ncol, nlay, clwp, ciwp, liqmsk, icemsk)

    def cloud_combine(self, ncol, nlay, nspec, twostr, liq, ice, tau, ssa, g):
        from .hiplib import ext_call

        ext_call(self.lib, "rte_hip_cloud_combine", ["i", "i", "i", "i"] + ["a"] * 9, ncol, nlay, nspec, 1 if twostr else 0,
                 *liq, *ice, tau, ssa if twostr else tau, g if twostr else tau)


def default_glue(lib, arrays):
    return NumpyGlue() if isinstance(arrays, NumpyArrays) else HipGlue(lib)


# --------------------------------------------------------------------------------------
# Cloud optics (rrtmgp/frontend/mo_cloud_optics_rrtmgp.F90) and the all-sky assembly
# --------------------------------------------------------------------------------------
class CloudOptics:
    """The kernel-facing part of ``ty_cloud_optics_rrtmgp%cloud_optics`` (look-up-table branch, :276-430):
    cloud optical properties per band from liquid / ice water paths and particle sizes."""

    def __init__(self, lib, tables, arrays):
        self.lib, self.xp, self.tb = lib, arrays, tables
        self.nbnd = tables["extliq"].shape[1]
        self.t = {k: arrays.asarray(v) for k, v in tables.items() if hasattr(v, "shape")}

    def cloud_optics(self, ncol, nlay, clwp, ciwp, reliq, deice, twostr, buffers=None, glue=None, fused=False,
                     delta_scale=False):
        """``fused`` (device containers): one extension kernel instead of masks + 2 look-ups + combination, optionally with
        the delta scaling the all-sky driver applies next (``delta_scale``; without ``fused`` the caller does it)."""
        xp, tb, nb = self.xp, self.tb, self.nbnd
        b = buffers if buffers is not None else {}

        def buf(name, shape, kind="f"):
            if name not in b:
                b[name] = xp.empty(shape, kind)
            return b[name]

        gl = glue or default_glue(self.lib, xp)
        if fused and hasattr(gl, "cloud_optics_fused"):
            tau = buf("cld_tau", (ncol, nlay, nb))
            ssa = buf("cld_ssa", (ncol, nlay, nb)) if twostr else None
            g = buf("cld_g", (ncol, nlay, nb)) if twostr else None
            gl.cloud_optics_fused(ncol, nlay, nb, twostr, delta_scale, clwp, ciwp, reliq, deice, tb, self.t, tau, ssa, g)
            b["delta_scaled"] = bool(delta_scale)
            return b
        b["delta_scaled"] = False
        liqmsk, icemsk = buf("liqmsk", (ncol, nlay), "b"), buf("icemsk", (ncol, nlay), "b")
        gl.cloud_masks(ncol, nlay, clwp, ciwp, liqmsk, icemsk)
        liq = [buf(n, (ncol, nlay, nb)) for n in ("ltau", "ltaussa", "ltaussag")]
        ice = [buf(n, (ncol, nlay, nb)) for n in ("itau", "itaussa", "itaussag")]
        t = self.t
        self.lib.rrtmgp_compute_cld_from_table(ncol, nlay, nb, liqmsk, clwp, reliq, tb["liq_nsteps"], tb["liq_step_size"],
                                               tb["radliq_lwr"], t["extliq"], t["ssaliq"], t["asyliq"], *liq)   # :372-375
        self.lib.rrtmgp_compute_cld_from_table(ncol, nlay, nb, icemsk, ciwp, deice, tb["ice_nsteps"], tb["ice_step_size"],
                                               tb["diamice_lwr"], t["extice"], t["ssaice"], t["asyice"], *ice)   # :379-384
        tau = buf("cld_tau", (ncol, nlay, nb))
        ssa = buf("cld_ssa", (ncol, nlay, nb)) if twostr else None
        g = buf("cld_g", (ncol, nlay, nb)) if twostr else None
        gl.cloud_combine(ncol, nlay, nb, twostr, liq, ice, tau, ssa, g)                                           # :392-425
        return b


def allsky_lw(lib, xp, go: "GasOptics", co: CloudOptics, ncol, nlay, atm, clouds, sfc_emis_gpt, gb=None, cb=None, rb=None,
              fuse: bool = True, factored_sources: bool = False):
    """LW half of examples/all-sky/rrtmgp_allsky.F90:362-380: clouds as absorbers (1scl) added to the gas optical
    depth band by band, then rte_lw without scattering.  ``fuse`` (device containers only): the library's fused
    extension kernels -- cloud optics in one pass, and the band-wise increment applied inside compute_tau_absorption;
    same values, each 3-D array written once."""
    fuse = fuse and not isinstance(xp, NumpyArrays)
    factored_sources = factored_sources and fuse  # (library extensions: the sources stay factored, see rte_lw_factored)
    if fuse:
        cb = co.cloud_optics(ncol, nlay, clouds["lwp"], clouds["iwp"], clouds["rel"], clouds["dei"], False, buffers=cb, fused=True)
        gb = go.gas_optics_lw(ncol, nlay, atm["play"], atm["plev"], atm["tlay"], atm["tsfc"], atm["col_gas"], atm["tlev"],
                              atm["top_at_1"], buffers=gb, tau_bybnd=cb["cld_tau"], factored_sources=factored_sources)  # :374 fused in
    else:
        gb = go.gas_optics_lw(ncol, nlay, atm["play"], atm["plev"], atm["tlay"], atm["tsfc"], atm["col_gas"], atm["tlev"],
                              atm["top_at_1"], buffers=gb)
        cb = co.cloud_optics(ncol, nlay, clouds["lwp"], clouds["iwp"], clouds["rel"], clouds["dei"], False, buffers=cb)
        lib.rte_inc_1scalar_by_1scalar_bybnd(ncol, nlay, go.ngpt, gb["tau"], cb["cld_tau"], go.nbnd, go.t["band_lims_gpt"])  # :374
    if factored_sources:
        rb = rte_lw_factored(lib, xp, ncol, nlay, go.ngpt, go.nbnd, go.t["band_lims_gpt"], atm["top_at_1"], gb["tau"], gb["pfrac"],
                             gb["planck_lay"], gb["planck_lev"], sfc_emis_gpt, gb["sfc_src"], buffers=rb)
        return gb, cb, rb
    rb = rte_lw(lib, xp, ncol, nlay, go.ngpt, atm["top_at_1"], gb["tau"], gb["lay_src"], gb["lev_src"], sfc_emis_gpt,
                gb["sfc_src"], buffers=rb)
    return gb, cb, rb


def allsky_sw(lib, xp, go: "GasOptics", co: CloudOptics, ncol, nlay, atm, clouds, mu0, sfc_alb_gpt, gb=None, cb=None, rb=None,
              fuse=True):
    """SW half (:382-404): two-stream clouds, delta-scaled, added to the gas optical properties band by band.
    ``fuse`` (device containers only): cloud optics + delta scaling in one pass, and compute_tau_rayleigh +
    combine_abs_and_rayleigh + the band-wise increment in one pass over the gas arrays; ``fuse="all"``: that pass is
    compute_tau_absorption's as well (the absorption optical depth never goes to memory)."""
    if isinstance(xp, NumpyArrays):
        fuse = False
    if fuse:
        cb = co.cloud_optics(ncol, nlay, clouds["lwp"], clouds["iwp"], clouds["rel"], clouds["dei"], True, buffers=cb, fused=True,
                             delta_scale=True)                                                                      # :394 fused in
        gb = go.gas_optics_sw(ncol, nlay, atm["play"], atm["plev"], atm["tlay"], atm["col_gas"], atm["col_dry"], buffers=gb,
                              fuse_rayleigh=("all" if fuse == "all" else True),
                              clouds_bybnd=(cb["cld_tau"], cb["cld_ssa"], cb["cld_g"]))                             # :395 fused in
    else:
        gb = go.gas_optics_sw(ncol, nlay, atm["play"], atm["plev"], atm["tlay"], atm["col_gas"], atm["col_dry"], buffers=gb)
        cb = co.cloud_optics(ncol, nlay, clouds["lwp"], clouds["iwp"], clouds["rel"], clouds["dei"], True, buffers=cb)
        lib.rte_delta_scale_2str_k(ncol, nlay, co.nbnd, cb["cld_tau"], cb["cld_ssa"], cb["cld_g"])                   # :394
        lib.rte_inc_2stream_by_2stream_bybnd(ncol, nlay, go.ngpt, gb["tau"], gb["ssa"], gb["g"], cb["cld_tau"], cb["cld_ssa"],
                                             cb["cld_g"], go.nbnd, go.t["band_lims_gpt"])                           # :395
    rb = rte_sw(lib, xp, ncol, nlay, go.ngpt, atm["top_at_1"], gb["tau"], gb["ssa"], gb["g"], mu0, gb["toa_src"],
                sfc_alb_gpt, sfc_alb_gpt, buffers=rb)
    return gb, cb, rb


# --------------------------------------------------------------------------------------
# RTE solvers
# --------------------------------------------------------------------------------------
def rte_lw(lib, xp, ncol, nlay, ngpt, top_at_1, tau, lay_src, lev_src, sfc_emis_gpt, sfc_src,
           n_gauss_angles: int = 1, inc_flux=None, sfc_src_jac=None, do_jacobians=False,
           lw_Ds=None, ssa=None, g=None, use_2stream=False, do_broadband=True,
           buffers: Optional[Dict[str, object]] = None):
    """Mirror of ``rte_lw`` (rte/frontend/mo_rte_lw.F90:79-473) for 1scl / 2str optical props.

    Returns the dict of output buffers: ``flux_up``/``flux_dn`` are broadband (ncol,nlay+1) when
    ``do_broadband`` else spectral (ncol,nlay+1,ngpt); ``flux_up_jac`` when requested."""
    b = buffers if buffers is not None else {}

    def buf(name, shape, kind="f"):
        if name not in b:
            b[name] = xp.empty(shape, kind)
        return b[name]

    if inc_flux is None:
        if "inc_flux_zero" not in b:
            b["inc_flux_zero"] = xp.zeros((ncol, ngpt))
        inc_flux = b["inc_flux_zero"]  # :297-305
    if use_2stream:
        # lw_solver_2stream path :388-409 (spectral fluxes, then reduce)
        gfu, gfd = buf("gpt_flux_up", (ncol, nlay + 1, ngpt)), buf("gpt_flux_dn", (ncol, nlay + 1, ngpt))
        lib.rte_lw_solver_2stream(ncol, nlay, ngpt, top_at_1, tau, ssa, g, lay_src, lev_src,
                                  sfc_emis_gpt, sfc_src, inc_flux, gfu, gfd)
        if do_broadband:
            fu, fd = buf("flux_up", (ncol, nlay + 1)), buf("flux_dn", (ncol, nlay + 1))
            lib.rte_sum_broadband(ncol, nlay + 1, ngpt, gfu, fu)
            lib.rte_sum_broadband(ncol, nlay + 1, ngpt, gfd, fd)
        return b
    nmus = n_gauss_angles
    key = ("secants", nmus, id(lw_Ds))
    if key not in b:
        if lw_Ds is not None:  # :346-356 user-provided secants (ncol,ngpt)
            b[key] = lw_Ds
        else:  # :357-365
            sec = np.empty((ncol, ngpt, nmus), order="F")
            for imu in range(nmus):
                sec[:, :, imu] = GAUSS_DS[nmus - 1][imu]
            b[key] = xp.asarray(sec)
        # the quadrature weights stay on the HOST, as in the reference (gauss_wts is a host table, mo_rte_lw.F90:367):
        # the library reads them on the host, and a device copy would cost a device-to-host copy + sync per call
        b[("weights", nmus)] = np.array(GAUSS_WTS[nmus - 1] if lw_Ds is None else [1.0], dtype=xp.ftype)
    secants, weights = b[key], b[("weights", nmus)]
    do_rescaling = ssa is not None and g is not None
    decoy2 = buf("decoy2D", (ncol, nlay + 1))
    if do_broadband:
        fu, fd = buf("flux_up", (ncol, nlay + 1)), buf("flux_dn", (ncol, nlay + 1))
        gfu = gfd = buf("decoy3D", (1,))  # never written in broadband mode
    else:
        gfu, gfd = buf("gpt_flux_up", (ncol, nlay + 1, ngpt)), buf("gpt_flux_dn", (ncol, nlay + 1, ngpt))
        fu = fd = decoy2
    jac = buf("flux_up_jac", (ncol, nlay + 1)) if do_jacobians else decoy2
    lib.rte_lw_solver_noscat(
        ncol, nlay, ngpt, top_at_1, nmus if lw_Ds is None else 1, secants, weights, tau, lay_src,
        lev_src, sfc_emis_gpt, sfc_src, inc_flux, gfu, gfd, do_broadband, fu, fd, do_jacobians,
        sfc_src_jac if do_jacobians else sfc_src, jac, do_rescaling, ssa if do_rescaling else tau,
        g if do_rescaling else tau)
    return b


def rte_lw_factored(lib, xp, ncol, nlay, ngpt, nbnd, band_lims_gpt, top_at_1, tau, pfrac, planck_lay, planck_lev, sfc_emis_gpt,
                    sfc_src, n_gauss_angles: int = 1, inc_flux=None, sfc_src_jac=None, do_jacobians=False,
                    buffers: Optional[Dict[str, object]] = None):
    """``rte_lw`` (broadband fluxes, no scattering) on FACTORED sources -- what ``GasOptics.gas_optics_lw(factored_sources=True)``
    leaves: the library extension ``rte_hip_lw_solver_noscat_factored`` forms lay_source / lev_source per g-point inside the
    solver with the operations of compute_Planck_source (bit-identical fluxes), so the two (ncol, nlay[+1], ngpt) source arrays are
    neither written nor read.  Shapes the extension does not take (-2) are expanded and go through ``rte_lw``."""
    from . import hiplib

    b = buffers if buffers is not None else {}

    def buf(name, shape, kind="f"):
        if name not in b:
            b[name] = xp.empty(shape, kind)
        return b[name]

    if inc_flux is None:
        if "inc_flux_zero" not in b:
            b["inc_flux_zero"] = xp.zeros((ncol, ngpt))
        inc_flux = b["inc_flux_zero"]
    nmus = n_gauss_angles
    key = ("secants", nmus, id(None))
    if key not in b:
        sec = np.empty((ncol, ngpt, nmus), order="F")
        for imu in range(nmus):
            sec[:, :, imu] = GAUSS_DS[nmus - 1][imu]
        b[key] = xp.asarray(sec)
        b[("weights", nmus)] = np.array(GAUSS_WTS[nmus - 1], dtype=xp.ftype)
    fu, fd = buf("flux_up", (ncol, nlay + 1)), buf("flux_dn", (ncol, nlay + 1))
    jac = buf("flux_up_jac", (ncol, nlay + 1)) if do_jacobians else buf("decoy2D", (ncol, nlay + 1))
    rc = hiplib.ext_call(lib, "rte_hip_lw_solver_noscat_factored", "iiiiiiaaaaaaaaaaaaiaa", ncol, nlay, ngpt, nbnd, int(top_at_1), nmus,
                         b[key], b[("weights", nmus)], band_lims_gpt, tau, pfrac, planck_lay, planck_lev, sfc_emis_gpt, sfc_src,
                         inc_flux, fu, fd, int(bool(do_jacobians)), sfc_src_jac if do_jacobians else sfc_src, jac)
    assert rc in (0, -2), rc
    if rc == -2:
        lay_src, lev_src = buf("lay_src", (ncol, nlay, ngpt)), buf("lev_src", (ncol, nlay + 1, ngpt))
        rc = hiplib.ext_call(lib, "rte_hip_expand_factored_sources", "iiiiaaaaaa", ncol, nlay, nbnd, ngpt, band_lims_gpt, pfrac,
                             planck_lay, planck_lev, lay_src, lev_src)
        assert rc == 0, rc
        return rte_lw(lib, xp, ncol, nlay, ngpt, top_at_1, tau, lay_src, lev_src, sfc_emis_gpt, sfc_src, n_gauss_angles=nmus,
                      inc_flux=inc_flux, sfc_src_jac=sfc_src_jac, do_jacobians=do_jacobians, buffers=b)
    return b


def rte_sw(lib, xp, ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, inc_flux_dir, sfc_alb_dir_gpt,
           sfc_alb_dif_gpt, inc_flux_dif=None, do_broadband=True, noscat=False,
           buffers: Optional[Dict[str, object]] = None):
    """Mirror of ``rte_sw_mu0_full`` (rte/frontend/mo_rte_sw.F90:103-394); ``mu0`` is (ncol,nlay)."""
    b = buffers if buffers is not None else {}

    def buf(name, shape, kind="f"):
        if name not in b:
            b[name] = xp.empty(shape, kind)
        return b[name]

    if noscat:  # 1scl branch :283-302: direct beam only
        fdir = buf("gpt_flux_dir", (ncol, nlay + 1, ngpt))
        lib.rte_sw_solver_noscat(ncol, nlay, ngpt, top_at_1, tau, mu0, inc_flux_dir, fdir)
        if do_broadband:
            lib.rte_sum_broadband(ncol, nlay + 1, ngpt, fdir, buf("flux_dir", (ncol, nlay + 1)))
        return b
    has_dif_bc = inc_flux_dif is not None
    if not has_dif_bc:
        if "inc_flux_zero" not in b:
            b["inc_flux_zero"] = xp.zeros((ncol, ngpt))
        inc_flux_dif = b["inc_flux_zero"]
    if do_broadband:
        d3 = buf("decoy3D", (1,))  # one aliased decoy, never written (mo_rte_sw.F90:204-207)
        gfu = gfd = gfdir = d3
        fu, fd, fdir = (buf(n, (ncol, nlay + 1)) for n in ("flux_up", "flux_dn", "flux_dir"))
    else:
        gfu, gfd, gfdir = (buf(n, (ncol, nlay + 1, ngpt)) for n in ("gpt_flux_up", "gpt_flux_dn", "gpt_flux_dir"))
        fu = fd = fdir = buf("decoy2D", (ncol, nlay + 1))
    lib.rte_sw_solver_2stream(ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, sfc_alb_dir_gpt,
                              sfc_alb_dif_gpt, inc_flux_dir, gfu, gfd, gfdir, has_dif_bc,
                              inc_flux_dif, do_broadband, fu, fd, fdir)
    return b


# --------------------------------------------------------------------------------------
# by-band fluxes (rte/extensions/mo_fluxes_byband.F90: ty_fluxes_byband%reduce)
# --------------------------------------------------------------------------------------
def rte_lw_byband(lib, xp, ncol, nlay, ngpt, nbnd, band_lims_gpt, top_at_1, tau, lay_src, lev_src, sfc_emis_gpt, sfc_src,
                  n_gauss_angles: int = 1, inc_flux=None, buffers: Optional[Dict[str, object]] = None):
    """``rte_lw`` with a ``ty_fluxes_byband`` flux object: by-band fluxes (ncol, nlay+1, nbnd).  The reference computes the
    spectral arrays and reduces them (mo_rte_lw.F90:297-321 -> mo_fluxes_byband.F90:46-137: ``rte_sum_byband``); with the HIP
    library's extension ``rte_hip_lw_solver_noscat_byband`` the segmented kernel accumulates per band and the spectral arrays
    never exist.  Libraries without the extension (the oracle) take the reference's route."""
    b = buffers if buffers is not None else {}
    if "bb_up" not in b:
        b["bb_up"], b["bb_dn"] = xp.empty((ncol, nlay + 1, nbnd)), xp.empty((ncol, nlay + 1, nbnd))
    if lib.has("rte_hip_lw_solver_noscat_byband") and nlay <= 80:
        from . import hiplib

        nmus = n_gauss_angles
        if ("secants", nmus) not in b:  # mo_rte_lw.F90:357-365
            sec = np.empty((ncol, ngpt, nmus), order="F")
            for imu in range(nmus):
                sec[:, :, imu] = GAUSS_DS[nmus - 1][imu]
            b[("secants", nmus)] = xp.asarray(sec)
        if inc_flux is None:
            if "inc_flux_zero" not in b:
                b["inc_flux_zero"] = xp.zeros((ncol, ngpt))
            inc_flux = b["inc_flux_zero"]
        weights = np.array(GAUSS_WTS[nmus - 1], dtype=xp.ftype)
        rc = hiplib.ext_call(lib, "rte_hip_lw_solver_noscat_byband", "iiiiiiaaaaaaaaaaa", ncol, nlay, ngpt, nbnd, int(top_at_1), nmus,
                             b[("secants", nmus)], weights, band_lims_gpt, tau, lay_src, lev_src, sfc_emis_gpt, sfc_src, inc_flux,
                             b["bb_up"], b["bb_dn"])
        assert rc in (0, -2), rc
        if rc == 0:
            return b
        # (-2: a shape the extension does not take, e.g. ncol * (nlay + 1) >= 2^29 -- the reference's route below)
    r = rte_lw(lib, xp, ncol, nlay, ngpt, top_at_1, tau, lay_src, lev_src, sfc_emis_gpt, sfc_src, n_gauss_angles=n_gauss_angles,
               inc_flux=inc_flux, do_broadband=False, buffers=b)
    lib.rte_sum_byband(ncol, nlay + 1, ngpt, nbnd, band_lims_gpt, r["gpt_flux_up"], b["bb_up"])
    lib.rte_sum_byband(ncol, nlay + 1, ngpt, nbnd, band_lims_gpt, r["gpt_flux_dn"], b["bb_dn"])
    return b


def rte_sw_byband(lib, xp, ncol, nlay, ngpt, nbnd, band_lims_gpt, top_at_1, tau, ssa, g, mu0, inc_flux_dir, sfc_alb_dir_gpt,
                  sfc_alb_dif_gpt, inc_flux_dif=None, buffers: Optional[Dict[str, object]] = None):
    """``rte_sw`` with a ``ty_fluxes_byband`` flux object (mo_rte_sw.F90 -> mo_fluxes_byband.F90): by-band up / down / direct
    fluxes; ``rte_hip_sw_solver_2stream_byband`` where the library has it, the spectral arrays + ``rte_sum_byband`` otherwise."""
    b = buffers if buffers is not None else {}
    if "bb_up" not in b:
        b["bb_up"], b["bb_dn"], b["bb_dir"] = (xp.empty((ncol, nlay + 1, nbnd)) for _ in range(3))
    if lib.has("rte_hip_sw_solver_2stream_byband") and nlay <= 96:
        from . import hiplib

        dif = inc_flux_dif if inc_flux_dif is not None else inc_flux_dir
        rc = hiplib.ext_call(lib, "rte_hip_sw_solver_2stream_byband", "iiiiiaaaaaaaaiaaaa", ncol, nlay, ngpt, nbnd, int(top_at_1),
                             band_lims_gpt, tau, ssa, g, mu0, sfc_alb_dir_gpt, sfc_alb_dif_gpt, inc_flux_dir,
                             1 if inc_flux_dif is not None else 0, dif, b["bb_up"], b["bb_dn"], b["bb_dir"])
        assert rc in (0, -2), rc
        if rc == 0:
            return b
        # (-2: a shape the extension does not take -- the spectral arrays + rte_sum_byband below)
    r = rte_sw(lib, xp, ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, inc_flux_dir, sfc_alb_dir_gpt, sfc_alb_dif_gpt,
               inc_flux_dif=inc_flux_dif, do_broadband=False, buffers=b)
    for k_in, k_out in (("gpt_flux_up", "bb_up"), ("gpt_flux_dn", "bb_dn"), ("gpt_flux_dir", "bb_dir")):
        lib.rte_sum_byband(ncol, nlay + 1, ngpt, nbnd, band_lims_gpt, r[k_in], b[k_out])
    return b
