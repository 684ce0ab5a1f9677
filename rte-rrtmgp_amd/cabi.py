"""ctypes binding of the kernel C ABI declared in include/rte_rrtmgp_kernels.h.

One signature table (``SIGNATURES``) describes every entry point: each argument is
``(name, kind)`` with kind one of

    "i"  int   scalar, passed by reference        (Fortran ``integer, intent(in)``)
    "f"  Float scalar, passed by reference        (``real(wp), intent(in)``)
    "b"  Bool  scalar (1 byte), by reference      (``logical(wl), intent(in)``)
    "a"  array, passed as a raw pointer           (numpy array, torch tensor, int address or None)

The reference's Fortran frontend passes *every* scalar by address (no ``value``
attribute in rte/kernels/api/*.F90, rrtmgp/kernels/api/*.F90), which is what "by
reference" reproduces here.  Because the reference build (oracle/_ref), the C restatement
(oracle/liboracle.so) and the HIP library all export these same symbols, the same table binds
all three; the oracle loaders live under ``oracle/`` and import this module, never the reverse.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Sequence, Tuple

import numpy as np

Arg = Tuple[str, str]


def _sig(spec: str) -> List[Arg]:
    out = []
    for tok in spec.split():
        kind, name = tok.split(":")
        out.append((name, kind))
    return out


# Argument order = the reference bind(C) interfaces (file:line in the header comments).
SIGNATURES: Dict[str, List[Arg]] = {
    "rrtmgp_interpolation": _sig(
        "i:ncol i:nlay i:ngas i:nflav i:neta i:npres i:ntemp a:flavor a:press_ref_log a:temp_ref "
        "f:press_ref_log_delta f:temp_ref_min f:temp_ref_delta f:press_ref_trop_log a:vmr_ref "
        "a:play a:tlay a:col_gas a:jtemp a:fmajor a:fminor a:col_mix a:tropo a:jeta a:jpress"),
    "rrtmgp_compute_tau_absorption": _sig(
        "i:ncol i:nlay i:nbnd i:ngpt i:ngas i:nflav i:neta i:npres i:ntemp "
        "i:nminorlower i:nminorklower i:nminorupper i:nminorkupper i:idx_h2o "
        "a:gpoint_flavor a:band_lims_gpt a:kmajor a:kminor_lower a:kminor_upper "
        "a:minor_limits_gpt_lower a:minor_limits_gpt_upper "
        "a:minor_scales_with_density_lower a:minor_scales_with_density_upper "
        "a:scale_by_complement_lower a:scale_by_complement_upper "
        "a:idx_minor_lower a:idx_minor_upper a:idx_minor_scaling_lower a:idx_minor_scaling_upper "
        "a:kminor_start_lower a:kminor_start_upper a:tropo a:col_mix a:fmajor a:fminor "
        "a:play a:tlay a:col_gas a:jeta a:jtemp a:jpress a:tau"),
    "rrtmgp_compute_tau_rayleigh": _sig(
        "i:ncol i:nlay i:nbnd i:ngpt i:ngas i:nflav i:neta i:npres i:ntemp a:gpoint_flavor "
        "a:band_lims_gpt a:krayl i:idx_h2o a:col_dry a:col_gas a:fminor a:jeta a:tropo a:jtemp "
        "a:tau_rayleigh"),
    "rrtmgp_compute_Planck_source": _sig(
        "i:ncol i:nlay i:nbnd i:ngpt i:nflav i:neta i:npres i:ntemp i:nPlanckTemp a:tlay a:tlev "
        "a:tsfc i:sfc_lay a:fmajor a:jeta a:tropo a:jtemp a:jpress a:gpoint_bands a:band_lims_gpt "
        "a:pfracin f:temp_ref_min f:totplnk_delta a:totplnk a:gpoint_flavor a:sfc_src a:lay_src "
        "a:lev_src a:sfc_source_Jac"),
    "rte_lw_solver_noscat": _sig(
        "i:ncol i:nlay i:ngpt b:top_at_1 i:nmus a:Ds a:weights a:tau a:lay_source a:lev_source "
        "a:sfc_emis a:sfc_src a:inc_flux a:flux_up a:flux_dn b:do_broadband a:broadband_up "
        "a:broadband_dn b:do_Jacobians a:sfc_srcJac a:flux_upJac b:do_rescaling a:ssa a:g"),
    "rte_lw_solver_2stream": _sig(
        "i:ncol i:nlay i:ngpt b:top_at_1 a:tau a:ssa a:g a:lay_source a:lev_source a:sfc_emis "
        "a:sfc_src a:inc_flux a:flux_up a:flux_dn"),
    "rte_sw_solver_noscat": _sig(
        "i:ncol i:nlay i:ngpt b:top_at_1 a:tau a:mu0 a:inc_flux_dir a:flux_dir"),
    "rte_sw_solver_2stream": _sig(
        "i:ncol i:nlay i:ngpt b:top_at_1 a:tau a:ssa a:g a:mu0 a:sfc_alb_dir a:sfc_alb_dif "
        "a:inc_flux_dir a:flux_up a:flux_dn a:flux_dir b:has_dif_bc a:inc_flux_dif b:do_broadband "
        "a:broadband_up a:broadband_dn a:broadband_dir"),
    "rte_sum_broadband": _sig("i:ncol i:nlev i:ngpt a:spectral_flux a:broadband_flux"),
    "rte_net_broadband_full": _sig(
        "i:ncol i:nlev i:ngpt a:spectral_flux_dn a:spectral_flux_up a:broadband_flux_net"),
    "rte_net_broadband_precalc": _sig("i:ncol i:nlev a:flux_dn a:flux_up a:broadband_flux_net"),
    "rte_compute_Planck_source_2D": _sig("i:ncol i:nlay i:nnu a:nus a:dnus a:T a:source"),
    "rte_compute_Planck_source_1D": _sig("i:ncol i:nnu a:nus a:dnus a:T a:source"),
    "rte_sum_byband": _sig("i:ncol i:nlev i:ngpt i:nbnd a:band_lims a:spectral_flux a:byband_flux"),
    "rte_net_byband_full": _sig(
        "i:ncol i:nlev i:ngpt i:nbnd a:band_lims a:spectral_flux_dn a:spectral_flux_up a:byband_flux_net"),
    "net_byband_precalc": _sig("i:ncol i:nlev i:nbnd a:byband_flux_dn a:byband_flux_up a:byband_flux_net"),
    "zero_array_1D": _sig("i:ni a:array"),
    "zero_array_2D": _sig("i:ni i:nj a:array"),
    "zero_array_3D": _sig("i:ni i:nj i:nk a:array"),
    "zero_array_4D": _sig("i:ni i:nj i:nk i:nl a:array"),
    "set_to_scalar_1D": _sig("i:ni a:array f:value"),
    "set_to_scalar_2D": _sig("i:ni i:nj a:array f:value"),
    "set_to_scalar_3D": _sig("i:ni i:nj i:nk a:array f:value"),
    "set_to_scalar_4D": _sig("i:ni i:nj i:nk i:nl a:array f:value"),
    "rte_delta_scale_2str_f_k": _sig("i:ncol i:nlay i:ngpt a:tau a:ssa a:g a:f"),
    "rte_delta_scale_2str_k": _sig("i:ncol i:nlay i:ngpt a:tau a:ssa a:g"),
    "rte_increment_1scalar_by_1scalar": _sig("i:ncol i:nlay i:ngpt a:tau1 a:tau2"),
    "rte_increment_1scalar_by_2stream": _sig("i:ncol i:nlay i:ngpt a:tau1 a:tau2 a:ssa2"),
    "rte_increment_1scalar_by_nstream": _sig("i:ncol i:nlay i:ngpt a:tau1 a:tau2 a:ssa2"),
    "rte_increment_2stream_by_1scalar": _sig("i:ncol i:nlay i:ngpt a:tau1 a:ssa1 a:tau2"),
    "rte_increment_2stream_by_2stream": _sig("i:ncol i:nlay i:ngpt a:tau1 a:ssa1 a:g1 a:tau2 a:ssa2 a:g2"),
    "rte_increment_2stream_by_nstream": _sig("i:ncol i:nlay i:ngpt i:nmom2 a:tau1 a:ssa1 a:g1 a:tau2 a:ssa2 a:p2"),
    "rte_increment_nstream_by_1scalar": _sig("i:ncol i:nlay i:ngpt a:tau1 a:ssa1 a:tau2"),
    "rte_increment_nstream_by_2stream": _sig("i:ncol i:nlay i:ngpt i:nmom1 a:tau1 a:ssa1 a:p1 a:tau2 a:ssa2 a:g2"),
    "rte_increment_nstream_by_nstream": _sig(
        "i:ncol i:nlay i:ngpt i:nmom1 i:nmom2 a:tau1 a:ssa1 a:p1 a:tau2 a:ssa2 a:p2"),
    "rte_inc_1scalar_by_1scalar_bybnd": _sig("i:ncol i:nlay i:ngpt a:tau1 a:tau2 i:nbnd a:gpt_lims"),
    "rte_inc_1scalar_by_2stream_bybnd": _sig("i:ncol i:nlay i:ngpt a:tau1 a:tau2 a:ssa2 i:nbnd a:gpt_lims"),
    "rte_inc_1scalar_by_nstream_bybnd": _sig("i:ncol i:nlay i:ngpt a:tau1 a:tau2 a:ssa2 i:nbnd a:gpt_lims"),
    "rte_inc_2stream_by_1scalar_bybnd": _sig("i:ncol i:nlay i:ngpt a:tau1 a:ssa1 a:tau2 i:nbnd a:gpt_lims"),
    "rte_inc_2stream_by_2stream_bybnd": _sig(
        "i:ncol i:nlay i:ngpt a:tau1 a:ssa1 a:g1 a:tau2 a:ssa2 a:g2 i:nbnd a:gpt_lims"),
    "rte_inc_2stream_by_nstream_bybnd": _sig(
        "i:ncol i:nlay i:ngpt i:nmom2 a:tau1 a:ssa1 a:g1 a:tau2 a:ssa2 a:p2 i:nbnd a:gpt_lims"),
    "rte_inc_nstream_by_1scalar_bybnd": _sig("i:ncol i:nlay i:ngpt a:tau1 a:ssa1 a:tau2 i:nbnd a:gpt_lims"),
    "rte_inc_nstream_by_2stream_bybnd": _sig(
        "i:ncol i:nlay i:ngpt i:nmom1 a:tau1 a:ssa1 a:p1 a:tau2 a:ssa2 a:g2 i:nbnd a:gpt_lims"),
    "rte_inc_nstream_by_nstream_bybnd": _sig(
        "i:ncol i:nlay i:ngpt i:nmom1 i:nmom2 a:tau1 a:ssa1 a:p1 a:tau2 a:ssa2 a:p2 i:nbnd a:gpt_lims"),
    "rte_extract_subset_dim1_3d": _sig("i:ncol i:nlay i:ngpt a:array_in i:colS i:colE a:array_out"),
    "rte_extract_subset_dim2_4d": _sig("i:nmom i:ncol i:nlay i:ngpt a:array_in i:colS i:colE a:array_out"),
    "rte_extract_subset_absorption_tau": _sig("i:ncol i:nlay i:ngpt a:tau_in a:ssa_in i:colS i:colE a:tau_out"),
    "rrtmgp_compute_cld_from_table": _sig(
        "i:ncol i:nlay i:ngpt a:mask a:lwp a:re i:nsteps f:step_size f:offset a:tau_table a:ssa_table "
        "a:asy_table a:tau a:taussa a:taussag"),
}


def header_symbols(header_path: str) -> List[str]:
    """Names of all functions declared in a C header (used by the symbol-export test)."""
    txt = open(header_path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return re.findall(r"\bvoid\s+(\w+)\s*\(", txt)


def as_pointer(x) -> ctypes.c_void_p:
    """Raw address of a numpy array / torch tensor / int / None."""
    if x is None:
        return ctypes.c_void_p(0)
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if isinstance(x, np.ndarray):
        if not (x.flags.f_contiguous or x.flags.c_contiguous):
            raise ValueError("array crossing the C ABI must be dense")
        return ctypes.c_void_p(x.ctypes.data)
    if hasattr(x, "data_ptr"):  # torch tensor (host or device)
        if not x.is_contiguous():
            raise ValueError("tensor crossing the C ABI must be contiguous")
        return ctypes.c_void_p(x.data_ptr())
    raise TypeError(f"cannot pass {type(x)} as an array argument")


class KernelLib:
    """A shared library exporting the kernel C ABI, callable with keyword or positional args.

    ``lib.rte_lw_solver_noscat(ncol, nlay, ...)`` converts scalars to by-reference ctypes
    values of the library's precision and arrays to raw pointers; it returns nothing (all
    entry points are ``void`` subroutines, like the reference's).
    """

    def __init__(self, path: str, precision: str = "dp", required: Sequence[str] = ()):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.precision = precision
        self.cfloat = ctypes.c_double if precision == "dp" else ctypes.c_float
        self.npfloat = np.float64 if precision == "dp" else np.float32
        self._dll = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL if False else ctypes.DEFAULT_MODE)
        self._fns = {}
        for name in required:
            self._get(name)

    def has(self, name: str) -> bool:
        try:
            self._get(name)
            return True
        except AttributeError:
            return False

    def raw(self, name: str):
        """The bare ctypes function for non-reference (library extension) symbols."""
        return getattr(self._dll, name)

    def _get(self, name: str):
        fn = self._fns.get(name)
        if fn is None:
            fn = getattr(self._dll, name)  # AttributeError if the symbol is missing
            fn.restype = None
            self._fns[name] = fn
        return fn

    def call(self, fname: str, *args, **kwargs):
        sig = SIGNATURES[fname]
        if len(args) > len(sig):
            raise TypeError(f"{fname}: too many arguments")
        vals = list(args)
        for name, _ in sig[len(args):]:
            if name not in kwargs:
                raise TypeError(f"{fname}: missing argument {name}")
            vals.append(kwargs[name])
        keep, cargs = [], []
        for (name, kind), v in zip(sig, vals):
            if kind == "i":
                c = ctypes.c_int(int(v))
                keep.append(c)
                cargs.append(ctypes.byref(c))
            elif kind == "f":
                c = self.cfloat(float(v))
                keep.append(c)
                cargs.append(ctypes.byref(c))
            elif kind == "b":
                c = ctypes.c_bool(bool(v))
                keep.append(c)
                cargs.append(ctypes.byref(c))
            else:
                cargs.append(as_pointer(v))
        self._get(fname)(*cargs)

    def __getattr__(self, name):
        if name in SIGNATURES:
            return lambda *a, **k: self.call(name, *a, **k)
        raise AttributeError(name)
