"""CPU tests of the drop-in boundary (no compute calls): the HIP library builds for gfx950, loads,
and exports every symbol include/rte_rrtmgp_kernels.h declares; the ctypes signature table agrees
with the header's argument lists; and the symbol set is the reference's own bind(C) name set."""
import ctypes
import os
import re

import pytest

from rte_rrtmgp_amd import cabi, hiplib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rte_rrtmgp_kernels.h")


def _header_decls():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return {m.group(1): [a.strip() for a in m.group(2).split(",")]
            for m in re.finditer(r"\bvoid\s+(\w+)\s*\((.*?)\)\s*;", txt, flags=re.S)}


def test_library_builds_and_exports_every_header_symbol():
    path = hiplib.build()
    assert os.path.exists(path)
    hiplib._one_hip_runtime()
    dll = ctypes.CDLL(path)
    for name in cabi.header_symbols(HEADER):
        assert hasattr(dll, name), f"{name} declared in the header but not exported"
    for ext in ("rte_hip_set_stream", "rte_hip_sync", "rte_hip_profile_enable", "rte_hip_profile_get",
                "rte_hip_combine_abs_and_rayleigh_2str", "rte_hip_broadcast_gpt", "rte_hip_release",
                "rte_hip_defer_zero", "rte_hip_force_direct_gather",
                "rte_hip_force_generic_lw", "rte_hip_force_generic_sw", "rte_hip_invalidate_plans",
                "rte_hip_set_lw2str_bugcompat", "rte_hip_device_count", "rte_hip_cloud_masks", "rte_hip_cloud_combine",
                "rte_hip_seg_groups", "rte_hip_get_layer_number", "rte_hip_get_layer_mass",
                "rte_hip_col_gas_fill", "rte_hip_tlev_interp", "rte_hip_compute_optimal_angles",
                "rte_hip_combine_abs_and_rayleigh_1scl", "rte_hip_combine_abs_and_rayleigh_nstr",
                "rte_hip_expand_and_transpose", "rte_hip_secants_fill", "rte_hip_rfmip_sw_toa_renorm",
                "rte_hip_rfmip_sw_mu0", "rte_hip_broadcast_cols", "rte_hip_mask_columns",
                "rte_hip_tau_rayleigh_combine_2str", "rte_hip_compute_tau_absorption_inc_bybnd",
                "rte_hip_cloud_optics_fused", "rte_hip_lw_sfc_lds", "rte_hip_stat", "rte_hip_overlap_planck", "rte_hip_share_geometry", "rte_hip_gas_optics_sw_2str",
                "rte_hip_aux_stream", "rte_hip_profile_only", "rte_hip_lw_solver_noscat_byband", "rte_hip_sw_solver_2stream_byband",
                "rte_hip_compute_Planck_source_factored", "rte_hip_lw_solver_noscat_factored", "rte_hip_expand_factored_sources"):
        assert hasattr(dll, ext)
    # the public extension header: contexts, error channel, host-mirror mode, opt-in modes, timing
    ext_h = open(os.path.join(ROOT, "include", "rte_hip_ext.h")).read()
    ext_h = re.sub(r"/\*.*?\*/", "", ext_h, flags=re.S)
    declared = re.findall(r"\b(rte_hip_\w+)\s*\(", ext_h)
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(dll, name), f"{name} declared in include/rte_hip_ext.h but not exported"


def test_signature_table_matches_header():
    decls = _header_decls()
    assert set(decls) == set(cabi.SIGNATURES), set(decls) ^ set(cabi.SIGNATURES)
    for name, args in decls.items():
        sig = cabi.SIGNATURES[name]
        assert len(args) == len(sig), name
        for a, (argname, kind) in zip(args, sig):
            toks = a.replace("*", " * ").split()
            assert toks[-1] == argname, (name, a, argname)
            base = [t for t in toks[:-1] if t not in ("const", "*")][0]
            is_array = kind == "a"
            if kind == "i":
                assert base == "int"
            elif kind == "f":
                assert base == "Float"
            elif kind == "b":
                assert base == "Bool"
            # scalars are `const T*` (by reference); output arrays are non-const
            assert "*" in toks
            if not is_array:
                assert "const" in toks


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """No CPU fallback: a missing extension is an error, not a silent eager path."""
    monkeypatch.setattr(hiplib, "PKG_DIR", str(tmp_path))
    monkeypatch.setattr(hiplib, "_loaded", {})
    monkeypatch.setattr(hiplib.shutil, "which", lambda *_: None)
    monkeypatch.setattr(hiplib.os.path, "exists", lambda p: False if "hipcc" in p or str(tmp_path) in p else os.path.lexists(p))
    with pytest.raises((RuntimeError, FileNotFoundError)):
        hiplib.load()


@pytest.mark.skipif(not os.path.isdir("/root/reference/rte/kernels/api"), reason="reference tree not present")
def test_names_are_the_reference_bind_c_names():
    """Every symbol we export under a reference name exists as bind(C) in the reference's api modules."""
    names = set()
    files = []
    for d in ("/root/reference/rte/kernels/api", "/root/reference/rrtmgp/kernels/api"):
        files += [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".F90")]
    files.append("/root/reference/rte/extensions/mo_fluxes_byband.F90")  # the by-band reducers are bind(C) there
    for f in files:
        txt = open(f).read()
        names |= set(re.findall(r'bind\s*\(\s*C\s*,\s*name\s*=\s*"(\w+)"', txt, flags=re.I))
        # bind(C) without a name: the binding label is the lower-cased procedure name
        names |= {m.lower() for m in re.findall(r'subroutine\s+(\w+)\s*\([^)]*\)\s*bind\s*\(\s*C\s*\)', txt, flags=re.I)}
    ours = set(cabi.header_symbols(HEADER))
    assert ours <= names, ours - names
