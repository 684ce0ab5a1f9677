"""Builder / loader of the product library ``librte_rrtmgp_hip.so`` (hand-written HIP, gfx950).

``build()`` compiles rte-rrtmgp_amd/csrc/*.hip with hipcc (cross-compiles without a GPU) into
rte-rrtmgp_amd/librte_rrtmgp_hip.so (in-tree so it travels with the repo snapshot; git-ignored).
``load()`` returns a ``cabi.KernelLib`` on it and FAILS LOUDLY when the library is missing or a
symbol is absent -- there is no CPU fallback anywhere in the product path.
"""
from __future__ import annotations

import ctypes
import glob
import os
import shutil
import subprocess
from typing import List, Optional, Sequence

from . import cabi

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
ROOT = os.path.dirname(PKG_DIR)
LIB_NAMES = {"dp": "librte_rrtmgp_hip.so", "sp": "librte_rrtmgp_hip_sp.so"}
if os.environ.get("RTE_HIP_VARIANT"):  # experiment builds of tools/variants.py (librte_rrtmgp_hip_x<tag>.so), A/B timing only
    LIB_NAMES["dp"] = f"librte_rrtmgp_hip_x{os.environ['RTE_HIP_VARIANT']}.so"
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function"]

_loaded = {}


def lib_path(precision: str = "dp") -> str:
    return os.path.join(PKG_DIR, LIB_NAMES[precision])


def sources() -> List[str]:
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(path: str) -> bool:
    if not os.path.exists(path):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps)


def build(precision: str = "dp", force: bool = False, verbose: bool = False, extra: Sequence[str] = ()) -> str:
    """Compile the HIP library for gfx950; returns its path."""
    out = lib_path(precision)
    if not force and not _stale(out):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build librte_rrtmgp_hip.so")
    # one builder at a time (several ranks of a multi-GPU launch may get here together): the others wait
    # and then find the library up to date; the link goes to a temporary name and is renamed into place
    import fcntl

    import hashlib
    import tempfile

    # the lock lives outside the package directory (nothing but sources and the product is left beside them)
    lock_path = os.path.join(tempfile.gettempdir(), "rte_hip_build_" + hashlib.sha1(out.encode()).hexdigest()[:16] + ".lock")
    with open(lock_path, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale(out):
                return out
            tmp = f"{out}.{os.getpid()}.tmp"
            cmd = [hipcc] + HIPCC_FLAGS + list(extra) + ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
            if precision == "sp":
                cmd.append("-DRTE_USE_SP")
            cmd += sources() + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, out)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out


def _one_hip_runtime() -> None:
    """torch ships its own libamdhip64; if this library were dlopen'ed first it would pull in
    /opt/rocm's copy and the process would end up with two HIP runtimes (the second one then
    reports "no ROCm-capable device").  Import torch first so both share torch's runtime.  A
    torch-free host program (e.g. the Fortran frontend) simply uses /opt/rocm's."""
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch absent: single runtime anyway
        pass


def load(precision: str = "dp", build_if_missing: bool = True) -> cabi.KernelLib:
    """Load the HIP library.  Raises (never falls back) when it cannot be built or loaded."""
    if precision in _loaded:
        return _loaded[precision]
    _one_hip_runtime()
    path = lib_path(precision)
    # (an experiment build named by RTE_HIP_VARIANT is used as it is: rebuilding it here would drop its flags)
    if _stale(path) and not (os.environ.get("RTE_HIP_VARIANT") and precision == "dp" and os.path.exists(path)):
        if not build_if_missing or not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
            if not os.path.exists(path):
                raise RuntimeError(f"{path} is missing and cannot be built: the HIP extension is "
                                   "mandatory (no CPU fallback in the product path)")
        else:
            build(precision)
    hdr = os.path.join(ROOT, "include", "rte_rrtmgp_kernels.h")
    lib = cabi.KernelLib(path, precision, required=cabi.header_symbols(hdr))
    _loaded[precision] = lib
    return lib


_KINDS = {"i": ctypes.c_int, "l": ctypes.c_longlong, "d": ctypes.c_double}


def ext_call(lib: cabi.KernelLib, name: str, kinds: Sequence[str], *args):
    """Call a library-extension entry point (``rte_hip_*``).  Extension symbols take scalars BY
    VALUE ('i' int, 'l' long long, 'd' double) and arrays as pointers ('a')."""
    fn = lib.raw(name)
    cargs = []
    for k, v in zip(kinds, args):
        cargs.append(cabi.as_pointer(v) if k == "a" else _KINDS[k](v))
    fn.restype = ctypes.c_int
    return fn(*cargs)


class CallGraph:
    """A sequence of library calls captured as one hipGraph (``rte_hip_graph_begin`` / ``_end``) and replayed with one
    submission: ``g = CallGraph(lib, fn)`` runs ``fn()`` once uncaptured (sizes the scratch arena) and once captured;
    ``g.launch()`` replays it on the library's stream.  ``fn`` must use device arrays that stay where they are (pass the
    ``buffers=`` dictionaries of the frontend functions) and must not read values back."""

    def __init__(self, lib: cabi.KernelLib, fn, warm: bool = True):
        self.lib = lib
        if warm:
            fn()
        if ext_call(lib, "rte_hip_graph_begin", []) != 0:
            raise RuntimeError("rte_hip_graph_begin failed")
        try:
            fn()
        finally:
            handle = ctypes.c_void_p(0)
            rc = lib.raw("rte_hip_graph_end")(ctypes.byref(handle))
        if rc != 0 or not handle.value:
            raise RuntimeError("rte_hip_graph_end failed")
        self.handle = handle

    def launch(self) -> None:
        fn = self.lib.raw("rte_hip_graph_launch")
        fn.restype = ctypes.c_int
        rc = fn(self.handle)
        if rc == -4:
            raise RuntimeError("rte_hip_graph_launch: the graph is stale -- library buffers it addresses were freed or reallocated "
                               "since the capture (a larger call on this context, rte_hip_release): capture it again")
        if rc != 0:
            raise RuntimeError("rte_hip_graph_launch failed")

    def close(self) -> None:
        if self.handle is not None and self.handle.value:
            self.lib.raw("rte_hip_graph_destroy")(self.handle)
            self.handle = None


def set_stream(lib: cabi.KernelLib, stream_handle: Optional[int]) -> None:
    """Make the library launch on the given hipStream_t (0/None = the null stream, which is also
    torch's default stream on ROCm)."""
    ext_call(lib, "rte_hip_set_stream", ["a"], int(stream_handle or 0))
