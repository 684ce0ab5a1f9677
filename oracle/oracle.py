"""Loaders for the CPU oracles.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package (rte-rrtmgp_amd/) never does.

  load_c(precision)    -> the plain-C restatement oracle/liboracle[_sp].so (built with gcc on demand)
  load_ref(precision)  -> the reference's own Fortran kernels oracle/_ref/librefkernels[_sp].so,
                          or None when neither a prebuilt binary nor /root/reference is available
  big_stack(fn, ...)   -> run ``fn`` in a thread with a 1 GiB stack: the flang-built reference keeps
                          automatic arrays such as pfrac(ncol,nlay,ngpt) on the stack
                          (reference rrtmgp/kernels/mo_gas_optics_rrtmgp_kernels.F90:613)
"""
from __future__ import annotations

import importlib.util
import os
import subprocess
import sys
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _pkg():
    """The product package, used here only for its ABI binding table (cabi.KernelLib)."""
    if "rte_rrtmgp_amd" not in sys.modules:
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        import rte_rrtmgp_amd  # noqa: F401  (root-level alias module)
    return sys.modules["rte_rrtmgp_amd"]


def build_c(force: bool = False) -> None:
    need = force or not all(os.path.exists(os.path.join(HERE, n)) for n in ("liboracle.so", "liboracle_sp.so"))
    if not need:
        need = any(os.path.getmtime(os.path.join(HERE, src)) > os.path.getmtime(os.path.join(HERE, n))
                   for src in ("rte_rrtmgp_oracle.c", "glue_oracle.c") for n in ("liboracle.so", "liboracle_sp.so"))
    if need:
        subprocess.check_call(["make", "-C", HERE, "liboracle.so", "liboracle_sp.so"],
                              stdout=subprocess.DEVNULL)


def build_ref() -> bool:
    """(Re)build oracle/_ref from /root/reference when that tree exists; report availability."""
    if os.path.isdir("/root/reference/rte/kernels") and not os.path.exists(
            os.path.join(HERE, "_ref", "librefkernels.so")):
        subprocess.check_call(["sh", os.path.join(HERE, "build_ref.sh")], stdout=subprocess.DEVNULL)
    return os.path.exists(os.path.join(HERE, "_ref", "librefkernels.so"))


def load_c(precision: str = "dp"):
    build_c()
    name = "liboracle.so" if precision == "dp" else "liboracle_sp.so"
    return _pkg().cabi.KernelLib(os.path.join(HERE, name), precision)


def load_ref(precision: str = "dp"):
    if not build_ref():
        return None
    name = "librefkernels.so" if precision == "dp" else "librefkernels_sp.so"
    return _pkg().cabi.KernelLib(os.path.join(HERE, "_ref", name), precision)


def big_stack(fn, *args, **kwargs):
    """Call ``fn(*args, **kwargs)`` on a thread with a 1 GiB stack and return its result."""
    box = {}

    def run():
        try:
            box["r"] = fn(*args, **kwargs)
        except BaseException as e:  # noqa: BLE001
            box["e"] = e

    old = threading.stack_size(1 << 30)
    try:
        t = threading.Thread(target=run)
        t.start()
        t.join()
    finally:
        threading.stack_size(old)
    if "e" in box:
        raise box["e"]
    return box.get("r")
