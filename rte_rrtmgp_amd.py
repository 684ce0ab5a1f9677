"""Importable alias for the product package directory ``rte-rrtmgp_amd/``.

The package directory carries the project's hyphenated name, which Python cannot import
directly; ``import rte_rrtmgp_amd`` executes this file, which loads ``rte-rrtmgp_amd/__init__.py``
as the package ``rte_rrtmgp_amd`` (submodules resolve inside that directory) and replaces this
module in ``sys.modules`` with it.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rte-rrtmgp_amd")
_spec = importlib.util.spec_from_file_location(
    "rte_rrtmgp_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rte_rrtmgp_amd"] = _mod
_spec.loader.exec_module(_mod)
