#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/*.npz from the REFERENCE kernels themselves.

Run in the build container (needs /root/reference and flang):  python tests/golden/make_golden.py

For every seeded case of tests/cases.py the reference's own `default` Fortran kernels
(compiled in place by oracle/build_ref.sh into oracle/_ref/librefkernels.so -- binary only, never
committed) are driven through the kernel C ABI, and the outputs are stored:
  * small arrays (<= 6000 elements) in full;
  * large arrays as a strided sample (every k-th element in Fortran order) plus
    [sum, sum of squares, min, max];
  * a SHA-256 of all inputs, so a fixture can never be compared against different inputs.
The tiny cases additionally store their complete inputs (k-distribution + atmosphere).
A fixture is data only: no reference source text is stored anywhere.

NOTE: the reference's default CPU lw_solver_2stream uses g-point 1's level source for every
g-point (rte/kernels/mo_rte_solver_kernels.F90:420-424 vs :929-930); the "lw2str.*" entries
therefore hold that bug-compatible result.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
from oracle import oracle as O  # noqa: E402
from rte_rrtmgp_amd import frontend  # noqa: E402

FULL_LIMIT = 6000


def summarize(arr):
    arr = np.asarray(arr)
    if arr.size <= FULL_LIMIT:
        return {"full": arr}
    flat = arr.ravel(order="F")
    stride = max(1, -(-arr.size // 1024))
    d = {"sample": flat[::stride].copy(), "stride": np.array(stride), "shape": np.array(arr.shape)}
    if arr.dtype.kind == "f":
        d["stats"] = np.array([flat.sum(), (flat * flat).sum(), flat.min(), flat.max()])
    else:
        d["stats"] = np.array([flat.astype(np.int64).sum(), 0, flat.min(), flat.max()], dtype=np.float64)
    return d


def main():
    ref = O.load_ref()
    if ref is None:
        raise SystemExit("reference build unavailable (needs /root/reference + flang)")
    xp = frontend.NumpyArrays()
    for name, case in cases.CASES.items():
        inp = cases.make_inputs(case)
        out = O.big_stack(cases.run_suite, ref, xp, case, inp)
        store = {"__digest__": np.array(cases.inputs_digest(*inp))}
        for k, v in out.items():
            for kk, vv in summarize(v).items():
                store[f"{k}|{kk}"] = vv
        if "tiny" in name:
            kd, atm, ex = inp
            for k, v in kd.arrays.items():
                store[f"in.kd.{k}"] = v
            for k, v in kd.scalars.items():
                store[f"in.kds.{k}"] = np.array(v)
            for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas", "col_dry"):
                store[f"in.atm.{k}"] = getattr(atm, k)
            for k, v in ex.items():
                store[f"in.ex.{k}"] = v
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **store)
        print(name, len(out), "arrays ->", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
