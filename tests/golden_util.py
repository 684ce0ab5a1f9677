"""Compare a run_suite() result with a committed golden fixture (tests/golden/<case>.npz)."""
import os

import numpy as np

import cases

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(case_name):
    return np.load(os.path.join(GOLDEN_DIR, case_name + ".npz"))


def compare(case_name, out, inputs, rtol, skip_prefixes=()):
    """Returns the worst relative error over all stored outputs; asserts the inputs match."""
    z = load(case_name)
    assert str(z["__digest__"]) == cases.inputs_digest(*inputs), "fixture was made from different inputs"
    names = sorted({k.split("|")[0] for k in z.files if "|" in k})
    assert names, "empty fixture"
    worst = (0.0, None)
    for n in names:
        if any(n.startswith(p) for p in skip_prefixes):
            continue
        assert n in out, f"suite did not produce {n}"
        a = np.asarray(out[n])
        if f"{n}|full" in z.files:
            e = max(cases.rel_err(a, z[f"{n}|full"]), cases.elem_err(a, z[f"{n}|full"], 1e-4) * 1e-2)  # elementwise at 100 x rtol
        else:
            stride = int(z[f"{n}|stride"])
            assert tuple(z[f"{n}|shape"]) == a.shape
            flat = a.ravel(order="F")
            e = cases.rel_err(flat[::stride], z[f"{n}|sample"])
            st = z[f"{n}|stats"]
            if a.dtype.kind == "f":
                mine = np.array([flat.sum(), (flat * flat).sum(), flat.min(), flat.max()])
                scale = np.maximum(np.abs(st), 1e-300)
                e = max(e, float(np.max(np.abs(mine - st) / scale) / 16.0))  # sums: allow reordering noise
            else:
                mine = np.array([flat.astype(np.int64).sum(), 0, flat.min(), flat.max()], dtype=np.float64)
                if not np.array_equal(mine, st):
                    e = float("inf")
        if e > worst[0]:
            worst = (e, n)
        assert e <= rtol, f"{case_name}: {n} differs from the golden fixture by {e:.3e} (> {rtol})"
    return worst
