"""GPU-box helper: the reference's frontend in single precision (oracle/build_extern_sp.sh) on the SP HIP library and on the
reference's SP CPU kernels, each against the double-precision CPU run of the same inputs: worst absolute flux deviation
[W/m2] (the reference accepts 3.5e-1 for SP results, examples/compare-to-reference.py) and relative to the largest flux.
usage: sp_frontend_errors.py [lw|sw] [block] [top_at_1]"""
import sys, tempfile, pathlib
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import stream_io, test_extern_frontend as T
kind = sys.argv[1] if len(sys.argv) > 1 else "sw"
block = int(sys.argv[2]) if len(sys.argv) > 2 else 512
top = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
ncol, nlay = 512, 60
ngpt, nbnd = (256, 16) if kind == "lw" else (224, 14)
with tempfile.TemporaryDirectory() as d:
    d = pathlib.Path(d)
    raw, kd, atm, kf, af = T._frontend_case(d, kind, ncol, nlay, block, top, True, True, ngpt=ngpt, nbnd=nbnd,
                                            nminor_lower=4 * nbnd, nminor_upper=2 * nbnd + 3, seed=19, checks=True)
    run = lambda b, **kw: stream_io.run_frontend_driver(b, kf, af, str(d / (b + ".bin")), T.GASES, ncol, nlay, kind == "lw", **kw)[0]
    dp = run("ref_frontend_driver_cpuref")
    res = {"cpu sp": run("ref_frontend_driver_sp_cpuref"), "hip sp": run("ref_frontend_driver_sp", env={"RTE_HIP_HOST_MIRROR": "0"}),
           "hip dp": run("ref_frontend_driver", env={"RTE_HIP_HOST_MIRROR": "0"})}
    for name, r in res.items():
        for k in dp:
            e = np.abs(r[k] - dp[k])
            i = np.unravel_index(np.argmax(e), e.shape)
            print(f"{kind} block {block} {name} {k:12s}: worst |d| {e.max():.3e} W/m2 at (col {i[0]}, lev {i[1]}), flux there {dp[k][i]:.4f}; relative to max flux {e.max() / np.abs(dp[k]).max():.2e}")
