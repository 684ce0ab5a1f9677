"""GPU-box helper: spectral-output solves (segmented kernels storing per-g-point fluxes) against the generic kernels,
1e5 x 60 columns: LW no-scattering (256 g-points, 1 angle) and SW two-stream (224 g-points)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nlay = int(sys.argv[2]) if len(sys.argv) > 2 else 60
def rnd(*sh, scale=1.0, off=0.0):
    return torch.rand(*reversed(sh), dtype=torch.float64, device="cuda").mul_(scale).add_(off)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for kind, ngpt in (("lw", 256), ("sw", 224)):
    tau, ssa, g = rnd(ncol, nlay, ngpt, scale=2.0), rnd(ncol, nlay, ngpt, scale=0.9), rnd(ncol, nlay, ngpt, scale=0.8)
    bufs = {}
    if kind == "lw":
        lay, lev = rnd(ncol, nlay, ngpt, scale=10, off=1), rnd(ncol, nlay + 1, ngpt, scale=10, off=1)
        emis, sfc = rnd(ncol, ngpt, scale=0.2, off=0.8), rnd(ncol, ngpt, scale=10)
        fn = lambda: frontend.rte_lw(lib, xp, ncol, nlay, ngpt, False, tau, lay, lev, emis, sfc, do_broadband=False, buffers=bufs)
        out_gb = 2 * 8 * ncol * (nlay + 1) * ngpt / 1e9; in_gb = (2 * nlay + nlay + 1) * 8 * ncol * ngpt / 1e9
        force = "rte_hip_force_generic_lw"
    else:
        mu0 = xp.full((ncol, nlay), 0.86); idir, alb = rnd(ncol, ngpt, scale=100), rnd(ncol, ngpt, scale=0.5)
        fn = lambda: frontend.rte_sw(lib, xp, ncol, nlay, ngpt, False, tau, ssa, g, mu0, idir, alb, alb, do_broadband=False, buffers=bufs)
        out_gb = 3 * 8 * ncol * (nlay + 1) * ngpt / 1e9; in_gb = 3 * nlay * 8 * ncol * ngpt / 1e9
        force = "rte_hip_force_generic_sw"
    t_seg = timed(fn)
    hiplib.ext_call(lib, force, ["i"], 1)
    t_gen = timed(fn)
    hiplib.ext_call(lib, force, ["i"], 0)
    print(f"{kind} spectral {ncol} x {nlay} x {ngpt}: segmented {t_seg:.2f} ms ({(in_gb + out_gb) / t_seg:.2f} TB/s on {in_gb + out_gb:.1f} GB; "
          f"outputs alone {out_gb:.1f} GB = {out_gb / 5.5:.2f} ms at 5.5 TB/s), generic {t_gen:.2f} ms")
    del tau, ssa, g, bufs; torch.cuda.empty_cache()
# by-band fluxes (16 bands) straight from the segmented kernels against spectral output + rte_sum_byband
for kind, ngpt, nbnd in (("lw", 256, 16), ("sw", 224, 14)):
    gpb = ngpt // nbnd
    bl = xp.asarray(np.asfortranarray(np.stack([1 + gpb * np.arange(nbnd), gpb * (1 + np.arange(nbnd))]).astype(np.int32)))
    tau, ssa, g = rnd(ncol, nlay, ngpt, scale=2.0), rnd(ncol, nlay, ngpt, scale=0.9), rnd(ncol, nlay, ngpt, scale=0.8)
    if kind == "lw":
        lay, lev = rnd(ncol, nlay, ngpt, scale=10, off=1), rnd(ncol, nlay + 1, ngpt, scale=10, off=1)
        emis, sfc = rnd(ncol, ngpt, scale=0.2, off=0.8), rnd(ncol, ngpt, scale=10)
        bufs = {}
        fn = lambda: frontend.rte_lw_byband(lib, xp, ncol, nlay, ngpt, nbnd, bl, False, tau, lay, lev, emis, sfc, buffers=bufs)
    else:
        mu0 = xp.full((ncol, nlay), 0.86); idir, alb = rnd(ncol, ngpt, scale=100), rnd(ncol, ngpt, scale=0.5)
        bufs = {}
        fn = lambda: frontend.rte_sw_byband(lib, xp, ncol, nlay, ngpt, nbnd, bl, False, tau, ssa, g, mu0, idir, alb, alb, buffers=bufs)
    print(f"{kind} by-band {ncol} x {nlay} x {ngpt} ({nbnd} bands): {timed(fn):.2f} ms (spectral output + rte_sum_byband: the spectral time above plus the reduction)")
    del tau, ssa, g, bufs; torch.cuda.empty_cache()
