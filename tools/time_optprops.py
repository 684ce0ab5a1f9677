"""GPU-box helper: the all-sky elementwise kernels at benchmark size (1e5 x 60 x 256): achieved HBM GB/s."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from rte_rrtmgp_amd import frontend, hiplib
lib = hiplib.load(); xp = frontend.TorchArrays("cuda:0")
ncol, nlay, ngpt, nbnd = 100000, 60, 256, 16
g = torch.Generator(device="cuda").manual_seed(1)
def R(*sh):
    t = xp.empty(sh); t.uniform_(0.01, 0.95, generator=g); return t
t1, s1, g1, t2, s2, g2 = (R(ncol, nlay, ngpt) for _ in range(6))
tb, sb, gb = (R(ncol, nlay, nbnd) for _ in range(3))
lims = xp.asarray(np.asfortranarray(np.array([[1 + 16 * b for b in range(nbnd)], [16 * (b + 1) for b in range(nbnd)]], dtype=np.int32)))
GB = ncol * nlay * ngpt * 8 / 1e9
def timed(name, f, gbytes, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{name:42s} {dt * 1e3:7.2f} ms  {gbytes / dt:7.0f} GB/s")
timed("rte_increment_1scalar_by_1scalar", lambda: lib.rte_increment_1scalar_by_1scalar(ncol, nlay, ngpt, t1, t2), 3 * GB)
timed("rte_increment_2stream_by_2stream", lambda: lib.rte_increment_2stream_by_2stream(ncol, nlay, ngpt, t1, s1, g1, t2, s2, g2), 9 * GB)
timed("rte_inc_2stream_by_2stream_bybnd", lambda: lib.rte_inc_2stream_by_2stream_bybnd(ncol, nlay, ngpt, t1, s1, g1, tb, sb, gb, nbnd, lims), 6 * GB + 3 * GB / 16)
timed("rte_delta_scale_2str_k", lambda: lib.rte_delta_scale_2str_k(ncol, nlay, ngpt, t1, s1, g1), 6 * GB)
