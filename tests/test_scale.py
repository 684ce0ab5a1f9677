"""Full-size and sharding checks on the GPU (size-independent properties; the elementwise comparison of the full-size
configurations with the reference kernels is tests/test_fullsize_oracle.py):
BASELINE configs[1] (1e5 columns), the per-GPU shard of configs[4] (1e6 / 8 = 125 000 columns), shard invariance of
the column decomposition, the RCCL reduction under a real (1-rank) nccl group, and the 32-bit-offset guard."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rte_rrtmgp_amd import frontend, hiplib, sharding, synth  # noqa: E402

pytestmark = pytest.mark.gpu
NLAY = 60


def _lw_chain(hip, xp, kd, atm_np, ncol, top_at_1, bufs=None, rb=None):
    A = xp.asarray
    go = frontend.GasOptics(hip, kd, xp)
    b = go.gas_optics_lw(ncol, NLAY, A(atm_np["play"]), A(atm_np["plev"]), A(atm_np["tlay"]), A(atm_np["tsfc"]),
                         A(atm_np["col_gas"]), A(atm_np["tlev"]), top_at_1, buffers=bufs)
    r = frontend.rte_lw(hip, xp, ncol, NLAY, kd.ngpt, top_at_1, b["tau"], b["lay_src"], b["lev_src"],
                        xp.full((ncol, kd.ngpt), 0.98), b["sfc_src"], buffers=rb)
    return b, r


@pytest.mark.parametrize("ncol", [100000, 125000])
def test_full_size_column_tiling_invariance(ncol):
    """The benchmark's own size: 100 distinct seeded columns (an RFMIP-like set of sites) tiled to 1e5 / 125 000
    columns, the chain run exactly as bench.py runs it (device-resident, deferred zero fill).  Columns are independent,
    so every copy must reproduce the 100-column result (reference tests/rte_lw_solver_unit_tests.F90:139-144 does this
    with 8 columns): broadband fluxes to 1e-12 elementwise (floor 1e-6 of the maximum), and to 5e-14 for a strided sample of tau / sources planes."""
    import torch

    hip = hiplib.load()
    xp = frontend.TorchArrays("cuda:0")
    kd = synth.make_kdist("lw")
    tile = 100
    reps = ncol // tile
    atm = synth.make_atmosphere(tile, NLAY, seed=21, kdist=kd)
    base = {k: getattr(atm, k) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas")}
    b1, r1 = _lw_chain(hip, xp, kd, base, tile, atm.top_at_1)
    up1, dn1 = xp.to_numpy(r1["flux_up"]).copy(), xp.to_numpy(r1["flux_dn"]).copy()
    tau1 = b1["tau"].clone()
    lev1 = b1["lev_src"].clone()
    big = {k: np.asfortranarray(np.concatenate([v] * reps, axis=0)) for k, v in base.items()}
    hiplib.ext_call(hip, "rte_hip_defer_zero", ["i"], 1)
    try:
        bN, rN = _lw_chain(hip, xp, kd, big, ncol, atm.top_at_1)
        torch.cuda.synchronize()
    finally:
        hiplib.ext_call(hip, "rte_hip_defer_zero", ["i"], 0)
    for name, small, bigt in (("flux_up", up1, rN["flux_up"]), ("flux_dn", dn1, rN["flux_dn"])):
        t = bigt.reshape(NLAY + 1, reps, tile)  # torch shape (nlev, ncol) -> (nlev, copy, column)
        ref = torch.from_numpy(np.ascontiguousarray(small.T)).to(t.device)[:, None, :]
        err = ((t - ref).abs() / ref.abs().clamp_min(1e-6 * float(ref.abs().max()))).max()
        assert float(err) <= 1e-12, (name, float(err))  # small- and large-batch calls take different gas-optics kernels
        assert bool(torch.isfinite(t).all()) and float(t.min()) >= 0.0
    # optical depths and level sources of g-points spread over the bands: every copy against the 100-column run
    # (production slab kernels vs the small-problem kernels: FMAs in a different association, a few ulp)
    for name, small, bigt, nl in (("tau", tau1, bN["tau"], NLAY), ("lev_src", lev1, bN["lev_src"], NLAY + 1)):
        for g in range(3, kd.ngpt, 37):
            t = bigt[g].reshape(nl, reps, tile)
            ref = small[g][:, None, :]
            err = ((t - ref).abs() / ref.abs().clamp_min(1e-300)).max()
            assert float(err) <= 5e-14, (name, g, float(err))
    torch.cuda.synchronize()


def test_column_shards_reproduce_the_unsharded_run():
    """What each rank of an N-GPU job computes (sharding.shard_columns: contiguous ranges, sizes differing by at most
    one) run here one shard after the other on one GPU: the concatenation equals the unsharded run.  Tiles of the
    production kernels start at the shard boundary, so a column can change between the slab kernel (FMAs) and the
    overflow worklist (reference association): agreement is to rounding (1e-14 elementwise), not bitwise."""
    import torch

    hip = hiplib.load()
    xp = frontend.TorchArrays("cuda:0")
    kd = synth.make_kdist("lw", ngpt=128, nbnd=8)
    ncol = 5003
    atm = synth.make_atmosphere(ncol, NLAY, seed=77, kdist=kd)
    full = {k: getattr(atm, k) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas")}
    _, r = _lw_chain(hip, xp, kd, full, ncol, atm.top_at_1)
    up, dn = xp.to_numpy(r["flux_up"]).copy(), xp.to_numpy(r["flux_dn"]).copy()
    for world in (2, 3):
        parts_up, parts_dn, covered = [], [], 0
        for rank in range(world):
            c0, n = sharding.shard_columns(ncol, rank, world)
            assert c0 == covered
            covered += n
            shard = {k: np.asfortranarray(v[c0:c0 + n]) for k, v in full.items()}
            _, rs = _lw_chain(hip, xp, kd, shard, n, atm.top_at_1)
            parts_up.append(xp.to_numpy(rs["flux_up"]).copy())
            parts_dn.append(xp.to_numpy(rs["flux_dn"]).copy())
        assert covered == ncol
        for whole, parts in ((up, parts_up), (dn, parts_dn)):
            cat = np.concatenate(parts, axis=0)
            assert np.max(np.abs(cat - whole) / np.maximum(np.abs(whole), 1e-300)) <= 1e-14
            assert np.mean(cat == whole) > 0.9  # and bit-identical for almost every value
    torch.cuda.synchronize()


def test_mean_profile_allreduce_under_nccl():
    """The path's only collective, under a real RCCL ("nccl") process group of one rank on this GPU."""
    import torch
    import torch.distributed as dist

    created = False
    if not dist.is_initialized():
        port = 29500 + os.getpid() % 2000
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        created = True
    try:
        g = torch.Generator(device="cuda").manual_seed(3)
        up = torch.rand(NLAY + 1, 777, dtype=torch.float64, device="cuda", generator=g)
        dn = torch.rand(NLAY + 1, 777, dtype=torch.float64, device="cuda", generator=g)
        prof = sharding.allreduce_mean_profile(up, dn, 777)
        assert prof.shape == (2, NLAY + 1)
        assert torch.allclose(prof[0], up.mean(dim=1), rtol=1e-14, atol=0) and torch.allclose(prof[1], dn.mean(dim=1), rtol=1e-14, atol=0)
        gathered = sharding.allgather_fluxes(up, 777)
        assert torch.equal(gathered, up)
    finally:
        if created:
            dist.destroy_process_group()


def test_two_ranks_of_bench_share_one_device_over_gloo():
    """The rank / seed / shard / collective plumbing of `bench.py --gpus N` on the HIP path: two ranks launched as the
    driver launches them (torch.distributed.run, 127.0.0.1), both on this box's one GPU, gloo for the collective (RCCL
    refuses two ranks on one device).  The JSON line must carry the whole-job rate, both ranks' step times and the
    all-reduce time measured inside the step."""
    import json
    import socket
    import subprocess

    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ncol", "5000", "--steps", "2",
           "--warmup", "1", "--dist-backend", "gloo", "--single-device", "--no-cpu-baseline", "--no-plain-abi"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["scaling"] == "weak"
    assert res["config"]["columns_per_gpu"] == 5000 and "10000 columns in all" in res["config"]["workload"]
    assert res["config"]["rccl_world_size"] == 2 and res["config"]["dist_backend"] == "gloo"
    pr = res["per_rank_ms_per_step"]
    assert len(pr["ranks"]) == 2 and all(t > 0 for t in pr["ranks"]) and pr["min"] <= pr["median"] <= pr["max"]
    assert abs(res["ms_per_step"] - pr["max"]) < 1e-3  # the job's step time is its slowest rank's
    assert res["value"] == pytest.approx(10000 * 2 / (res["ms_per_step"] * 2 * 1e-3), rel=1e-3)
    assert res["allreduce_ms_per_step"]["max_over_ranks"] > 0
    assert res["step_ms"]["n"] == 2 and res["step_ms"]["min"] <= res["step_ms"]["median"]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the N > 1 analogue of the driver's 1-GPU command): bench.py starts
    its ranks under torch.distributed.run itself; here both ranks on this box's one GPU over gloo.  And without
    --single-device, two ranks on a one-GPU box: a "skipped" line and status 0, never an assertion."""
    import json
    import subprocess

    import torch

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--dist-backend", "gloo", "--ncol", "4096",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-plain-abi", "--no-factored"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["rccl_world_size"] == 2 and res["config"]["columns_per_gpu"] == 4096
    assert len(res["per_rank_ms_per_step"]["ranks"]) == 2 and res["allreduce_ms_per_step"]["max_over_ranks"] > 0
    assert res["value"] > 0
    want = torch.cuda.device_count() + 1  # one rank more than the box has devices
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == want and res["value"] is None and "visible" in res["skipped"]


def test_c_level_flux_reduction_over_rccl():
    """The exchange step for host programs without torch (csrc/collectives.hip): rte_hip_allreduce_mean_profile and
    rte_hip_allgather_columns on a communicator made with RCCL's own C API -- one rank (RCCL refuses two ranks on this
    box's one device; N ranks differ in the communicator only) -- and the communicator-free single-rank form, against numpy.
    Device and host pointers."""
    import ctypes

    import torch

    hip = hiplib.load()
    ncol, nlev = 70001, 61
    rng = np.random.default_rng(5)
    up, dn = rng.random((nlev, ncol)) * 400.0, rng.random((nlev, ncol)) * 300.0  # Fortran (ncol, nlev)
    ref_up, ref_dn = up.sum(axis=1) / (3.0 * ncol), dn.sum(axis=1) / (3.0 * ncol)  # "a third of a 3-rank domain"
    fn = hip.raw("rte_hip_allreduce_mean_profile")
    fn.restype = ctypes.c_int
    P, I, LL = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong

    def mean(comm, a, b, on_device):
        if on_device:
            ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
            mu, md = torch.empty(nlev, dtype=torch.float64, device="cuda"), torch.empty(nlev, dtype=torch.float64, device="cuda")
            rc = fn(P(comm), I(ncol), I(nlev), P(ta.data_ptr()), P(tb.data_ptr()), LL(3 * ncol), P(mu.data_ptr()), P(md.data_ptr()))
            torch.cuda.synchronize()
            return rc, mu.cpu().numpy(), md.cpu().numpy()
        mu, md = np.empty(nlev), np.empty(nlev)
        rc = fn(P(comm), I(ncol), I(nlev), P(a.ctypes.data), P(b.ctypes.data), LL(3 * ncol), P(mu.ctypes.data), P(md.ctypes.data))
        return rc, mu, md

    for on_device in (True, False):  # no communicator: one rank
        rc, mu, md = mean(None, up, dn, on_device)
        assert rc == 0
        assert np.allclose(mu, ref_up, rtol=1e-13, atol=0) and np.allclose(md, ref_dn, rtol=1e-13, atol=0)
    assert hiplib.ext_call(hip, "rte_hip_rccl_available", []) == 1
    # a one-rank communicator from RCCL's C API (the library resolves the same librccl)
    try:
        rccl = ctypes.CDLL("librccl.so", mode=ctypes.RTLD_GLOBAL)
    except OSError:
        rccl = ctypes.CDLL("librccl.so.1", mode=ctypes.RTLD_GLOBAL)

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid, comm = UniqueId(), ctypes.c_void_p()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        rc, mu, md = mean(comm.value, up, dn, True)
        assert rc == 0
        assert np.allclose(mu, ref_up, rtol=1e-13, atol=0) and np.allclose(md, ref_dn, rtol=1e-13, atol=0)
        ag = hip.raw("rte_hip_allgather_columns")
        ag.restype = ctypes.c_int
        loc = torch.from_numpy(up).cuda()
        glob = torch.zeros_like(loc)
        assert ag(P(comm.value), I(ncol), I(nlev), P(loc.data_ptr()), P(glob.data_ptr())) == 0
        torch.cuda.synchronize()
        assert torch.equal(glob, loc)
        assert ag(P(comm.value), I(ncol), I(nlev), P(up.ctypes.data), P(glob.data_ptr())) == -2  # host memory: refused
        # slabs of unequal width (shard boundaries on multiples of 64 columns): this rank's columns in a wider, agreed slab
        agv = hip.raw("rte_hip_allgatherv_columns")
        agv.restype = ctypes.c_int
        glob.zero_()
        assert agv(P(comm.value), I(ncol), I(nlev), P(loc.data_ptr()), I(ncol + 59), LL(ncol), P(glob.data_ptr())) == 0
        torch.cuda.synchronize()
        assert torch.equal(glob, loc)
        part = torch.full((nlev, ncol + 7), -1.0, dtype=torch.float64, device="cuda")  # "global" field with 7 columns of another rank behind
        assert agv(P(comm.value), I(ncol), I(nlev), P(loc.data_ptr()), I(ncol), LL(ncol + 7), P(part.data_ptr())) == 0
        torch.cuda.synchronize()
        assert torch.equal(part[:, :ncol], loc) and bool((part[:, ncol:] == -1.0).all())
        assert agv(P(comm.value), I(ncol), I(nlev), P(loc.data_ptr()), I(ncol - 1), LL(ncol), P(glob.data_ptr())) == -2  # wider than the slab
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_32bit_offset_guard_falls_back():
    """The segmented solver addresses a g-point plane with 32-bit byte offsets (8 * ncol * (nlay+1) < 2^32).  A call
    beyond that must take the generic kernel, not abort or wrap: 2^24 columns x 32 layers x 1 g-point (13 GB of
    inputs); the first and the last 4096 columns are checked against a small call on the segmented kernel."""
    import torch

    hip = hiplib.load()
    xp = frontend.TorchArrays("cuda:0")
    ncol, nlay, ngpt, sub = 1 << 24, 32, 1, 4096
    assert ncol * (nlay + 1) >= 1 << 29
    g = torch.Generator(device="cuda").manual_seed(11)

    def R(*sh):
        t = xp.empty(sh)
        t.uniform_(0.0, 1.0, generator=g)
        return t

    tau, lay, lev = R(ncol, nlay, ngpt).mul_(2), R(ncol, nlay, ngpt).mul_(10).add_(1), R(ncol, nlay + 1, ngpt).mul_(10).add_(1)
    emis, sfc = R(ncol, ngpt).mul_(0.1).add_(0.9), R(ncol, ngpt).mul_(10)
    rb = frontend.rte_lw(hip, xp, ncol, nlay, ngpt, False, tau, lay, lev, emis, sfc)
    torch.cuda.synchronize()
    for c0 in (0, ncol - sub):
        sl = slice(c0, c0 + sub)
        cut = lambda t: t[..., sl].contiguous()  # noqa: E731  (column is the last torch axis)
        rs = frontend.rte_lw(hip, xp, sub, nlay, ngpt, False, cut(tau), cut(lay), cut(lev), cut(emis), cut(sfc))
        for k in ("flux_up", "flux_dn"):
            a, b = rb[k][..., sl], rs[k]
            assert float(((a - b).abs() / b.abs().clamp_min(1e-300)).max()) <= 1e-13, (k, c0)
    torch.cuda.synchronize()


def test_planck_on_the_side_stream_gives_the_same_arrays():
    """``rte_hip_overlap_planck(1)``: compute_Planck_source runs on a second stream concurrently with the
    compute_tau_absorption call it follows (waiting only for what was queued before that call), and the library stream
    joins it.  Same kernels on the same inputs: every array of the LW chain must be bit-identical to the serial run,
    repeatedly (a missing dependency would show as a race), at a size where both kernels fill the chip."""
    import torch

    hip = hiplib.load()
    xp = frontend.TorchArrays("cuda:0")
    kd = synth.make_kdist("lw")
    ncol = 40000
    atm = synth.make_atmosphere(ncol, NLAY, seed=5, kdist=kd)
    inp = {k: getattr(atm, k) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas")}

    def run(overlap, reps):
        hiplib.ext_call(hip, "rte_hip_overlap_planck", ["i"], overlap)
        hiplib.ext_call(hip, "rte_hip_defer_zero", ["i"], 1)
        try:
            bufs, rb, out = {}, {}, None
            for _ in range(reps):
                b, r = _lw_chain(hip, xp, kd, inp, ncol, atm.top_at_1, bufs=bufs, rb=rb)
                # something of the caller's queued right behind the call, on the library stream, must see the sources
                chk = b["lay_src"].sum() + b["lev_src"].sum() + b["sfc_src"].sum()
                out = {k: b[k].clone() for k in ("tau", "lay_src", "lev_src", "sfc_src", "sfc_src_jac")}
                out["flux_up"], out["flux_dn"], out["chk"] = r["flux_up"].clone(), r["flux_dn"].clone(), chk.clone()
            torch.cuda.synchronize()
            return out
        finally:
            hiplib.ext_call(hip, "rte_hip_overlap_planck", ["i"], 0)
            hiplib.ext_call(hip, "rte_hip_defer_zero", ["i"], 0)

    serial = run(0, 1)
    for _ in range(3):
        forked = run(1, 3)
        for k, v in serial.items():
            assert torch.equal(v, forked[k]), k


def test_planck_on_the_geometry_of_tau_gives_the_same_arrays():
    """``rte_hip_share_geometry(1)``: compute_Planck_source takes the per-(tile, layer) bounding boxes left by the
    compute_tau_absorption call right before it instead of deriving them again.  The boxes only say which table rows
    are staged: every array must be bit-identical to the run without sharing -- on the benchmark atmosphere, and on a
    shuffled site-like one whose wide boxes send work to the direct-gather worklists; a call in between (here: a
    zero_array on an unrelated buffer) must switch the sharing off for that step, not break it."""
    import torch

    hip = hiplib.load()
    xp = frontend.TorchArrays("cuda:0")
    kd = synth.make_kdist("lw")
    ncol = 20000
    for climate, seed in (("rce", 3), ("sites", 4)):
        atm = synth.make_atmosphere(ncol, NLAY, seed=seed, kdist=kd, climate=climate)
        inp = {k: getattr(atm, k) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas")}
        if climate == "sites":  # random column order: every tile spans the whole climatological range
            perm = np.random.default_rng(1).permutation(ncol)
            inp = {k: np.asfortranarray(v[perm]) for k, v in inp.items()}

        def run(share):
            hiplib.ext_call(hip, "rte_hip_share_geometry", ["i"], share)
            try:
                b, r = _lw_chain(hip, xp, kd, inp, ncol, atm.top_at_1)
                torch.cuda.synchronize()
                out = {k: b[k].clone() for k in ("tau", "lay_src", "lev_src", "sfc_src", "sfc_src_jac")}
                out["flux_up"], out["flux_dn"] = r["flux_up"].clone(), r["flux_dn"].clone()
                return out
            finally:
                hiplib.ext_call(hip, "rte_hip_share_geometry", ["i"], 0)

        plain = run(0)
        shared = run(1)
        for k, v in plain.items():
            assert torch.equal(v, shared[k]), (climate, k)
    # a foreign library call between the two: the Planck call must fall back to its own geometry
    hiplib.ext_call(hip, "rte_hip_share_geometry", ["i"], 1)
    try:
        go = frontend.GasOptics(hip, kd, xp)
        A = xp.asarray
        st = go.interpolation(ncol, NLAY, A(inp["play"]), A(inp["tlay"]), A(inp["col_gas"]), None)
        tau = xp.empty((ncol, NLAY, kd.ngpt))
        hip.zero_array_3D(ncol, NLAY, kd.ngpt, tau)
        go.compute_tau_absorption(ncol, NLAY, st, A(inp["play"]), A(inp["tlay"]), A(inp["col_gas"]), tau)
        other = xp.empty((ncol, 4))
        hip.zero_array_2D(ncol, 4, other)  # <- in between
        bufs = [xp.empty((ncol, kd.ngpt)), xp.empty((ncol, NLAY, kd.ngpt)), xp.empty((ncol, NLAY + 1, kd.ngpt)), xp.empty((ncol, kd.ngpt))]
        go.source(ncol, NLAY, st, A(inp["tlay"]), A(inp["tlev"]), A(inp["tsfc"]), atm.top_at_1, bufs[0], bufs[1], bufs[2], bufs[3])
        torch.cuda.synchronize()
        assert torch.equal(bufs[1], plain["lay_src"]) and torch.equal(bufs[2], plain["lev_src"])
    finally:
        hiplib.ext_call(hip, "rte_hip_share_geometry", ["i"], 0)


def test_tau_on_the_masks_of_interpolation_gives_the_same_arrays():
    """``rte_hip_share_geometry(1)``, first half of the step: rrtmgp_interpolation leaves per (256-column block, layer)
    the bit masks of the table rows its columns touch, and the compute_tau_absorption call that directly follows builds
    its tile geometry from them instead of reading the index arrays again.  Masks only say which rows are staged:
    every array and the size of the direct-gather worklist must be those of the run without sharing -- on the benchmark
    atmosphere, on a shuffled site-like one, and on columns whose pressure is not monotone in the layer index (the
    interpolation's masks are keyed by the tropo flag, the geometry kernel's by the layer ranges: there the kernel must
    notice and derive its own)."""
    import torch

    hip = hiplib.load()
    xp = frontend.TorchArrays("cuda:0")
    kd = synth.make_kdist("lw")
    ncol = 20000
    ptrop = float(np.exp(kd.press_ref_trop_log))
    for case, seed in (("rce", 5), ("sites", 6), ("kinked", 7)):
        atm = synth.make_atmosphere(ncol, NLAY, seed=seed, kdist=kd, climate="sites" if case == "sites" else "rce")
        inp = {k: np.array(getattr(atm, k), order="F") for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas")}
        if case == "sites":
            perm = np.random.default_rng(2).permutation(ncol)
            inp = {k: np.asfortranarray(v[perm]) for k, v in inp.items()}
        if case == "kinked":
            # every 7th column: exchange the pressures of the two tropospheric layers nearest the tropopause -- the layer
            # of lowest tropospheric pressure is then not the first tropospheric layer, so that layer lies in neither range
            play = inp["play"]
            trop = play > ptrop
            for c in range(0, ncol, 7):
                idx = np.nonzero(trop[c])[0]
                a, b = (idx[0], idx[1]) if play[c, idx[0]] < play[c, idx[1]] else (idx[-1], idx[-2])
                play[c, a], play[c, b] = play[c, b], play[c, a]
            assert ((play > ptrop) == trop).all()

        def run(share):
            hiplib.ext_call(hip, "rte_hip_share_geometry", ["i"], share)
            hiplib.ext_call(hip, "rte_hip_defer_zero", ["i"], 1)
            try:
                b, r = _lw_chain(hip, xp, kd, inp, ncol, atm.top_at_1)
                torch.cuda.synchronize()
                out = {k: b[k].clone() for k in ("tau", "lay_src", "lev_src", "sfc_src")}
                out["flux_up"], out["flux_dn"] = r["flux_up"].clone(), r["flux_dn"].clone()
                return out, hiplib.ext_call(hip, "rte_hip_stat", ["i"], 0), hiplib.ext_call(hip, "rte_hip_stat", ["i"], 2)
            finally:
                hiplib.ext_call(hip, "rte_hip_defer_zero", ["i"], 0)
                hiplib.ext_call(hip, "rte_hip_share_geometry", ["i"], 0)

        plain, n_plain, src_plain = run(0)
        shared, n_shared, src_shared = run(1)
        assert src_plain == 2
        assert src_shared == (2 if case == "kinked" else 1), case
        assert n_plain == n_shared, (case, n_plain, n_shared)
        for k, v in plain.items():
            assert torch.equal(v, shared[k]), (case, k)


def test_worklist_beside_the_slab_kernel_gives_the_same_arrays():
    """``rte_hip_aux_stream``: the direct-gather worklist of compute_tau_absorption runs on a second stream beside the
    slab kernel (forked after the geometry pre-pass, joined before the call returns) -- or after it on the library
    stream.  The two kernels write disjoint (tile, layer, band) entries: every array must be bit-identical, on an
    atmosphere that actually fills the worklist (shuffled site-like columns), with the LW and the one-pass SW gas optics."""
    import torch

    hip = hiplib.load()
    xp = frontend.TorchArrays("cuda:0")
    ncol = 20000
    A = xp.asarray
    for kind in ("lw", "sw"):
        kd = synth.make_kdist(kind)
        atm = synth.make_atmosphere(ncol, NLAY, seed=8, kdist=kd, climate="sites")
        perm = np.random.default_rng(3).permutation(ncol)
        inp = {k: np.asfortranarray(getattr(atm, k)[perm]) for k in ("play", "plev", "tlay", "tlev", "tsfc", "col_gas", "col_dry")}

        def run(beside):
            hiplib.ext_call(hip, "rte_hip_aux_stream", ["i"], beside)
            try:
                go = frontend.GasOptics(hip, kd, xp)
                if kind == "lw":
                    b = go.gas_optics_lw(ncol, NLAY, A(inp["play"]), A(inp["plev"]), A(inp["tlay"]), A(inp["tsfc"]), A(inp["col_gas"]),
                                         A(inp["tlev"]), atm.top_at_1)
                    keys = ("tau", "lay_src", "lev_src")
                else:
                    b = go.gas_optics_sw(ncol, NLAY, A(inp["play"]), A(inp["plev"]), A(inp["tlay"]), A(inp["col_gas"]), A(inp["col_dry"]),
                                         fuse_rayleigh="all")
                    keys = ("tau", "ssa", "g")
                torch.cuda.synchronize()
                return {k: b[k].clone() for k in keys}, hiplib.ext_call(hip, "rte_hip_stat", ["i"], 0)
            finally:
                hiplib.ext_call(hip, "rte_hip_aux_stream", ["i"], 1)

        after, n_after = run(0)
        beside, n_beside = run(1)
        assert n_after == n_beside and n_after > 0, (kind, n_after, n_beside)
        for k, v in after.items():
            assert torch.equal(v, beside[k]), (kind, k)
