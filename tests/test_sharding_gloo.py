"""Multi-process CPU test of the N>1 path (gloo, world_size 2): columns sharded across ranks, each
rank runs the same kernel chain on its shard (through the C oracle here -- no GPU in this
container), the domain-mean profile is all-reduced and the field all-gathered; both must equal
the single-process result."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCOL, NLAY = 37, 12


def _fluxes(first, count):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    from rte_rrtmgp_amd import frontend, synth

    kd = synth.make_kdist("lw", ngpt=32, nbnd=4, nminor_lower=6, nminor_upper=4)
    atm = synth.make_atmosphere(NCOL, NLAY, seed=3, kdist=kd)
    sl = slice(first, first + count)
    xp = frontend.NumpyArrays()
    lib = O.load_c()
    go = frontend.GasOptics(lib, kd, xp)
    F = synth.F
    b = go.gas_optics_lw(count, NLAY, F(atm.play[sl]), F(atm.plev[sl]), F(atm.tlay[sl]), F(atm.tsfc[sl]),
                         F(atm.col_gas[sl]), F(atm.tlev[sl]), atm.top_at_1)
    r = frontend.rte_lw(lib, xp, count, NLAY, kd.ngpt, atm.top_at_1, b["tau"], b["lay_src"], b["lev_src"],
                        xp.full((count, kd.ngpt), 0.98), b["sfc_src"])
    return r["flux_up"], r["flux_dn"]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from rte_rrtmgp_amd import sharding

    first, count = sharding.shard_columns(NCOL, rank, world)
    up, dn = _fluxes(first, count)
    tu, td = torch.from_numpy(np.ascontiguousarray(up.T)), torch.from_numpy(np.ascontiguousarray(dn.T))
    mean = sharding.allreduce_mean_profile(tu, td, NCOL)
    full_up = sharding.allgather_fluxes(tu, NCOL)
    if rank == 0:
        np.save(out + "_mean.npy", mean.numpy())
        np.save(out + "_up.npy", full_up.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_columns_partition():
    sys.path.insert(0, ROOT)
    from rte_rrtmgp_amd import sharding

    for n in (1, 7, 100000, 1000000, 1000003):
        for w in (1, 2, 4, 8):
            for align in (None, 1, 64):
                parts = [sharding.shard_columns(n, r, w, align) for r in range(w)]
                assert parts[0][0] == 0 and sum(c for _, c in parts) == n
                for (s0, c0), (s1, _) in zip(parts, parts[1:]):
                    assert s0 + c0 == s1
                a = align if align else (64 if n >= 1024 * w else 1)  # the default: 64-column boundaries for shards of >= 1024 columns
                assert max(c for _, c in parts) - min(c for _, c in parts) < max(2, 2 * a)  # (one block, and the last block may be short)
                assert all(s0 % a == 0 or (s0 == n and c0 == 0) for s0, c0 in parts)  # (ranks past the end: empty)
    # BASELINE configs[4]: 1e6 columns on 8 ranks -- no rank gets the 125 000 columns whose rows are 64-byte aligned only
    assert [sharding.shard_columns(1000000, r, 8)[1] for r in range(8)] == [125056] + [124992] * 7


def test_two_rank_gloo_matches_single_process(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r0")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    up, dn = _fluxes(0, NCOL)
    mean = np.load(out + "_mean.npy")
    full_up = np.load(out + "_up.npy").T
    assert np.allclose(mean[0], up.mean(axis=0), rtol=1e-13)
    assert np.allclose(mean[1], dn.mean(axis=0), rtol=1e-13)
    assert np.array_equal(full_up, up)  # per-column results do not depend on the sharding


def _worker_uneven(rank, world, port, out, ncol):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from rte_rrtmgp_amd import sharding

    nlev = 5
    whole = torch.arange(nlev * ncol, dtype=torch.float64).reshape(nlev, ncol) * 0.25 + 1.0  # torch shape of Fortran (ncol, nlev)
    first, count = sharding.shard_columns(ncol, rank, world)
    mine = whole[:, first:first + count].contiguous()
    mean = sharding.allreduce_mean_profile(mine, 2.0 * mine, ncol)
    full = sharding.allgather_fluxes(mine, ncol)
    ok = bool(torch.equal(full, whole)) and bool(torch.allclose(mean[0], whole.mean(dim=1), rtol=1e-14, atol=0)) \
        and bool(torch.allclose(mean[1], 2.0 * whole.mean(dim=1), rtol=1e-14, atol=0))
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag)  # every rank must have assembled the same field
    if rank == 0:
        np.save(out + f"_{ncol}.npy", np.array([int(flag), count]))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_gloo_uneven_widths(tmp_path):
    """The 8-GPU job's exchange step on CPU: 8 gloo ranks, column counts that do not divide by 8 (shards of different
    widths, incl. fewer columns than ranks: empty shards), all-gather of the slabs and all-reduce of the mean profile."""
    out = str(tmp_path / "r")
    for ncol in (1003, 13, 5):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_worker_uneven, args=(8, port, out, ncol), nprocs=8, join=True)
        got = np.load(out + f"_{ncol}.npy")
        assert got[0] == 8, f"ncol={ncol}: {8 - got[0]} rank(s) assembled a different field"


def test_bench_without_a_gpu_or_with_too_few_prints_a_skipped_line():
    """`python bench.py --gpus 8` on a box with fewer devices (here: none) must print one "skipped" JSON line and exit 0."""
    import json
    import subprocess

    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 8 and res["value"] is None and res["skipped"]
