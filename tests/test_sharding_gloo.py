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

    for n in (1, 7, 100000, 1000003):
        for w in (1, 2, 4, 8):
            parts = [sharding.shard_columns(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            for (s0, c0), (s1, _) in zip(parts, parts[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


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
