"""CPU-only, world_size 2 over gloo: the sharding and the gather+group-sum of partial commitments.
The per-rank partial is produced by the oracle here (no GPU in this container); on the GPU box the
same code path carries HIP partials (tests/test_gpu_msm.py::test_point_sum, bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q, exchange="rccl"):
    try:
        os.environ["LURK_PARTIALS_EXCHANGE"] = exchange  # "host": the 96-byte partials go through a gloo side group (distributed.py)
        _worker_body(rank, world, port, n, q)
    except Exception as e:  # surface failures instead of hanging the parent
        q.put((rank, repr(e)))
        raise


def _worker_body(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lurk_beta_amd import point_to_affine
    from lurk_beta_amd.distributed import allreduce_commitment, shard_range
    from oracle import coracle as C

    lo, hi = shard_range(n, world, rank)
    B = C.synth_bases(0, hi - lo, first=lo)
    S = C.synth_scalars(1, 1, 1, hi - lo, first=lo)
    partial = C.msm_pippenger(0, B, S, 2)  # stands in for this rank's HIP commit
    full = allreduce_commitment(0, partial)
    q.put((rank, point_to_affine(0, full)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,exchange", [(1001, "rccl"), (4096, "rccl"), (1001, "host")])
def test_sharded_commitment_world2_gloo(n, exchange):
    from oracle import coracle as C

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = C.jac_to_affine(0, C.msm_pippenger(0, C.synth_bases(0, n), C.synth_scalars(1, 1, 1, n)))
    assert got[0] == want and got[1] == want


def test_shard_range_covers_everything():
    from lurk_beta_amd.distributed import shard_range

    for n in (0, 1, 7, 8, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _tree_worker(rank, world, port, log_n, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from lurk_beta_amd.distributed import shard_range, sharded_tree8_root
        from oracle import coracle as C

        n = 1 << log_n
        lo, hi = shard_range(n, world, rank)
        leaves = C.synth_scalars(1, 2, 0, n)[lo:hi]  # this rank's contiguous leaves
        # the oracle stands in for the HIP kernels (no GPU here); the sharding, gather and final hash8 are the code under test
        root = sharded_tree8_root(1, leaves, tree_root=lambda f, lv: C.poseidon_tree8(f, lv),
                                  hash8=lambda f, pre: C.poseidon_batch(f, 8, pre.reshape(-1, 4)))
        q.put((rank, [int(x) for x in np.asarray(root).reshape(4)]))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, repr(e)))
        raise


@pytest.mark.parametrize("log_n", [3, 9])
def test_sharded_poseidon_tree_world2_gloo(log_n):
    from oracle import coracle as C

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tree_worker, args=(r, world, port, log_n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [int(x) for x in np.asarray(C.poseidon_tree8(1, C.synth_scalars(1, 2, 0, 1 << log_n))).reshape(-1)[:4]]
    assert got[0] == want and got[1] == want
