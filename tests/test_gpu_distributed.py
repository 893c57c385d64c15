"""N > 1 on the GPU box.  The box has one MI355X, so the ranks share GPU 0 and talk over gloo: what is exercised is
the real code path of a rank - its HIP commit of its own key slice, the all_gather of the 96-byte partials, the group
sum - and bench.py's own launcher (`--gpus N` with no WORLD_SIZE in the environment)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    try:
        import torch
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from lurk_beta_amd import point_to_affine
        from lurk_beta_amd.distributed import ShardedCommitmentKey, shard_range
        from oracle import coracle as C

        lo, hi = shard_range(n, world, rank)
        B = C.synth_bases(0, hi - lo, first=lo)
        S = C.synth_scalars(1, 1, 1, hi - lo, first=lo)
        ck = ShardedCommitmentKey(0, B, precompute=bool(rank))  # rank 0 plain key, rank 1 table key: partials must still add up
        full = ck.commit(S)                                      # HIP partial on this rank, gather, group sum
        q.put((rank, point_to_affine(0, full)))
        ck.close()
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
        raise


@pytest.mark.parametrize("n", [1001, 1 << 15])
def test_sharded_commitment_world2_hip_partials(hip, n):
    import torch.multiprocessing as mp

    from oracle import coracle as C

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = C.jac_to_affine(0, C.msm_pippenger(0, C.synth_bases(0, n), C.synth_scalars(1, 1, 1, n)))
    assert got[0] == want and got[1] == want


def test_bench_spawns_its_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment must run TWO ranks and say so."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--verify", "--log-n", "16",
                          "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--pmc", "off"], capture_output=True, text=True, timeout=900,
                         env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["verified"] is True and rec["config"]["total_points"] == 2 << 16
    assert rec["scaling"] == "weak" and rec["steps"] == 3


def test_bench_refuses_a_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stdout + out.stderr)


def _nccl_worker(port, n, q):
    try:
        import torch
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from lurk_beta_amd import point_to_affine
        from lurk_beta_amd.distributed import ShardedCommitmentKey, _all_gather_rows, gather_partials, sharded_tree8_root
        from oracle import coracle as C

        B = C.synth_bases(0, n)
        S = C.synth_scalars(1, 1, 1, n)
        ck = ShardedCommitmentKey(0, B, precompute=True)
        part = ck.ck.commit(S)
        gathered = gather_partials(part)          # the RCCL branch: device tensor in, all_gather, host array out
        full = ck.commit(S)
        leaves = C.synth_scalars(1, 2, 0, 8 ** 4)
        root = sharded_tree8_root(1, leaves)
        rows = _all_gather_rows(leaves[:8])       # the tree path's gather of subtree roots on the same backend (world 1: itself)
        q.put(("ok", gathered.shape, bool((gathered[0] == part).all()) and bool((rows == leaves[:8]).all()), point_to_affine(0, full),
               [int(x) for x in np.asarray(root).reshape(-1)]))
        ck.close()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put(("error", repr(e)))
        raise


def test_rccl_gather_branch_world1(hip):
    """The `nccl` (= RCCL) branch of gather_partials / the sharded tree has to have EXECUTED on the hardware it is written for: one rank,
    one GPU, world size 1 - the all_gather is a device-to-device copy through RCCL, the code path (device tensors, stream ordering, the
    host round trip of the 96-byte partial) is the one 8 ranks take."""
    import torch.multiprocessing as mp

    from oracle import coracle as C

    n = 5000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), n, q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert res[0] == "ok", res
    assert p.exitcode == 0
    assert res[1] == (1, 12) and res[2]
    assert res[3] == C.jac_to_affine(0, C.msm_pippenger(0, C.synth_bases(0, n), C.synth_scalars(1, 1, 1, n)))
    want_root = C.poseidon_tree8(1, C.synth_scalars(1, 2, 0, 8 ** 4))
    assert res[4] == [int(x) for x in np.asarray(want_root).reshape(-1)]
