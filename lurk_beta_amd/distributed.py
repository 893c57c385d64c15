"""Multi-GPU layer for the commitment path: one process per GPU, points sharded across ranks.

A Pedersen commitment is linear in its (scalar, base) pairs, so a key of n points is cut into
`world` contiguous slices; each rank keeps its slice of the key resident in its own HBM and commits to
its slice of the witness.  The only exchange is the gather of the 96-byte partial commitments
(RCCL all_gather over xGMI when the group backend is "nccl"; "gloo" on CPU for tests) followed by
the group sum on every rank - never a reduction of bucket arrays (RCCL cannot
reduce elliptic-curve points, and bucket arrays are ~100 MB per GPU).

The dense arity-8 Poseidon tree shards the same way (sharded_tree8_root): the 8 subtrees below the root are
dealt to the ranks, each rank reduces its subtrees to their roots on its own GPU, the 8 x 32-byte roots are
all-gathered and every rank hashes them once more.

Folding steps themselves do not shard: each prove_step mutates the running instance
(/root/reference/src/proof/nova.rs:282-295); what shards is the work inside one commitment."""
from __future__ import annotations

import numpy as np

from .msm import CommitmentKey, point_sum_gathered


_GATHER_BUFS: dict = {}
_SIDE_GROUPS: dict = {}


def _host_side_group(group):
    """A gloo group over the same ranks (created collectively at the first exchange): LURK_PARTIALS_EXCHANGE=host sends the 96-byte partials
    through it instead of host -> device -> RCCL -> host (the collective then costs no device round trip; the
    default stays the RCCL exchange the path is specified with)."""
    import torch.distributed as dist

    key = id(group)
    if key not in _SIDE_GROUPS:
        ranks = dist.get_process_group_ranks(group) if group is not None else None
        _SIDE_GROUPS[key] = dist.new_group(ranks=ranks, backend="gloo")
    return _SIDE_GROUPS[key]


def shard_range(n_total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous slice [lo, hi) of rank `rank`; the first n_total % world ranks get one extra point."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_partials(partial: np.ndarray, group=None) -> np.ndarray:
    """all_gather of one 96-byte Jacobian point per rank -> (world, 12) uint64 on every rank."""
    import torch
    import torch.distributed as dist

    import os

    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    if os.environ.get("LURK_PARTIALS_EXCHANGE", "rccl") == "host":
        group, backend = _host_side_group(group), "gloo"
    if backend == "nccl":
        # RCCL moves device buffers: one pinned staging row in, ONE collective into a resident (world, 12) tensor, one copy out -
        # the buffers are kept per (group, world) so that a commitment allocates nothing (a step makes two of these exchanges)
        key = (id(group), world)
        bufs = _GATHER_BUFS.get(key)
        if bufs is None:
            bufs = (torch.empty(12, dtype=torch.int64).pin_memory(), torch.empty(12, dtype=torch.int64, device="cuda"),
                    torch.empty((world, 12), dtype=torch.int64, device="cuda"), torch.empty((world, 12), dtype=torch.int64).pin_memory())
            _GATHER_BUFS[key] = bufs
        h_in, d_in, d_out, h_out = bufs
        h_in.numpy()[:] = np.ascontiguousarray(partial, dtype=np.uint64).view(np.int64)
        d_in.copy_(h_in, non_blocking=True)
        dist.all_gather_into_tensor(d_out, d_in, group=group)
        h_out.copy_(d_out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return h_out.numpy().view(np.uint64).copy()
    t = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.uint64).view(np.int64).copy())
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return torch.stack(out).numpy().view(np.uint64)


def allreduce_commitment(curve: int, partial: np.ndarray, group=None) -> np.ndarray:
    """Group-sum of the per-rank partial commitments; every rank gets the full commitment."""
    return point_sum_gathered(curve, gather_partials(partial, group))  # the ABI's one call for this exchange


class ShardedCommitmentKey:
    """Rank-local slice of a commitment key.  `bases` is this rank's slice (host array, or a device
    tensor with device=True); commit() takes this rank's slice of the scalar vector."""

    def __init__(self, curve: int, bases, n_local: int | None = None, group=None, **kw):
        self.curve = curve
        self.group = group
        self.ck = CommitmentKey(curve, bases, n=n_local, **kw)

    def commit(self, scalars, is_mont: bool = False) -> np.ndarray:
        return allreduce_commitment(self.curve, self.ck.commit(scalars, is_mont), self.group)

    def commit_device(self, d_scalars, n: int, is_mont: bool = False, stream=None) -> np.ndarray:
        return allreduce_commitment(self.curve, self.ck.commit_device(d_scalars, n, is_mont, stream), self.group)

    def close(self):
        self.ck.close()


def _all_gather_rows(rows: np.ndarray, group=None) -> np.ndarray:
    """all_gather of a small (k, 4) uint64 array per rank -> (world * k, 4), rank order."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.uint64).view(np.int64).copy())
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return torch.cat(out).cpu().numpy().view(np.uint64).reshape(-1, 4)


def sharded_tree8_root(field_id: int, local_leaves: np.ndarray, group=None, tree_root=None, hash8=None) -> np.ndarray:
    """Root of the dense arity-8 Poseidon tree whose 8^h leaves are dealt contiguously to the ranks
    (world in {1, 2, 4, 8}: rank r holds the 8/world subtrees number r*8/world ...).  One exchange of
    8 x 32 bytes.  `tree_root(field, leaves) -> root` and
    `hash8(field, preimages(k,8,4)) -> digests(k,4)` default to the HIP kernels; the CPU tests inject
    the oracle."""
    import torch.distributed as dist

    from . import poseidon as _p

    tree_root = tree_root or _p.poseidon_tree8
    hash8 = hash8 or (lambda f, pre: _p.poseidon_batch(f, 8, pre))
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world not in (1, 2, 4, 8):
        raise ValueError("an arity-8 tree shards over 1, 2, 4 or 8 ranks")
    lv = np.ascontiguousarray(local_leaves, dtype=np.uint64).reshape(-1, 4)
    per_rank = 8 // world
    n_sub, rem = divmod(lv.shape[0], per_rank)
    if rem or n_sub < 1 or (n_sub & (n_sub - 1)) or (n_sub.bit_length() - 1) % 3:
        raise ValueError("each rank must hold 8/world subtrees of 8^k leaves")
    if n_sub == 1:  # the leaves are the root's children themselves
        roots = lv
    else:
        roots = np.stack([np.asarray(tree_root(field_id, lv[i * n_sub:(i + 1) * n_sub])).reshape(4) for i in range(per_rank)])
    if world > 1:
        roots = _all_gather_rows(roots, group)
    return np.asarray(hash8(field_id, roots.reshape(1, 8, 4))).reshape(4)
