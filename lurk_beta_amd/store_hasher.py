"""Level-synchronous store hydration over the HIP batch hasher.

Mirrors the hashing side of lurk-beta's content-addressed store: ``StoreHasher`` preimage layouts
(/root/reference/src/lem/store.rs:29-78) and ``StoreCore::hydrate_z_cache`` (/root/reference/src/lem/
store_core.rs:256-269), which walks the DAG recursively and hashes node by node.  The device wants
batches, so nodes are hashed level by level: every node whose children are already digested goes
into one ``lurk_hip_poseidon_batch`` call per arity.  Nodes are (tag, kind, children) tuples:

  ("atom", tag, value)                         digest = value               (store_core.rs:204)
  ("tuple2", tag, a, b)                        hash4(tag_a,h_a,tag_b,h_b)   (store.rs:32-36)
  ("tuple3", tag, a, b, c)                     hash6(...)                   (store.rs:37-49)
  ("tuple4", tag, a, b, c, d)                  hash8(...)                   (store.rs:50-65)
  ("compact", tag, a, b, c)                    hash4(h_a, tag_b, h_b, h_c)  (store.rs:75-77)
  ("comm", secret, a)                          hash3(secret, tag_a, h_a)    (store.rs:70-73)
"""
from __future__ import annotations

from .poseidon import PoseidonCache


def hydrate(cache: PoseidonCache, nodes: list[tuple]) -> list[int]:
    """nodes[i] may only reference children with a smaller index.  Returns the digest of every node."""
    n = len(nodes)
    digest: list[int | None] = [None] * n
    tag = [0] * n
    level = [0] * n
    for i, nd in enumerate(nodes):
        kind = nd[0]
        if kind == "atom":
            tag[i], digest[i] = nd[1], nd[2]
        elif kind == "comm":
            tag[i] = 8  # ExprTag::Comm
            level[i] = level[nd[2]] + 1
        else:
            tag[i] = nd[1]
            level[i] = 1 + max(level[c] for c in nd[2:])
    for lv in range(1, max(level, default=0) + 1):
        by_arity: dict[int, list[tuple[int, list[int]]]] = {}
        for i, nd in enumerate(nodes):
            if level[i] != lv:
                continue
            kind = nd[0]
            if kind in ("tuple2", "tuple3", "tuple4"):
                pre = [x for c in nd[2:] for x in (tag[c], digest[c])]
            elif kind == "compact":
                a, b, c = nd[2:]
                pre = [digest[a], tag[b], digest[b], digest[c]]
            elif kind == "comm":
                pre = [nd[1], tag[nd[2]], digest[nd[2]]]
            else:
                raise ValueError(kind)
            by_arity.setdefault(len(pre), []).append((i, pre))
        for arity, items in by_arity.items():
            outs = cache.hash_many(arity, [p for _, p in items])
            for (i, _), d in zip(items, outs):
                digest[i] = d
    return digest  # type: ignore[return-value]


_KIND = {"atom": 0, "tuple2": 2, "tuple3": 3, "tuple4": 4, "compact": 5, "comm": 6}


def hydrate_device(field_id: int, nodes: list[tuple]) -> list[int]:
    """The same DAG through ``lurk_hip_store_hydrate``: every level hashed on the device, one copy back at the end."""
    import ctypes

    import numpy as np

    from . import _lib

    lib = _lib.load()
    n = len(nodes)
    rec = np.zeros((n, 8), dtype=np.uint32)  # kind, tag, child[4], value, reserved
    values: list[int] = []
    tag_of = [0] * n
    for i, nd in enumerate(nodes):
        kind = nd[0]
        rec[i, 0] = _KIND[kind]
        if kind == "atom":
            tag_of[i] = nd[1]
            rec[i, 6] = len(values)
            values.append(int(nd[2]))
        elif kind == "comm":
            tag_of[i] = 8  # ExprTag::Comm
            rec[i, 2] = nd[2]
            rec[i, 6] = len(values)
            values.append(int(nd[1]))
        else:
            tag_of[i] = nd[1]
            for k, c in enumerate(nd[2:]):
                rec[i, 2 + k] = c
        rec[i, 1] = tag_of[i]
    vals = np.zeros((max(len(values), 1), 4), dtype=np.uint64)
    for k, v in enumerate(values):
        for w in range(4):
            vals[k, w] = (v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    out = np.zeros((n, 4), dtype=np.uint64)
    levels = ctypes.c_size_t()
    _lib.check(lib.lurk_hip_store_hydrate(field_id, _lib.ptr(rec), n, _lib.ptr(vals), len(values), _lib.ptr(out), ctypes.byref(levels)))
    hydrate_device.last_levels = levels.value
    return [int(out[i, 0]) | int(out[i, 1]) << 64 | int(out[i, 2]) << 128 | int(out[i, 3]) << 192 for i in range(n)]
