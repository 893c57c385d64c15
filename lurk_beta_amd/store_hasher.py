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


def encode(nodes: list[tuple]):
    """nodes -> (records (n, 8) u32 as ``lurk_hip_store_node`` lays them out, values (m, 4) u64 canonical)."""
    import numpy as np

    n = len(nodes)
    rec = np.zeros((n, 8), dtype=np.uint32)  # kind, tag, child[4], value, reserved
    values: list[int] = []
    for i, nd in enumerate(nodes):
        kind = nd[0]
        rec[i, 0] = _KIND[kind]
        if kind == "atom":
            rec[i, 1] = nd[1]
            rec[i, 6] = len(values)
            values.append(int(nd[2]))
        elif kind == "comm":
            rec[i, 1] = 8  # ExprTag::Comm
            rec[i, 2] = nd[2]
            rec[i, 6] = len(values)
            values.append(int(nd[1]))
        else:
            rec[i, 1] = nd[1]
            for k, c in enumerate(nd[2:]):
                rec[i, 2 + k] = c
    vals = np.zeros((max(len(values), 1), 4), dtype=np.uint64)
    for k, v in enumerate(values):
        for w in range(4):
            vals[k, w] = (v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return rec, vals


def hydrate_records(field_id: int, rec, vals):
    """``lurk_hip_store_hydrate`` on encoded records: (digests (n, 4) u64 canonical, levels)."""
    import ctypes

    import numpy as np

    from . import _lib

    out = np.zeros((rec.shape[0], 4), dtype=np.uint64)
    levels = ctypes.c_size_t()
    _lib.check(_lib.load().lurk_hip_store_hydrate(field_id, _lib.ptr(rec), rec.shape[0], _lib.ptr(vals), vals.shape[0], _lib.ptr(out), ctypes.byref(levels)))
    return out, levels.value


# ---- synthetic DAGs of the two shapes that matter (tests and bench.py --workload store_hydrate) ----
_TAG_NIL, _TAG_CONS, _TAG_SYM, _TAG_THUNK, _TAG_STR, _TAG_CHAR, _TAG_ENV = 0, 1, 2, 5, 6, 7, 12  # ExprTag discriminants (src/tag.rs), as oracle/pyref.py


def _string_nodes(nodes, s):
    nodes.append(("atom", _TAG_STR, 0))
    cur = len(nodes) - 1
    for ch in reversed(s):
        nodes.append(("atom", _TAG_CHAR, ord(ch)))
        nodes.append(("tuple2", _TAG_STR, len(nodes) - 1, cur))
        cur = len(nodes) - 1
    return cur


def _symbol_nodes(nodes, path, tag=_TAG_SYM):
    nodes.append(("atom", _TAG_SYM, 0))
    cur = len(nodes) - 1
    for k, name in enumerate(path):
        sn = _string_nodes(nodes, name)
        nodes.append(("tuple2", tag if k == len(path) - 1 else _TAG_SYM, sn, cur))
        cur = len(nodes) - 1
    return cur


def list_dag(k: int = 400) -> list[tuple]:
    """DEEP: a proper list of k distinct symbols (each a string-hashing chain ~10 levels deep, k of them side by side), the list's
    spine of k conses (one node per level), a thunk, an env binding and a commitment on top: the shape on which a level-synchronous
    device hasher pays one dependency chain per spine cell."""
    big: list[tuple] = []
    syms = [_symbol_nodes(big, ["lurk", "user", "sym%d" % j]) for j in range(k)]
    big.append(("atom", _TAG_NIL, 0))
    lst = len(big) - 1
    for sy in reversed(syms):
        big.append(("tuple2", _TAG_CONS, sy, lst))
        lst = len(big) - 1
    big.append(("tuple3", _TAG_THUNK, syms[0], syms[1], lst))
    big.append(("compact", _TAG_ENV, syms[2], syms[3], lst))
    big.append(("comm", 12345, len(big) - 2))
    return big


def wide_dag(k: int = 12000) -> list[tuple]:
    """WIDE: k distinct symbols under a balanced binary tree of conses: ~10^5.6 nodes at k = 12 000, ~30 levels."""
    big: list[tuple] = []
    cur = [_symbol_nodes(big, ["lurk", "user", "s%d" % j]) for j in range(k)]
    while len(cur) > 1:
        nxt = []
        for j in range(0, len(cur) - 1, 2):
            big.append(("tuple2", _TAG_CONS, cur[j], cur[j + 1]))
            nxt.append(len(big) - 1)
        if len(cur) & 1:
            nxt.append(cur[-1])
        cur = nxt
    big.append(("comm", 777, cur[0]))
    return big


def hydrate_device(field_id: int, nodes: list[tuple]) -> list[int]:
    """The same DAG through ``lurk_hip_store_hydrate``: wide levels hashed on the device, narrow ones by the library's host Poseidon."""
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
