"""Host-side mirror of lurk-beta's sparse Poseidon trie over the HIP hasher.

Mirrors ``coprocessor::trie::Trie<F, ARITY, HEIGHT>`` (/root/reference/src/coprocessor/trie/mod.rs):
``StandardTrie`` = arity 8, height 85 (:43); ``init_empty`` (:464-481), ``path`` (:589-608),
``lookup`` (:635-652), ``insert`` (:745-800).  Every node hash is ``hash8`` on the GPU through
``PoseidonCache``; the child map (hash -> preimage) is the reference's ``children``/inverse cache
(:453-458)."""
from __future__ import annotations

from .poseidon import PoseidonCache

NUM_BITS = {0: 255, 1: 255, 2: 254}


class Trie:
    def __init__(self, field_id: int, height: int = 85, arity: int = 8, cache: PoseidonCache | None = None):
        assert arity == 8, "lurk-beta instantiates the trie with arity 8 only"
        self.field_id, self.height, self.arity = field_id, height, arity
        self.cache = cache or PoseidonCache(field_id)
        self.children: dict[int, tuple[int, ...]] = {}
        self.empty_roots: list[int] = []
        self._init_empty()
        self.root = self.empty_roots[height - 1] if height else 0

    # -- init_empty (:464-481): empty_roots[0] = hash8([0;8]), empty_roots[i] = hash8([empty_roots[i-1];8])
    def _init_empty(self):
        cur = 0
        for _ in range(self.height):
            cur = self._register([cur] * self.arity)
            self.empty_roots.append(cur)

    def _register(self, preimage) -> int:
        h = self.cache.compute_hash(list(preimage))
        self.children[h] = tuple(preimage)
        return h

    def empty_root_for_height(self, h: int) -> int:
        return 0 if h == 0 else self.empty_roots[h - 1]

    def leaves(self) -> int:
        return self.arity ** self.height

    # -- path (:589-608): MSB-first bits, keep the last 3*H bits, 3-bit big-endian digits
    def path(self, key: int) -> list[int]:
        nbits = NUM_BITS[self.field_id]
        be = [(key >> i) & 1 for i in reversed(range(nbits))]
        need = 3 * self.height
        if need > len(be):
            be = [0] * (need - len(be)) + be
        tail = be[len(be) - need:]
        return [tail[i] << 2 | tail[i + 1] << 1 | tail[i + 2] for i in range(0, need, 3)]

    def _preimages_along(self, path: list[int]) -> list[tuple[int, ...]]:
        """Preimage of every node from the root down the path (an absent subtree is the empty one)."""
        out, node = [], self.root
        for level, digit in enumerate(path):
            pre = self.children.get(node)
            if pre is None:  # empty subtree of height (height - level)
                sub = self.empty_root_for_height(self.height - level - 1)
                pre = (sub,) * self.arity
            out.append(pre)
            node = pre[digit]
        return out

    # -- lookup (:635-652): the leaf-level preimage entry is the payload; 0 = absent
    def lookup(self, key: int):
        path = self.path(key)
        pres = self._preimages_along(path)
        v = pres[-1][path[-1]]
        return None if v == 0 else v

    # -- insert (:745-800): replace the entry, re-hash bottom-up
    def insert(self, key: int, value: int) -> bool:
        path = self.path(key)
        pres = self._preimages_along(path)
        existed = pres[-1][path[-1]] != 0
        cur = value
        for pre, digit in zip(reversed(pres), reversed(path)):
            new = list(pre)
            new[digit] = cur
            cur = self._register(new)
        self.root = cur
        return existed
