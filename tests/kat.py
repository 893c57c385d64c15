"""Shared KAT recipes: each golden value of tests/golden/bn254_poseidon_kats.json expressed as a
list of (arity, preimage) Poseidon calls over BN254 Fr, so the same recipe can be replayed through
the oracle (CPU) and through the HIP library (GPU)."""
import json
import os

from oracle import pyref as R

BN = 2  # field id of BN254 Fr

with open(os.path.join(os.path.dirname(__file__), "golden", "bn254_poseidon_kats.json")) as f:
    GOLDEN = {k: v["value"] for k, v in json.load(f).items() if not k.startswith("_")}


def golden_int(name):
    return int(GOLDEN[name], 16)


def compute_all(hash_fn):
    """hash_fn(preimage: list[int]) -> int, over BN254 Fr.  Returns {kat name: value}."""
    H = hash_fn

    def hstr(s):
        h = 0
        for ch in reversed(s):
            h = H([R.TAG_CHAR, ord(ch), R.TAG_STR, h])
        return h

    def hsym(path):
        h = 0
        for name in path:
            h = H([R.TAG_STR, hstr(name), R.TAG_SYM, h])
        return h

    out = {}
    roots = []
    cur = 0
    for _ in range(85):
        cur = H([cur] * 8)
        roots.append(cur)
    out["hash8_zeros"] = roots[0]
    out["empty_root_2"] = roots[1]
    out["empty_root_3"] = roots[2]
    out["empty_root_4"] = roots[3]
    out["empty_root_85"] = roots[84]
    # insert(123 -> 456) into the empty StandardTrie (trie/mod.rs:745-800)
    empty = [0] + roots
    cur = 456
    for level, digit in enumerate(reversed(R.trie_path(BN, 123, 85))):
        pre = [empty[level]] * 8
        pre[digit] = cur
        cur = H(pre)
    out["trie_insert_123_456"] = cur
    out["commit_num_0"] = H([0, R.TAG_NUM, 0])
    out["commit_123"] = H([0, R.TAG_NUM, 123])
    nil = hsym(["lurk", "nil"])
    out["commit_nil"] = H([0, R.TAG_NIL, nil])
    x = hsym(["lurk", "user", "x"])
    args = H([R.TAG_SYM, x, R.TAG_NIL, nil])  # (x) = cons(x, nil)
    fun = H([R.TAG_CONS, args, R.TAG_SYM, x, R.TAG_ENV, 0, R.TAG_NIL, 0])  # store.rs:623-626
    out["commit_lambda_x_x"] = H([0, R.TAG_FUN, fun])
    # a zero-argument function: vars = nil, body = nil (src/lem/eval.rs:1178-1181; eval_tests.rs:461)
    fun0 = H([R.TAG_NIL, nil, R.TAG_NIL, nil, R.TAG_ENV, 0, R.TAG_NIL, 0])
    out["commit_lambda_noargs_nil"] = H([0, R.TAG_FUN, fun0])
    # the REPL's own examples: '(13 . 21) committed without and with a secret (src/cli/repl/meta_cmd.rs:246-247, 262-264)
    cons = H([R.TAG_NUM, 13, R.TAG_NUM, 21])
    out["commit_cons_13_21"] = H([0, R.TAG_CONS, cons])
    out["hide_12345_cons_13_21"] = H([12345, R.TAG_CONS, cons])
    return out
