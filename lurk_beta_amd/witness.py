"""Host-side mirror of the slot-witness half of ``MultiFrame`` witness generation over the HIP library.

Reference: ``generate_slots_witnesses`` (/root/reference/src/lem/multiframe.rs:520-592) builds, per slot, the aux block
``allocate_slot`` (/root/reference/src/lem/circuit.rs:242-315) would allocate; ``synthesize_frame`` then opens every frame's
witness with its slots' blocks in (hash4, hash6, hash8, commitment, bit_decomp) order (circuit.rs:1429-1451) and the frames
are concatenated after the globals (multiframe.rs:699-702).  Here the blocks are computed on the GPU straight into the
device-resident witness vector ``W`` (``lurk_hip_slot_witness_dev``); only the globals and the non-slot remainder of each
frame - what the CPU circuit synthesis still produces - are copied in (``lurk_hip_witness_blocks_dev``).  All values are
32-byte Montgomery field elements, the in-memory form arecibo commits to."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

SLOT_BIT_DECOMP, SLOT_COMMITMENT, SLOT_HASH4, SLOT_HASH6, SLOT_HASH8 = 1, 3, 4, 6, 8
# the order generate_slots_witnesses walks a frame's hints in (multiframe.rs:527-536)
SLOT_ORDER = (("hash4", SLOT_HASH4), ("hash6", SLOT_HASH6), ("hash8", SLOT_HASH8), ("commitment", SLOT_COMMITMENT), ("bit_decomp", SLOT_BIT_DECOMP))
# eval_step's slot counts (/root/reference/src/lem/eval.rs:1960-1964)
STEP_SLOT_COUNTS = {"hash4": 14, "hash6": 0, "hash8": 6, "commitment": 1, "bit_decomp": 3}


def slot_preimage_len(slot_type: int) -> int:
    return 1 if slot_type == SLOT_BIT_DECOMP else slot_type


def slot_witness_size(field_id: int, slot_type: int) -> int:
    """``compute_witness_size`` (multiframe.rs:503-516).  Host computation: needs no GPU."""
    out = ctypes.c_size_t()
    _lib.check(_lib.load().lurk_hip_slot_witness_size(field_id, slot_type, ctypes.byref(out)))
    return out.value


def slot_witness(field_id: int, slot_type: int, preimages: np.ndarray, mont: bool = False) -> np.ndarray:
    """Host arrays in and out: (n, arity | 1, 4) u64 preimages -> (n, size, 4) u64 Montgomery blocks."""
    lib = _lib.load()
    pre = np.ascontiguousarray(preimages, dtype=np.uint64)
    n = pre.size // (4 * slot_preimage_len(slot_type))
    out = np.zeros((n, slot_witness_size(field_id, slot_type), 4), dtype=np.uint64)
    _lib.check(lib.lurk_hip_slot_witness(field_id, slot_type, _lib.ptr(pre), n, int(mont), _lib.ptr(out)))
    return out


class MultiFrameWitness:
    """Layout of one MultiFrame's aux vector W = [globals | frame 0 | ... | frame rc-1], frame = [slot blocks | body], and
    its assembly on the device.  The layout is fixed for a proof (same step circuit every step), so the per-slot offsets
    are uploaded once."""

    def __init__(self, field_id: int, num_frames: int, globals_len: int, body_len: int, slot_counts: dict | None = None):
        import torch

        self.field_id, self.num_frames, self.globals_len, self.body_len = field_id, num_frames, globals_len, body_len
        self.counts = dict(STEP_SLOT_COUNTS if slot_counts is None else slot_counts)
        self.sizes = {name: slot_witness_size(field_id, st) for name, st in SLOT_ORDER}
        self.slots_len = sum(self.counts.get(name, 0) * self.sizes[name] for name, _ in SLOT_ORDER)
        self.frame_len = self.slots_len + body_len
        self.w_len = globals_len + num_frames * self.frame_len
        self.offsets = {}
        off = 0
        for name, _ in SLOT_ORDER:
            cnt = self.counts.get(name, 0)
            if cnt:
                o = (globals_len + np.arange(num_frames, dtype=np.uint64)[:, None] * np.uint64(self.frame_len) + np.uint64(off)
                     + np.arange(cnt, dtype=np.uint64)[None, :] * np.uint64(self.sizes[name])).reshape(-1)
                self.offsets[name] = torch.from_numpy(o.view(np.int64)).cuda()
            off += cnt * self.sizes[name]

    def assemble(self, d_w, d_preimages: dict, globals_host: np.ndarray | None = None, bodies_host: np.ndarray | None = None, mont: bool = True,
                 stream=None, per_type_offsets: bool = False):
        """d_w: (>= w_len, 4) int64 device tensor.  d_preimages[name]: device tensor with num_frames * count preimages of that
        slot type, frame-major.  globals_host: (globals_len, 4); bodies_host: (num_frames, body_len, 4) - Montgomery values."""
        import torch

        lib = _lib.load()
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        assert d_w.shape[0] >= self.w_len
        if per_type_offsets:  # the general form: one call per slot type with explicit per-slot offsets
            for name, st in SLOT_ORDER:
                cnt = self.counts.get(name, 0)
                if not cnt:
                    continue
                pre = d_preimages[name]
                n = self.num_frames * cnt
                assert pre.numel() == n * slot_preimage_len(st) * 4
                _lib.check(lib.lurk_hip_slot_witness_dev(self.field_id, st, _lib.ptr(pre), n, int(mont), _lib.ptr(d_w), _lib.ptr(self.offsets[name]), 0, 0,
                                                         _lib.ptr(s)))
        else:  # every slot block of the MultiFrame in one call (the per-type launches run side by side inside)
            counts = (ctypes.c_size_t * 5)(*[self.counts.get(name, 0) for name, _ in SLOT_ORDER])
            ptrs = (ctypes.c_void_p * 5)(*[_lib.ptr(d_preimages[name]) if self.counts.get(name, 0) else None for name, _ in SLOT_ORDER])
            for name, st in SLOT_ORDER:
                if self.counts.get(name, 0):
                    assert d_preimages[name].numel() == self.num_frames * self.counts[name] * slot_preimage_len(st) * 4
            _lib.check(lib.lurk_hip_frames_witness_dev(self.field_id, self.num_frames, counts, ptrs, int(mont), _lib.ptr(d_w), self.globals_len,
                                                       self.frame_len, _lib.ptr(s)))
        if globals_host is not None and self.globals_len:
            g = np.ascontiguousarray(globals_host, dtype=np.uint64)
            _lib.check(lib.lurk_hip_witness_blocks_dev(_lib.ptr(d_w), 0, self.globals_len, _lib.ptr(g), 1, 1, self.globals_len, _lib.ptr(s)))
            self._keep_g = g
        if bodies_host is not None and self.body_len:
            b = np.ascontiguousarray(bodies_host, dtype=np.uint64)
            _lib.check(lib.lurk_hip_witness_blocks_dev(_lib.ptr(d_w), self.globals_len + self.slots_len, self.frame_len, _lib.ptr(b), 1, self.num_frames,
                                                       self.body_len, _lib.ptr(s)))
            self._keep_b = b
        return d_w
