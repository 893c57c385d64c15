"""--workload fold_step: the synthetic stand-in for one Nova folding step of benches/fibonacci.rs through the step entry points."""
from __future__ import annotations

import ctypes
import json
import os
import sys
import time

from .common import BENCH, ROOT, cpu_leg_threads


def _frame_structured_columns(rng, row_of_entry, num_cons, num_vars, num_io):
    """Column pattern of the Lurk step circuit: W = [globals | frame 0 aux | frame 1 aux | ...] (frames are
    synthesized independently and their aux concatenated, /root/reference/src/lem/multiframe.rs:699-702, 11 141
    constraints and 9 119 aux per frame, src/lem/eval.rs:1966-1967), so frame f's rows touch frame f's block (88 %),
    the globals at the front (6 %), the previous frame's block (4 %: its outputs) and the constant-one column u (2 %)."""
    import numpy as np

    nf = max(1, num_cons // 11141)
    cons_pf, vars_pf = -(-num_cons // nf), max(1, num_vars // nf)
    frame = np.minimum(row_of_entry // cons_pf, nf - 1)
    kind = rng.random(row_of_entry.size)
    local = frame * vars_pf + rng.integers(0, vars_pf, row_of_entry.size)
    prev = np.maximum(frame - 1, 0) * vars_pf + rng.integers(0, vars_pf, row_of_entry.size)
    glob = rng.integers(0, min(256, num_vars), row_of_entry.size)
    cols = np.where(kind < 0.88, local, np.where(kind < 0.94, glob, np.where(kind < 0.98, prev, num_vars)))
    return np.minimum(cols, num_vars + num_io).astype(np.uint64)


def synth_r1cs_shape(field_id, p, num_cons, num_vars, num_io, seed=7, uniform_columns=False):
    """Synthetic CSR triple shaped like the Lurk step circuit (3-4 entries per row, one row in 300 a 255-entry
    bit decomposition, coefficients mostly +-1 / small): the bench's own generator (numpy), values in Montgomery form."""
    import numpy as np

    rng = np.random.default_rng(seed)
    ncols = num_vars + 1 + num_io
    table_ints = [1, p - 1, 2, p - 2, 3, 4, 8, 16, 256, 1 << 32, p - (1 << 16)] + [int(rng.integers(1, 1 << 62)) ** 4 % p for _ in range(21)]
    table = np.array([[(v << 256) % p >> (64 * w) & 0xFFFFFFFFFFFFFFFF for w in range(4)] for v in table_ints], dtype=np.uint64)
    weights = np.array([40, 25, 5, 2, 2, 1, 1, 1, 1, 1, 1] + [1] * 21, dtype=np.float64)
    weights /= weights.sum()

    def sparse(one_per_row=False):
        cnt = np.ones(num_cons, dtype=np.uint64) if one_per_row else rng.integers(3, 5, num_cons).astype(np.uint64)
        if not one_per_row:
            cnt[rng.integers(0, num_cons, max(1, num_cons // 300))] = min(ncols, 255)
        indptr = np.zeros(num_cons + 1, dtype=np.uint64)
        np.cumsum(cnt, out=indptr[1:])
        nnz = int(indptr[-1])
        rows = np.repeat(np.arange(num_cons, dtype=np.int64), cnt.astype(np.int64))
        if one_per_row:
            indices = np.full(nnz, num_vars, dtype=np.uint64)
        elif uniform_columns:
            indices = rng.integers(0, ncols, nnz).astype(np.uint64)
        else:
            indices = _frame_structured_columns(rng, rows, num_cons, num_vars, num_io)
        data = np.ascontiguousarray(table[rng.choice(len(table_ints), size=nnz, p=weights)])
        return indptr, indices, data

    return sparse(), sparse(), sparse(one_per_row=True)


def fold_step_workload(args, lib, world, rank):
    """Synthetic stand-in for the device work of ONE Nova folding step of benches/fibonacci.rs on the primary (Pallas) curve
    (the bench itself needs cargo + arecibo and cannot run here), through the step entry points:
      W2 assembled in HBM: 14 hash4 + 6 hash8 + 1 commitment + 3 bit-decomposition slot blocks per frame written by the trace
        kernels (lurk_hip_slot_witness_dev), the non-slot remainder of every frame (1 311 aux, what the CPU synthesis produces)
        copied in over PCIe                                                      (src/lem/multiframe.rs:520-592, 699-702)
      lurk_hip_fold_step_begin: commit(W2) || cross term T over the step circuit's rows || commit(T)   (nova.rs:287-293)
      lurk_hip_fold_step_finish(r): [W | u | X] <- z1 + r z2, E <- E1 + r T
    Sizes on Pallas: 8 951 aux per frame (7 640 slot aux: bit decompositions are 298 instead of BN254's 354; + 1 311) and
    10 973 constraints per frame (11 141 - 3 x 56), from src/lem/eval.rs:1960-1967 and multiframe.rs:495-497.
    Reported as "equivalent Lurk iterations/s" = rc / t(step).  Left out: what stays on the CPU in the reference (the transcript,
    circuit synthesis of the frame bodies, the small secondary-curve fold): an upper bound on the end-to-end rate, flagged synthetic."""
    import numpy as np
    import torch

    import lurk_beta_amd as L
    from lurk_beta_amd import _lib, synth

    rc = args.rc
    F = L.FIELD_PALLAS_FQ
    mf = L.MultiFrameWitness(F, rc, globals_len=64, body_len=1311)
    n_w, n_t, n_io = mf.w_len, 10973 * rc, 6
    n_key = max(n_w, n_t)
    q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
    stream = torch.cuda.current_stream().cuda_stream
    d_bases = synth.bases(L.CURVE_PALLAS, n_key)
    pre = {"hash4": synth.scalars(F, 3, 1, 14 * rc * 4, mont=True), "hash8": synth.scalars(F, 4, 1, 6 * rc * 8, mont=True),
           "commitment": synth.scalars(F, 5, 1, rc * 3, mont=True), "bit_decomp": synth.scalars(F, 6, 1, 3 * rc, mont=True)}
    globals_pinned = torch.empty((mf.globals_len, 4), dtype=torch.int64).pin_memory()   # (a pageable source would make the async copy wait for the stream)
    globals_pinned.copy_(synth.scalars(F, 7, 1, mf.globals_len, mont=True).cpu())
    globals_host = globals_pinned.numpy().view(np.uint64)
    bodies_host = torch.empty((rc, mf.body_len, 4), dtype=torch.int64).pin_memory()
    bodies_host.copy_(synth.scalars(F, 8, 1, rc * mf.body_len, mont=True).reshape(rc, mf.body_len, 4).cpu())
    bodies_np = bodies_host.numpy().view(np.uint64)
    d_w2s = [torch.zeros((n_w, 4), dtype=torch.int64, device="cuda") for _ in range(3 if args.stage_ahead == 3 else 2)]
    d_w2 = d_w2s[0]
    x2 = synth.scalars(F, 9, 0, n_io, mont=True).cpu().numpy().view(np.uint64)
    t_setup = time.perf_counter()
    host_mats = synth_r1cs_shape(F, q, n_t, n_w, n_io)
    shape = L.R1CSShape(F, n_t, n_w, n_io, *host_mats)
    shape_setup_s = time.perf_counter() - t_setup
    if not args.verify:
        host_mats = None
    info = shape.info()
    # the challenge of every step is derived by the library's transcript (arecibo's PoseidonRO over pp_digest, U1, U2, comm_T: 128 bits),
    # on the host between begin and finish, as NIFS::prove does
    pp_digest = 0x1F3C5A7990B2D4E6F8123456789ABCDEF0FEDCBA9876543210AA55AA55AA55
    r_chal = 0x0FEDCBA0987654321234567890ABCDEF  # the secondary-curve leg below still feeds a constant of that size
    r_mont = np.array([((r_chal << 256) % q) >> (64 * w) & 0xFFFFFFFFFFFFFFFF for w in range(4)], dtype=np.uint64)
    last_r = [r_mont]
    torch.cuda.synchronize()
    devices = [int(x) for x in args.devices.split(",")] if args.devices else None
    if devices:  # the key cut across a device list inside this process: slices commit concurrently, 96-byte partials summed on the host
        assert not args.stage_ahead, "--devices: staging ahead is not available with a multi-device key"
        ck = L.MultiCommitmentKey(L.CURVE_PALLAS, d_bases.cpu().numpy().view(np.uint64), devices, precompute=bool(args.precompute), window_bits=args.window_bits,
                                  auto_slices=bool(args.auto_slices))
    else:
        ck = L.CommitmentKey(L.CURVE_PALLAS, d_bases, n=n_key, device=True, precompute=bool(args.precompute), window_bits=args.window_bits)
    ctx = L.FoldingContext(L.CURVE_PALLAS, shape, ck)
    ctx.set_pp_digest(pp_digest)
    helper_keys = []
    if args.helper_devices:
        assert args.stage_ahead and not devices, "--helper-devices goes with --stage-ahead 1 and a single-device key"
        for hd in [int(x) for x in args.helper_devices.split(",")]:
            _lib.check(lib.lurk_hip_set_device(hd))
            with torch.cuda.device(hd):
                hb = d_bases if d_bases.device.index == hd else d_bases.to(f"cuda:{hd}")
                hk = L.CommitmentKey(L.CURVE_PALLAS, hb, n=n_key, device=True, precompute=bool(args.precompute), window_bits=args.window_bits)
                hk.reserve(n_key, 3)
            helper_keys.append(hk)
            ctx.add_helper(hk)
        _lib.check(lib.lurk_hip_set_device(torch.cuda.current_device()))
    z1 = synth.scalars(F, 1, 1, n_w + 1 + n_io, mont=True).cpu().numpy().view(np.uint64)   # a running instance with witness-like values
    e1 = synth.scalars(F, 2, 0, n_t, mont=True).cpu().numpy().view(np.uint64)              # a running error vector (uniform, like any folded T)
    ident = np.zeros(12, dtype=np.uint64)
    # the running instance's commitments are the commitments of the running vectors (what RecursiveSNARK::verify re-computes)
    if devices:
        ctx.set_running(z1, e1, ck.commit(z1[:n_w], is_mont=True), ck.commit(e1, is_mont=True))
    else:
        ctx.set_running(z1, e1, ck.commit_device(torch.from_numpy(z1[:n_w].view(np.int64)).cuda(), n_w, is_mont=True),
                        ck.commit_device(torch.from_numpy(e1.view(np.int64)).cuda(), n_t, is_mont=True))

    # --stage-ahead 1: the step circuit's range of the NEXT witness is traced and its commitment started before this
    # step opens (lurk-beta synthesizes witnesses ahead of the folding loop, nova.rs:304-326); the augmented circuit's own
    # variables depend on the previous fold and arrive with begin: modelled as the first 9 000 and the last 3 000 positions
    if not devices:
        ck.reserve(n_key, 4)
    lo, hi = (9000, n_w - 3000) if args.stage_ahead and args.late_ranges else (0, n_w)
    late_host = synth.scalars(F, 10, 1, lo + n_w - hi, mont=True).cpu().numpy().view(np.uint64)
    patches = [(0, late_host[:lo]), (hi, late_host[lo:])] if lo else []
    staged_k = [0]

    def stage():
        buf = d_w2s[staged_k[0] & 1]
        staged_k[0] += 1
        mf.assemble(buf, pre, globals_host, bodies_np, mont=True, stream=stream)     # W2 in HBM (slot traces on the device)
        ctx.prefetch(buf[lo:hi], lo, stream=stream)                                  # commit(step circuit's range) starts now

    # --stage-ahead 3: the witness producer runs TWO steps ahead (lurk-beta's producer thread synthesizes frames independently of the
    # folding loop, nova.rs:304-326).  Step k's submit hook - called inside begin once the step's own launches are enqueued, so nothing
    # of it stands between a step's finish and the next cross term - traces witness k + 2 and stages instance k + 1 (traced during step
    # k - 1, so complete): its commitment is submitted at once, behind commit(T), in the FOLLOW class: sort and plan beside the cross
    # term's end and commit(T)'s sort, the accumulation once commit(T)'s has ended, in the window the step's serial chain leaves idle
    # (T's bucket reduction, the transcript, the next cross term).
    trace_k = [0]
    tstreams = [torch.cuda.Stream() for _ in range(3)] if args.stage_ahead == 3 else []

    def trace_next():
        k = trace_k[0]
        trace_k[0] += 1
        mf.assemble(d_w2s[k % 3], pre, globals_host, bodies_np, mont=True, stream=tstreams[k % 3].cuda_stream)

    def stage_traced():
        k = staged_k[0]
        staged_k[0] += 1
        ctx.prefetch(d_w2s[k % 3][lo:hi], lo, stream=tstreams[k % 3].cuda_stream)

    phase = {"assemble_and_stage": 0.0, "begin": 0.0, "transcript": 0.0, "finish": 0.0}  # host wall time per call site (begin blocks on the commitments)

    # the secondary-curve half of a step (Vesta, scalars in Fp): arecibo's augmented circuit on the other curve of the cycle, ~10^4
    # constraints.  In prove_step its NIFS::prove runs BEFORE the primary's and the two depend on each other through the circuits
    # (the primary's augmented variables - the late ranges here - hash the secondary's folded instance; the next secondary circuit is
    # synthesized from the primary's).  W2 comes from host memory (that circuit is synthesized on the CPU).
    sec = None
    if args.secondary and rank == 0:
        P_MOD = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
        nc2, nv2, nio2 = 10_000, 10_000, 2
        sec_mats = synth_r1cs_shape(L.FIELD_PALLAS_FP, P_MOD, nc2, nv2, nio2, seed=11, uniform_columns=True)
        shape2 = L.R1CSShape(L.FIELD_PALLAS_FP, nc2, nv2, nio2, *sec_mats)
        sec_bases = synth.bases(L.CURVE_VESTA, max(nc2, nv2))
        ck2 = L.CommitmentKey(L.CURVE_VESTA, sec_bases, n=max(nc2, nv2), device=True, precompute=bool(args.precompute))
        ck2.reserve(max(nc2, nv2), 3)
        ctx2 = L.FoldingContext(L.CURVE_VESTA, shape2, ck2)
        pp_digest2 = 0x0A5B3C7D9E1F2A4B6C8D0E1F3A5B7C9D1E3F5A7B9C0D2E4F6A8B0C2D4E6F
        ctx2.set_pp_digest(pp_digest2)
        sz1 = synth.scalars(L.FIELD_PALLAS_FP, 31, 1, nv2 + 1 + nio2, mont=True).cpu().numpy().view(np.uint64)
        se1 = synth.scalars(L.FIELD_PALLAS_FP, 32, 0, nc2, mont=True).cpu().numpy().view(np.uint64)
        ctx2.set_running(sz1, se1, ck2.commit_device(torch.from_numpy(sz1[:nv2].view(np.int64)).cuda(), nv2, is_mont=True),
                         ck2.commit_device(torch.from_numpy(se1.view(np.int64)).cuda(), nc2, is_mont=True))
        w2_sec = torch.empty((nv2, 4), dtype=torch.int64).pin_memory()
        w2_sec.copy_(synth.scalars(L.FIELD_PALLAS_FP, 33, 1, nv2, mont=True).cpu())
        w2_sec_np = w2_sec.numpy().view(np.uint64)
        x2_sec = synth.scalars(L.FIELD_PALLAS_FP, 34, 0, nio2, mont=True).cpu().numpy().view(np.uint64)

        def sec_step():  # one NIFS::prove on the secondary curve: begin, r from the library's transcript, finish
            ctx2.begin(w2_sec_np, x2_sec)
            ctx2.finish(ctx2.challenge())

        sec = dict(step=sec_step, ctx=ctx2, ck=ck2, shape=shape2, mats=sec_mats, bases=sec_bases, dims=(nv2, nc2, nio2), pp=pp_digest2, w2=w2_sec_np, x2=x2_sec, p=P_MOD)

    def step():
        t_a = time.perf_counter()
        if args.stage_ahead:
            if args.stage_ahead == 1:
                stage()                                                               # the next step's, under this step's work
            t_b = time.perf_counter()
            cw, ct = ctx.begin_prefetched(x2, patches)                               # late ranges + cross term + commit(T) (2: + stage() from the submit hook)
        elif args.witness_ahead:
            # the witness of step k+1 is produced while step k folds (lurk-beta's producer thread, nova.rs:304-326); this step's W2 was
            # produced a step ago.  --witness-ahead 1: its trace kernels are enqueued BEFORE this step's commitments and run beside
            # them; 2: AFTER begin has returned, so that they run while the host derives r (the device is idle there)
            k = staged_k[0]
            staged_k[0] += 1
            if args.witness_ahead == 1:
                mf.assemble(d_w2s[(k + 1) & 1], pre, globals_host, bodies_np, mont=True, stream=wstreams[(k + 1) & 1].cuda_stream)
            t_b = time.perf_counter()
            if args.witness_ahead == 3:  # traced from the step's submit hook: behind the step's opening kernels, beside its commitments
                hook_k[0] = k + 1
            cw, ct = ctx.begin(d_w2s[k & 1], x2, stream=wstreams[k & 1].cuda_stream)  # both commitments + the cross term
            if args.witness_ahead == 2:
                mf.assemble(d_w2s[(k + 1) & 1], pre, globals_host, bodies_np, mont=True, stream=wstreams[(k + 1) & 1].cuda_stream)
        else:
            mf.assemble(d_w2, pre, globals_host, bodies_np, mont=True, stream=stream)
            t_b = time.perf_counter()
            cw, ct = ctx.begin(d_w2, x2, stream=stream)                              # both commitments + the cross term
        t_c = time.perf_counter()
        r = ctx.challenge()  # r = RO(pp_digest, U1, U2, comm_T): U1 and U2 were absorbed inside begin, one permutation is left (lurk_hip_fold_step_challenge)
        t_r = time.perf_counter()
        ctx.finish(r)
        last_r[0] = r
        t_d = time.perf_counter()
        phase["assemble_and_stage"] += t_b - t_a
        phase["begin"] += t_c - t_b
        phase["transcript"] += t_r - t_c
        phase["finish"] += t_d - t_r
        return cw, ct

    def hook_sa3():
        trace_next()      # witness k + 2
        stage_traced()    # instance k + 1: staged and its commitment submitted (FOLLOW) from inside step k's begin

    if args.stage_ahead == 3:
        trace_next()
        trace_next()
        torch.cuda.synchronize()
        stage_traced()
        ctx.set_submit_hook(hook_sa3)
    elif args.stage_ahead:
        stage()
        if args.stage_ahead == 2:  # the next instance is traced, staged and its commitment started from inside begin (the submit hook)
            ctx.set_submit_hook(stage)
    elif args.witness_ahead:
        wstreams = [torch.cuda.Stream(), torch.cuda.Stream()]  # witness k is produced on stream k & 1, into buffer k & 1
        hook_k = [0]
        if args.witness_ahead == 3:
            ctx.set_submit_hook(lambda: mf.assemble(d_w2s[hook_k[0] & 1], pre, globals_host, bodies_np, mont=True, stream=wstreams[hook_k[0] & 1].cuda_stream))
        mf.assemble(d_w2s[0], pre, globals_host, bodies_np, mont=True, stream=wstreams[0].cuda_stream)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    import gc
    gc.collect()  # (as timeit does: no cyclic-garbage collection inside the timed region)
    gc.disable()
    # the `steps`-step region is repeated (as the msm workload repeats its region) and the MEDIAN reported: one region of twenty 3 ms
    # steps is 60 ms, short enough for a clock ramp or a neighbour's burst to move it by 5-7 % (seen once in six default lines); the
    # library's per-kernel profile covers the last repetition
    reps = max(1, min(getattr(args, "reps", 1) or 1, 3))
    region_ms = []
    for rep in range(reps):
        if rep == reps - 1:
            lib.lurk_hip_profile_enable(1)
            lib.lurk_hip_profile_reset()
            for k in phase:
                phase[k] = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        # (the folds are ordered on the context's own stream; torch.cuda.synchronize() is device-wide)
        torch.cuda.synchronize()
        region_ms.append((time.perf_counter() - t0) / args.steps * 1e3)
    elapsed = sorted(region_ms)[len(region_ms) // 2] * 1e-3 * args.steps
    lib.lurk_hip_profile_enable(0)
    phase_primary = dict(phase)  # (the both-curve loops below run step() again)
    both = {}
    if sec is not None:
        # (a) the secondary half alone, (b) BOTH halves as ONE timed loop in prove_step's order - secondary NIFS::prove, then the primary
        # step, every iteration (nova.rs:287-293): the secondary's two 10^4-point commitments run beside whatever the primary left in
        # flight (the staged commit(W2) of its next step) -, (c) the secondary half issued from the primary's submit hook, i.e. while
        # the primary's commitments are in flight: an upper bound on what overlapping the two halves could give; it is NOT prove_step's
        # order (the primary's augmented variables depend on the secondary's fold), so (b) is the both-curve number of this line.
        def timed(fn, n):  # the median of as many n-step regions as the primary loop takes
            out = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                out.append((time.perf_counter() - t) / n * 1e3)
            return sorted(out)[len(out) // 2]

        for _ in range(3):
            sec["step"]()
        both["secondary_alone"] = timed(sec["step"], max(args.steps, 10))

        def both_serial():
            sec["step"]()
            step()

        both_serial()
        both["serial_order"] = timed(both_serial, args.steps)
        hook_before = {2: stage, 3: hook_sa3}.get(args.stage_ahead) if args.stage_ahead else \
            ((lambda: mf.assemble(d_w2s[hook_k[0] & 1], pre, globals_host, bodies_np, mont=True, stream=wstreams[hook_k[0] & 1].cuda_stream)) if args.witness_ahead == 3 else None)

        def hook_both():
            if hook_before is not None:
                hook_before()
            sec["step"]()

        ctx.set_submit_hook(hook_both)
        step()
        both["from_the_submit_hook"] = timed(step, args.steps)
        ctx.set_submit_hook(hook_before)
    gc.enable()
    if args.stage_ahead >= 2 or (not args.stage_ahead and args.witness_ahead == 3):
        ctx.set_submit_hook(None)
    if args.stage_ahead:  # drain the instance staged by the last timed step (one was staged before the region: K stagings inside it)
        ctx.begin_prefetched(x2, patches)
        ctx.finish(r_mont)
    verified = None
    if args.verify and rank == 0:
        verified = verify_fold_step(L, ctx, host_mats, F, q, n_w, n_t, n_io, d_bases, pp_digest, x2,
                                    lambda buf: mf.assemble(buf, pre, globals_host, bodies_np, mont=True, stream=stream), d_w2s[0])

    def kernel_ms(name):
        tot, cnt = ctypes.c_double(), ctypes.c_uint64()
        _lib.check(lib.lurk_hip_profile_get(name.encode(), ctypes.byref(tot), ctypes.byref(cnt)))
        return tot.value / max(cnt.value, 1), cnt.value

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        nnz = sum(info["nnz"])
        ct_ms, _ = kernel_ms("r1cs_cross_term")
        fv_ms, _ = kernel_ms("fold_vec")
        tr_ms, tr_n = kernel_ms("poseidon_trace")
        bd_ms, _ = kernel_ms("bit_decomp_trace")
        acc_ms, acc_n = kernel_ms("msm_accumulate")       # mean launch over the timed region (HIP events on its launch stream): 2 per step
        acc_bytes = 96.0 * (n_w + n_t) / 2.0              # algorithmic bytes of the mean launch: 32 B scalar + 64 B base per point
        # algorithmic HBM bytes of the cross-term kernel: 8 B per CSR record + 4 B per row pointer, two 32-byte gathers
        # per record (z1, z2), 32 B of T per row; fold_vec: two reads + one write of 32 B per element
        ct_bytes = nnz * 8.0 + 3 * 4.0 * n_t + 2 * 32.0 * nnz + 32.0 * n_t
        cached = os.environ.get("LURK_FOLD_CACHED_PRODUCTS", "1") != "0" and not devices
        if cached:
            # the cached-products kernel (round 6): ONE 32-byte gather per record (z2 alone); per row the three cached products read (96 B),
            # T written (32 B), the step's three products written (96 B), and - the cache's own fold riding in the launch - the previous
            # step's products read (96 B) and the cache written back (96 B)
            ct_bytes = nnz * 8.0 + 3 * 4.0 * n_t + 32.0 * nnz + n_t * (96.0 + 32.0 + 96.0 + 96.0 + 96.0)
        fv_bytes = 96.0 * ((n_w + 1 + n_io) + n_t) / (1 if cached else 2)  # cached flow: z and E fold in ONE launch (on a side stream)
        res = {
            "metric": "equivalent Lurk iterations/s (synthetic stand-in for one Nova folding step, Pallas)",
            "value": round(rc / (ms * 1e-3), 1), "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong" if devices else "weak", "vs_baseline": None,
            "dtype": "u32x8 (255-bit Montgomery, integer VALU)", "data": "synthetic",
            "config": {"staged_ahead": args.stage_ahead, "witness_ahead": 0 if args.stage_ahead else args.witness_ahead,
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "devices": devices, "distinct_devices": len(set(devices)) if devices else 1,
                       "slices": len(ck.shards()) if devices else 1, "auto_slices": bool(args.auto_slices) if devices else None,
                       "helper_devices": args.helper_devices or None,
                       "workload": f"fold-step stand-in rc={rc} through lurk_hip_fold_step_{'prefetch/begin_prefetched' if args.stage_ahead else 'begin'}/finish: W2 ({n_w} aux: {21 * rc} Poseidon + {3 * rc} bit-decomposition "
                                   f"slot blocks traced on the device + {rc} x 1311 body aux over PCIe) -> MSM(W2) + cross term over {n_t} rows ({nnz} non-zeros, "
                                   f"{info['distinct_coefficients']} distinct coefficients) + MSM(T) -> fold of [W|u|X] and E",
                       "note": "device work + the transcript (r derived per step by the library's PoseidonRO on the host); body synthesis not modelled; "
                               "the secondary-curve half is the separate secondary_curve_step record",
                       "verified": verified,
                       "r1cs_columns": "frame-structured (88 % frame-local, 6 % globals, 4 % previous frame, 2 % u): a builder-chosen model of the step circuit's sparsity, "
                                       "see fold_kernels.r1cs_cross_term_uniform_columns for the structure-free case",
                       "shape_setup_s_once": round(shape_setup_s, 2)},
            "host_ms_per_step": {k: round(v / args.steps * 1e3, 3) for k, v in phase_primary.items()},
            # the step's dominant kernel is the bucket accumulation of its two commitments; the cross term (fold_kernels below) is the HBM-side one
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate_kernel", "achieved": round(acc_bytes / (acc_ms * 1e-3) / 1e9, 3) if acc_ms else None,
                         "peak": 8000.0, "unit": "GB/s", "frac": round(acc_bytes / (acc_ms * 1e-3) / 8e12, 6) if acc_ms else None, "traffic": None,
                         "avg_launch_ms": round(acc_ms, 4), "launches_per_step": acc_n // max(args.steps, 1), "algorithmic_bytes_per_launch": acc_bytes,
                         "note": "96 B per point over the mean of the step's two commitments (W2 and T), launches timed inside the step (they share the device with "
                                 "the cross term and each other's sort); integer-VALU bound as in the msm workload: see its roofline_valu"},
            "fold_kernels": {
                "r1cs_cross_term": {"kernel": "r1cs_cross_term_cached_kernel (z2's gathers + the cached A z1, B z1, C z1; the cache's fold in the same launch)" if cached
                                    else "r1cs_cross_term_kernel (six gathers per row)",
                                    "note": "launches timed INSIDE the step, beside the staged commitment's kernels and the witness producer; alone: bench_tools/fold_bench.py "
                                            "(profiles/r06_fold_kernels_isolated.txt: 0.20 ms = 43 % of 8 TB/s for the cached kernel at rc = 100)",
                                    "ms": round(ct_ms, 4), "algorithmic_bytes": ct_bytes, "achieved_GBps": round(ct_bytes / (ct_ms * 1e-3) / 1e9, 1) if ct_ms else None,
                                    "hbm_frac": round(ct_bytes / (ct_ms * 1e-3) / 8e12, 4) if ct_ms else None},
                "fold_vec": {"ms_per_launch": round(fv_ms, 4), "algorithmic_bytes_per_launch": fv_bytes,
                             "achieved_GBps": round(fv_bytes / (fv_ms * 1e-3) / 1e9, 1) if fv_ms else None,
                             "hbm_frac": round(fv_bytes / (fv_ms * 1e-3) / 8e12, 4) if fv_ms else None},
                "slot_witness_trace": {"poseidon_ms_per_launch": round(tr_ms, 4), "launches_per_step": tr_n // max(args.steps, 1),
                                       "bit_decomp_ms_per_launch": round(bd_ms, 4), "bytes_written_per_step": mf.slots_len * rc * 32.0},
            },
        }
        res["config"]["timed_region"] = ("median of %d repetitions of the %d-step region (each: synchronize, %d steps, synchronize); the per-kernel times are "
                                         "the last repetition's" % (len(region_ms), args.steps, args.steps))
        res["config"]["ms_per_step_by_repetition"] = [round(x, 4) for x in region_ms]
        if sec is not None:
            nv2, nc2, nio2 = sec["dims"]
            verified2 = None
            if args.verify:
                d_w2_sec = torch.from_numpy(sec["w2"].view(np.int64)).cuda()
                verified2 = verify_fold_step(L, sec["ctx"], sec["mats"], L.FIELD_PALLAS_FP, sec["p"], nv2, nc2, nio2, sec["bases"], sec["pp"], sec["x2"],
                                             lambda buf: None, d_w2_sec, curve=L.CURVE_VESTA)
            res["secondary_curve_step"] = {"ms_per_step": round(both["secondary_alone"], 4), "curve": "vesta", "constraints": nc2, "variables": nv2, "verified": verified2,
                                           "note": "arecibo's augmented circuit on the secondary curve is ~10^4 constraints [SURVEY 8: MEM]; two latency-bound "
                                                   "commitments of 10^4 points + cross term + transcript + folds, W2 from host memory"}
            res["both_curves_ms_per_step"] = round(both["serial_order"], 4)
            res["both_curves_iterations_per_s"] = round(rc / (both["serial_order"] * 1e-3), 1)
            res["both_curves"] = {"measured_as": "ONE timed loop, every iteration = the secondary curve's NIFS::prove then the primary step (prove_step's order, "
                                                 "/root/reference/src/proof/nova.rs:287-293); not a sum of two loops",
                                  "sum_of_the_two_separate_loops_ms": round(ms + both["secondary_alone"], 4),
                                  "secondary_issued_from_the_primary_submit_hook_ms": round(both["from_the_submit_hook"], 4),
                                  "note": "the hook form overlaps the two halves (an upper bound on that overlap); it is not prove_step's order - the primary's augmented "
                                          "variables depend on the secondary's fold - and is not the line's both-curve number"}
            sec["ctx"].close()
            sec["ck"].close()
            sec["shape"].close()
        # the step's own cross-term launch ALONE on the device (the in-step figure above shares the chip with the staged commitment's
        # accumulation, its tail and the witness producer): the cached kernel with the cache's fold riding in it, as the step launches it
        if cached:
            lib.lurk_hip_profile_enable(1)
            lib.lurk_hip_profile_reset()
            d_z1a = torch.from_numpy(z1.view(np.int64)).cuda()
            d_z2a = torch.cat([d_w2, d_z1a[n_w:]])
            torch.cuda.synchronize()
            abc1 = shape.multiply_vec(d_z1a, stream=stream)
            u1a = z1[n_w:n_w + 1].reshape(4)
            _, abc2a = shape.cross_term_cached(d_z2a, abc1, u1a, stream=stream)
            ra = np.array([3, 5, 7, 11], dtype=np.uint64)
            torch.cuda.synchronize()
            lib.lurk_hip_profile_reset()
            for _ in range(10):
                shape.cross_term_cached(d_z2a, abc1, u1a, prev=abc2a, r_prev_mont=ra, stream=stream)
            torch.cuda.synchronize()
            ca_ms, _ = kernel_ms("r1cs_cross_term")
            lib.lurk_hip_profile_enable(0)
            del abc1, abc2a, d_z1a, d_z2a
            res["fold_kernels"]["r1cs_cross_term"]["alone_on_the_device"] = {
                "ms": round(ca_ms, 4), "algorithmic_bytes": ct_bytes, "achieved_GBps": round(ct_bytes / (ca_ms * 1e-3) / 1e9, 1) if ca_ms else None,
                "hbm_frac": round(ct_bytes / (ca_ms * 1e-3) / 8e12, 4) if ca_ms else None,
                "note": "mean of 10 launches with nothing else on the device, the cache's fold in the launch (HIP events on the launch stream)"}
        # the same cross term over a structure-free shape (uniformly random columns): the other end of the sparsity range
        lib.lurk_hip_profile_enable(1)
        lib.lurk_hip_profile_reset()
        shape_u = L.R1CSShape(F, n_t, n_w, n_io, *synth_r1cs_shape(F, q, n_t, n_w, n_io, uniform_columns=True))
        d_z1 = torch.from_numpy(z1.view(np.int64)).cuda()
        d_t = torch.empty((n_t, 4), dtype=torch.int64, device="cuda")
        for _ in range(3):
            shape_u.cross_term(d_z1, torch.cat([d_w2, d_z1[n_w:]]), out=d_t, stream=stream)
        torch.cuda.synchronize()
        cu_ms, _ = kernel_ms("r1cs_cross_term")
        lib.lurk_hip_profile_enable(0)
        shape_u.close()
        cu_bytes = nnz * 8.0 + 3 * 4.0 * n_t + 2 * 32.0 * nnz + 32.0 * n_t  # (this leg runs the six-gather kernel, alone on the device)
        res["fold_kernels"]["r1cs_cross_term_uniform_columns"] = {"kernel": "r1cs_cross_term_kernel (six gathers per row), alone on the device", "ms": round(cu_ms, 4),
                                                                   "hbm_frac": round(cu_bytes / (cu_ms * 1e-3) / 8e12, 4) if cu_ms else None}
        if not args.no_cpu_baseline:
            from oracle import coracle as C

            m = min(n_t, 1 << 22)
            B = C.synth_bases(0, m)
            s_w, s_t = C.synth_scalars(1, 1, 1, min(n_w, m)), C.synth_scalars(1, 2, 0, m)
            threads, cpu_info = cpu_leg_threads()
            C.msm_fast(0, B[:4096], s_t[:4096], nthreads=threads)
            t1 = time.perf_counter()
            C.msm_fast(0, B[: min(n_w, m)], s_w, nthreads=threads)
            C.msm_fast(0, B, s_t, nthreads=threads)
            dt = time.perf_counter() - t1
            scale = (n_w + n_t) / (min(n_w, m) + m)
            res["cpu_baseline"] = {"value": round(rc / (dt * scale), 2), "unit": "iterations/s", "cores": threads, **cpu_info, "kind": "port",
                                   "sample": f"the step's two MSMs ({min(n_w, m)} and {m} points{'' if scale == 1 else ', scaled linearly to the full sizes'}) in {dt:.2f} s with oracle/msm_fast.c "
                                             f"(pasta-msm-shaped Pippenger, {threads} threads); fold arithmetic, witness generation and transcript not included"}
        print(json.dumps(res), flush=True)
    ctx.close()
    for hk in helper_keys:
        hk.close()
    ck.close()
    shape.close()


def verify_fold_step(L, ctx, host_mats, F, q, n_w, n_t, n_io, d_bases, pp_digest, x2, assemble, d_w2, curve=0):
    """--verify: ONE more step after the timed loop through lurk_hip_fold_step, every output against the oracle at the bench's own size:
    comm_W2 and comm_T (oracle/msm_fast.c), r (the oracle's transcript over the oracle's instance), T and the folded (z, E) element by
    element (oracle/oracle.c), the folded instance's commitments.  The checker only: nothing here is timed."""
    import numpy as np
    import torch

    from oracle import coracle as C
    from oracle import pyref as R

    f = F  # the scalar field of `curve` (Pallas: Fq = 1, Vesta: Fp = 0)
    t0 = time.perf_counter()
    mats = [(ip, ix, C.from_mont(f, d)) for ip, ix, d in host_mats]
    bases = d_bases.cpu().numpy().view(np.uint64).reshape(-1, 8)
    z1m, e1m = ctx.read()
    z1, e1 = C.from_mont(f, z1m), C.from_mont(f, e1m)
    cw1, ce1, _, _ = ctx.instance()
    pt = lambda a: None if a == (0, 0) else a
    cw1_o = pt(C.jac_to_affine(curve, C.msm_fast(curve, bases[:n_w], z1[:n_w])))   # the running instance's commitments, recomputed
    ce1_o = pt(C.jac_to_affine(curve, C.msm_fast(curve, bases[:n_t], e1)))
    ok = {"running_comm_W": pt(L.point_to_affine(curve, cw1)) == cw1_o, "running_comm_E": pt(L.point_to_affine(curve, ce1)) == ce1_o}
    assemble(d_w2)
    torch.cuda.synchronize()
    w2 = C.from_mont(f, d_w2.cpu().numpy().view(np.uint64).reshape(-1, 4))
    x2c = C.from_mont(f, x2.reshape(-1, 4))
    cw, ct, r_mont = ctx.step(d_w2, x2, pp_digest, stream=torch.cuda.current_stream().cuda_stream)
    z2 = np.concatenate([w2, C.ints_to_limbs([1]), x2c])
    u1 = C.limbs_to_ints(z1[n_w:n_w + 1])[0]
    t = C.cross_term(f, *[C.spmv(f, *M, z1) for M in mats], *[C.spmv(f, *M, z2) for M in mats], u1, 1)
    cw2_o, ct_o = C.jac_to_affine(curve, C.msm_fast(curve, bases[:n_w], w2)), C.jac_to_affine(curve, C.msm_fast(curve, bases[:n_t], t))
    ok["comm_W2"] = L.point_to_affine(curve, cw) == cw2_o
    ok["comm_T"] = L.point_to_affine(curve, ct) == ct_o
    r = R.nifs_challenge("pallas" if curve == 0 else "vesta", pp_digest, cw1_o, ce1_o, u1, C.limbs_to_ints(z1[n_w + 1:]), pt(cw2_o), C.limbs_to_ints(x2c), pt(ct_o))
    ok["challenge"] = C.limbs_to_ints(C.from_mont(f, r_mont.reshape(1, 4)))[0] == r
    zf, ef = C.axpy(f, z1, z2, r), C.axpy(f, e1, t, r)
    gz, ge = ctx.read()
    ok["folded_z"] = bool(np.array_equal(C.from_mont(f, gz), zf))
    ok["folded_E"] = bool(np.array_equal(C.from_mont(f, ge), ef))
    gcw, gce, _, _ = ctx.instance()
    ok["folded_comm_W"] = L.point_to_affine(curve, gcw) == C.jac_to_affine(curve, C.msm_fast(curve, bases[:n_w], zf[:n_w]))
    ok["folded_comm_E"] = L.point_to_affine(curve, gce) == C.jac_to_affine(curve, C.msm_fast(curve, bases[:n_t], ef))
    if not all(ok.values()):
        diag = {}
        try:  # which side is off: the device vector, the host-folded instance, or the checker?  (the GPU's own commitment of the read-back vector)
            import torch as _t

            g = L.point_to_affine(curve, ctx.key.commit_device(_t.from_numpy(np.ascontiguousarray(z1m[:n_w]).view(np.int64)).cuda(), n_w, is_mont=True))
            diag = {"gpu_commit_of_read_back_z1_equals_instance": pt(g) == pt(L.point_to_affine(curve, cw1)), "gpu_commit_of_read_back_z1_equals_oracle": pt(g) == cw1_o}
        except Exception as e:  # noqa: BLE001
            diag = {"diagnostic_failed": repr(e)}
        raise SystemExit(f"bench.py --verify: the fold step does not match the oracle: {ok} {diag}")
    return {"ok": True, "checks": sorted(ok), "oracle_s": round(time.perf_counter() - t0, 1),
            "against": "oracle/oracle.c (spmv, cross term, axpy), oracle/msm_fast.c (6 commitments), oracle/pyref.py (transcript), one extra step after the timed loop"}
