"""The default line's verified sub-records (child runs of bench.py) and the launcher for --gpus N."""
from __future__ import annotations

import ctypes
import json
import os
import sys
import time

from .common import BENCH, ROOT


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher in the environment: re-run this command line under
    torch.distributed.run with one rank per GPU (rendezvous on 127.0.0.1); rank 0 of that job prints the JSON line."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def sub_records(args):
    """The default line's sub-records: child runs of this file (a fresh process each: its own HIP context, nothing shared with the
    timed region above), each with --verify, so that the driver's one command witnesses the folding step (BASELINE.json's first
    metric, through its synthetic stand-in), the 2^24 Poseidon tree (configs[2]), the 2^24 NTT and the compressing proof (f3) with
    their parity checks."""
    import subprocess

    common = ["--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--sub-records", "off", "--pmc", "off", "--verify"]
    legs = {
        "fold_step_rc100": ["--workload", "fold_step", "--rc", "100"],
        # BASELINE.json configs[1] (the standalone 2^20 MSM) synchronous and with two commitments in flight, and configs[3] (the rc = 900
        # step), on the driver's clock since round 6: --verify = the discrete-log checksum / every output of one more step vs the oracle
        "msm_2_20_sync": ["--workload", "msm", "--log-n", "20", "--pipeline", "1", "--no-plain-leg", "--cpu-sample-log-n", "20"],
        "msm_2_20_inflight2": ["--workload", "msm", "--log-n", "20", "--pipeline", "2", "--no-plain-leg", "--no-cpu-baseline"],
        "fold_step_rc900": ["--workload", "fold_step", "--rc", "900", "--secondary", "0"],
        "poseidon_tree_2_24": ["--workload", "poseidon_tree", "--log-n", "24"],
        "ntt_2_24": ["--workload", "ntt", "--log-n", "24"],
        "compress_2_20": ["--workload", "compress", "--log-n", "20"],  # the compressing proof of a 2^20 x 2^20 instance; --verify = the oracle's verifier
    }
    if args.no_cpu_baseline:
        common.append("--no-cpu-baseline")
    out = {}
    for name, extra in legs.items():
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, BENCH] + extra + [a for a in common if not (a == "--no-cpu-baseline" and a in extra)], capture_output=True, text=True,
                               timeout=420)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-400:]}"}
            else:
                out[name] = json.loads(lines[-1])
        except Exception as e:  # noqa: BLE001 (a sub-record must never cost the headline line)
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        out[name]["wall_s"] = round(time.perf_counter() - t0, 1)
    return out


def summarize(subs):
    """{leg: [ms_per_step, verified]} - the LAST key of the default line, so that the tail of the line a log keeps still shows every
    sub-record's time and parity flag (a failed leg: [None, False, the error's tail])."""
    out = {}
    for name, r in subs.items():
        if "error" in r:
            out[name] = [None, False, str(r["error"])[-120:]]
            continue
        v = (r.get("config") or {}).get("verified")
        if v is None:
            v = r.get("verified")  # the msm workload reports its discrete-log checksum at the top level
        ok = bool(v.get("ok")) if isinstance(v, dict) else (bool(v) if v is not None else None)
        if ok and (r.get("cpu_baseline") or {}).get("matches_gpu_result") is False:
            ok = False
        if ok and isinstance((r.get("secondary_curve_step") or {}).get("verified"), dict) and not r["secondary_curve_step"]["verified"].get("ok"):
            ok = False
        out[name] = [r.get("ms_per_step"), ok]
        if "both_curves_ms_per_step" in r:
            out[name].append({"both_curves_ms_per_step": r["both_curves_ms_per_step"]})
    return out


def any_failed(summary):
    """a leg that errored, or whose parity check did not come back true"""
    return any(v[1] is not True for v in summary.values())
