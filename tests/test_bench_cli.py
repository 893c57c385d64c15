"""bench.py's launcher contract, checked without a GPU: `--gpus N` must either find N ranks or make them."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_a_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stdout + out.stderr)


def test_bench_gpus_flag_is_read():
    """With --gpus 2 and no WORLD_SIZE the script re-executes itself under torch.distributed.run (two ranks); here,
    without a GPU, those ranks fail - but they must have been started."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--log-n", "10", "--steps", "1",
                          "--warmup", "0", "--no-cpu-baseline", "--pmc", "off"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    import torch

    if not torch.cuda.is_available():
        assert out.returncode != 0
        assert "local_rank" in out.stderr or "rank" in out.stderr.lower()  # torch.distributed.run's failure report names the ranks
