"""The multi-GPU prediction, written down BEFORE any N-GPU run exists so that the first SCALE_rNN.json can
falsify it.  Every constant is a single-GPU measurement of rounds 5-6 (profiles/r05_*, profiles/r06_*); nothing here was fitted to a
multi-GPU number - there is none.  `bench.py --gpus N` prints `predicted_value` beside `value`."""
from __future__ import annotations

ACC_MS_PER_2_20 = 0.83      # bucket accumulation per 2^20 points of a table key (13 windows; 3.3-3.5 ms at 2^22)
SORT_MS_PER_2_20 = 0.14     # the three sort passes per 2^20 points (0.56 ms at 2^22)
CHAIN_MS = 0.64             # what a commitment pays whatever its size: plan, finalize, 19 one-addition-deep reduction levels, host tail
PIPELINE_FACTOR = 1.06      # commitments in flight: measured per-commitment share / (accumulate + sort) at 2^22 with 4 in flight (4.12 / 3.88)
EXCHANGE_MS = {1: 0.0, 2: 0.045, 4: 0.055, 8: 0.075}  # ONE all-gather of 96 B per rank + the host sum: a small-message RCCL latency over xGMI, not bandwidth


# weak scaling: what the exchange costs the pipeline besides its latency - every step puts one RCCL kernel (a few workgroups, ~20 us) on
# each device beside two resident accumulations, and N ranks share the host's cores for their host tails.  A GUESS (there is no N-GPU
# measurement to take it from), stated so that it can be wrong: 1 % per doubling.
WEAK_INTERFERENCE = {1: 1.0, 2: 0.99, 4: 0.98, 8: 0.97}


def _exchange(n_gpus):
    return EXCHANGE_MS.get(n_gpus, 0.01 * n_gpus)


def msm_ms_per_step(log_n, n_gpus, scaling, pipeline):
    """milliseconds per step of `bench.py --workload msm --gpus N`: weak = every rank its own 2^log_n points, commitments in flight,
    the exchange on the host thread while the device works on the others; strong = ONE 2^log_n commitment cut across the ranks."""
    per = (1 << log_n) / float(1 << 20)
    if scaling == "weak":
        body = (ACC_MS_PER_2_20 + SORT_MS_PER_2_20) * per
        if pipeline > 1:
            return max(body * PIPELINE_FACTOR, _exchange(n_gpus)) / WEAK_INTERFERENCE.get(n_gpus, 0.95)  # the exchange hides behind the commitments still in flight
        return body + CHAIN_MS + _exchange(n_gpus)
    slice_per = per / n_gpus
    return (ACC_MS_PER_2_20 + SORT_MS_PER_2_20) * slice_per + CHAIN_MS + _exchange(n_gpus)


def msm_predicted_value(log_n, n_gpus, scaling, pipeline):
    """Mscalar-mul/s of the whole job"""
    ms = msm_ms_per_step(log_n, n_gpus, scaling, pipeline)
    total = (1 << log_n) * (n_gpus if scaling == "weak" else 1)
    return total / (ms * 1e-3) / 1e6


def table():
    """the numbers DESIGN.md section 6 quotes"""
    rows = []
    for log_n in (22, 24):
        for scaling, pipe in (("weak", 4), ("strong", 1)):
            rows.append((log_n, scaling, [round(msm_predicted_value(log_n, n, scaling, pipe)) for n in (1, 2, 4, 8)]))
    return rows


if __name__ == "__main__":
    for log_n, scaling, vals in table():
        print(f"2^{log_n} {scaling:6s} N=1/2/4/8: " + " / ".join(str(v) for v in vals) + " Mscalar-mul/s")
