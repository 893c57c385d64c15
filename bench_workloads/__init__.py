"""bench.py's workloads, one module each (bench.py at the repo root is the entry point the driver runs: arguments, the launcher for
--gpus N, process-group set-up and dispatch).  Every workload prints ONE JSON line on rank 0; only the --verify legs and the cpu_baseline
legs import oracle/ (the checker, never inside a timed region)."""
