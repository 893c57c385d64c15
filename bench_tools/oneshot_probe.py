import time, numpy as np, torch
import sys; sys.path.insert(0, '.')
import lurk_beta_amd as L
from lurk_beta_amd import synth
for log_n in (20, 22):
    n = 1 << log_n
    B = synth.bases(0, n).cpu().numpy().view(np.uint64)
    S = synth.scalars(1, 1, 0, n, mont=True).cpu().numpy().view(np.uint64)
    L.msm(0, B[:1024], S[:1024], is_mont=True)
    ts = []
    for _ in range(4):
        t = time.perf_counter(); r = L.msm(0, B, S, is_mont=True); ts.append(time.perf_counter() - t)
    print(log_n, 'oneshot ms', [round(x * 1e3, 2) for x in ts], 'M/s', round(n / min(ts) / 1e6, 1))
    ck = L.CommitmentKey(0, B)
    ts = []
    for _ in range(4):
        t = time.perf_counter(); r = ck.commit(S, is_mont=True); ts.append(time.perf_counter() - t)
    print(log_n, 'ctx_run (host scalars) ms', [round(x * 1e3, 2) for x in ts], 'M/s', round(n / min(ts) / 1e6, 1))
    # pinned
    Sp = torch.from_numpy(S.view(np.int64)).pin_memory().numpy().view(np.uint64)
    ts = []
    for _ in range(4):
        t = time.perf_counter(); r = ck.commit(Sp, is_mont=True); ts.append(time.perf_counter() - t)
    print(log_n, 'ctx_run (pinned host scalars) ms', [round(x * 1e3, 2) for x in ts])
    ck.close()
