import time, numpy as np, torch, sys
sys.path.insert(0, '.')
import lurk_beta_amd as L
from lurk_beta_amd import synth, _lib
lib = _lib.load()
for log_n in (16, 20, 22):
    n = 1 << log_n
    d_b = synth.bases(0, n); d_s = synth.scalars(1, 1, 0, n, mont=True)
    ck = L.CommitmentKey(0, d_b, n=n, device=True, precompute=True); ck.reserve(n, 3)
    st = torch.cuda.current_stream().cuda_stream
    for prof in (0, 1):
        lib.lurk_hip_profile_enable(prof)
        ts = []
        for i in range(12):
            t = time.perf_counter(); ck.submit_device(i % 3, d_s, n, is_mont=True, stream=st); ts.append(time.perf_counter() - t)
            if i % 3 == 2:
                for k in range(3): ck.wait(k)
        print(log_n, 'profiling', prof, 'submit host us', [round(x * 1e6) for x in ts[3:]])
    lib.lurk_hip_profile_enable(0); lib.lurk_hip_profile_reset()
    ck.close()
