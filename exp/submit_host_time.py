"""Host time of one pipelined submission (h2e_submit) and of the wait + per-step consumer work around it, for a small pairing batch:
is the pipelined small-batch step (1.2-1.8 ms whatever the ring) the host's?"""
import os
import sys
import time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halo2ecc_s_amd import Engine, Program, synth

curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_381"
units = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ring = int(sys.argv[3]) if len(sys.argv) > 3 else 3
profiling = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
eng = Engine(0)
eng.set_option(6, 60)
if curve == "bn256":
    prog = Program.pairing_check_bn256(emit_shape=False)
    ins = np.stack([synth.pairing_check_bn256_inputs(instance=k) for k in range(units)])
else:
    prog = Program.pairing_check_bls12_381(emit_shape=False)
    ins = np.stack([synth.pairing_check_bls12_381_inputs(instance=k) for k in range(units)])
d_in = eng.upload_inputs(prog, ins)
bufs = [eng.alloc(prog, units) for _ in range(ring)]
eng.set_option(4, ring)   # H2E_OPT_PIPELINE_DEPTH
eng.set_profiling(profiling)
pending = []
t_submit, t_wait = [], []
N = 60
t_all0 = None
for k in range(N + 10):
    if k == 10:
        torch.cuda.synchronize()
        t_all0 = time.perf_counter()
        t_submit.clear(); t_wait.clear()
    base, rng, sel, status = bufs[k % ring]
    t0 = time.perf_counter()
    if len(pending) >= ring:
        job = pending.pop(0)
        eng.wait(job)
        if profiling:
            eng.job_launch_ms(job)
        else:
            torch.cuda.current_stream().synchronize() if False else None
    t1 = time.perf_counter()
    status.zero_()
    job = eng.submit(prog, d_in, base, rng, sel, status)
    t2 = time.perf_counter()
    pending.append(job)
    t_wait.append(t1 - t0)
    t_submit.append(t2 - t1)
for job in pending:
    eng.wait(job)
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all0
print(f"{curve} x {units}, ring {ring}, profiling {profiling}: {1e3 * t_all / N:.3f} ms per step; host: submit {1e3 * np.mean(t_submit):.3f} ms (max {1e3 * np.max(t_submit):.3f}), wait {1e3 * np.mean(t_wait):.3f} ms")
