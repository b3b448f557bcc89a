"""Device timeline of pipelined submissions from the debug build's stamp kernels (H2E_DEBUG_STAMPS; no profiler, the host runs free):
   python exp/dev_timeline.py <bn256|bls12_381|msm> <units> <ring> [bench-like host waits: 0/1]      (run by exp/dev_timeline.sh)"""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("H2E_DEBUG_STAMPS", "/tmp/h2e_stamps.txt")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from halo2ecc_s_amd import Engine, Program, synth, engine as engine_mod

what, units, ring = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
host_waits = len(sys.argv) > 4 and sys.argv[4] == "1"
eng = Engine(0)
if what == "bn256":
    prog = Program.pairing_check_bn256(emit_shape=False)
    ins = np.stack([synth.pairing_check_bn256_inputs(instance=k) for k in range(units)])
elif what == "bls12_381":
    prog = Program.pairing_check_bls12_381(emit_shape=False)
    ins = np.stack([synth.pairing_check_bls12_381_inputs(instance=k) for k in range(units)])
else:
    prog = Program.msm_bn256_tile(1024, emit_shape=False)
    ins = np.stack([synth.msm_bn256_tile_inputs(1024, tile=k, with_expected=False)[0] for k in range(units)])
d_in = eng.upload_inputs(prog, ins)
bufs = [eng.alloc(prog, units) for _ in range(ring)]
eng.set_option(4, ring)
eng.set_profiling(host_waits)
pending = []
N = 40
for k in range(N):
    if k == 8:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    base, rng, sel, status = bufs[k % ring]
    if len(pending) >= ring:
        job = pending.pop(0)
        eng.wait(job)
        if host_waits:
            eng.job_launch_ms(job)
    status.zero_()
    pending.append(eng.submit(prog, d_in, base, rng, sel, status))
for job in pending:
    eng.wait(job)
torch.cuda.synchronize()
print(f"{what} x {units} ring {ring} host waits {host_waits}: {1e3 * (time.perf_counter() - t0) / (N - 8):.3f} ms per step")
n = engine_mod.lib().h2e_debug_dump_stamps()
rows = [tuple(int(x) for x in l.split()) for l in open(os.environ["H2E_DEBUG_STAMPS"])]
t_first = min(r[2] for r in rows if r[0] >= 20)
names = {1000: "done", 1001: "start"}
by_run = {}
for run, tag, t in rows:
    by_run.setdefault(run, []).append((tag, (t - t_first) / 100.0))   # 100 MHz -> us
for run in sorted(by_run):
    if 20 <= run < 32:
        ev = sorted(by_run[run], key=lambda x: x[1])
        print("run %3d: " % run + "  ".join("%s %.0f" % (names.get(tag, "s%d%s" % (tag // 4, ["cb", "ce", "xb", "xe"][tag % 4])), t) for tag, t in ev))
