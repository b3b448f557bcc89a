"""A/B of an env knob that is read when a program is recorded, on ONE allocation of the advice arrays:
exp/ab_program.py VAR value [value ...]   ('-' = unset)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from halo2ecc_s_amd import Engine, Program, synth
var, values = sys.argv[1], sys.argv[2:]
n, tiles = 1024, 64
eng = Engine(0)
progs = {}
for v in values:
    if v == '-': os.environ.pop(var, None)
    else: os.environ[var] = v
    progs[v] = Program.msm_bn256_tile(n, emit_shape=False)
os.environ.pop(var, None)
ins = np.stack([synth.msm_bn256_tile_inputs(n, tile=t, cheap_points=True, with_expected=False)[0] for t in range(tiles)])
d_in = eng.upload_inputs(progs[values[0]], ins)
base, rng, sel, status = eng.alloc(progs[values[0]], tiles)
eng.set_profiling(True)
for rep in range(int(os.environ.get("AB_REPS", "3"))):
    for v in values:
        ms = []
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.run(progs[v], d_in, base, rng, sel, status); torch.cuda.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0))
        lm = eng.last_run_launch_ms()
        print(f"{var}={v}: step {np.mean(ms[1:]):.2f} ms  tail: chain {lm[-1][0]:.2f} expansion {lm[-1][1]:.2f}  windows x {lm[-2][1]:.2f}", flush=True)
