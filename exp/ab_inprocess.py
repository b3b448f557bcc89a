"""A/B of an engine env knob that is read on every h2e_run call, on ONE allocation of the advice arrays (the step time
depends on the allocation, exp/layout_variance.py): exp/ab_inprocess.py VAR value [value ...]   ('-' = unset)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from halo2ecc_s_amd import Engine, Program, synth
var, values = sys.argv[1], sys.argv[2:]
n, tiles = 1024, 64
eng = Engine(0)
prog = Program.msm_bn256_tile(n, emit_shape=False)
ins = np.stack([synth.msm_bn256_tile_inputs(n, tile=t, cheap_points=True, with_expected=False)[0] for t in range(tiles)])
d_in = eng.upload_inputs(prog, ins)
base, rng, sel, status = eng.alloc(prog, tiles)
eng.set_profiling(True)
for rep in range(int(os.environ.get("AB_REPS", "3"))):
    for v in values:
        if v == '-': os.environ.pop(var, None)
        else: os.environ[var] = v
        ms, xs = [], []
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.run(prog, d_in, base, rng, sel, status); torch.cuda.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0))
            xs.append(max(x[1] for x in eng.last_run_launch_ms()))
        print(f"{var}={v}: step {np.mean(ms[1:]):.2f} ms  window expansion {np.mean(xs[1:]):.2f} ms", flush=True)
