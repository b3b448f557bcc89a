"""A/B of an env knob that is read when an engine context creates its streams, on ONE allocation of the advice arrays:
exp/ab_engine.py VAR value [value ...]   ('-' = unset).  Prints the device's stream priority range first."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from halo2ecc_s_amd import Engine, Program, synth
var, values = sys.argv[1], sys.argv[2:]
n, tiles = 1024, 64
prog = Program.msm_bn256_tile(n, emit_shape=False)
ins = np.stack([synth.msm_bn256_tile_inputs(n, tile=t, cheap_points=True, with_expected=False)[0] for t in range(tiles)])
engs = {}
base = None
for v in values:
    if v == '-': os.environ.pop(var, None)
    else: os.environ[var] = v
    e = Engine(0)
    if base is None:
        d_in = e.upload_inputs(prog, ins)
        base, rng, sel, status = e.alloc(prog, tiles)
    e.run(prog, d_in, base, rng, sel, status); torch.cuda.synchronize()   # streams are created on the first run
    e.set_profiling(True)
    engs[v] = e
os.environ.pop(var, None)
for rep in range(int(os.environ.get("AB_REPS", "3"))):
    for v in values:
        e = engs[v]
        ms = []
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            e.run(prog, d_in, base, rng, sel, status); torch.cuda.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0))
        lm = e.last_run_launch_ms()
        print(f"{var}={v}: step {np.mean(ms[1:]):.2f} ms  windows: chain {lm[-2][0]:.2f} x {lm[-2][1]:.2f}  tail chain {lm[-1][0]:.2f}", flush=True)
