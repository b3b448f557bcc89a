"""Does the step time depend on where the advice arrays live?  Within one process: allocate the arrays, time 6 steps,
free them, keep a dummy allocation of a different size to shift the next placement, repeat.  (Between processes on
one box the step time differs by up to +-3 ms while it is constant to +-0.1 ms within a process.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from halo2ecc_s_amd import Engine, Program, synth
n, tiles = 1024, 64
eng = Engine(0)
prog = Program.msm_bn256_tile(n, emit_shape=False)
ins = np.stack([synth.msm_bn256_tile_inputs(n, tile=t, cheap_points=True, with_expected=False)[0] for t in range(tiles)])
d_in = eng.upload_inputs(prog, ins)
keep = []
for trial, pad_mb in enumerate([0, 0, 777, 3001, 1, 12345, 64, 0]):
    if pad_mb:
        keep.append(torch.empty(pad_mb << 20, dtype=torch.uint8, device="cuda"))
    base, rng, sel, status = eng.alloc(prog, tiles)
    ms = []
    for it in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.run(prog, d_in, base, rng, sel, status); torch.cuda.synchronize()
        ms.append(1e3 * (time.perf_counter() - t0))
    print(f"trial {trial} pad {pad_mb} MB: base {base.data_ptr():#x} range {rng.data_ptr():#x} select {sel.data_ptr():#x}  "
          f"steps {[round(x, 2) for x in ms[1:]]}", flush=True)
    del base, rng, sel, status
    torch.cuda.empty_cache()
