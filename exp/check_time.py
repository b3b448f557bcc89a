"""time of h2e_check (the device-side constraint check) over BASELINE's batches: 64 x 1024-point tiles, 64 bn256 checks"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halo2ecc_s_amd import Engine, Program, synth
from halo2ecc_s_amd import engine as E

eng = Engine(0)
eng.set_option(6, 60)
for name in ("pairing_bn256", "msm"):
    if name == "msm":
        n, units = 1024, 64
        prog = Program.msm_bn256_tile(n)
        ins = np.stack([synth.msm_bn256_tile_inputs(n, tile=t, cheap_points=True, with_expected=False)[0] for t in range(units)])
    else:
        units = 64
        prog = Program.pairing_check_bn256()
        ins = np.stack([synth.pairing_check_bn256_inputs(instance=k) for k in range(units)])
    d_in = eng.upload_inputs(prog, ins)
    base, rng, sel, status = eng.alloc(prog, units)
    eng.run(prog, d_in, base, rng, sel, status)
    torch.cuda.synchronize()
    out = None
    for cls, what in ((0, "all"), (1 << E.CHECK_BASE_GATE, "base gate"), ((1 << E.CHECK_RANGE_GATE) | (1 << E.CHECK_RANGE_LOOKUP), "range"),
                      (1 << E.CHECK_SELECT_LOOKUP, "select"), (1 << E.CHECK_COPY, "copy")):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = eng.check(prog, d_in, base, rng, sel, classes=cls, out=out)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        print(name, units, "instances,", what, "ms", [round(t, 2) for t in ts], "failing (expected-point mismatch of the MSM test body aside):",
              int((out[:, :5] != 0).any(dim=1).sum()), flush=True)
    del base, rng, sel
    torch.cuda.empty_cache()
