"""Side measurement (not the headline bench): 64 x bn256 / 16 x bls12_381 check_pairing witness on one GPU."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from halo2ecc_s_amd import Engine, Program, synth
eng = Engine(0)
out = {}
for name, n, mk, gen, cells in (("bn256", int(os.environ.get("N_BN", "64")), Program.pairing_check_bn256, synth.pairing_check_bn256_inputs, 6165013),
                                ("bls12_381", int(os.environ.get("N_BLS", "16")), Program.pairing_check_bls12_381, synth.pairing_check_bls12_381_inputs, 7952811)):
    prog = mk(emit_shape=False)
    base_ins = [gen(instance=k) for k in range(4)]
    ins = np.stack([base_ins[k % 4] for k in range(n)])
    d = eng.upload_inputs(prog, ins)
    b, r, s, st = eng.alloc(prog, n)
    eng.set_profiling(True)
    for it in range(3):
        st.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.run(prog, d, b, r, s, st); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if not os.environ.get("H2E_DEBUG_LEVELS_NOSTORE"):
        assert int(st.abs().max()) == 0
    print(name, "value chain / expansion ms per launch:", [(round(a, 2), round(b_, 2)) for a, b_ in eng.last_run_launch_ms()], file=sys.stderr)
    out[name] = {"instances": n, "ms": dt * 1e3, "cells_per_s": cells * n / dt, "pairing_checks_per_s": n / dt}
    del d, b, r, s, st
    torch.cuda.empty_cache()
print(json.dumps(out))
