"""Driver hooks: build() compiles every HIP extension for gfx950 (+ the oracle, which is the checker,
not the product); smoke() runs one small hot-path invocation on cuda:0 and checks it against the oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build():
    from halo2ecc_s_amd import build as b
    b.build(verbose=True)
    # the oracle's C++ restatement (test infrastructure); the reference itself is Rust with un-vendored
    # git dependencies and there is no Rust toolchain in the image, so there is no oracle/_ref.
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    import halo2ecc_s_amd  # noqa: F401
    from halo2ecc_s_amd.engine import lib, EXPORTED_SYMBOLS
    L = lib()
    for s in EXPORTED_SYMBOLS:
        getattr(L, s)


def smoke():
    import numpy as np
    import oracle_lib
    from halo2ecc_s_amd import Engine, Program, synth
    from parity import compare_advice
    eng = Engine(0)
    n = 4
    inp, _ = synth.msm_bn256_tile_inputs(n)
    prog = Program.msm_bn256_tile(n)
    d_in = eng.upload_inputs(prog, np.stack([inp]))
    base, rng, sel, status = eng.alloc(prog, 1)
    eng.run(prog, d_in, base, rng, sel, status)
    eng.torch.cuda.synchronize()
    assert int(status[0]) == 0, int(status[0])
    orun = oracle_lib.run_msm_bn256_tile(n, inp)
    compare_advice(prog, orun, base, rng, sel)
    print(f"smoke OK: {n}-point bn256 MSM tile, {prog.n_advice_cells} advice cells bit-exact vs oracle")


if __name__ == "__main__":
    build()
