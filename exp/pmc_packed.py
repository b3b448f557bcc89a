"""One h2e_run of a small pairing batch (for rocprofv3 --pmc): python3 exp/pmc_packed.py <bn256|bls12_381> <units>"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from halo2ecc_s_amd import Engine, Program, synth

curve, units = sys.argv[1], int(sys.argv[2])
eng = Engine(0)
if curve == "bn256":
    prog, gen = Program.pairing_check_bn256(emit_shape=False), synth.pairing_check_bn256_inputs
else:
    prog, gen = Program.pairing_check_bls12_381(emit_shape=False), synth.pairing_check_bls12_381_inputs
d = eng.upload_inputs(prog, np.stack([gen(instance=k) for k in range(units)]))
arrs = eng.alloc(prog, units)
for _ in range(2):
    eng.run(prog, d, *arrs)
    torch.cuda.synchronize()
assert int(arrs[3].abs().max()) == 0
