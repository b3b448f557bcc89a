"""One h2e_run (twice) of a full batch for rocprofv3 --pmc: python3 exp/pmc_run.py <msm|bn256|bls12_381> <units>"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from halo2ecc_s_amd import Engine, Program, synth

what, units = sys.argv[1], int(sys.argv[2])
eng = Engine(0)
if what == "msm":
    prog = Program.msm_bn256_tile(1024, emit_shape=False)
    ins = [synth.msm_bn256_tile_inputs(1024, tile=k)[0] for k in range(units)]
elif what == "bn256":
    prog = Program.pairing_check_bn256(emit_shape=False)
    ins = [synth.pairing_check_bn256_inputs(instance=k) for k in range(units)]
else:
    prog = Program.pairing_check_bls12_381(emit_shape=False)
    ins = [synth.pairing_check_bls12_381_inputs(instance=k) for k in range(units)]
d = eng.upload_inputs(prog, np.stack(ins))
arrs = eng.alloc(prog, units)
for _ in range(2):
    eng.run(prog, d, *arrs)
    torch.cuda.synchronize()
assert int(arrs[3].abs().max()) == 0
