import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from halo2ecc_s_amd import Engine, Program, synth
eng = Engine(0)
def run(prog, ins):
    d = eng.upload_inputs(prog, np.stack(ins))
    b, r, s, st = eng.alloc(prog, len(ins))
    eng.run(prog, d, b, r, s, st); torch.cuda.synchronize()
    return st.cpu().numpy()
# dispatch A: 64 strands x (2 assign_w + int_mul)
print(run(Program.int_mul_batch(0, 64, emit_shape=False), [synth.int_mul_batch_inputs(0, 64)]))
# dispatch B: integer_chip_st (1 lane): ~25 ops incl 2 int_div
print(run(Program.integer_chip_st(0, emit_shape=False), [synth.integer_chip_st_inputs(0)]))
