import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from halo2ecc_s_amd import Engine, Program, synth
n = int(sys.argv[1]); tiles = int(sys.argv[2])
eng = Engine(0)
prog = Program.msm_bn256_tile(n, emit_shape=False)
A = synth.msm_bn256_tile_inputs(n, tile=0, cheap_points=True)[0]
B = synth.msm_bn256_tile_inputs(n, tile=1, cheap_points=True)[0]
dA = eng.upload_inputs(prog, np.stack([A] * tiles)); dB = eng.upload_inputs(prog, np.stack([B] * tiles))
b, r, s, st = eng.alloc(prog, tiles)
for name, d in (("A", dA), ("B", dB), ("A", dA), ("B", dB)):
    st.zero_(); eng.run(prog, d, b, r, s, st); torch.cuda.synchronize()
    print(name, "status", np.unique(st.cpu().numpy()))
