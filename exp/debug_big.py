"""parity of one large MSM tile against the oracle (debugging aid; the oracle takes a few seconds at n = 1024)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib
from parity import compare_advice
from halo2ecc_s_amd import Engine, Program, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = Engine(0)
inp, _ = synth.msm_bn256_tile_inputs(n, tile=0, cheap_points=True)
prog = Program.msm_bn256_tile(n)
d = eng.upload_inputs(prog, np.stack([inp, inp]))
b, r, s, st = eng.alloc(prog, 2)
eng.run(prog, d, b, r, s, st); torch.cuda.synchronize()
print("status", st.cpu().numpy())
orun = oracle_lib.run_msm_bn256_tile(n, inp, threads=os.cpu_count())
print("oracle status", orun.info.status)
try:
    compare_advice(prog, orun, b, r, s, instance=0)
    print("PARITY OK")
except AssertionError as e:
    print("PARITY FAIL", str(e)[:600])
