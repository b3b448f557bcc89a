import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib
from halo2ecc_s_amd import Engine, Program, synth
n = 1024; tiles = 64
eng = Engine(0)
inp, _ = synth.msm_bn256_tile_inputs(n, tile=0, cheap_points=True)
prog = Program.msm_bn256_tile(n)
d = eng.upload_inputs(prog, np.stack([inp] * tiles))
b, r, s, st = eng.alloc(prog, tiles)
eng.run(prog, d, b, r, s, st); torch.cuda.synchronize()
print("status", st.cpu().numpy())
orun = oracle_lib.run_msm_bn256_tile(n, inp, threads=os.cpu_count())
rows = (prog.base_rows, prog.range_rows, prog.select_rows)
outs = (b, r, s)
for region in range(3):
    ovals, oflags = orun.adv(region, rows[region])
    for inst in (0, 17, 63):
        got = outs[region][inst].cpu().numpy().view(np.uint64)
        bad = np.argwhere((got != ovals).any(axis=2))
        print("region", region, "instance", inst, "bad cells", len(bad), "first", bad[:3].tolist(), "last", bad[-2:].tolist() if len(bad) else None)
launches = prog.launches()
print(launches[-2:])
ovals, _ = orun.adv(0, rows[0])
got = b[0].cpu().numpy().view(np.uint64)
bad = np.argwhere((got != ovals).any(axis=2))
for (rr, cc) in bad[:6].tolist():
    print(rr, cc, [hex(int(x)) for x in got[rr, cc]], [hex(int(x)) for x in ovals[rr, cc]])
# second run: does it heal?
st.zero_(); eng.run(prog, d, b, r, s, st); torch.cuda.synchronize()
print("second run status", st.cpu().numpy()[:4])
