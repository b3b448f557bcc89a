import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib
from halo2ecc_s_amd import Engine, Program, synth
n = int(sys.argv[1]); tiles = int(sys.argv[2])
eng = Engine(0)
inp, _ = synth.msm_bn256_tile_inputs(n, tile=0, cheap_points=True)
prog = Program.msm_bn256_tile(n)
d = eng.upload_inputs(prog, np.stack([inp] * tiles))
b, r, s, st = eng.alloc(prog, tiles)
eng.run(prog, d, b, r, s, st); torch.cuda.synchronize()
orun = oracle_lib.run_msm_bn256_tile(n, inp, threads=os.cpu_count())
L = prog.launches()[-2]
b0, db, ns = L['base0'], L['dbase'], L['n_strands']
ovals, _ = orun.adv(0, prog.base_rows)
got = b[tiles - 1].cpu().numpy().view(np.uint64)
# cells of the windows segment that are non-zero on the GPU (= stored by the replay) and differ from the oracle
seg = slice(b0, b0 + db * ns)
g, o = got[seg], ovals[seg]
nz = (g != 0).any(axis=2)
bad = nz & (g != o).any(axis=2)
print("replay-stored cells", int(nz.sum()), "wrong", int(bad.sum()))
idx = np.argwhere(bad)
for (rr, cc) in idx[:5].tolist():
    print("row", rr + b0, "strand", rr // db, "rel row", rr % db, "col", cc, [hex(int(x)) for x in g[rr, cc]], [hex(int(x)) for x in o[rr, cc]])
if len(idx): print("strands with bad cells", np.unique(idx[:, 0] // db)[:20], "rel rows", np.unique(idx[:, 0] % db)[:20])
print("status", np.unique(st.cpu().numpy()))
t0 = b0 + db * ns
gt, ot = got[t0:], ovals[t0:]
badt = np.argwhere((gt != ot).any(axis=2))
print("tail rows", gt.shape[0], "bad cells", len(badt), "first", (badt[:2] + [t0, 0]).tolist())
