"""run 64 bn256 pairing checks once through the stamped build and print cycles per round kind (exp/wave_stamps.sh)"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["H2E_LIB"] = os.path.join(ROOT, "exp/_dbg/libh2e_stamps.so")
from halo2ecc_s_amd import Engine, Program, synth   # noqa: E402
from halo2ecc_s_amd.engine import lib   # noqa: E402
import torch   # noqa: E402
units = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = Engine(0)
prog = Program.pairing_check_bn256(emit_shape=False)
ins = np.stack([synth.pairing_check_bn256_inputs(instance=t) for t in range(units)])
d_in = eng.upload_inputs(prog, ins)
arrs = eng.alloc(prog, units)
for rep in range(3):
    arrs[3].zero_()
    torch.cuda.synchronize()
    eng.run(prog, d_in, *arrs)
    torch.cuda.synchronize()
assert int(arrs[3].abs().max()) == 0
buf = (C.c_ulonglong * 32)()
lib().h2e_engine_wave_stamps_fp0.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib().h2e_engine_wave_stamps_fp0(buf) == 0
names = ["light / linear combinations", "medium / loads", "mul", "div", "through cells", "chunk switch", "-", "-"]   # (field chain: kinds 0-3)
tot = sum(buf[k] for k in range(8))
for k in range(6):
    if buf[8 + k]:
        print(f"{names[k]:24s} rounds {buf[8 + k]:6d}  cycles {buf[k]:10d}  per round {buf[k] / buf[8 + k]:8.0f}  ({100.0 * buf[k] / tot:.1f} %)")
print("total cycles", tot, "=", tot / 100e6 * 1e3, "ms at the 100 MHz s_memtime clock" if False else "")
n_light = max(1, buf[8])
print("(digits kernel, wave 0) light rounds: header %.0f, records %.0f, barrier %.0f cycles per round; product rounds: barrier %.0f" % (buf[16] / n_light, buf[17] / n_light, buf[18] / n_light, buf[19] / max(1, buf[10])))
print("(lane kernels) light rounds, lane 0: header %.0f, record %.0f, op %.0f, barrier %.0f cycles per round" % tuple(buf[16 + k] / n_light for k in range(4)))
if buf[20]:
    print("(digits kernel, wave 0) a linear combination: digits in %.0f, columns %.0f, carry resolve %.0f, quotient estimate %.0f, q w and subtraction %.0f, conditional subtraction %.0f cycles (%d records)"
          % tuple([buf[21 + k] / buf[20] for k in range(6)] + [buf[20]]))
