"""run pairing checks once through the stamped build (exp/wave_stamps.sh) and print where the digit chain's cycles go:
   python exp/wave_stamps.py [units] [bn256|bls12_381]"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import halo2ecc_s_amd.engine as E   # noqa: E402
E.lib_path = lambda: os.path.join(ROOT, "exp/_dbg/libh2e_stamps.so")   # (the product has no library switch: exp builds patch the loader)
from halo2ecc_s_amd import Engine, Program, synth   # noqa: E402
from halo2ecc_s_amd.engine import lib   # noqa: E402
import torch   # noqa: E402
units = int(sys.argv[1]) if len(sys.argv) > 1 else 64
curve = sys.argv[2] if len(sys.argv) > 2 else "bn256"
eng = Engine(0)
if curve == "bn256":
    prog = Program.pairing_check_bn256(emit_shape=False)
    ins = np.stack([synth.pairing_check_bn256_inputs(instance=t) for t in range(units)])
    acc = "h2e_engine_wave_stamps_fp0"
else:
    prog = Program.pairing_check_bls12_381(emit_shape=False)
    ins = np.stack([synth.pairing_check_bls12_381_inputs(instance=t) for t in range(units)])
    acc = "h2e_engine_wave_stamps_fp1"
d_in = eng.upload_inputs(prog, ins)
arrs = eng.alloc(prog, units)
for rep in range(3):
    arrs[3].zero_()
    torch.cuda.synchronize()
    eng.run(prog, d_in, *arrs)
    torch.cuda.synchronize()
assert int(arrs[3].abs().max()) == 0
buf = (C.c_ulonglong * 128)()
fn = getattr(lib(), acc)
fn.argtypes = [C.POINTER(C.c_ulonglong)]
assert fn(buf) == 0
names = ["light / linear combinations", "medium / loads", "mul", "div", "through cells", "chunk switch", "-", "-"]   # (field chain: kinds 0-3)
tot = sum(buf[k] for k in range(8))
print(f"{curve}, {units} checks; stamps of the LAST launched chain (the final exponentiation's), workgroup 0; s_memtime ticks (100 MHz: x ~24 = shader cycles)")
for k in range(6):
    if buf[8 + k]:
        print(f"{names[k]:24s} rounds {buf[8 + k]:6d}  ticks {buf[k]:10d}  per round {buf[k] / buf[8 + k]:8.1f}  ({100.0 * buf[k] / tot:.1f} %)")
print("total ticks", tot, "=", tot / 100e6 * 1e3, "ms at 100 MHz")
n_light = max(1, buf[8])
print("(wave 0) light rounds: header %.1f, records %.1f, barrier %.1f ticks per round" % (buf[16] / n_light, buf[17] / n_light, buf[18] / n_light))
if buf[20]:
    print("(wave 0) a linear combination: digits in %.1f, columns %.1f, carry resolve %.1f, quotient estimate %.1f, q w and subtraction %.1f, rest %.1f ticks (%d records)"
          % tuple([buf[21 + k] / buf[20] for k in range(6)] + [buf[20]]))
print("per computing wave: rounds with records, ticks per such round in its records, barrier wait per round (all rounds), header part per round")
n_rounds = max(1, sum(buf[104 + k] for k in range(8)))
for w in range(15):
    b = buf[32 + 4 * w: 36 + 4 * w]
    print(f"  wave {w:2d}: {b[1]:5d} rounds, {b[0] / max(1, b[1]):7.1f} in records, barrier {b[2] / n_rounds:7.1f}, header {b[3] / n_rounds:6.1f}")
print("whole rounds by record count (<= 4, 8, 16, 24, 32, 40, 48, more): rounds / ticks per round")
for k in range(8):
    if buf[104 + k]:
        print(f"  bucket {k}: {buf[104 + k]:5d} rounds, {buf[96 + k] / buf[104 + k]:7.1f} ticks")
