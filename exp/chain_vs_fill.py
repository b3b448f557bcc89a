"""What does the pairings' value chain feel next to an expansion: the memory system, or something the expansion kernel itself does?
One bn256 batch at a time (H2E_PAIRING_SPLITS=0: one chain of 2 261 rounds), its chain bracket (a) alone, (b) while a torch fill kernel -
tiny code, the same HBM write stream - runs on another stream, (c) while a torch copy (reads + writes) runs."""
import os
import sys
import numpy as np
os.environ.setdefault("H2E_PAIRING_SPLITS", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halo2ecc_s_amd import Engine, Program, synth

units = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = Engine(0)
eng.set_option(6, 60)
prog = Program.pairing_check_bn256(emit_shape=False)
ins = np.stack([synth.pairing_check_bn256_inputs(instance=k) for k in range(units)])
d_in = eng.upload_inputs(prog, ins)
base, rng, sel, status = eng.alloc(prog, units)
eng.set_profiling(True)
big = torch.empty(int(24e9) // 8, dtype=torch.int64, device="cuda")
big2 = torch.empty_like(big)
side = torch.cuda.Stream()


def measure(mode, reps=6):
    out = []
    for i in range(reps):
        status.zero_()
        torch.cuda.synchronize()
        if mode != "alone":
            with torch.cuda.stream(side):
                for _ in range(2):
                    if mode == "fill":
                        big.fill_(i)          # 24 GB of stores: ~4.5 ms
                    else:
                        big2.copy_(big)       # 24 GB read + 24 GB written
        eng.run(prog, d_in, base, rng, sel, status)
        torch.cuda.current_stream().synchronize()
        ms = eng.last_run_launch_ms()
        out.append((round(sum(a for a, _ in ms), 3), round(sum(b for _, b in ms), 3)))
        torch.cuda.synchronize()
        assert int(status.abs().max()) == 0
    return out[1:]


for mode in ("alone", "fill", "copy", "alone"):
    r = measure(mode)
    print(mode, "chain ms", [a for a, _ in r], "expansion ms", [b for _, b in r], flush=True)
