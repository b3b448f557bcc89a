"""side measurement: h2e_export_columns on the base array of 16 x 1024-point MSM tiles (HBM-bound transpose)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halo2ecc_s_amd import Engine
eng = Engine(0)
for cols, rows in ((5, 6471309), (3, 6826643), (2, 470000)):
    x = torch.zeros((16, rows, cols, 4), dtype=torch.int64, device="cuda")
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = eng.export_columns(x); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"cols {cols}: {x.numel() * 8 / 1e9:.1f} GB in {dt * 1e3:.1f} ms = {2 * x.numel() * 8 / dt / 1e12:.2f} TB/s (read + write)")
    del x, y
