"""Measurements of the SURVEY 8(f) "next" rows on one MI355X (side numbers for DESIGN.md 7b-d; not the headline bench):
  8f-1  h2e_export of a 64-tile MSM batch (NEXT_ROWS_TILES): rows / columns layout, canonical / Montgomery form (HBM read + write)
  8f-2  general-scalar MSM (bls12_381 G1 over bn256 Fr, the reference's 50-point shape), 64 instances per run
  8f-3  the operator API: msm_unsafe as an op on a device-resident context vs the same tile as a whole program
  8f-4  shape artefacts on the device: fixed columns, range lookup table, copy constraints
Writes one JSON line; run on the GPU box from the repo root:  python exp/next_rows_bench.py > gpurun_out/next_rows.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from halo2ecc_s_amd import Engine, Program, Records, synth  # noqa: E402
from halo2ecc_s_amd import engine as E  # noqa: E402

eng = Engine(0)
dev = "cuda:0"
free_b, _ = torch.cuda.mem_get_info(0)
scratch = torch.empty((int(free_b * 0.9) // 8,), dtype=torch.int64, device=dev)   # first touch (see bench.py)
scratch.fill_(-1)
torch.cuda.synchronize()
del scratch
torch.cuda.empty_cache()
out = {}


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


# ---- 8f-1: export ------------------------------------------------------------------------------------------------
n, tiles = 1024, int(os.environ.get("NEXT_ROWS_TILES", "64"))
prog = Program.msm_bn256_tile(n)
ins = np.stack([synth.msm_bn256_tile_inputs(n, tile=900 + k, cheap_points=True, with_expected=False)[0] for k in range(tiles)])
d_in = eng.upload_inputs(prog, ins)
arrs = eng.alloc(prog, tiles)
eng.run(prog, d_in, *arrs)
torch.cuda.synchronize()
exp = {}
for region, name in ((0, "base"), (1, "range")):
    batch = arrs[region]
    buf = None
    for layout, lname in ((E.LAYOUT_ROWS, "rows"), (E.LAYOUT_COLUMNS, "columns")):
        for form, fname in ((E.FORM_CANONICAL, "canonical"), (E.FORM_MONTGOMERY, "montgomery")):
            shape = (tiles, batch.shape[0], batch.shape[1], 4) if layout == E.LAYOUT_ROWS else (tiles, batch.shape[1], batch.shape[0], 4)
            if buf is None or tuple(buf.shape) != shape:
                buf = torch.empty(shape, dtype=batch.dtype, device=dev)
            dt = timed(lambda: eng.export(prog, region, batch, layout=layout, form=form, out=buf), reps=3, warm=1)
            gb = 2 * batch.numel() * 8 / 1e9   # every cell read (or masked) and written
            exp[f"{name}_{lname}_{fname}"] = {"ms": round(dt * 1e3, 2), "GB_moved": round(gb, 2), "TB_per_s": round(gb / dt / 1e3, 2)}
    del buf
out[f"export_{tiles}_tiles"] = exp
dg = {}
total = 0.0
for region, name in ((0, "base"), (1, "range"), (2, "select")):
    dout = torch.zeros((tiles, 4), dtype=torch.int64, device=dev)
    dt = timed(lambda: eng.digest(prog, region, arrs[region], out=dout), reps=3, warm=1)
    flags = (prog.base_flags, prog.range_flags, prog.select_flags)[region]()
    assigned = int((np.asarray(flags) & 1).sum())
    gb = assigned * tiles * 32 / 1e9
    total += dt
    dg[name] = {"ms": round(dt * 1e3, 2), "GB_read": round(gb, 2), "TB_per_s": round(gb / dt / 1e3, 2)}
dg["all_ms"] = round(total * 1e3, 2)
out[f"digest_{tiles}_tiles"] = dg
cc = timed(lambda: eng.export_copy_constraints(prog), reps=5)
out["copy_constraints"] = {"n": prog.n_permutations, "ms": round(cc * 1e3, 3), "G_pairs_per_s": round(prog.n_permutations / cc / 1e9, 2)}
fx = {}
for region, name in ((0, "base"), (1, "range"), (2, "select")):
    dt = timed(lambda: eng.export_fixed(prog, region, 1, d_in[:1], layout=E.LAYOUT_COLUMNS, form=E.FORM_MONTGOMERY), reps=3, warm=1)
    rows = (prog.base_rows, prog.range_rows, prog.select_rows)[region]
    gb = rows * (9, 2, 2)[region] * 32 / 1e9
    fx[name] = {"ms": round(dt * 1e3, 2), "GB_written": round(gb, 2), "TB_per_s": round(gb / dt / 1e3, 2)}
out["fixed_columns_montgomery"] = fx
rt = timed(lambda: eng.range_table(form=E.FORM_MONTGOMERY), reps=10)
out["range_table"] = {"rows": 524287, "ms": round(rt * 1e3, 3)}
del arrs, d_in, batch
torch.cuda.empty_cache()

# ---- 8f-2: general-scalar MSM ------------------------------------------------------------------------------------
npts, inst = 50, 64
gprog = Program.msm_bls12_381_tile(npts)
gins = np.stack([synth.msm_bls12_381_tile_inputs(npts, tile=k)[0] for k in range(inst)])
gd = eng.upload_inputs(gprog, gins)
garr = eng.alloc(gprog, inst)
dt = timed(lambda: eng.run(gprog, gd, *garr), reps=5, warm=2)
assert int(garr[3].abs().max()) == 0
cells = gprog.n_advice_cells * inst
out["general_scalar_msm_bls12_381"] = {"points": npts, "instances": inst, "ms_per_run": round(dt * 1e3, 2), "advice_cells_per_run": cells,
                                       "cells_per_s": round(cells / dt), "GB_per_s": round(cells * 32 / dt / 1e9, 1), "mode": "h2e_run, sequential"}
del garr, gd
torch.cuda.empty_cache()

# ---- 8f-3: operator API -------------------------------------------------------------------------------------------
n, inst = 256, 16
ins = np.stack([synth.msm_bn256_tile_inputs(n, tile=950 + k, cheap_points=True)[0] for k in range(inst)])
wprog = Program.msm_bn256_tile(n, emit_shape=False)
rows = (wprog.base_rows + 64, wprog.range_rows + 64, wprog.select_rows + 64)
d_in = eng.upload_inputs(wprog, ins)
warr = eng.alloc(wprog, inst)
dt_prog = timed(lambda: eng.run(wprog, d_in, *warr), reps=5, warm=2)


def ops_once():
    rec = Records(eng, E.FIELD_BN256_FQ, inst, rows, emit_shape=False)
    pts = rec.assign_points(n, ins[:, 0:3 * n])
    scs = rec.assign_scalars(n, ins[:, 3 * n:4 * n])
    rec.msm_unsafe(pts, scs, ins[:, 4 * n:4 * n + 6])
    torch.cuda.synchronize()
    st = int(rec.arrays()[3].abs().max())
    rec.close()
    return st


ops_once()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    assert ops_once() == 0
dt_ops = (time.perf_counter() - t0) / reps
out["operator_api_msm_256_points_x16"] = {"whole_program_ms": round(dt_prog * 1e3, 2), "three_ops_ms_incl_recording_and_context_setup": round(dt_ops * 1e3, 2)}
print(json.dumps(out))
