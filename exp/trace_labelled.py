"""label a rocprofv3 kernel trace with the debug build's launch log (exp/trace_labelled.sh): both are in the host's enqueue order.
   python exp/trace_labelled.py run_kernel_trace.csv launch_log.txt [runs to print, from the middle of the trace]
prints, per kernel of those runs: start / end / duration (ms, relative to the first), run, slot, segment, what, stream role, grid"""
import csv
import re
import sys

trace, log, n_show = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = [r for r in csv.DictReader(open(trace)) if "h2e_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Correlation_Id"]))   # the host's API call order (Dispatch_Id is the order on the device)
PRE = {1: ("candidates", ["h2e_predict<"], ["h2e_finalize_hints"]), 2: ("windows", ["h2e_predict_windows"], ["h2e_finalize_ecc"]),
       3: ("tail", ["h2e_predict_tail"], ["h2e_finalize_ecc"]), 4: ("select", ["h2e_select"], []),
       5: ("field_chain", ["h2e_field_chain"], ["h2e_field_finalize", "h2e_field_sinks"])}
entries = []
streams = {}
slot_of_run = {}
for ln in open(log):
    t = ln.split()
    if t[2] == "begin":
        run = int(t[1])
        slot_of_run[run] = int(t[4])
        names = ["chain", "expand", "side", "fixup", "small"]
        for nm, ptr in zip(names, [t[9], t[11], t[13], t[15], t[17]]):
            if ptr not in ("(nil)", "0"):
                streams.setdefault(ptr, nm + (str(slot_of_run[run]) if nm in ("chain", "side") else ""))
        continue
    run, seg, what, a = int(t[1]), int(t[3]), t[4], int(t[6])
    b, c, d, st = int(t[8]), int(t[10]), int(t[12]), t[14]
    if what == "launch":
        pats = {1: ["h2e_replay", "h2e_hint_store", "h2e_run_tape<"], 2: ["h2e_run_tape"], 4: ["h2e_fixup_inverses"]}[a]
        label = {1: "values", 2: "expansion", 4: "fixup"}[a] + f" ops={b} strands={c} n={d}"
    else:
        nm, p1, p2 = PRE.get(b, ("pre%d" % b, ["h2e_"], ["h2e_"]))
        pats = (p1 if a & 1 else []) + (p2 if a & 2 else [])
        label = f"{nm} phase={a}"
    entries.append((run, seg, label, pats, st))
out = []
k = 0
misc = 0
for run, seg, label, pats, st in entries:
    took = 0
    while k < len(rows):
        name = rows[k]["Kernel_Name"]
        if any(p in name for p in pats):
            out.append((rows[k], run, seg, label, st))
            took += 1
            k += 1
            if took >= 4:
                break
            continue
        if took == 0 and not any(x in name for x in ("h2e_run_tape", "h2e_replay", "h2e_hint_store", "h2e_fixup", "h2e_predict", "h2e_finalize", "h2e_select", "h2e_field")):
            out.append((rows[k], -1, -1, "misc", "?"))   # gate, status, digest reduce, unit records ...
            misc += 1
            k += 1
            continue
        break
    if took == 0:
        print(f"unmatched log entry run {run} seg {seg} {label} at trace row {k}: {rows[k]['Kernel_Name'][:50] if k < len(rows) else 'end'}", file=sys.stderr)
print(f"{len(entries)} log entries, {len(rows)} h2e kernels, matched {len(out) - misc}, misc {misc}, left {len(rows) - k}", file=sys.stderr)
runs = sorted({r for _, r, _, _, _ in out if r >= 0})
mid = runs[len(runs) // 2 - 1: len(runs) // 2 - 1 + n_show] if len(runs) > n_show else runs
sel = [o for o in out if o[1] in mid]
t0 = min(int(o[0]["Start_Timestamp"]) for o in sel)
sel.sort(key=lambda o: int(o[0]["Start_Timestamp"]))
print("# start end dur(ms) run slot seg stream grid(workgroups) kernel | label")
for r, run, seg, label, st in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    kn = re.sub(r"void |<.*", "", r["Kernel_Name"])[:26]
    print(f"{(s - t0) / 1e6:8.3f} {(e - t0) / 1e6:8.3f} {(e - s) / 1e6:7.3f} run {run:3d} slot {slot_of_run.get(run, -1)} seg {seg:2d} {streams.get(st, st):8s} wg={wg:6d} {kn:26s} | {label}")
