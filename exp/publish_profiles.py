"""copy the judged summaries of gpurun_out/<tag> (exp/collect_profiles.sh) into profiles/"""
import csv, json, shutil, subprocess, sys
tag = sys.argv[1]
src = f"gpurun_out/{tag}"
d = json.load(open(f"{src}/bench.json"))
print("bench:", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["launch_ms"])
BIG = 1000000   # the window expansion's dispatches (one, or two when split: h2e_capi.cpp `expand`) are the only h2e_run_tape grids this large
def pmc(path, name):
    got = []
    for r in csv.DictReader(open(path)):
        if 'h2e_run_tape' in r['Kernel_Name'] and 'false' in r['Kernel_Name'] and r['Counter_Name'] == name and int(r['Grid_Size']) > BIG:
            got.append((int(r['Grid_Size']), float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
    return got[-d["roofline"].get("launches_per_step", 1):]   # the last step's dispatches
w = pmc(f"{src}/pmc_wr/run_counter_collection.csv", 'WRITE_SIZE'); f = pmc(f"{src}/pmc_rd/run_counter_collection.csv", 'FETCH_SIZE')
out = {"kernel": "h2e_run_tape<FP_BN256_FQ,false> (expansion of the MSM window strands), the dispatches of one step, 64 tiles",
       "grids": [x[0] for x in w], "launches": len(w),
       "WRITE_SIZE_raw": sum(x[1] for x in w), "FETCH_SIZE_raw": sum(x[1] for x in f),
       "WRITE_SIZE_raw_per_dispatch": [x[1] for x in w], "FETCH_SIZE_raw_per_dispatch": [x[1] for x in f],
       "duration_ms_under_pmc": [[x[2] for x in w], [x[2] for x in f]],
       "note": "rocprofv3 --pmc, separate passes (exp/collect_profiles.sh), one step; *_raw = sum over the step's window-expansion dispatches; units KB as reported; FETCH_SIZE must be doubled on gfx950 for wide coalesced reads (MI355X_MICROARCH.md)",
       "file": f"{tag}_msm64x1024_hbm_pmc.json"}
json.dump(out, open(f"profiles/{tag}_msm64x1024_hbm_pmc.json", "w"), indent=1)
json.dump(out, open("profiles/hbm_pmc_latest.json", "w"), indent=1)
shutil.copy(f"{src}/stats/run_kernel_stats.csv", f"profiles/{tag}_msm64x1024_kernel_stats.csv")
shutil.copy(f"{src}/bench.json", f"profiles/{tag}_bench.json")
rows = list(csv.DictReader(open(f"{src}/stats/run_kernel_trace.csv")))
dd = [(int(r['Grid_Size_X']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6) for r in rows
      if 'h2e_run_tape' in r['Kernel_Name'] and 'false' in r['Kernel_Name'] and int(r['Grid_Size_X']) > BIG]
line = [l for l in open(f"{src}/stats.log").read().split('\n') if l.startswith('{"metric"')][0]
b2 = json.loads(line)
nl = len(w); timed = dd[-nl * b2["steps"]:]
json.dump({"kernel": f"h2e_run_tape<FP_BN256_FQ,false>, grids {sorted(set(g for g, _ in dd))} (expansion of the MSM window strands, 64 tiles, {nl} launch(es) per step)",
           "dispatch_ms": dd, "timed_dispatches_avg_ms": sum(t for _, t in timed) / len(timed),
           "timed_per_step_sum_ms": sum(t for _, t in timed) / b2["steps"],
           "source": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline (exp/collect_profiles.sh); the last {nl * b2['steps']} dispatches are the timed steps",
           "bench_events_ms_same_command": b2["roofline"]["launch_ms"], "bench_launches_per_step": b2["roofline"].get("launches_per_step", 1),
           "bench_ms_per_step_same_command": b2["ms_per_step"]},
          open(f"profiles/{tag}_msm64x1024_dominant_kernel.json", "w"), indent=1)
open(f"profiles/{tag}_msm64x1024_last_step_timeline.txt", "w").write(
    subprocess.run([sys.executable, "exp/timeline.py", f"{src}/stats/run_kernel_trace.csv", "6"], capture_output=True, text=True).stdout)
print(out["WRITE_SIZE_raw"], out["FETCH_SIZE_raw"], timed, b2["roofline"]["launch_ms"])
