"""copy the judged summaries of gpurun_out/<tag> (exp/collect_profiles.sh) into profiles/:
   <tag>_bench.json                 the bench line of the un-profiled run (PMC traffic measured by its own child passes)
   <tag>_<wl>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command
   <tag>_<wl>_dominant_kernel.json  per-dispatch durations of the dominant kernel from that trace vs the bench's HIP events
   <tag>_<wl>_timeline.txt          per-dispatch start/end of the last two (pipelined) steps"""
import csv, json, shutil, subprocess, sys
tag = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else "msm64x1024"
src = f"gpurun_out/{tag}"
d = json.load(open(f"{src}/bench.json"))
r = d["roofline"]
print("bench:", d["ms_per_step"], d["value"], "whole", d["whole_step"]["frac"], "dom", r["frac"], r["launch_ms"], "traffic", r.get("traffic"))
shutil.copy(f"{src}/stats/run_kernel_stats.csv", f"profiles/{tag}_{wl}_kernel_stats.csv")
shutil.copy(f"{src}/bench.json", f"profiles/{tag}_bench.json")
line = [l for l in open(f"{src}/stats.log").read().split('\n') if l.startswith('{"metric"')][0]
b2 = json.loads(line)
r2 = b2["roofline"]
nl = r2["launches_per_step"]
rows = list(csv.DictReader(open(f"{src}/stats/run_kernel_trace.csv")))
xs = [(int(x['Grid_Size_X']), (int(x['End_Timestamp']) - int(x['Start_Timestamp'])) / 1e6) for x in rows
      if 'h2e_run_tape' in x['Kernel_Name'] and 'false' in x['Kernel_Name']]
big = sorted(set(g for g, _ in xs))[-nl:]              # the dominant launch's dispatches are the largest grids
dd = [(g, t) for g, t in xs if g in big]
timed = dd[-nl * b2["steps"]:]
json.dump({"kernel": f"{r2['kernel']}, grids {big}, {nl} launch(es) per step",
           "timed_dispatches_avg_ms": sum(t for _, t in timed) / len(timed),
           "timed_per_step_sum_ms": sum(t for _, t in timed) / b2["steps"],
           "dispatch_ms": dd,
           "source": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --workload ... --no-cpu-baseline --traffic off (exp/collect_profiles.sh); the last {nl * b2['steps']} dispatches are the timed steps",
           "bench_events_ms_same_command": r2["launch_ms"], "bench_launches_per_step": nl,
           "bench_ms_per_step_same_command": b2["ms_per_step"],
           "bench_events_ms_unprofiled_run": r["launch_ms"], "bench_ms_per_step_unprofiled_run": d["ms_per_step"]},
          open(f"profiles/{tag}_{wl}_dominant_kernel.json", "w"), indent=1)
open(f"profiles/{tag}_{wl}_timeline.txt", "w").write(
    subprocess.run([sys.executable, "exp/timeline.py", f"{src}/stats/run_kernel_trace.csv", str((b2["steps"] + b2["warmup"]) // 2)], capture_output=True, text=True).stdout)
print("trace avg", sum(t for _, t in timed) / len(timed), "events", r2["launch_ms"])
