"""copy the judged summaries of gpurun_out/<tag> (exp/r3_profiles.sh) into profiles/:
   <tag>_bench.json                       the default `python bench.py` line (children under "also", PMC traffic, CPU baseline)
   <tag>_<wl>_kernel_stats.csv            rocprofv3 --kernel-trace --stats of `bench.py --sub --suite main --workload <wl> ...`
   <tag>_<wl>_dominant_kernel.json        per-dispatch durations of the dominant kernel(s) from that trace vs the bench's HIP events
   <tag>_msm_timeline.txt                 per-dispatch start / end of two pipelined steps in the middle of the timed region"""
import csv, json, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r3_p"
src = f"gpurun_out/{tag}"
d = json.loads(open(f"{src}/bench.json").read().strip().splitlines()[-1])
json.dump(d, open(f"profiles/{tag}_bench.json", "w"), indent=1)
print("bench:", d["ms_per_step"], "whole", d["whole_step"]["frac"], "dom", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"))


def bench_line(wl):
    line = [ln for ln in open(f"{src}/stats_{wl}.log").read().split("\n") if ln.startswith('{"metric"')][0]
    return json.loads(line)


def trace(wl):
    rows = list(csv.DictReader(open(f"{src}/stats_{wl}/run_kernel_trace.csv")))
    rows = [r for r in rows if "h2e_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


for wl in ["msm", "pairing_bn256", "pairing_bls12_381", "pairing_bn256_ring1", "pairing_bls12_381_ring1", "job"]:
    shutil.copy(f"{src}/stats_{wl}/run_kernel_stats.csv", f"profiles/{tag}_{wl}_kernel_stats.csv")
    b = bench_line(wl)
    r = b["roofline"]
    rows = trace(wl)
    out = {"command": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --sub --suite main --workload ... ({wl}; exp/r3_profiles.sh)",
           "bench_ms_per_step_same_command": b["ms_per_step"], "bench_steps": b["steps"]}
    dur = lambda x: (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6   # noqa: E731
    if wl.startswith("pairing"):
        ch = [dur(x) for x in rows if "h2e_field_chain_digits" in x["Kernel_Name"]][-b["steps"]:]
        xs = [dur(x) for x in rows if "h2e_run_tape" in x["Kernel_Name"] and "false" in x["Kernel_Name"]][-b["steps"]:]
        out.update(kernel="h2e_field_chain_digits (the value chain) and h2e_run_tape<.., false> (the expansion); the timed steps' dispatches",
                   chain_dispatch_avg_ms=sum(ch) / len(ch), chain_dispatch_ms=ch, expansion_dispatch_avg_ms=sum(xs) / len(xs),
                   bench_events_value_chain_ms=r["value_chain_ms"], bench_events_expansion_ms=r["expansion_ms"],
                   note="the bench's value-chain bracket also holds h2e_field_finalize and h2e_hint_store (0.25-0.5 ms)")
    else:
        nl = r["launches_per_step"]
        xs = [(int(x["Grid_Size_X"]), dur(x)) for x in rows if "h2e_run_tape" in x["Kernel_Name"] and "false" in x["Kernel_Name"]]
        big = sorted(set(g for g, _ in xs))[-nl:]
        dd = [(g, t) for g, t in xs if g in big]
        timed = dd[-nl * (b["steps"] + (4 if wl == "msm" else 0)):][:nl * b["steps"]] if wl == "msm" else dd[-nl * b["steps"]:]
        out.update(kernel=f"{r['kernel']}, grids {big}, {nl} launches per step", timed_dispatches_avg_ms=sum(t for _, t in timed) / len(timed),
                   timed_per_step_sum_ms=sum(t for _, t in timed) / b["steps"], bench_events_ms_same_command=r["launch_ms"],
                   bench_events_ms_unprofiled_run=d["roofline"]["launch_ms"] if wl == "msm" else d["also"]["msm_job_2e20"]["roofline"]["launch_ms"],
                   bench_ms_per_step_unprofiled_run=d["ms_per_step"] if wl == "msm" else d["also"]["msm_job_2e20"]["ms_per_step"])
        if wl == "job":
            out["note"] = "the streaming job runs slower under the profiler (its per-step record kernels and copies are traced); the un-profiled figure is the one in <tag>_bench.json"
    json.dump(out, open(f"profiles/{tag}_{wl}_dominant_kernel.json", "w"), indent=1)
    print(wl, {k: v for k, v in out.items() if "avg" in k or "per_step" in k})

# timeline of two pipelined MSM steps: from the 12th window-expansion launch of the trace on (warm-up 4 steps + 8 timed steps in)
rows = trace("msm")
big = max(int(x["Grid_Size_X"]) for x in rows if "h2e_run_tape" in x["Kernel_Name"])
starts = [i for i, x in enumerate(rows) if int(x["Grid_Size_X"]) == big and "h2e_run_tape" in x["Kernel_Name"]]
first_big = min(g for g in set(int(x["Grid_Size_X"]) for x in rows if "h2e_run_tape" in x["Kernel_Name"]) if g > big * 0.7)
wins = [i for i, x in enumerate(rows) if int(x["Grid_Size_X"]) == first_big and "h2e_run_tape" in x["Kernel_Name"]]
lo, hi = wins[12], wins[14]
t0 = int(rows[lo]["Start_Timestamp"])
with open(f"profiles/{tag}_msm_timeline.txt", "w") as f:
    f.write("# start ms, end ms, duration ms, grid, queue, stream, kernel - two pipelined steps of `bench.py --workload msm` (ring 2), from one\n"
            "# window-expansion launch to the one two steps later; streams: one expansion stream, a chain and a side stream per job slot\n")
    for x in rows[lo:hi + 1]:
        s, e = int(x["Start_Timestamp"]), int(x["End_Timestamp"])
        f.write(f"{(s - t0) / 1e6:8.3f} {(e - t0) / 1e6:8.3f} {(e - s) / 1e6:8.3f}  grid={x['Grid_Size_X']} q={x['Queue_Id']} s={x['Stream_Id']} {x['Kernel_Name'].replace('void ', '')[:40]}\n")
print("timeline rows", hi - lo + 1)
