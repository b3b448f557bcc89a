"""copy the judged summaries of gpurun_out/<tag> (exp/r6_profiles.sh) into profiles/:
   <tag>_bench.json                       the default `python bench.py` line (children under "also", PMC traffic, CPU baseline, summary)
   <tag>_<wl>_kernel_stats.csv            rocprofv3 --kernel-trace --stats of `bench.py --sub --suite main --workload <wl> ...`
   <tag>_<wl>_dominant_kernel.json        per-dispatch durations of the dominant kernel(s) from that trace vs the bench's HIP events
   <tag>_msm_timeline.txt                 per-dispatch start / end of two pipelined steps in the middle of the timed region
   <tag>_spread.json                      ms_per_step / single_batch_ms of three fresh processes per headline workload"""
import csv, glob, json, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r6_p"
src = f"gpurun_out/{tag}"
# a set that was not taken from the tree as it is is not published (round 5's review: profiles of a kernel that no longer shipped)
import subprocess
have = subprocess.run([sys.executable, "exp/source_hash.py"], capture_output=True, text=True).stdout.strip()
want = open(f"{src}/source_hash.txt").read().strip()
if have != want and "--force" not in sys.argv:
    sys.exit(f"publish: {src} was taken from sources {want[:16]}, the tree is {have[:16]} - re-run exp/r6_profiles.sh (or --force)")
headline = open(f"{src}/bench.json").read().strip().splitlines()[-1]
assert len(headline) < 4096
open(f"profiles/{tag}_bench_headline.json", "w").write(headline + "\n")   # the driver's record: the LAST stdout line of `python bench.py`
d = json.load(open(f"{src}/bench_detail.json"))
json.dump(d, open(f"profiles/{tag}_bench.json", "w"), indent=1)
open(f"profiles/{tag}_source_hash.txt", "w").write(want + "\n")
print("bench:", d["ms_per_step"], "whole", d["whole_step"]["frac"], "dom", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"))
print(json.dumps(d["summary"]))


def bench_line(wl):
    line = [ln for ln in open(f"{src}/stats_{wl}.log").read().split("\n") if ln.startswith('{"metric"')][0]
    return json.loads(line)


def trace(wl):
    rows = list(csv.DictReader(open(f"{src}/stats_{wl}/run_kernel_trace.csv")))
    rows = [r for r in rows if "h2e_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


dur = lambda x: (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6   # noqa: E731
shutil.copy(f"{src}/stats_consumer_ready/run_kernel_stats.csv", f"profiles/{tag}_consumer_ready_kernel_stats.csv")
cr = bench_line("consumer_ready")
json.dump({"two_pass_montgomery_ms": cr["consumer_ready_ms_per_step"], "first_pass": {k: v for k, v in cr["consumer_ready_first_pass"].items() if k != "launch_ms"},
           "first_pass_launch_ms_last": cr["consumer_ready_first_pass"].get("launch_ms", [None])[-1], "note": "under rocprofv3 --kernel-trace"},
          open(f"profiles/{tag}_consumer_ready.json", "w"), indent=1)
traffic = {}
for f in sorted(glob.glob(f"{src}/traffic_*.json")):
    b = json.loads([ln for ln in open(f).read().split("\n") if ln.startswith('{"metric"')][-1])
    x = b["roofline"].get("expansion", b["roofline"])
    traffic[f.split("traffic_")[1][:-5]] = {"ms_per_step": b["ms_per_step"], "kernel": x["kernel"][:60], "algorithmic_bytes_per_launch": x["algorithmic_bytes_per_launch"],
                                            "traffic_bytes_per_launch": x.get("traffic"), "traffic_over_algorithmic": (x["traffic"] / x["algorithmic_bytes_per_launch"]) if x.get("traffic") else None,
                                            "traffic_detail": x.get("traffic_detail"), "note": x.get("traffic_note")}
json.dump({"what": "HBM traffic of the packed expansion's dominant launch (two rocprofv3 --pmc child passes of the bench command, WRITE_SIZE + 2 x FETCH_SIZE)", "runs": traffic},
          open(f"profiles/{tag}_packed_traffic.json", "w"), indent=1)
print("packed traffic:", {k: v["traffic_over_algorithmic"] for k, v in traffic.items()})
for wl in ["msm", "pairing_bn256", "pairing_bls12_381", "pairing_bn256_ring1", "pairing_bls12_381_ring1", "job"]:
    shutil.copy(f"{src}/stats_{wl}/run_kernel_stats.csv", f"profiles/{tag}_{wl}_kernel_stats.csv")
    b = bench_line(wl)
    r = b["roofline"]
    rows = trace(wl)
    out = {"command": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --sub --suite main ... ({wl}; exp/r6_profiles.sh)",
           "bench_ms_per_step_same_command": b["ms_per_step"], "bench_steps": b["steps"]}
    if wl.startswith("pairing"):
        ch = [dur(x) for x in rows if "h2e_field_chain_digits" in x["Kernel_Name"]]
        xs = [dur(x) for x in rows if "h2e_run_tape" in x["Kernel_Name"] and "h2e_hint_store" not in x["Kernel_Name"] and int(x["Grid_Size_X"]) > 64 * 64]
        n_seg = max(1, sum(1 for v in r["value_chain_ms"] if v > 0.05))
        ch, xs = ch[-b["steps"] * n_seg:], xs[-b["steps"] * n_seg:]
        out.update(kernel="h2e_field_chain_digits (the value chain: one dispatch per launch of the check) and h2e_run_tape / h2e_run_tape_packed (the expansion); the timed steps' dispatches",
                   launches_per_check=n_seg, chain_dispatch_avg_ms=sum(ch) / max(1, len(ch)), chain_ms_per_check=sum(ch) / b["steps"],
                   expansion_dispatch_avg_ms=sum(xs) / max(1, len(xs)), expansion_ms_per_check=sum(xs) / b["steps"],
                   bench_events_value_chain_ms=r["value_chain_ms"], bench_events_expansion_ms=r["expansion_ms"],
                   note="the bench's value-chain brackets also hold h2e_field_finalize, h2e_field_sinks and h2e_hint_store")
    else:
        nl = r["launches_per_step"]
        xs = [(int(x["Grid_Size_X"]), dur(x)) for x in rows if "h2e_run_tape" in x["Kernel_Name"] and "false" in x["Kernel_Name"]]
        big = sorted(set(g for g, _ in xs))[-nl:]
        dd = [(g, t) for g, t in xs if g in big]
        timed = dd[-nl * b["steps"]:]
        also = d["also"].get("msm_job_2e20", {})
        out.update(kernel=f"{r['kernel']}, grids {big}, {nl} launches per step", timed_dispatches_avg_ms=sum(t for _, t in timed) / len(timed),
                   timed_per_step_sum_ms=sum(t for _, t in timed) / b["steps"], bench_events_ms_same_command=r["launch_ms"],
                   bench_events_ms_unprofiled_run=d["roofline"]["launch_ms"] if wl == "msm" else also.get("roofline", {}).get("launch_ms"),
                   bench_ms_per_step_unprofiled_run=d["ms_per_step"] if wl == "msm" else also.get("ms_per_step"))
    json.dump(out, open(f"profiles/{tag}_{wl}_dominant_kernel.json", "w"), indent=1)
    print(wl, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items() if "avg" in k or "per_step" in k or "per_check" in k})

# timeline of two pipelined MSM steps in the middle of the timed region
rows = trace("msm")
big = max(int(x["Grid_Size_X"]) for x in rows if "h2e_run_tape" in x["Kernel_Name"])
first_big = min(g for g in set(int(x["Grid_Size_X"]) for x in rows if "h2e_run_tape" in x["Kernel_Name"]) if g > big * 0.7)
wins = [i for i, x in enumerate(rows) if int(x["Grid_Size_X"]) == first_big and "h2e_run_tape" in x["Kernel_Name"]]
lo, hi = wins[len(wins) // 2], wins[len(wins) // 2 + 2]
t0 = int(rows[lo]["Start_Timestamp"])
with open(f"profiles/{tag}_msm_timeline.txt", "w") as f:
    f.write("# start ms, end ms, duration ms, grid (workgroups), stream, kernel - two pipelined steps of `bench.py --workload msm` (three runs in flight: h2e_ring), from one\n"
            "# window-expansion launch to the one two steps later; stream 1 = the shared expansion stream, a chain and a side stream per job slot\n")
    for x in rows[lo:hi + 1]:
        s, e = int(x["Start_Timestamp"]), int(x["End_Timestamp"])
        f.write(f"{(s - t0) / 1e6:8.3f} {(e - t0) / 1e6:8.3f} {(e - s) / 1e6:8.3f}  grid={int(x['Grid_Size_X']) // 64} s={x['Stream_Id']} {x['Kernel_Name'].replace('void ', '')[:44]}\n")
print("timeline rows", hi - lo + 1, "span ms", (int(rows[hi]["Start_Timestamp"]) - t0) / 1e6)

spread = {}
for w in ("msm", "pairing_bn256", "pairing_bls12_381"):
    runs = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f"{src}/spread_{w}_*.json"))]
    spread[w] = {"ms_per_step": [round(x["ms_per_step"], 3) for x in runs], "single_batch_ms": [round(x["single_batch_ms"], 3) for x in runs],
                 "whole_step_frac": [round(x["whole_step"]["frac"], 4) for x in runs]}
json.dump({"what": "three fresh processes per headline workload on one box (exp/r6_profiles.sh)", "runs": spread}, open(f"profiles/{tag}_spread.json", "w"), indent=1)
print(spread)
