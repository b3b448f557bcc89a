#!/bin/bash
# round 4, call 8: assigned-only column export (consumer-ready step), the new tests, and a kernel timeline of the pipelined MSM step
cd "$(dirname "$0")/.."
O=gpurun_out/r4_8; mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "export or without_the_select or general_scalars" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 900 python bench.py --sub --suite main --workload msm --ring 1 --steps 3 --warmup 1 --latency-steps 0 --consumer-ready 3 --no-cpu-baseline --traffic off > $O/consumer.json 2> $O/consumer.err
python -c "
import json; d=json.loads(open('$O/consumer.json').read().strip().splitlines()[-1]); print('consumer_ready_ms_per_step', d['consumer_ready_ms_per_step'], 'ms_per_step', d['ms_per_step'])" || tail -3 $O/consumer.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/stats_msm -o run --output-format csv -- python3 bench.py --sub --suite main --workload msm --no-cpu-baseline --traffic off --steps 10 --warmup 4 --latency-steps 0 > $O/stats_msm.log 2>&1
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r4_8/stats_msm/run_kernel_trace.csv")))
rows = [r for r in rows if "h2e_" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = max(int(x["Grid_Size_X"]) for x in rows if "h2e_run_tape" in x["Kernel_Name"])
first_big = min(g for g in set(int(x["Grid_Size_X"]) for x in rows if "h2e_run_tape" in x["Kernel_Name"]) if g > big * 0.7)
wins = [i for i, x in enumerate(rows) if int(x["Grid_Size_X"]) == first_big and "h2e_run_tape" in x["Kernel_Name"]]
lo, hi = wins[14], wins[16]
t0 = int(rows[lo]["Start_Timestamp"])
with open("gpurun_out/r4_8/msm_timeline.txt", "w") as f:
    for x in rows[lo:hi + 1]:
        s, e = int(x["Start_Timestamp"]), int(x["End_Timestamp"])
        f.write(f"{(s - t0) / 1e6:8.3f} {(e - t0) / 1e6:8.3f} {(e - s) / 1e6:8.3f}  grid={int(x['Grid_Size_X'])//64} q={x['Queue_Id']} s={x['Stream_Id']} {x['Kernel_Name'].replace('void ', '')[:44]}\n")
print("timeline rows", hi - lo + 1, "two steps span ms", (int(rows[hi]["Start_Timestamp"]) - t0) / 1e6)
PY
grep '"metric"' $O/stats_msm.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('traced run ms/step', round(d['ms_per_step'],2))"
