#!/bin/bash
# where does the host spend a pipelined submission?  debug build's launch log with host clocks (H2E_DEBUG_LOG), 8 bn256 checks, ring 8, no host waits
cd "$(dirname "$0")/.."
O=gpurun_out/host_gaps; mkdir -p $O; rm -f $O/log.txt
cp exp/_dbg/libh2e_dbg.so halo2ecc_s_amd/libh2e.so
H2E_DEBUG_LOG=$O/log.txt python exp/submit_host_time.py bn256 8 ${RING:-8} 0 2>&1 | tail -1
python - <<'PY'
import re
lines = open("gpurun_out/host_gaps/log.txt").read().splitlines()
ev = []
for l in lines:
    m = re.search(r"host_us ([0-9.]+)", l)
    r = re.match(r"run (\d+)", l)
    if m and r:
        ev.append((int(r.group(1)), float(m.group(1)), l[:70]))
runs = sorted({r for r, _, _ in ev})
mid = runs[len(runs) * 2 // 3]
for run in (mid, mid + 1):
    e = [x for x in ev if x[0] == run]
    print("run", run, "lines", len(e), "total host us", round(e[-1][1] - e[0][1], 1))
    for a, b in zip(e, e[1:]):
        if b[1] - a[1] > 20.0:
            print("   %8.1f us after: %s" % (b[1] - a[1], a[2]))
# between the end of one submission and the first line of the next
first = {}
last = {}
for r, t, l in ev:
    first.setdefault(r, t)
    last[r] = t
print("between submissions:", [round(first[r + 1] - last[r], 1) for r in runs[len(runs) // 2:len(runs) // 2 + 8] if r + 1 in first])
PY
