"""print one line per bench JSON of a directory: ms per step, single-batch latency, the value-chain and expansion brackets per launch
   python exp/bench_lines.py gpurun_out/<dir>"""
import glob
import json
import sys
for f in sorted(glob.glob(sys.argv[1].rstrip("/") + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1][:-5], "ms/step %.3f single %s chain %s x %s" % (d["ms_per_step"], d.get("single_batch_ms") and round(d["single_batch_ms"], 3),
              [round(x, 2) for x in r["value_chain_ms"] if x > 0.3], [round(x, 2) for x in r["expansion_ms"] if x > 0.3]))
    except Exception as e:   # noqa: BLE001
        err = f[:-5] + ".err"
        print(f, "failed", e, open(err).read()[-300:] if __import__("os").path.exists(err) else "")
