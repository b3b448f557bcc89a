"""per kernel of the consumer-ready run: dispatches, HBM bytes written / fetched per dispatch (largest dispatch and total), from the two --pmc passes"""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
tot = {}
for counter, scale in (("WRITE_SIZE", 1024.0), ("FETCH_SIZE", 2 * 1024.0)):   # KiB -> bytes; FETCH_SIZE counts 128-byte requests as 64 on gfx950
    files = glob.glob(os.path.join(out, "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(counter, "pass: no counter file")
        continue
    per = defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        per[name].append((int(row["Dispatch_Id"]), float(row["Counter_Value"]) * scale))
    tot[counter] = per
names = sorted(set(tot.get("WRITE_SIZE", {})) | set(tot.get("FETCH_SIZE", {})), key=lambda n: -sum(v for _, v in tot.get("WRITE_SIZE", {}).get(n, [])))
print("source hash", open(os.path.join(out, "source_hash.txt")).read().strip())
print("bench.py --workload msm --ring 1 --consumer-ready 1 (64 tiles x 1024 points): per kernel, over the whole process")
print(f"{'kernel':60s} {'calls':>6s} {'written GB':>11s} {'fetched GB':>11s} {'largest dispatch: written / fetched GB':>40s}")
for n in names[:14]:
    w = tot.get("WRITE_SIZE", {}).get(n, [])
    f = tot.get("FETCH_SIZE", {}).get(n, [])
    wl = max((v for _, v in w), default=0.0)
    fl = max((v for _, v in f), default=0.0)
    print(f"{n[:60]:60s} {max(len(w), len(f)):6d} {sum(v for _, v in w) / 1e9:11.2f} {sum(v for _, v in f) / 1e9:11.2f} {wl / 1e9:20.2f} / {fl / 1e9:.2f}")
