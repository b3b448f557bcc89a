"""The published profile set describes the code that ships: profiles/r6_p_source_hash.txt (written on the GPU box by exp/r6_profiles.sh
before it measured anything, copied by exp/publish_profiles_r6.py - which refuses a set whose hash is not the tree's) equals the
sha256 of the product's sources as they stand (exp/source_hash.py: csrc, include/h2e.h, the Python binding, bench.py).  A source edit
after the last profile run turns this red until the set is re-taken."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_published_profiles_were_taken_from_these_sources():
    tree = subprocess.run([sys.executable, os.path.join(ROOT, "exp", "source_hash.py")], capture_output=True, text=True, check=True).stdout.strip()
    with open(os.path.join(ROOT, "profiles", "r6_p_source_hash.txt")) as f:
        published = f.read().strip()
    assert published == tree, "profiles/r6_p_* were taken from other sources: re-run exp/r6_profiles.sh on the GPU and exp/publish_profiles_r6.py"


def test_published_headline_is_the_contract_line():
    """the line the driver would have parsed, as published: below 4 KB, the MSM's roofline names the expansion with the counters' traffic,
    the rocprofv3 stats of the same command agree with the live launch duration within 5 %"""
    with open(os.path.join(ROOT, "profiles", "r6_p_bench_headline.json")) as f:
        line = f.read().strip()
    assert len(line) < 4096 and "\n" not in line
    d = json.loads(line)
    r = d["roofline"]
    assert r["kernel"].startswith("h2e_run_tape<") and r["traffic"] and r["traffic"] >= 21.7e9   # (algorithmic bytes of a window launch: 21.717 GB)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and d["cpu_baseline"]["value"] > 0
    with open(os.path.join(ROOT, "profiles", "r6_p_msm_dominant_kernel.json")) as f:
        k = json.load(f)
    assert abs(k["timed_dispatches_avg_ms"] - r["launch_ms"]) / r["launch_ms"] < 0.05
