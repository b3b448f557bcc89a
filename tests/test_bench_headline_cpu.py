"""The one stdout line of bench.py is what the driver parses - and it gives up on a long one: round 5's default invocation printed
25 KB and `BENCH_r05.json.parsed` was null.  The headline is built from the whole measurement by `bench.headline`; here it is
applied to that very record (profiles/r5_t_bench.json, every `also` child and traffic detail included) and must stay under 4 KB
with every key the contract names."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_headline_of_round5_record_fits_and_keeps_the_contract():
    import bench
    with open(os.path.join(ROOT, "profiles", "r5_t_bench.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    h = bench.headline(full, os.path.join(ROOT, "bench_detail.json"))
    line = json.dumps(h)
    assert len(line) < bench.HEADLINE_MAX_BYTES == 4096, len(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "single_batch_ms", "consumer_ready_ms_per_step", "summary"):
        assert key in h, key
    assert list(h)[-1] == "summary" and "workload" in h["config"] and "model" not in h["config"]
    r = h["roofline"]
    assert r["traffic"] is not None and r["traffic_source"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert not any(isinstance(v, (list, dict)) for v in r.values())          # scalars only: the per-segment arrays stay in the detail file
    cb = h["cpu_baseline"]
    assert all(k in cb for k in ("value", "unit", "cores", "kind", "sample"))
    assert set(h["also_steps"]) == set(full["also"])


def test_headline_keeps_the_expansion_beside_a_dominant_value_chain():
    import bench
    with open(os.path.join(ROOT, "profiles", "r5_t_bench.json")) as f:
        full = json.load(f)
    blk = full["also"]["pairing_bls12_381"]
    h = bench.headline(blk, "x")
    assert "expansion" in h["roofline"] and h["roofline"]["expansion"]["kernel"].startswith("h2e_run_tape")
    assert len(json.dumps(h)) < 4096
