"""bench.py's own paths at a small size: the JSON contract the driver reads, the streaming-job mode (ring of output buffers,
digest consumer, gathered per-unit records) and the N = 2 launch (two ranks on one GPU over gloo - the collective path
without a second device)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*args, env=None, launcher=None):
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + list(args)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _check_contract(d, n_gpus, steps, warmup):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9


def test_bench_small_msm_with_cpu_baseline():
    d = _bench("--units", "2", "--points", "33", "--steps", "3", "--warmup", "1", "--traffic", "off", "--cpu-sample-points", "33")
    _check_contract(d, 1, 3, 1)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]


def test_bench_streaming_job_digest_and_records():
    # 8 tiles of 33 points through a ring of two 2-tile buffer sets: 4 steps, digest consumer on, records gathered
    d = _bench("--units", "2", "--points", "33", "--job-tiles", "8", "--warmup", "1", "--traffic", "off", "--no-cpu-baseline")
    _check_contract(d, 1, 4, 1)
    assert d["gathered_records"]["shape"] == [2, 29] and d["gathered_records"]["status_or"] == 0
    assert len(d["digest_sample"]) == 4 and any(d["digest_sample"])


@pytest.mark.parametrize("workload", ["pairing_bn256", "pairing_bls12_381"])
def test_bench_small_pairing(workload):
    d = _bench("--workload", workload, "--units", "2", "--steps", "3", "--warmup", "1", "--ring", "2", "--traffic", "off", "--no-cpu-baseline")
    _check_contract(d, 1, 3, 1)


def test_bench_two_ranks_one_gpu_gloo():
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                "--master-port", str(29600 + os.getpid() % 300)]
    d = _bench("--gpus", "2", "--units", "2", "--points", "33", "--steps", "2", "--warmup", "1", "--traffic", "off", "--no-cpu-baseline",
               "--dist-backend", "gloo", "--device", "0", "--digest", launcher=launcher)
    _check_contract(d, 2, 2, 1)
    assert d["gathered_records"]["shape"] == [4, 29] and d["gathered_records"]["status_or"] == 0
