"""bench.py in-process with the host time of every Engine.submit / wait / job_launch_ms call printed: python exp/bench_submit_times.py -- <bench args>"""
import os
import runpy
import sys
import time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from halo2ecc_s_amd import engine as E
log = []
def wrap(name):
    f = getattr(E.Engine, name)
    def g(self, *a, **k):
        t = time.perf_counter()
        r = f(self, *a, **k)
        log.append((name, 1e3 * (time.perf_counter() - t)))
        return r
    setattr(E.Engine, name, g)
for n in ("submit", "wait", "job_launch_ms", "unit_records"):
    wrap(n)
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[sys.argv.index("--") + 1:]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit:
    pass
for n in ("submit", "wait", "job_launch_ms", "unit_records"):
    v = [round(ms, 2) for k, ms in log if k == n]
    print(n, len(v), v)
