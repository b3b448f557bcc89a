"""bench.py in-process with the debug build's device stamps (H2E_DEBUG_STAMPS), then the device timeline of some timed steps:
   python exp/bench_timeline.py <first run to print> <runs> -- <bench.py arguments>      (the debug library must be the loaded libh2e.so)"""
import os
import runpy
import sys
os.environ.setdefault("H2E_DEBUG_STAMPS", "/tmp/h2e_stamps.txt")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
first, count = int(sys.argv[1]), int(sys.argv[2])
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[sys.argv.index("--") + 1:]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit:
    pass
from halo2ecc_s_amd import engine as engine_mod
engine_mod.lib().h2e_debug_dump_stamps()
rows = [tuple(int(x) for x in l.split()) for l in open(os.environ["H2E_DEBUG_STAMPS"])]
t_first = min(r[2] for r in rows if r[0] >= first)
names = {1000: "done", 1001: "start"}
by_run = {}
for run, tag, t in rows:
    by_run.setdefault(run, []).append((tag, (t - t_first) / 100.0))   # 100 MHz -> us
for run in sorted(by_run):
    if first <= run < first + count:
        ev = sorted(by_run[run], key=lambda x: x[1])
        print("run %3d: " % run + "  ".join("%s %.0f" % (names.get(tag, "s%d%s" % (tag // 4, ["cb", "ce", "xb", "xe"][tag % 4])), t) for tag, t in ev))
