#!/usr/bin/env python3
"""Print LDS / scratch / SGPR / VGPR / spill counts of every gfx950 kernel in libh2e.so (llvm-objdump --offloading
extracts the code object, llvm-readelf --notes lists the kernel descriptors)."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
lib = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "halo2ecc_s_amd", "libh2e.so"))
tmp = tempfile.mkdtemp()
try:
    shutil.copy(lib, os.path.join(tmp, "lib.so"))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "lib.so"], cwd=tmp, capture_output=True)
    for co in glob.glob(os.path.join(tmp, "*gfx950*")):
        t = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in t.split("- .agpr_count")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "-"])[1]
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            print(f"{name[:80]:80s} lds={g('group_segment_fixed_size'):>6} scratch={g('private_segment_fixed_size'):>5} "
                  f"sgpr={g('sgpr_count'):>3} vgpr={g('vgpr_count'):>3} spill={g('vgpr_spill_count')}")
finally:
    shutil.rmtree(tmp)
