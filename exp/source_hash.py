"""sha256 over the product's sources (csrc, include, the Python binding, bench.py): what a profile set was taken from"""
import glob
import hashlib
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
files = sorted(glob.glob(os.path.join(root, "halo2ecc_s_amd", "csrc", "*.h*")) + glob.glob(os.path.join(root, "halo2ecc_s_amd", "csrc", "*.cpp")) +
               glob.glob(os.path.join(root, "halo2ecc_s_amd", "*.py")) + [os.path.join(root, "include", "h2e.h"), os.path.join(root, "bench.py")])
for f in files:
    h.update(os.path.relpath(f, root).encode())
    h.update(open(f, "rb").read())
print(h.hexdigest())
