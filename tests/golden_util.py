import glob
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_all():
    docs = []
    for p in sorted(glob.glob(os.path.join(HERE, "golden", "*.json"))):
        with open(p) as f:
            docs.append(json.load(f))
    return docs


def inputs_of(doc):
    return np.array([[int(w, 16) for w in slot] for slot in doc["inputs_hex"]], dtype=np.uint64)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
