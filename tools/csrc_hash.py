"""SHA-256 over the sources librt_hip.so is built from (raytracing_amd/csrc/*.{hip,cpp,h}, its Makefile, include/rt_hip.h).

profiles/pmc_constants.json stores the hash of the tree its counters were collected from; bench.py recomputes it at run time
and reports the PMC-derived roofline fields as null when the two differ (the instruction counts then belong to another kernel).

usage: python tools/csrc_hash.py      -> prints the hash
"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash(root=ROOT):
    csrc = os.path.join(root, "raytracing_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".h")) or f == "Makefile")
    files.append(os.path.join(root, "include", "rt_hip.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_hash())
