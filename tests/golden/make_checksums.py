"""Writes the small entries of tests/golden/frame_checksums.json from the CPU oracle (sharding.frame_checksum of the
oracle's image).  The full-size entries (C2 1920x1080x64, C4 3840x2160x1024) cannot come from the oracle in reasonable
time; they are single-GPU results printed by `python bench.py --gpus 1 --workload c2|c4 --print-checksum` on an MI355X and
pasted in by hand — the sharded and unsharded GPU images are bit-identical (tests/test_gpu_parity.py), and the small entries
tie the GPU's images to the oracle's."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from raytracing_amd import sharding, synth  # noqa: E402

SMALL = [("c2", 16, 9, 3), ("c2", 15, 7, 3), ("c4", 24, 16, 2), ("c2", 96, 54, 4)]


def key(name, w, h, spp, cfg):
    return f"{name}:{w}x{h}:spp{spp}:seed{cfg['seed']}:depth{cfg['max_bounces']}"


def main():
    path = os.path.join(ROOT, "tests", "golden", "frame_checksums.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for name, w, h, spp in SMALL:
        cfg = (synth.config_c2 if name == "c2" else synth.config_c4)(w, h)
        fb, *_ = ob.OracleScene(cfg["objects"]).render(cfg["gen"], spp, cfg["infinity_distance"], cfg["max_bounces"], seed=cfg["seed"], threads=8)
        out[key(name, w, h, spp, cfg)] = f"{sharding.frame_checksum(torch.from_numpy(fb)):#018x}"
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
