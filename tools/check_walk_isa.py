"""Where do the grid walk's item loads sit in the compiled kernels?  ("the if (active) build")

rt_grid.h's walk is ONE wave-uniform loop (`__ballot(active)` in its header; mbcnt queue and 64-lane confirm batches in its body).
Compiled right, its item loads (ds_read_u16 / global_load_ushort) sit in an inner loop at depth 2 of the kernel (the main loop is
depth 1).  With the start cell's record read under `if (active)` in front of the loop (-DRT_GRID_ACTIVE_LOAD), ROCm 7.2's clang
merges that block with the identical one at the loop's end; the loop then has two entries, and the structuriser turns it into a
NEST: an outer loop that alternates between "lanes that fetch a record" and "lanes that walk", and the walk as a depth-3 loop that
lanes leave through the exec mask -- so the ballot, the mbcnt ranks and the 64-lane confirm run on a subset of the wave, and
candidates handed to absent lanes are never confirmed (lost hits).

usage: [RT_CSRC_DIR=/path/to/csrc] python tools/check_walk_isa.py [-DRT_GRID_ACTIVE_LOAD ...]   -> one line per kernel: depth of the loop that holds the item loads
exit code 1 if any kernel's walk is not a depth-2 inner loop
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.environ.get("RT_CSRC_DIR", os.path.join(ROOT, "raytracing_amd", "csrc"))  # RT_CSRC_DIR: another checkout's csrc (e.g. a git worktree of 6720931)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-w",
         "-S", "--cuda-device-only"]


def walk_depths(extra=()):
    """{kernel name: set of loop-header comments above the walk's item loads}"""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("rt_pinhole.hip", "rt_paths.hip", "rt_kernels.hip"):
            asm = os.path.join(tmp, src + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-o", asm, src], cwd=CSRC, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            kernel, header = None, None
            for line in open(asm):
                m = re.match(r"^(_ZN2rt\S*(pinhole_kernel|paths_kernel|trace_kernel)\S*):", line)
                if m:
                    kernel, header = m.group(1), None
                    continue
                m = re.search(r"(Inner Loop Header|Loop Header): Depth=(\d+)", line)
                if m:
                    header = (m.group(1), int(m.group(2)))
                    continue
                # the walk's item list: 16-bit loads (LDS or global); nothing else in these kernels reads 16-bit words
                if kernel and re.search(r"\b(ds_read_u16|global_load_ushort|flat_load_ushort)\b", line):
                    out.setdefault(kernel, set()).add(header)
    return out


if __name__ == "__main__":
    bad = 0
    for kernel, headers in sorted(walk_depths(sys.argv[1:]).items()):
        ok = headers and all(h is not None and h == ("Inner Loop Header", 2) for h in headers)
        bad += not ok
        print(("ok   " if ok else "NEST ") + kernel + "  " + ", ".join(sorted(f"{h[0]} depth {h[1]}" if h else "no loop" for h in headers)))
    sys.exit(1 if bad else 0)
