"""The fuzz scenes of tools/fuzz_pinhole.py rendered on the GPU and by the CPU oracle (every case, not only mismatching ones): framebuffer,
first-hit ids / t, per-cell segment counts and totals must be identical.  Slower than the GPU-against-GPU fuzzers, but independent of
anything the GPU kernels share (grid, cull lists).  It lives under tests/ because it uses the oracle (test infrastructure).  usage: python tests/fuzz_oracle.py [cases] [first seed]; RT_HIP_GRID_IN_MEMORY etc. apply."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import numpy as np

import fuzz_pinhole
from oracle import binding as ob
from raytracing_amd import engine


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    fuzz_pinhole.seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    engine.init(0)
    bad = 0
    t0 = time.time()
    for case in range(n_cases):
        gen, objs, st, scale, n_sph, cu, cv = fuzz_pinhole.make_case(case)
        got = engine.Scene(objs).render(gen, 2, st, seed=case)
        want = ob.OracleScene(objs).render(gen, 2, st.infinity_distance, st.max_bounces, seed=case, threads=16)
        same = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(got[:4], want[:4])) and got[4] == want[4]
        if not same:
            bad += 1
            print(f"case {case} (seed {fuzz_pinhole.seed0 + case}): GPU != oracle; {n_sph} spheres, scale {scale:.3g}, {cu}x{cv}, segments {got[4]} vs {want[4]}", flush=True)
        if case % 100 == 99:
            print(f"{case + 1} cases, {bad} mismatching, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {n_cases} cases, {bad} mismatching", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
