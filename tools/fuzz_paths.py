"""Differential fuzz of the explicit-ray kernel (rt_paths.hip: bundle bound, cluster cull, list scan, LDS / global grid, generator-driven
renders) against the general kernel (RT_HIP_NO_PATHS=1: trace_kernel), which shares none of that: the fuzz cases of tools/fuzz_pinhole.py
(random projectors near / far from the coordinate origin, skewed, tiny and huge targets; sphere clouds with spheres around the origin,
huge spheres, crowds, boxes, mirror planes; every seventh case beyond an LDS tile) traced three ways --
  * the projector's rays as a ray list, in generator order (thin fans, column ends inside batches) and shuffled (no coherent batch);
  * a parallel sheet through the same cloud as a ray list;
  * a ParallelGrid render of that sheet (paths_kernel<GM, RENDER>).
Polylines, vertex counts, bounce records, framebuffers, ids and segment totals must be identical.  GPU only, no oracle.
usage: fuzz_paths.py [cases] [first seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

import fuzz_pinhole
from raytracing_amd import abi, engine
from raytracing_amd.engine import make_rays, parallel_grid


def both(fn):
    a = fn()
    os.environ["RT_HIP_NO_PATHS"] = "1"
    try:
        b = fn()
    finally:
        del os.environ["RT_HIP_NO_PATHS"]
    return a, b


def equal(a, b):
    for x, y in zip(a, b):
        x, y = np.asarray(x), np.asarray(y)
        if x.dtype.fields:
            for f in x.dtype.names:
                if not equal([x[f]], [y[f]]):
                    return False
        elif x.dtype.kind == "f":
            if not np.all((x == y) | (np.isnan(x) & np.isnan(y))):
                return False
        elif not np.array_equal(x, y):
            return False
    return True


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    fuzz_pinhole.seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    engine.init(0)
    bad = hits = traced = 0
    t0 = time.time()
    for case in range(n_cases):
        gen, objs, st, scale, n_sph, cu, cv = fuzz_pinhole.make_case(case)
        rng = np.random.default_rng(1000 + case)
        O, corner = np.array(gen.a[:], np.float32), np.array(gen.corner[:], np.float32)
        if case % 10 == 3:  # a dense one: a few hundred batches, rings exchanged between waves
            cu, cv = cu * 6, cv * 6
        i, j = np.meshgrid(np.arange(cu, dtype=np.float32), np.arange(cv, dtype=np.float32), indexing="ij")
        tgt = corner + i[..., None] * (np.array(gen.u[:], np.float32) / np.float32(cu)) + j[..., None] * (np.array(gen.v[:], np.float32) / np.float32(cv))
        fan = make_rays(np.tile(O, (cu * cv, 1)), (tgt - O).reshape(-1, 3))
        shuffled = fan[rng.permutation(len(fan))]
        d0 = (tgt.reshape(-1, 3).mean(0) - O).astype(np.float32)
        sheet_o = (tgt.reshape(-1, 3) - d0).astype(np.float32)  # the target grid moved back along the mean direction
        sheet = make_rays(sheet_o, np.tile(d0, (len(sheet_o), 1)))
        pg = parallel_grid(corner - d0, gen.u[:], gen.v[:], cu, cv, d0)
        sc = engine.Scene(objs)
        what = []
        for name, rays in (("fan", fan), ("shuffled", shuffled), ("sheet", sheet)):
            a, b = both(lambda: sc.simulate_rays_raw(st, rays=rays, seed=case))
            if not equal(a, b):
                what.append(name)
            hits += int((a[1] > 2).sum())
            traced += len(rays)
        spp = int(rng.integers(1, 9))
        a, b = both(lambda: sc.render(pg, spp, st, seed=case))
        if not (equal(a[:4], b[:4]) and a[4] == b[4]):
            what.append("render")
        if what:
            bad += 1
            print(f"case {case} (seed {fuzz_pinhole.seed0 + case}): MISMATCH in {what}; {n_sph} spheres, scale {scale:.3g}, {cu}x{cv}", flush=True)
        if case % 50 == 49:
            print(f"{case + 1} cases, {bad} mismatching, {time.time() - t0:.0f} s, {hits / traced:.2f} of the listed rays bounce", flush=True)
    print(f"done: {n_cases} cases, {bad} mismatching", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
