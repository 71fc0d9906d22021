"""Differential fuzz of the Projector kernel's shared-origin filter: the same frame rendered with the default filters
(pinhole kernel) and with rt_hip_set_scan_mode(1) (general kernel, the reference's own discriminant as the filter) must be
bit-identical.  Random projectors (near / far from the coordinate origin, skewed, tiny and huge targets), random sphere
clouds including spheres that contain the projector origin and spheres at grazing distance; some scenes add a mirror
plane (continuing rays that start far outside the sphere grid), rotation-0 boxes (box entries of the grid) or a crowd of
spheres around the origin (more spheres in reach than a cull list holds).  GPU only, no oracle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from raytracing_amd import abi, engine
from raytracing_amd.engine import SimulationSettingsConfig, plane, projector, shape_object, sphere

def make_case(case):
    """The projector, objects and settings of fuzz case `case` (deterministic)."""
    rng = np.random.default_rng(seed0 + case)
    scale = 10.0 ** rng.uniform(-1, 3)
    centre = rng.normal(size=3) * scale * 10.0 ** rng.uniform(-1, 1.5)
    O = centre + rng.normal(size=3) * scale
    fwd = rng.normal(size=3)
    fwd /= np.linalg.norm(fwd)
    dist = scale * 10.0 ** rng.uniform(-1.5, 0.5)
    u = np.cross(fwd, rng.normal(size=3))
    u /= np.linalg.norm(u)
    v = np.cross(fwd, u) + rng.normal(size=3) * 0.2 * rng.random()
    w_, h_ = dist * 10.0 ** rng.uniform(-0.5, 0.7), dist * 10.0 ** rng.uniform(-0.5, 0.7)
    uvec, vvec = u * w_ + fwd * rng.normal() * 0.1 * dist, v * h_
    corner = O + fwd * dist - 0.5 * uvec - 0.5 * vvec
    cu, cv = int(rng.integers(8, 70)), int(rng.integers(8, 50))
    gen = projector(O, corner, uvec, vvec, cu, cv)
    n_sph = int(rng.integers(1, 400)) if case % 7 else int(rng.integers(1030, 1300))
    ctr = O + fwd * dist * rng.uniform(1.5, 12.0, (n_sph, 1)) + rng.normal(size=(n_sph, 3)) * max(w_, h_) * rng.uniform(0.2, 3.0)
    rad = max(w_, h_) * 10.0 ** rng.uniform(-2.2, -0.4, n_sph)
    if case % 3 == 0 and n_sph >= 8:  # a few spheres around the projector origin
        ctr[:3] = O + rng.normal(size=(3, 3)) * rad[:3, None] * 0.5
    if case % 9 == 0 and n_sph >= 8:  # and some huge ones
        rad[-2:] *= 30.0
    kinds = [abi.RT_MAT_GLASS, abi.RT_MAT_MIRROR, abi.RT_MAT_HALF_MIRROR]
    if case % 11 == 5 and n_sph >= 120:  # the origin inside a crowd: 60+ spheres contain it
        ctr[:70] = O + rng.normal(size=(70, 3)) * max(w_, h_) * 0.02
        rad[:70] = max(w_, h_) * rng.uniform(0.1, 0.3, 70)
    objl = [sphere(ctr[k], rad[k], kinds[k % 3], [1.5, 0.0, 0.5][k % 3]) for k in range(n_sph)]
    if case % 5 == 1:  # boxes among the spheres (object order matters for ties: put them in the middle)
        for k in range(0, n_sph, 5):
            objl[k] = shape_object(abi.RT_SHAPE_BOX, [2 * rad[k], 1.5 * rad[k], 3 * rad[k], 0], ctr[k], kinds[k % 3], [1.5, 0.0, 0.5][k % 3])
    if case % 4 == 2:  # a mirror plane behind / below the cloud
        objl.insert(n_sph // 2, plane(O + fwd * dist * 14.0, -fwd + rng.normal(size=3) * 0.3, abi.RT_MAT_MIRROR, 0.0))
    objs = np.array(objl, dtype=abi.OBJECT_DTYPE)
    st = SimulationSettingsConfig(float(scale * 50.0), int(rng.integers(1, 7)))
    return gen, objs, st, scale, n_sph, cu, cv


def main():
    global seed0
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    # third argument "oracle": every mismatching case is also rendered by the CPU oracle to say which side is wrong
    with_oracle = len(sys.argv) > 3 and sys.argv[3] == "oracle"
    engine.init(0)
    bad = 0
    t0 = time.time()
    for case in range(n_cases):
        gen, objs, st, scale, n_sph, cu, cv = make_case(case)
        sc = engine.Scene(objs)
        engine.set_scan_mode(0)
        a = sc.render(gen, 2, st, seed=case)
        engine.set_scan_mode(1)
        b = sc.render(gen, 2, st, seed=case)
        engine.set_scan_mode(0)
        same = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a[:4], b[:4])) and a[4] == b[4]
        if not same:
            bad += 1
            diff = int((a[1] != b[1]).sum())
            print(f"case {case}: MISMATCH  first_id differs in {diff} cells, segs {a[4]} vs {b[4]}  (scale {scale:.3g}, n_sph {n_sph}, {cu}x{cv})", flush=True)
            if with_oracle:
                from oracle import binding as ob

                o = ob.OracleScene(objs).render(gen, 2, st.infinity_distance, st.max_bounces, seed=case, threads=8)
                ok0 = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a[:4], o[:4])) and a[4] == o[4]
                ok1 = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(b[:4], o[:4])) and b[4] == o[4]
                print(f"      oracle segs {o[4]}: default filters {'==' if ok0 else '!='} oracle, exact-filter mode {'==' if ok1 else '!='} oracle", flush=True)
        if case % 200 == 199:
            print(f"{case + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s, hit fraction last {float((a[1] >= 0).mean()):.2f}", flush=True)
    print(f"done: {n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


seed0 = 0
if __name__ == "__main__":
    main()
