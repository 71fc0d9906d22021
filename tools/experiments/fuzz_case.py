"""One fuzz case of tools/fuzz_pinhole.py against the oracle, cell by cell.  usage: fuzz_case.py CASE SEED0"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import fuzz_pinhole as fz
from oracle import binding as ob
from raytracing_amd import engine

case, seed0 = int(sys.argv[1]), int(sys.argv[2])
fz.seed0 = seed0
engine.init(0)
gen, objs, st, scale, n_sph, cu, cv = fz.make_case(case)
sc = engine.Scene(objs)
a = sc.render(gen, 2, st, seed=case)
o = ob.OracleScene(objs).render(gen, 2, st.infinity_distance, st.max_bounces, seed=case, threads=8)
bad = np.nonzero((a[3] != o[3]) | (a[0] != o[0]).any(axis=1))[0]
print(f"case {case}: n_sph {n_sph}, {cu}x{cv}, depth {st.max_bounces}; segs {a[4]} vs oracle {o[4]}; {len(bad)} cells differ", flush=True)
for m in bad[:10]:
    print(f"  cell {m} ({m // cv},{m % cv}): seg {a[3][m]} want {o[3][m]}, fid {a[1][m]} want {o[1][m]}, t {a[2][m]} want {o[2][m]}, fb {a[0][m]} want {o[0][m]}", flush=True)
