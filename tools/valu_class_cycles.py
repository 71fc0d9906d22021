"""profiles/valu_class_cycles.json from the output of tools/ubench/valu_rate.hip (its VALU_CLASS_CYCLES line): the issue cost,
in cycles per wave instruction per SIMD at 8 waves per SIMD, of each instruction class rocprofv3's SQ_INSTS_VALU_* counters
report.  bench.py weights a kernel's measured instruction mix with these numbers to get the ceiling of roofline.frac.

usage: python tools/valu_class_cycles.py profiles/r03_ubench_valu_rate.txt
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
line = [ln for ln in open(src) if ln.startswith("VALU_CLASS_CYCLES")][-1]
raw = json.loads(line.split(" ", 1)[1])
out = {
    "source": os.path.relpath(src, ROOT),
    "unit": "cycles per wave-level VALU instruction per SIMD (8 waves per SIMD, independent instructions)",
    "cycles": {k: raw[k] for k in ("FMA_F32", "ADD_F32", "MUL_F32", "TRANS_F32", "INT32", "INT64", "CVT", "OTHER")},
    "detail": {"INT32_MUL": raw["INT32_MUL"], **raw["other_detail"]},
    "note": "OTHER = SQ_INSTS_VALU minus the named classes (compares, v_cndmask, moves, v_min/v_max, mbcnt, readlane ...): the mean of "
            "v_cndmask_b32_e64, v_mov_b32, v_mbcnt and v_cmp_e64.  INT32 = add / shift / and (v_mul_lo_u32 costs INT32_MUL).  "
            "Spec rate for FMA / ADD / MUL is 2 cycles (MI355X_MICROARCH.md); the chip sustains 2.35-2.45.",
}
json.dump(out, open(os.path.join(ROOT, "profiles", "valu_class_cycles.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out["cycles"]))
