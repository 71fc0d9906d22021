"""Derives the PMC constants bench.py reports (roofline numerator by VALU instruction class, traffic, valu_issue_util,
valu_lane_util, sustained clock) from a rocprofv3 summary written by tools/profile_bench.sh and merges them into
profiles/pmc_constants.json.  The entry is stamped with the source hash the summary carries ("# csrc_sha256:", the tree the
library that ran was built from) and with the workload configuration of the bench line in it; bench.py refuses to use an
entry whose stamp does not match the run.

usage: python tools/pmc_constants.py NAME SUMMARY.txt [KERNEL_SUBSTRING] [LAUNCHES_PER_FRAME]
  valu_issue_util = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * SQ_BUSY_CYCLES / 32 SEs)      (quad-cycles -> cycles; busy cycles per SE)
  valu_lane_util  = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
  clock_ghz       = SQ_BUSY_CYCLES / 32 / (kernel time of the same pass)                  (cross-check: GRBM_GUI_ACTIVE / 8 / time)
  hbm_bytes_per_launch = WRITE_SIZE KB * 1024 + 2 * FETCH_SIZE KB * 1024 per dispatch     (gfx950: FETCH_SIZE reports half the bytes,
                                                                                           MI355X_MICROARCH.md §HBM)
  valu_class_per_frame[c] = SQ_INSTS_VALU_<c> per launch * launches; OTHER = SQ_INSTS_VALU of the same pass - the named classes
  flops_fp32_per_frame = SQ_INSTS_VALU_FLOPS_FP32 (+ _TRANS) per launch * launches
"""
import json
import os
import re
import sys

name, path = sys.argv[1], sys.argv[2]
kernel = sys.argv[3] if len(sys.argv) > 3 else "pinhole_kernel"
launches_per_frame = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # C2: one launch per frame; C4 on one GPU: four launches of 2^21 cells
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASS = ("FMA_F32", "ADD_F32", "MUL_F32", "TRANS_F32", "INT32", "INT64", "CVT")
passes = []
cur = None
sha = None
config = None
for line in open(path):
    if line.startswith("# csrc_sha256:"):
        sha = line.split(":", 1)[1].strip()
        continue
    if line.startswith("{") and config is None:
        try:
            c = json.loads(line)["config"]
            config = {k: c[k] for k in ("width", "height", "spp", "rays", "max_bounces") if k in c}
        except Exception:
            pass
        continue
    if line.startswith("# source:"):
        cur = {"counters": {}, "ns": None, "n": None}
        passes.append(cur)
        continue
    if kernel not in line or cur is None:
        continue
    m = re.search(r"\s([A-Z][A-Za-z0-9_]+)\s+(\d+)\s+dispatches=(\d+)\s*$", line)
    if m:
        # several template instantiations may match the substring (e.g. table + render kernels never do; CX / non-CX could): add them up
        v, n = cur["counters"].get(m.group(1), (0.0, 0))
        cur["counters"][m.group(1)] = (v + float(m.group(2)), n + int(m.group(3)))
        continue
    m = re.search(r"dispatches=\s*(\d+)\s+total_ns=\s*(\d+)", line)
    if m:
        cur["n"], cur["ns"] = (cur["n"] or 0) + int(m.group(1)), (cur["ns"] or 0.0) + float(m.group(2))
out = {"source": os.path.relpath(path, ROOT), "kernel": kernel, "csrc_sha256": sha, "config": config}
for p in passes:
    c = p["counters"]

    def per_launch(k):
        return c[k][0] / c[k][1]

    if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c and p["ns"]:
        busy_cycles = c["SQ_BUSY_CYCLES"][0] / 32.0
        out["valu_issue_util"] = c["SQ_ACTIVE_INST_VALU"][0] * 4.0 / (1024.0 * busy_cycles)
        out["clock_ghz"] = busy_cycles / p["ns"]
        out["valu_insts_per_launch"] = per_launch("SQ_INSTS_VALU") if "SQ_INSTS_VALU" in c else None
        if "SQ_THREAD_CYCLES_VALU" in c:
            out["valu_lane_util"] = c["SQ_THREAD_CYCLES_VALU"][0] / (64.0 * c["SQ_ACTIVE_INST_VALU"][0])
        if "SQ_INSTS_SALU" in c:
            out["salu_insts_per_launch"] = per_launch("SQ_INSTS_SALU")
        if "SQ_INSTS_LDS" in c:
            out["lds_insts_per_launch"] = per_launch("SQ_INSTS_LDS")
    if "SQ_INSTS_VALU_FMA_F32" in c and "SQ_INSTS_VALU" in c:
        mix = {k: per_launch("SQ_INSTS_VALU_" + k) * launches_per_frame for k in CLASS if "SQ_INSTS_VALU_" + k in c}
        total = per_launch("SQ_INSTS_VALU") * launches_per_frame
        mix["OTHER"] = total - sum(mix.values())
        out["valu_class_per_frame"] = mix
        out["valu_insts_per_frame_mix_pass"] = total
    if "SQ_INSTS_VALU_FLOPS_FP32" in c:
        f = per_launch("SQ_INSTS_VALU_FLOPS_FP32")
        if "SQ_INSTS_VALU_FLOPS_FP32_TRANS" in c:
            f += per_launch("SQ_INSTS_VALU_FLOPS_FP32_TRANS")
        out["flops_fp32_per_frame"] = f * launches_per_frame
    if "GRBM_GUI_ACTIVE" in c and p["ns"]:
        out["clock_ghz_grbm"] = c["GRBM_GUI_ACTIVE"][0] / 8.0 / p["ns"]
    if "WRITE_SIZE" in c:
        out["write_bytes_per_launch"] = c["WRITE_SIZE"][0] * 1024.0 / c["WRITE_SIZE"][1]
    if "FETCH_SIZE" in c:
        out["fetch_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"][0] * 1024.0 / c["FETCH_SIZE"][1]
    if "TCC_EA0_ATOMIC_sum" in c:
        out["atomic_requests_per_launch"] = c["TCC_EA0_ATOMIC_sum"][0] / c["TCC_EA0_ATOMIC_sum"][1]
if "write_bytes_per_launch" in out and "fetch_bytes_per_launch" in out:
    out["hbm_bytes_per_launch"] = out["write_bytes_per_launch"] + out["fetch_bytes_per_launch"]
if out.get("valu_insts_per_launch") is not None:
    out["valu_insts_per_frame"] = out["valu_insts_per_launch"] * launches_per_frame
if "hbm_bytes_per_launch" in out:
    out["hbm_bytes_per_frame"] = out["hbm_bytes_per_launch"] * launches_per_frame
out["launches_per_frame"] = launches_per_frame
dst = os.path.join(ROOT, "profiles", "pmc_constants.json")
allc = json.load(open(dst)) if os.path.exists(dst) else {}
allc[name] = out
json.dump(allc, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
