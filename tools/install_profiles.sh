#!/bin/bash
# after tools/r03_final.sh on the GPU box: copy the four summaries into profiles/ and rebuild profiles/pmc_constants.json from them
set -e
cp gpurun_out/profile_r03f.txt profiles/r03_pinhole_kernel_c2.txt
cp gpurun_out/profile_r03f_sim.txt profiles/r03_paths_kernel_simulate.txt
cp gpurun_out/profile_r03f_c3.txt profiles/r03_mesh_c3.txt
cp gpurun_out/profile_r03f_c4.txt profiles/r03_pinhole_kernel_c4.txt
python tools/pmc_constants.py c2 profiles/r03_pinhole_kernel_c2.txt pinhole_kernel 1 > /dev/null
python tools/pmc_constants.py simulate profiles/r03_paths_kernel_simulate.txt paths_kernel 1 > /dev/null
python tools/pmc_constants.py c3 profiles/r03_mesh_c3.txt pinhole_kernel 1 > /dev/null
python tools/pmc_constants.py c4 profiles/r03_pinhole_kernel_c4.txt pinhole_kernel 4 > /dev/null
python - <<'PY'
import json
d = json.load(open("profiles/pmc_constants.json")); r = json.load(open("profiles/valu_class_cycles.json"))["cycles"]
for k, v in d.items():
    mix = v["valu_class_per_frame"]; tot = sum(mix.values()); mc = sum(mix[c] * r[c] for c in mix) / tot
    print(k, v["csrc_sha256"][:8], "insts %.4g issue %.3f lane %.3f clk %.3f hbm %.3g flops %.3g ceiling %.1f" % (v["valu_insts_per_frame"], v["valu_issue_util"], v["valu_lane_util"], v["clock_ghz"], v["hbm_bytes_per_frame"], v["flops_fp32_per_frame"], 1024 * 2.4 / mc))
PY
for t in r03f r03f_sim r03f_c3 r03f_c4; do grep -m1 '^{' gpurun_out/profile_$t.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', '%.0f Mrays/s %.3f ms/step kernel %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"; done
