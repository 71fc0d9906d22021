#!/bin/bash
# round-3 measurement pass: bench line + kernel trace + PMC passes of C2, the simulate call shape, C4 and C3 (tools/profile_bench.sh)
set -e
tools/profile_bench.sh r03f c2
tools/profile_bench.sh r03f_sim simulate
tools/profile_bench.sh r03f_c3 c3
tools/profile_bench.sh r03f_c4 c4
