"""The grid walk must compile to ONE wave-uniform loop.  ROCm 7.2's clang once merged two identical
`if (active) { read the cell record }` blocks (in front of the loop and at its end) and the structuriser then nested the walk
inside a divergent loop, where its ballots and 64-lane confirm batches ran on a subset of the wave: hits were lost
(profiles/r03_walk_isa_if_vs_select.txt).  tools/check_walk_isa.py reads the loop structure out of the compiler's own comments;
this test keeps the shipped sources in the good shape, whatever is edited around the walk next."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_grid_walk_is_one_inner_loop_in_every_kernel():
    from check_walk_isa import walk_depths

    depths = walk_depths()
    names = " ".join(depths)
    assert "pinhole_kernel" in names and "paths_kernel" in names and "trace_kernel" in names
    for kernel, headers in depths.items():
        assert headers == {("Inner Loop Header", 2)}, f"{kernel}: the walk's item loads sit in {headers}"
