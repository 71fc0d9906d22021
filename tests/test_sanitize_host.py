"""AddressSanitizer + UBSan run of the hand-written host code — CPU only, no GPU call is made.

tools/sanitize/Makefile builds (g++ -fsanitize=address,undefined -fno-sanitize-recover=all)
  * build/host_fuzz       — toml_lite.h + scene_config.h on the golden TOMLs and thousands of mutated copies, then
                            rt_host_grid.h / rt_host_bvh.h / rt_host_shapes.h on random and degenerate inputs, checking the
                            built index structures' ranges as it goes;
  * build/raytracing_asan — the native CLI (raytracing_amd/host/main.cpp) itself, run with --dump (parse + flatten + print).
A sanitizer report aborts the binary, which fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import TEMPLATE_LIKE_TOML  # noqa: E402

ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.fixture(scope="module")
def sanitized():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tools", "sanitize")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return os.path.join(ROOT, "build")


def test_host_code_under_asan_ubsan_fuzz(sanitized, tmp_path):
    tl = tmp_path / "template_like.toml"
    tl.write_text(TEMPLATE_LIKE_TOML, encoding="utf-8")
    r = subprocess.run([os.path.join(sanitized, "host_fuzz"), "2500", os.path.join(ROOT, "tests", "golden", "c1_scene.toml"), str(tl)],
                       capture_output=True, text=True, timeout=600, env=ENV)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "no sanitizer report" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    accepted = int(r.stdout.split("host_fuzz:")[1].split()[0])
    assert accepted >= 2  # the two unmutated files at least


@pytest.mark.parametrize("text,ok", [
    (None, True),                       # golden C1 file
    (TEMPLATE_LIKE_TOML, True),
    ("[simulation_settings]\ninfinity_distance = 1.0\n", False),            # missing max_bounces / [scene]
    ("[simulation_settings]\ninfinity_distance = 1.0\nmax_bounces = 2\n[scene]\n[[scene.objects]]\nshape = { type = \"Sphere\" }\n", False),
    ("[simulation_settings\n", False),
    ("[simulation_settings]\ninfinity_distance = 1.0\nmax_bounces = 2\n[scene]\n[[scene.object_generators]]\ntype = \"ObjectGrid\"\ncount_x = 4000000000\n"
     "count_z = 4000000000\nposition_start = [0.0, 0.0, 0.0]\nstep_x = [1.0, 0.0, 0.0]\nstep_z = [0.0, 0.0, 1.0]\n"
     "template.shape = { type = \"Sphere\", radius = 1.0 }\ntemplate.material = { type = \"Mirror\" }\n"
     "template.transform = { position = [0.0, 0.0, 0.0], rotation_y_deg = 0.0 }\n", False),  # refused, not expanded
])
def test_native_cli_dump_under_asan_ubsan(sanitized, tmp_path, text, ok):
    if text is None:
        text = open(os.path.join(ROOT, "tests", "golden", "c1_scene.toml"), encoding="utf-8").read()
    (tmp_path / "simulation.toml").write_text(text, encoding="utf-8")
    r = subprocess.run([os.path.join(sanitized, "raytracing_asan"), "--dump"], cwd=tmp_path, capture_output=True, text=True, timeout=120, env=ENV)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert (r.returncode == 0) == ok, (r.returncode, r.stderr[-2000:])
    if ok:
        assert r.stdout.count("object ") >= 10
