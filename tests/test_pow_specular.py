"""pow_specular (raytracer_amd/csrc/rt_pow.h), the x^y of the specular term the kernels use
instead of the library's pow (light.rs:78 calls powf): compiled for the host with the kernels'
floating-point flags and compared with the C library over four million arguments."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pow_specular_against_libm(tmp_path):
    exe = tmp_path / "pow_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                    "-I", os.path.join(REPO, "raytracer_amd", "csrc"),
                    os.path.join(REPO, "tests", "pow", "pow_check.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    tried, worst_abs, worst_rel, edge_bad = int(out[0]), float(out[1]), float(out[2]), int(out[3])
    assert tried > 3_000_000
    # x in (0, 1 + 5e-10], y in [0, 1000]: the result is in [0, 1] and within 1.2e-16 of pow's
    assert worst_abs <= 1.2e-16
    # results that can show in a colour (> 1e-6): a few ulps
    assert worst_rel <= 5e-15
    assert edge_bad == 0
