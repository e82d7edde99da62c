"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the CPU code (host side only, never GPU
code): the oracle's restatement, and the product's host mirror with its OBJ loader, writers and
host BVH builder."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(target_dir, target):
    r = subprocess.run(["make", "-s", "-C", os.path.join(REPO, target_dir), target], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sanitizer run complete" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_oracle_under_asan_ubsan():
    run("oracle", "asan-run")


def test_host_mirror_under_asan_ubsan(rt):
    run("raytracer_amd", "asan-host-run")
