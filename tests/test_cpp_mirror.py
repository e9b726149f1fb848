"""The C++ mirror of the reference's record types (include/nisshi.hpp): its own
test program, tests/cpp/mirror_test.cpp, follows the reference's unit tests.
The CPU part covers what needs no device (the 61-byte header split, framing);
the GPU part inflates and deflates through the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "mirror_test")


def _binary():
    if not os.path.exists(BIN):
        # built by __graft_entry__.build(); build here when run on its own
        from tansu_amd import build
        build.build()
        subprocess.check_call(["make", "-C", os.path.dirname(BIN)])
    return BIN


def test_mirror_cpu_cases():
    out = subprocess.run([_binary(), "cpu"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.startswith("ok ")


@pytest.mark.gpu
def test_mirror_gpu_cases():
    out = subprocess.run([_binary(), "gpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    assert out.stdout.startswith("ok ") and "(gpu)" in out.stdout
