"""The reference's caller is C (lantern_hnsw): include/lantern_gpu.h must be valid C11 and liblantern_gpu.so must link
and run from a plain C program that uses the usearch_* calls the way build.c / scan.c / hnsw.c do."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "lantern_c_caller.c")


@pytest.fixture(scope="module")
def binary(tmp_path_factory):
    from lantern_amd import build

    lib = build.build()
    out = str(tmp_path_factory.mktemp("c_abi") / "lantern_c_caller")
    libdir = os.path.dirname(lib)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), SRC, "-o", out,
                           "-L" + libdir, "-llantern_gpu", "-Wl,-rpath," + libdir])
    return out


def test_header_is_valid_c99_and_c11():
    for std in ("c99", "c11"):
        subprocess.check_call(["gcc", "-std=" + std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c",
                               os.path.join(ROOT, "include", "lantern_gpu.h")])


def test_c_caller_without_a_device_gets_error_strings(binary):
    from lantern_amd import capi

    if capi.device_count() > 0:
        pytest.skip("a device is present")
    p = subprocess.run([binary], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and "no HIP device" in p.stdout, p.stdout + p.stderr


@pytest.mark.gpu
def test_c_caller_builds_and_searches_on_the_device(binary):
    p = subprocess.run([binary], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.startswith("ok: 5 results, nearest label 18"), p.stdout + p.stderr
