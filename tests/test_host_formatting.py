"""The product's host-side writers (in-place row formatting, f64 printing, SHA-1 ids with the CPU's SHA extensions) produce the
bytes of the plain writers the CPU oracle uses (oracle/oracle_util.hpp - the oracle's own, the product does not include it): a small C++ check, built with g++ and run here."""
import os
import subprocess

from conftest import ROOT


def test_row_writers_and_ids_match_the_plain_ones(tmp_path):
    exe = str(tmp_path / "host_format_check")
    src = os.path.join(ROOT, "tests", "host_format_check.cpp")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "microphaser_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"), "-o", exe, src], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rows: identical" in r.stdout and "total mismatches 0" in r.stdout
