"""The hot kernels must not touch scratch (private memory). Twice this round a harmless-looking construct - a store in each arm of
an `if` chain, a loop over an array of structs - made the compiler keep cursors or a whole struct in a dynamically indexed private
array: `.private_segment_fixed_size` > 0, `scratch_load / scratch_store` in the inner loop, +0.3 GB of HBM writes per pass or a 5x
slower window phase, and nothing failed. This test reads the kernel metadata of the built library's gfx950 code object."""
import os
import re
import shutil
import subprocess

from conftest import ROOT

LLVM = "/opt/rocm/lib/llvm/bin"

# kernel name prefixes (demangled by eye: mp::<name><template arguments>) that run in the timed pass of configs B / C / D
HOT = [
    r"_ZN2mp14k1_pileup_bitsILi[12]E",
    r"_ZN2mp13k2a_admissionILi[12]E",
    r"_ZN2mp18k2a_admission_flatILi[12]ELi[12]E",
    r"_ZN2mp16k2l_window_lanesILi(6|8|16)ELi(0|256)ELi(8|16)E",
    r"_ZN2mp15k2w_window_rowsE",
    r"_ZN2mp21k2w_window_rows_multiILi\dELi[12]E",
    r"_ZN2mp13k3_window_seqILi(32|48)ELi[0123]ELi64ELi[12]E",
]


def test_hot_kernels_use_no_scratch(built, tmp_path):
    import microphaser_amd as m
    lib = str(tmp_path / "lib.so")
    shutil.copy(m.LIB_PATH, lib)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], check=True, capture_output=True, cwd=str(tmp_path))
    objs = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert objs, os.listdir(tmp_path)
    seen = {}
    for f in objs:
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
            if name and scratch:
                seen[name.group(1)] = int(scratch.group(1))
    assert len(seen) > 50
    hot = {k: v for k, v in seen.items() if any(re.match(p, k) for p in HOT)}
    assert len(hot) >= 20, sorted(hot)
    bad = {k: v for k, v in hot.items() if v != 0}
    assert not bad, "kernels with scratch: %r" % bad
