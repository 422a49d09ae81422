"""The host consumer (consume.cpp: frameshift algebra, termination, splice-side merges, row formatting - the part of print_haplotypes /
phase_gene that stays on the host, src/microphasing.rs:604-880, :1345-1941, src/common.rs:376-568) on the CPU, without a GPU: device
results recorded on an MI355X (tools/record_consumer_dumps.py; written only after the GPU run's streams equalled the oracle's) are
consumed on a host-only context with a batch planned here from the same synthetic inputs, and must give the CPU oracle's bytes. This
also puts the consumer under the sanitizer builds (tools/run_sanitized.sh)."""
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, ORACLE_CLI

CASES = json.load(open(os.path.join(GOLDEN, "consumer", "cases.json")))


def unpack(name, tmp_path):
    """The committed dumps are gzip-compressed (records are mostly padding): unpack one into the test's directory."""
    import gzip
    path = str(tmp_path / (name + ".bin"))
    with gzip.open(os.path.join(GOLDEN, "consumer", name + ".bin.gz"), "rb") as src, open(path, "wb") as dst:
        dst.write(src.read())
    return path


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_consumer_on_recorded_device_results_gives_the_oracle_output(built, tmp_path, case):
    import microphaser_amd as m
    dump = unpack(case["name"], tmp_path)
    prefix = str(tmp_path / "o")
    cmd = [ORACLE_CLI, "synth", "--seed", str(case["seed"]), "--transcripts", str(case["transcripts"]), "--depth", str(case["depth"]),
           "--spacing", str(case["spacing"]), "--indel-rate", str(case["indel_rate"]), "--multiallelic-rate", str(case["multiallelic_rate"]),
           "--softmask-rate", str(case["softmask_rate"]), "--mate-rate", str(case["mate_rate"]), "--window-len", str(case["window_len"]),
           "--prefix", prefix] + (["--mode", "normal"] if case["mode"] == "normal" else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    st = json.loads(r.stdout)
    exp = {e: open(prefix + "." + e, "rb").read() for e in ("fa", "normal.fa", "tsv")}
    ctx = m.Context(-1)    # host-only: planner and consumer, no device
    ds = ctx.synth(case["seed"], case["transcripts"], case["depth"], case["spacing"], indel_rate=case["indel_rate"],
                   multiallelic_rate=case["multiallelic_rate"], softmask_rate=case["softmask_rate"], mate_rate=case["mate_rate"])
    mode = m.MODE_NORMAL if case["mode"] == "normal" else m.MODE_SOMATIC
    b = ds.batch(window_len=case["window_len"], mode=mode)
    res = b.results_from_dump(dump)
    assert res.windows == st["windows"]
    assert res.fasta == exp["fa"] and res.tsv == exp["tsv"]
    if mode == m.MODE_SOMATIC:
        assert res.normal_fasta == exp["normal.fa"]
    assert exp["tsv"].count(b"\n") > 20
    # a stream mask leaves the other streams unwritten
    only = b.results_from_dump(dump, m.STREAM_FASTA)
    assert only.fasta == exp["fa"] and only.tsv == b""


def test_a_dump_of_another_batch_is_refused(built, tmp_path):
    import microphaser_amd as m
    ctx = m.Context(-1)
    b = ctx.synth(5, 4).batch()
    with pytest.raises(m.MicrophaserError):
        b.results_from_dump(unpack(CASES[0]["name"], tmp_path))
    junk = tmp_path / "junk.bin"
    junk.write_bytes(b"not a dump")
    with pytest.raises(m.MicrophaserError):
        b.results_from_dump(str(junk))
