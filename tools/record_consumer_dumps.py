#!/usr/bin/env python3
"""Record device-result dumps for the CPU-only consumer tests (run on the GPU box):

  python tools/record_consumer_dumps.py        -> gpurun_out/consumer/<case>.bin   (commit them as tests/golden/consumer/<case>.bin.gz: gzip -9)

For each small synthetic exome of tests/golden/consumer/cases.json the hot path runs on the GPU and the device results - the seam
between the device pass and the host consumer - are written with mp_batch_results_dump, after checking that the GPU run's streams
equal the CPU oracle's. tests/test_consumer_replay.py then consumes the committed dumps on a host-only context and must reproduce
the oracle's bytes: the host consumer (splice merges, frameshift algebra, row formatting; 1000 lines of threaded C++) is covered by
the CPU suite and by the sanitizer builds (tools/run_sanitized.sh) without a GPU.
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import microphaser_amd as m

ORACLE = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
CASES = json.load(open(os.path.join(ROOT, "tests", "golden", "consumer", "cases.json")))


def oracle(case, tmp):
    prefix = os.path.join(tmp, case["name"])
    cmd = [ORACLE, "synth", "--seed", str(case["seed"]), "--transcripts", str(case["transcripts"]), "--depth", str(case["depth"]),
           "--spacing", str(case["spacing"]), "--indel-rate", str(case["indel_rate"]), "--multiallelic-rate", str(case["multiallelic_rate"]),
           "--softmask-rate", str(case["softmask_rate"]), "--mate-rate", str(case["mate_rate"]), "--window-len", str(case["window_len"]),
           "--prefix", prefix]
    if case["mode"] == "normal":
        cmd += ["--mode", "normal"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("oracle failed on %s: %s" % (case["name"], r.stderr[-1000:]))
    return {e: open(prefix + "." + e, "rb").read() for e in ("fa", "normal.fa", "tsv")}, json.loads(r.stdout)


def main():
    out_dir = os.path.join(ROOT, "gpurun_out", "consumer")
    os.makedirs(out_dir, exist_ok=True)
    ctx = m.Context(0)
    tmp = tempfile.mkdtemp(prefix="mp_dump_")
    for case in CASES:
        exp, st = oracle(case, tmp)
        ds = ctx.synth(case["seed"], case["transcripts"], case["depth"], case["spacing"], indel_rate=case["indel_rate"],
                       multiallelic_rate=case["multiallelic_rate"], softmask_rate=case["softmask_rate"], mate_rate=case["mate_rate"])
        mode = m.MODE_NORMAL if case["mode"] == "normal" else m.MODE_SOMATIC
        b = ds.batch(window_len=case["window_len"], mode=mode)
        b.run()
        r = b.results()
        ok = (r.fasta, r.tsv) == (exp["fa"], exp["tsv"]) and (mode == m.MODE_NORMAL or r.normal_fasta == exp["normal.fa"]) and r.windows == st["windows"]
        if not ok:
            raise SystemExit("GPU run of %s differs from the oracle: no dump written" % case["name"])
        path = os.path.join(out_dir, case["name"] + ".bin")
        b.run()
        b.dump_results(path)
        print("%s: %d windows, dump %d bytes" % (case["name"], r.windows, os.path.getsize(path)), flush=True)


if __name__ == "__main__":
    main()
