"""Randomised config E pipelines (test infrastructure): `normal` -> `build_reference` -> `somatic` -> `filter`, every stage on the GPU, the
`filter` stage (and the peptidome it reads) compared with the CPU oracle on the same bytes - the filter's five streams must be identical.
  python tools/fuzz_config_e.py [first_seed] [n_cases] [time_budget_s]
"""
import os, random, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import microphaser_amd as m
from microphaser_amd.shard import merge_streams
ORACLE_CLI = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
rng = random.Random(first)
ctx = m.Context(0)
bad = done = 0
t0 = time.time()
for k in range(count):
    if time.time() - t0 > budget: break
    seed = first + k
    n = rng.choice([10, 20, 30]); depth = rng.choice([10, 20, 30]); spacing = rng.choice([2.0, 3.5, 5.4, 9.0])
    indel = rng.choice([0, 0.03, 0.1]); multi = rng.choice([0, 0.08]); plen = rng.choice([9, 9, 8, 10, 11])
    tmp = tempfile.mkdtemp(prefix="mpcfge")
    ds = ctx.synth(seed, n, float(depth), spacing, indel_rate=indel, multiallelic_rate=multi)
    wl = 3 * plen
    normal_parts, som_parts = [], []
    for g in range(ds.num_genes):
        for mode, acc in ((m.MODE_NORMAL, normal_parts), (m.MODE_SOMATIC, som_parts)):
            try:
                b = ds.batch(window_len=wl, gene_lo=g, gene_hi=g + 1, mode=mode); b.run(); r = b.results()
                acc.append(r.fasta if mode == m.MODE_NORMAL else dict(fasta=r.fasta, normal_fasta=r.normal_fasta, tsv=r.tsv))
                b.close()
            except m.MicrophaserError:
                pass
    nfa = os.path.join(tmp, "normal.fa"); open(nfa, "wb").write(b"".join(normal_parts))
    pep = ctx.build_reference(nfa, plen)
    ref_bin = os.path.join(tmp, "ref.bin"); open(ref_bin, "wb").write(pep.binary)
    # the peptidome itself: set-equal to the oracle's
    ro = subprocess.run([ORACLE_CLI, "build_reference", "-r", nfa, "-o", os.path.join(tmp, "o.bin"), "-l", str(plen)], capture_output=True)
    diffs = []
    if ro.returncode != 0: diffs.append("oracle build_reference failed")
    elif m.decode_bincode_set(open(os.path.join(tmp, "o.bin"), "rb").read()) != m.decode_bincode_set(pep.binary): diffs.append("peptidome differs")
    info = os.path.join(tmp, "info.tsv"); open(info, "wb").write(merge_streams(som_parts)["tsv"] if som_parts else b"")
    rows = open(info, "rb").read().count(b"\n")
    f = ctx.filter(info, ref_bin, plen)
    r = subprocess.run([ORACLE_CLI, "filter", "-r", ref_bin, "-l", str(plen), "-t", info, "-o", tmp + "/o.tsv", "-n", tmp + "/o.normal.fa",
                        "-s", tmp + "/o.removed.tsv", "-p", tmp + "/o.removed.fa"], capture_output=True)
    if r.returncode != 0: diffs.append("oracle filter failed: " + r.stderr.decode()[-200:])
    else:
        for name, got, exp in (("fasta", f.fasta, r.stdout), ("normal_fasta", f.normal_fasta, open(tmp + "/o.normal.fa", "rb").read()),
                               ("tsv", f.tsv, open(tmp + "/o.tsv", "rb").read()), ("removed_tsv", f.removed_tsv, open(tmp + "/o.removed.tsv", "rb").read()),
                               ("removed_fasta", f.removed_fasta, open(tmp + "/o.removed.fa", "rb").read())):
            if got != exp: diffs.append(name + " differs")
    done += 1; bad += 1 if diffs else 0
    print("DIFF" if diffs else "ok  ", "seed=%d n=%d depth=%d spacing=%s indel=%s multi=%s peptide_len=%d info_rows=%d kept=%d removed=%d" % (
        seed, n, depth, spacing, indel, multi, plen, rows, f.kept, f.removed), "; ".join(diffs), flush=True)
    ds.close()
print("cases: %d, mismatches: %d, %.0f s" % (done, bad, time.time() - t0))
sys.exit(1 if bad else 0)
