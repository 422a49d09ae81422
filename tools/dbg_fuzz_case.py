"""Re-run one case of tools/fuzz_vs_oracle.py gene by gene and report the genes whose output differs or fails.
  python tools/dbg_fuzz_case.py mode seed n depth spacing indel multi soft wl [read_len [mate_rate [isoform_rate]]]
"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import microphaser_amd as m
ORACLE_CLI = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
mode, seed, n, depth, spacing, indel, multi, soft, wl = sys.argv[1:10]
rl = sys.argv[10] if len(sys.argv) > 10 else "101"
mate = sys.argv[11] if len(sys.argv) > 11 else "0"
iso = sys.argv[12] if len(sys.argv) > 12 else "0"
tmp = tempfile.mkdtemp(prefix="mpdbg")
ctx = m.Context(0)
ds = ctx.synth(int(seed), int(n), float(depth), float(spacing), indel_rate=float(indel), multiallelic_rate=float(multi), softmask_rate=float(soft), read_len=int(rl), mate_rate=float(mate), isoform_rate=float(iso))
md = m.MODE_SOMATIC if mode == "somatic" else m.MODE_NORMAL
for g in range(ds.num_genes):
    prefix = os.path.join(tmp, "g%d" % g)
    r = subprocess.run([ORACLE_CLI, "synth", "--mode", mode, "--seed", seed, "--transcripts", n, "--depth", depth, "--spacing", spacing,
                        "--indel-rate", indel, "--multiallelic-rate", multi, "--softmask-rate", soft, "--window-len", wl, "--read-len", rl, "--mate-rate", mate, "--isoform-rate", iso, "--skip-panics",
                        "--genes", "%d:%d" % (g, g + 1), "--prefix", prefix], capture_output=True)
    st = json.loads(r.stdout) if r.returncode == 0 else None
    try:
        b = ds.batch(window_len=int(wl), gene_lo=g, gene_hi=g + 1, mode=md); b.run(); res = b.results()
        got = dict(fa=res.fasta, tsv=res.tsv); stats = b.stats.as_dict() if hasattr(b, "stats") else {}
        err = None
    except m.MicrophaserError as e:
        got, err = None, str(e)
    if st is None: print(g, "oracle failed", r.stderr.decode()[-200:]); continue
    exp = {e: open(prefix + "." + e, "rb").read() for e in ("fa", "tsv")}
    if err is not None: print(g, "ENGINE-ERR", err, "oracle skipped:", st["skipped"], "oracle tsv rows", exp["tsv"].count(b"\n"))
    elif st["skipped"]: print(g, "oracle skipped the gene, engine ran")
    else:
        print(g, "ok" if got == exp else "DIFF", exp["tsv"].count(b"\n"))
        if got != exp:
            for k in ("tsv", "fa"):
                a, b2 = got[k].split(b"\n"), exp[k].split(b"\n")
                for i in range(max(len(a), len(b2))):
                    if i >= len(a) or i >= len(b2) or a[i] != b2[i]:
                        print("  first", k, "difference at line", i); print("   engine:", a[i][:400] if i < len(a) else None); print("   oracle:", b2[i][:400] if i < len(b2) else None); break
