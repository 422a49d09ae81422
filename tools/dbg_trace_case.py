"""Trace one gene of a fuzz case on both sides (MP_TRACE: one line per print_haplotypes call with depth and haplotype keys) and
print the first differing trace lines.
  python tools/dbg_trace_case.py mode seed n depth spacing indel multi soft wl read_len gene [mate_rate [isoform_rate]]
"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import microphaser_amd as m
ORACLE_CLI = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
mode, seed, n, depth, spacing, indel, multi, soft, wl, rl, g = sys.argv[1:12]
mate = sys.argv[12] if len(sys.argv) > 12 else "0"
iso = sys.argv[13] if len(sys.argv) > 13 else "0"
g = int(g)
tmp = tempfile.mkdtemp(prefix="mptrace")
to, tg = os.path.join(tmp, "o.txt"), os.path.join(tmp, "g.txt")
env = dict(os.environ, MP_TRACE=to)
subprocess.run([ORACLE_CLI, "synth", "--mode", mode, "--seed", seed, "--transcripts", n, "--depth", depth, "--spacing", spacing, "--indel-rate", indel,
                "--multiallelic-rate", multi, "--softmask-rate", soft, "--window-len", wl, "--read-len", rl, "--mate-rate", mate, "--isoform-rate", iso, "--skip-panics", "--genes", "%d:%d" % (g, g + 1),
                "--prefix", os.path.join(tmp, "o")], capture_output=True, check=True, env=env)
os.environ["MP_TRACE"] = tg
ctx = m.Context(0)
ds = ctx.synth(int(seed), int(n), float(depth), float(spacing), indel_rate=float(indel), multiallelic_rate=float(multi), softmask_rate=float(soft), read_len=int(rl), mate_rate=float(mate), isoform_rate=float(iso))
b = ds.batch(window_len=int(wl), gene_lo=g, gene_hi=g + 1, mode=m.MODE_SOMATIC if mode == "somatic" else m.MODE_NORMAL); b.run(); b.results()
a, c = open(to).read().split("\n"), open(tg).read().split("\n")
k = 0
for i in range(max(len(a), len(c))):
    x, y = (a[i] if i < len(a) else None), (c[i] if i < len(c) else None)
    if x != y:
        print("trace line", i); print(" oracle:", x); print(" engine:", y); k += 1
        if k >= 4: break
print("trace lines:", len(a), len(c))
