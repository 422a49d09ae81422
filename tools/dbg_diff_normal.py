"""Debug helper: first differing TSV rows between the CPU oracle and the device in `normal` mode on a synthetic exome."""
import sys, os, subprocess, json
sys.path.insert(0, os.getcwd())
import microphaser_amd as m
from microphaser_amd.shard import merge_streams
a = sys.argv[1:]
seed, n, depth, spacing = int(a[0]), int(a[1]), float(a[2]), float(a[3])
indel, multi, soft = (float(a[4]), float(a[5]), float(a[6])) if len(a) > 6 else (0.0, 0.0, 0.0)
os.makedirs("gpurun_out", exist_ok=True)
r = subprocess.run(["oracle/_build/oracle_cli", "synth", "--mode", "normal", "--seed", str(seed), "--transcripts", str(n), "--depth", str(depth),
                    "--spacing", str(spacing), "--indel-rate", str(indel), "--multiallelic-rate", str(multi), "--softmask-rate", str(soft),
                    "--skip-panics", "--prefix", "gpurun_out/on"], capture_output=True, check=True)
st = json.loads(r.stdout)
ctx = m.Context(0)
ds = ctx.synth(seed, n, depth, spacing, indel_rate=indel, multiallelic_rate=multi, softmask_rate=soft)
parts, lo = [], 0
for g in st["skipped"] + [ds.num_genes]:
    if g > lo:
        b = ds.batch(gene_lo=lo, gene_hi=g, mode=m.MODE_NORMAL); b.run(); r = b.results()
        parts.append(dict(fasta=r.fasta, normal_fasta=r.normal_fasta, tsv=r.tsv))
    lo = g + 1
got = merge_streams(parts)
open("gpurun_out/gn.tsv", "wb").write(got["tsv"])
x = open("gpurun_out/on.tsv", "rb").read().split(b"\n"); y = got["tsv"].split(b"\n")
k = 0
for i, (p, q) in enumerate(zip(x, y)):
    if p != q:
        print("line", i); print("O", p.decode()); print("G", q.decode()); k += 1
        if k > 5: break
print(len(x), len(y), st["skipped"])
