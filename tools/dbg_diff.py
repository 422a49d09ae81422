import sys, os, subprocess, json
sys.path.insert(0, os.getcwd())
import microphaser_amd as m
from microphaser_amd.shard import merge_streams
seed, n, indel, multi, soft = [float(x) for x in sys.argv[1:6]] if len(sys.argv) > 5 else (17, 60, 0.05, 0, 0)
seed, n = int(seed), int(n)
os.makedirs("gpurun_out", exist_ok=True)
for f in ("gpurun_out/trace_g.txt", "gpurun_out/trace_o.txt"):
    if os.path.exists(f): os.remove(f)
os.environ["MP_TRACE"] = "gpurun_out/trace_o.txt"
r = subprocess.run(["oracle/_build/oracle_cli", "synth", "--seed", str(seed), "--transcripts", str(n), "--indel-rate", str(indel),
                    "--multiallelic-rate", str(multi), "--softmask-rate", str(soft), "--skip-panics", "--prefix", "gpurun_out/o"], capture_output=True, check=True)
st = json.loads(r.stdout)
os.environ["MP_TRACE"] = "gpurun_out/trace_g.txt"
ctx = m.Context(0)
ds = ctx.synth(seed, n, indel_rate=indel, multiallelic_rate=multi, softmask_rate=soft)
parts, lo = [], 0
for g in st["skipped"] + [ds.num_genes]:
    if g > lo:
        b = ds.batch(gene_lo=lo, gene_hi=g); b.run(); r = b.results()
        parts.append(dict(fasta=r.fasta, normal_fasta=r.normal_fasta, tsv=r.tsv))
    lo = g + 1
got = merge_streams(parts)
open("gpurun_out/g.tsv", "wb").write(got["tsv"]); open("gpurun_out/g.fa", "wb").write(got["fasta"])
a = open("gpurun_out/o.tsv", "rb").read().split(b"\n"); b = got["tsv"].split(b"\n")
k = 0
for i, (x, y) in enumerate(zip(a, b)):
    if x != y:
        print("line", i); print("O", x.decode()); print("G", y.decode()); k += 1
        if k > 2: break
print(len(a), len(b), st["skipped"])
