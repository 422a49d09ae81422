"""End-to-end CLI timing (SURVEY 8d ii): write a synthetic exome as BAM / VCF / GTF / FASTA files, then time
`microphaser somatic` (product, GPU) and `oracle_cli somatic` (CPU restatement, one thread) on those files.

  python tools/e2e_cli.py <config B|C|...> [transcripts]
"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import microphaser_amd as m
cfg = {"B": (1001, 1000, 30.0, 5.4), "C": (2020, 20000, 30.0, 5.4)}[sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "B"]
pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(pos[1]) if len(pos) > 1 else cfg[1]
import tempfile
out = tempfile.mkdtemp(prefix="mp_e2e_")   # GB-sized files: not under gpurun_out/ (64 MiB copy-back limit)
prefix = os.path.join(out, "synth")
ctx = m.Context(-1)
t = time.perf_counter(); ds = ctx.synth(cfg[0], n, cfg[2], cfg[3]); t_gen = time.perf_counter() - t
t = time.perf_counter(); ds.write(prefix); t_write = time.perf_counter() - t
files = {e: prefix + "." + e for e in ("bam", "vcf", "gtf", "fa")}
sizes = {e: os.path.getsize(p) for e, p in files.items()}
def run(cmd, stdout_path):
    t = time.perf_counter()
    with open(files["gtf"], "rb") as g, open(stdout_path, "wb") as o:
        r = subprocess.run(cmd, stdin=g, stdout=o, stderr=subprocess.PIPE)
    if os.environ.get("MP_DEBUG"): sys.stderr.write(r.stderr.decode())
    dt = time.perf_counter() - t
    if r.returncode != 0: raise SystemExit(r.stderr.decode()[-2000:])
    return dt
cli = os.path.join(ROOT, "microphaser_amd", "_lib", "microphaser")
ocli = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
args = [files["bam"], "--variants", files["vcf"], "--ref", files["fa"]]
t_gpu = run([cli, "somatic"] + args + ["--tsv", out + "/g.tsv", "--normal-output", out + "/g.normal.fa"], out + "/g.fa")
res = {"transcripts": n, "file_bytes": sizes, "generate_s": t_gen, "write_files_s": t_write, "microphaser_somatic_cli_s": t_gpu}
if "--no-oracle" not in sys.argv:
    t_cpu = run([ocli, "somatic"] + args + ["--tsv", out + "/o.tsv", "--normal-output", out + "/o.normal.fa"], out + "/o.fa")
    res["oracle_cli_s"] = t_cpu
    res["identical_outputs"] = all(open(out + "/g" + e, "rb").read() == open(out + "/o" + e, "rb").read() for e in (".fa", ".tsv", ".normal.fa"))
windows = sum(1 for _ in open(out + "/g.tsv")) - 1
res["tsv_rows"] = windows
if "--md5" in sys.argv:   # checksums of the product's streams (golden values for the full-size test once the oracle run beside them agreed)
    import hashlib
    def md5(path):
        h = hashlib.md5()
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        return h.hexdigest()
    res["md5"] = {"fasta": md5(out + "/g.fa"), "normal_fasta": md5(out + "/g.normal.fa"), "tsv": md5(out + "/g.tsv")}
print(json.dumps(res))
