"""Repeat the product CLI on one set of synthetic files (written once) to separate run-to-run noise from changes.
  python tools/e2e_repeat.py <config B|C> <repeats> [ENV=VALUE ...]   # each ENV=VALUE set is timed `repeats` times, interleaved
"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import microphaser_amd as m
cfg = {"B": (1001, 1000, 30.0, 5.4), "C": (2020, 20000, 30.0, 5.4)}[sys.argv[1]]
reps = int(sys.argv[2])
variants = [dict()] + [dict([a.split("=", 1)]) for a in sys.argv[3:]]
out = tempfile.mkdtemp(prefix="mp_e2e_")
prefix = os.path.join(out, "synth")
ctx = m.Context(-1)
ds = ctx.synth(cfg[0], cfg[1], cfg[2], cfg[3]); ds.write(prefix); ds.close()
cli = os.path.join(ROOT, "microphaser_amd", "_lib", "microphaser")
times = {i: [] for i in range(len(variants))}
for r in range(reps):
    for i, v in enumerate(variants):
        t = time.perf_counter()
        with open(prefix + ".gtf", "rb") as g, open(out + "/g.fa", "wb") as o:
            rc = subprocess.run([cli, "somatic", prefix + ".bam", "--variants", prefix + ".vcf", "--ref", prefix + ".fa", "--tsv", out + "/g.tsv",
                                 "--normal-output", out + "/g.normal.fa"], stdin=g, stdout=o, env=dict(os.environ, **v)).returncode
        times[i].append(time.perf_counter() - t)
        if rc != 0: raise SystemExit("CLI failed")
for i, v in enumerate(variants):
    print(v or "default", " ".join("%.2f" % t for t in times[i]), "min %.2f s" % min(times[i]), flush=True)
