"""Differential stress test of the device paths against each other (no oracle needed, so it can sweep many exomes):
default plan (window-parallel replay + byte-substitution sequences) vs MP_SEQUENTIAL_REPLAY=1 + MP_GENERAL_WALK=1
(sequential state machine + general sequence walk). Any difference in FASTA / normal FASTA / TSV is a bug in one of them.

  python tools/stress_paths.py [first_seed] [n_exomes]
"""
import os, random, subprocess, sys
sys.path.insert(0, os.getcwd())
CHILD = r'''
import os, sys, hashlib
sys.path.insert(0, os.getcwd())
import microphaser_amd as m
seed, n, depth, spacing, indel, multi, soft = sys.argv[1:8]
ctx = m.Context(0)
ds = ctx.synth(int(seed), int(n), float(depth), float(spacing), indel_rate=float(indel), multiallelic_rate=float(multi), softmask_rate=float(soft))
h = hashlib.sha1()
errs = 0
for g in range(ds.num_genes):
    try:
        b = ds.batch(gene_lo=g, gene_hi=g + 1); b.run(); r = b.results()
        for x in (r.fasta, r.normal_fasta, r.tsv): h.update(x); h.update(b"|")
    except m.MicrophaserError as e:
        errs += 1; h.update(("ERR:" + str(e)[:40]).encode())
print(h.hexdigest(), errs)
'''
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = random.Random(first)
bad = 0
for k in range(count):
    seed = first + k
    depth = rng.choice([8, 20, 30, 45, 60, 110, 220])     # > ~90x: more than 64 candidate reads per window (multi-block replay)
    spacing = rng.choice([1.35, 2.0, 3.5, 5.4, 9.0, 20.0])   # 1.35 nt: more than 64 variants under a read (two mask words)
    indel = rng.choice([0, 0, 0, 0.03])
    multi = rng.choice([0, 0, 0.08])
    soft = rng.choice([0, 0, 0.4])
    n = rng.choice([12, 20]) if depth < 100 else 6
    args = [str(x) for x in (seed, n, depth, spacing, indel, multi, soft)]
    outs = []
    for env_extra in ({}, {"MP_SEQUENTIAL_REPLAY": "1", "MP_GENERAL_WALK": "1"}):
        env = dict(os.environ); env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", CHILD] + args, capture_output=True, text=True, env=env)
        outs.append(r.stdout.strip() if r.returncode == 0 else "FAILED " + r.stderr[-300:])
    ok = outs[0] == outs[1] and not outs[0].startswith("FAILED")
    bad += 0 if ok else 1
    print(("ok  " if ok else "DIFF"), " ".join(args), outs[0][:60] if ok else outs, flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
