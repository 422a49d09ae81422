"""Host-leg timing of one batch (MP_DEBUG=1 prints the phases inside each call): plan + pack + H2D, run, D2H + consume.
  MP_DEBUG=1 python tools/time_e2e_phases.py [transcripts] [seed]
"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import microphaser_amd as m
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2020
ctx = m.Context(0)
t = time.perf_counter(); ds = ctx.synth(seed, n); print("synth %.2f s" % (time.perf_counter() - t), flush=True)
t = time.perf_counter(); b = ds.batch(); print("batch_create %.2f s" % (time.perf_counter() - t), flush=True)
t = time.perf_counter(); b.run(); print("run (cold) %.3f s" % (time.perf_counter() - t), flush=True)
t = time.perf_counter(); b.run(); print("run (warm) %.3f s" % (time.perf_counter() - t), flush=True)
t = time.perf_counter(); r = b.results(); print("results %.2f s" % (time.perf_counter() - t), flush=True)
print("bytes: fasta %d normal %d tsv %d" % (r.size("fasta"), r.size("normal_fasta"), r.size("tsv")))
