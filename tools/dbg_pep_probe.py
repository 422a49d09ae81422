import sys, os, time
sys.path.insert(0, os.getcwd())
import microphaser_amd as m
ctx = m.Context(0)
ds = ctx.synth(2020, 2500, 30.0, 5.4, gene_streams=True)
b = ds.batch(window_len=27, mode=m.MODE_NORMAL); b.run(); r = b.results(m.STREAM_FASTA); fa = r.fasta
print("fasta MB", len(fa)/1e6)
for k in range(2):
    t = time.perf_counter(); pep = ctx.peptidome(fa, 9); print("peptidome %.3f s, %d keys" % (time.perf_counter() - t, pep.keys_np.size))
