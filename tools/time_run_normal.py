"""Timing helper: `normal` mode on a synthetic exome (seed, transcripts, depth, spacing from argv)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import microphaser_amd as m
a = sys.argv[1:]
seed, n = int(a[0]), int(a[1])
depth, spacing = (float(a[2]), float(a[3])) if len(a) > 3 else (30.0, 5.4)
ctx = m.Context(0)
t = time.perf_counter(); ds = ctx.synth(seed, n, depth, spacing); print("synth %.1f s" % (time.perf_counter() - t), flush=True)
t = time.perf_counter(); b = ds.batch(mode=m.MODE_NORMAL); print("plan %.2f s" % (time.perf_counter() - t), flush=True)
for i in range(3):
    t = time.perf_counter(); st = b.run(); dt = time.perf_counter() - t
    print("run %d: wall %.2f ms  k1 %.2f k2 %.2f k3 %.2f k3b %.2f total_ev %.2f attempts %d rpl %d groups %d recs %d windows_planned %d" % (
        i, dt * 1e3, st.k1_ms, st.k2_ms, st.k3_ms, st.k3b_ms, st.total_ms, st.attempts, st.rows_per_lane, st.n_groups, st.n_records, st.n_windows_planned), flush=True)
t = time.perf_counter(); r = b.results(); dt = time.perf_counter() - t
print("results %.2f s  windows %d  fasta %.1f MB tsv %.1f MB" % (dt, r.windows, len(r.fasta) / 1e6, len(r.tsv) / 1e6), flush=True)
