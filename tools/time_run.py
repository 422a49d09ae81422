import sys, os, time
sys.path.insert(0, os.getcwd())
import microphaser_amd as m
ctx = m.Context(0)
ds = ctx.synth(2020, int(sys.argv[1]) if len(sys.argv) > 1 else 4000)
b = ds.batch()
for i in range(5):
    t = time.perf_counter(); st = b.run(); dt = time.perf_counter() - t
    print("run %d: wall %.2f ms  kernels %.2f ms (k1 %.2f k2 %.2f k3 %.2f k3b %.2f) total_ev %.2f attempts %d groups %d recs %d" % (i, dt*1e3, st.k1_ms+st.k2_ms+st.k3_ms+st.k3b_ms, st.k1_ms, st.k2_ms, st.k3_ms, st.k3b_ms, st.total_ms, st.attempts, st.n_groups, st.n_records))
