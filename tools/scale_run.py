"""Timing helper: kernel times of one `somatic` pass for growing synthetic exomes (config C shape)."""
import sys, os
sys.path.insert(0, os.getcwd())
import microphaser_amd as m
ctx = m.Context(0)
for n in [int(x) for x in sys.argv[1:]] or [1000, 2000, 4000, 8000]:
    ds = ctx.synth(2020, n)
    b = ds.batch()
    best = None
    for i in range(4):
        st = b.run()
        if best is None or st.k2_ms < best.k2_ms: best = st
    print("n %6d  k1 %.3f k2 %.3f k3 %.3f k3b %.3f  steps %d windows %d groups %d recs %d" % (
        n, best.k1_ms, best.k2_ms, best.k3_ms, best.k3b_ms, best.n_steps, best.n_windows_planned, best.n_groups, best.n_records), flush=True)
    del b, ds
