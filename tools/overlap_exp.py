"""Experiment: two half-size batches on two contexts (streams) of one GPU, run back to back vs concurrently from two host threads."""
import sys, os, time, threading
sys.path.insert(0, os.getcwd())
import microphaser_amd as m
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ctxs = [m.Context(0), m.Context(0)]
batches = []
for i, c in enumerate(ctxs):
    ds = c.synth(2020 + i, n)
    b = ds.batch()
    b.run()
    batches.append((ds, b))
def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t)
    return best * 1e3
def serial():
    for _, b in batches: b.run()
def conc(delay=0.0):
    def f(b, d):
        if d: time.sleep(d)
        b.run()
    th = [threading.Thread(target=f, args=(b, delay * i)) for i, (_, b) in enumerate(batches)]
    for t in th: t.start()
    for t in th: t.join()
print("serial      %.2f ms" % timed(serial))
for d in (0.0, 0.0005, 0.001, 0.0015, 0.002):
    print("concurrent stagger %.1f ms: %.2f ms" % (d * 1e3, timed(lambda: conc(d))))
