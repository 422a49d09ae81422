import sys, os, subprocess, json
sys.path.insert(0, os.getcwd())
import microphaser_amd as m
for f in ("gpurun_out/trace_g.txt","gpurun_out/trace_o.txt"):
    if os.path.exists(f): os.remove(f)
os.environ["MP_TRACE"]="gpurun_out/trace_g.txt"
ctx = m.Context(0)
seed, n = 7, 40
res = ctx.synth(seed, n).phase()
os.environ["MP_TRACE"]="gpurun_out/trace_o.txt"
subprocess.run(["oracle/_build/oracle_cli", "synth", "--seed", str(seed), "--transcripts", str(n), "--prefix", "gpurun_out/o"], check=True)
open("gpurun_out/g.tsv","wb").write(res.tsv)
a = open("gpurun_out/o.tsv","rb").read().split(b"\n"); b = res.tsv.split(b"\n")
k = 0
for i,(x,y) in enumerate(zip(a,b)):
    if x != y:
        print("line", i); print("O", x.decode()); print("G", y.decode()); k += 1
        if k > 2: break
print(len(a), len(b))
