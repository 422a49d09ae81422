"""The N>1 path on CPU: world_size-2 gloo run of gene sharding + ordered merge (no GPU).

Each rank plans its shard with the host-only context (the product's planner) and produces its shard's output streams
with the CPU oracle standing in for the kernels; rank 0 gathers and merges them and compares with a single-process run.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ORACLE_CLI, ROOT

WORKER = r'''
import json, os, subprocess, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
import microphaser_amd as m
from microphaser_amd.shard import shard_range, merge_streams, gather_streams
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank, world = dist.get_rank(), dist.get_world_size()
seed, n = 41, 10
ctx = m.Context(-1)
ds = ctx.synth(seed, n)
lo, hi = shard_range(ds.num_genes, rank, world)
ds.batch(gene_lo=lo, gene_hi=hi)   # the planner accepts the shard
prefix = os.path.join(%(tmp)r, "shard%%d" %% rank)
r = subprocess.run([%(cli)r, "synth", "--seed", str(seed), "--transcripts", str(n), "--genes", "%%d:%%d" %% (lo, hi), "--prefix", prefix],
                   capture_output=True, check=True)
st = json.loads(r.stdout)
local = dict(fasta=open(prefix + ".fa", "rb").read(), normal_fasta=open(prefix + ".normal.fa", "rb").read(),
             tsv=open(prefix + ".tsv", "rb").read(), windows=st["windows"])
parts = gather_streams(local, dist)
if rank == 0:
    merged = merge_streams(parts)
    open(os.path.join(%(tmp)r, "merged.json"), "w").write(json.dumps({"windows": sum(p["windows"] for p in parts),
        "ranges": [shard_range(ds.num_genes, r_, world) for r_ in range(world)]}))
    for k, ext in (("fasta", "fa"), ("normal_fasta", "normal.fa"), ("tsv", "tsv")):
        open(os.path.join(%(tmp)r, "merged." + ext), "wb").write(merged[k])
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_range_tiles_in_order():
    from microphaser_amd.shard import shard_range
    for n in (0, 1, 7, 8, 20000):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_two_rank_gloo_shard_and_merge(built, tmp_path):
    port = 29500 + (os.getpid() % 2000)
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, port=port, tmp=str(tmp_path), cli=ORACLE_CLI))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err.decode()[-2000:]
    whole = tmp_path / "whole"
    r = subprocess.run([ORACLE_CLI, "synth", "--seed", "41", "--transcripts", "10", "--prefix", str(whole)], capture_output=True, check=True)
    st = json.loads(r.stdout)
    meta = json.loads((tmp_path / "merged.json").read_text())
    assert meta["windows"] == st["windows"]
    for ext in ("fa", "normal.fa", "tsv"):
        assert (tmp_path / ("merged." + ext)).read_bytes() == open(str(whole) + "." + ext, "rb").read()
    assert (tmp_path / "merged.tsv").read_bytes().count(b"\n") > 50


UNION_WORKER = r'''
import json, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from microphaser_amd.shard import union_keys
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
local = [3, 5, 9, 1 << 44] if rank == 0 else [5, 7, 11, 12, (1 << 44) + 1, 9]
u = union_keys(local, dist)
open(%(tmp)r + "/union%%d.json" %% rank, "w").write(json.dumps(u))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_peptidome_union(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    script = tmp_path / "uworker.py"
    script.write_text(UNION_WORKER % dict(root=ROOT, port=port, tmp=str(tmp_path)))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err.decode()[-2000:]
    want = sorted({3, 5, 9, 1 << 44, 7, 11, 12, (1 << 44) + 1})
    for r in range(2):
        assert json.loads((tmp_path / ("union%d.json" % r)).read_text()) == want
