"""The N>1 path on CPU: world_size-2 gloo runs (no GPU).

test_two_rank_gloo_shard_and_merge: ONE synthetic exome, its genes dealt to the two ranks by the product's cost-weighted partition
(lpt_partition on mp_synth_gene_costs); every rank materialises only its genes (per-gene random streams), plans them with the
product's planner (mp_batch_create_genes on a host-only context), and - the kernels need a GPU - takes its genes' output bytes
from the CPU oracle, gene by gene, in the shard format the product's consumer reports (streams + per-gene offsets). The shards
travel as tensors (gather_shards), rank 0 merges them by gene ordinal (merge_by_gene) and the result must be the single-process
output of the whole exome. The GPU suite checks the same merge on shards the device produced.
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
from microphaser_amd.shard import lpt_partition, merge_by_gene, gather_shards
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank, world = dist.get_rank(), dist.get_world_size()
seed, n = 41, 10
ctx = m.Context(-1)
costs = ctx.synth_gene_costs(seed, n)
parts = lpt_partition(costs, world)
mine = parts[rank]
ds = ctx.synth(seed, n, keep=mine)                 # only this rank's genes of the exome
assert ds.num_genes == len(mine)
ds.batch_genes(range(len(mine)))                   # the product planner takes the shard
# the shard's bytes, gene by gene, from the oracle run on the WHOLE exome (global gene ordinals)
shard = dict(genes=mine, fasta=b"", normal_fasta=b"", tsv=b"", off=[[0], [0], [0]], windows=0)
header = b""
for g in mine:
    prefix = os.path.join(%(tmp)r, "g%%d" %% g)
    r = subprocess.run([%(cli)r, "synth", "--seed", str(seed), "--transcripts", str(n), "--gene-streams", "--genes", "%%d:%%d" %% (g, g + 1), "--prefix", prefix],
                       capture_output=True, check=True)
    shard["windows"] += json.loads(r.stdout)["windows"]
    fa, nfa, tsv = (open(prefix + e, "rb").read() for e in (".fa", ".normal.fa", ".tsv"))
    if tsv:
        h, body = tsv.split(b"\n", 1)
        if not shard["tsv"]:
            shard["tsv"] = h + b"\n"
            shard["off"][2] = [len(shard["tsv"])] * len(shard["off"][2])
        shard["tsv"] += body
    shard["fasta"] += fa
    shard["normal_fasta"] += nfa
    for k, name in enumerate(("fasta", "normal_fasta", "tsv")):
        shard["off"][k].append(len(shard[name]))
got = gather_shards(shard, dist)
if rank == 0:
    merged = merge_by_gene(got)
    open(os.path.join(%(tmp)r, "merged.json"), "w").write(json.dumps({"windows": merged["windows"], "parts": parts}))
    for k, ext in (("fasta", "fa"), ("normal_fasta", "normal.fa"), ("tsv", "tsv")):
        open(os.path.join(%(tmp)r, "merged." + ext), "wb").write(merged[k])
dist.barrier()
dist.destroy_process_group()
'''


def test_cost_ranges_tile_the_genes_in_order():
    """pipeline.cost_ranges: contiguous chunks of about equal cost, none empty, covering every gene once (phase_chunked's chunks)."""
    import random
    from microphaser_amd.pipeline import cost_ranges
    rnd = random.Random(5)
    for n in (0, 1, 2, 7, 100, 1000):
        costs = [rnd.randrange(1, 1000) for _ in range(n)]
        for k in (1, 2, 4, 8, 200):
            r = cost_ranges(costs, k)
            assert len(r) <= max(1, k) and all(hi > lo for lo, hi in r)
            assert [lo for lo, _ in r] == ([0] + [hi for _, hi in r][:-1] if r else [])
            assert (r[-1][1] if r else 0) == n
            if n >= 100 and k <= 8:
                total = sum(costs)
                assert max(sum(costs[lo:hi]) for lo, hi in r) <= total / k + 1000


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
    r = subprocess.run([ORACLE_CLI, "synth", "--seed", "41", "--transcripts", "10", "--gene-streams", "--prefix", str(whole)], capture_output=True, check=True)
    st = json.loads(r.stdout)
    meta = json.loads((tmp_path / "merged.json").read_text())
    assert meta["windows"] == st["windows"]
    assert sorted(g for p in meta["parts"] for g in p) == list(range(10)) and all(len(p) >= 3 for p in meta["parts"])
    assert meta["parts"][0] != list(range(len(meta["parts"][0])))     # a cost-weighted deal, not a contiguous range
    for ext in ("fa", "normal.fa", "tsv"):
        assert (tmp_path / ("merged." + ext)).read_bytes() == open(str(whole) + "." + ext, "rb").read()
    assert (tmp_path / "merged.tsv").read_bytes().count(b"\n") > 50


UNION_WORKER = r'''
import json, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
import microphaser_amd as m
from microphaser_amd.shard import union_keys
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
local = [3, 5, 9, 1 << 44] if rank == 0 else [5, 7, 9, 11, 12, (1 << 44) + 1]
u = union_keys(m.Context(-1), np.array(local, dtype=np.uint64), 9, dist)    # tensors over gloo, merged by mp_peptides_union
open(%(tmp)r + "/union%%d.json" %% rank, "w").write(json.dumps({"keys": u.keys, "n_binary": len(m.decode_bincode_set(u.binary))}))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_peptidome_union(built, tmp_path):
    port = 31500 + (os.getpid() % 2000)
    script = tmp_path / "uworker.py"
    script.write_text(UNION_WORKER % dict(root=ROOT, port=port, tmp=str(tmp_path)))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err.decode()[-2000:]
    want = sorted({3, 5, 9, 1 << 44, 7, 11, 12, (1 << 44) + 1})
    for r in range(2):
        got = json.loads((tmp_path / ("union%d.json" % r)).read_text())
        assert got["keys"] == want and got["n_binary"] == len(want)
