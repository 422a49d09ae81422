#!/usr/bin/env python3
"""BASELINE.json config E at full size on ONE GPU, stage by stage, with wall times (measurement tool):

  normal (germline + somatic calls, every haplotype of every window) -> build_reference -l 9 -> somatic -> filter

on the synthetic 20k-transcript exome (seed 2020, per-gene random streams). `normal` emits every window (~4x the text of `somatic`),
so the exome is walked in gene chunks - exactly what the ranks of a multi-GPU run do with their shards (microphaser_amd/pipeline.py):
per chunk normal -> FASTA -> build_reference -> sorted distinct keys; the chunks' key arrays are merged by mp_peptides_union; then
`somatic` per chunk, shards merged by gene, and one `filter` over the merged TSV.

  python tools/config_e_run.py [--transcripts 20000] [--chunks 8] [--out gpurun_out/config_e.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import microphaser_amd as m
from microphaser_amd.shard import merge_by_gene, shard_of


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--transcripts", type=int, default=20000)
    ap.add_argument("--chunks", type=int, default=8)
    ap.add_argument("--peptide-len", type=int, default=9)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    L = a.peptide_len
    ctx = m.Context(0)
    t = {}
    t0 = time.perf_counter()
    ds = ctx.synth(2020, a.transcripts, 30.0, 5.4, gene_streams=True)
    t["generate_s"] = time.perf_counter() - t0
    n = ds.num_genes
    cuts = [n * k // a.chunks for k in range(a.chunks + 1)]
    stats = dict(normal_windows=0, normal_fasta_bytes=0, normal_tsv_bytes=0, k2n_ms=0.0, k3_ms=0.0, k3b_ms=0.0, k1_ms=0.0, peptide_windows=0)
    key_arrays = []
    t_normal = t_build = 0.0
    for c in range(a.chunks):
        genes = list(range(cuts[c], cuts[c + 1]))
        t0 = time.perf_counter()
        b = ds.batch_genes(genes, window_len=3 * L, mode=m.MODE_NORMAL)
        st = b.run()
        res = b.results(m.STREAM_FASTA)       # build_reference reads the FASTA only
        fa = res.fasta
        stats["normal_windows"] += res.windows
        stats["normal_fasta_bytes"] += len(fa)
        stats["normal_tsv_bytes"] += res.size("tsv")
        stats["k1_ms"] += st.k1_ms; stats["k2n_ms"] += st.k2seq_ms; stats["k3_ms"] += st.k3_ms; stats["k3b_ms"] += st.k3b_ms
        res.close(); b.close()
        t_normal += time.perf_counter() - t0
        t0 = time.perf_counter()
        pep = ctx.peptidome(fa, L)            # keys only: nobody reads the translated FASTA in this pipeline
        stats["peptide_windows"] += pep.count
        key_arrays.append(pep.keys_np)
        del fa, pep
        t_build += time.perf_counter() - t0
        print("chunk %d/%d: normal %.1f s, build_reference %.1f s so far" % (c + 1, a.chunks, t_normal, t_build), flush=True)
    t["normal_s"], t["build_reference_s"] = t_normal, t_build
    t0 = time.perf_counter()
    peptidome = ctx.peptides_union(key_arrays, L)
    t["peptides_union_s"] = time.perf_counter() - t0
    stats["peptidome_size"] = int(peptidome.keys_np.size)
    t0 = time.perf_counter()
    shards = []
    som_windows = 0
    for c in range(a.chunks):
        genes = list(range(cuts[c], cuts[c + 1]))
        b = ds.batch_genes(genes)
        b.run()
        r = b.results()
        som_windows += r.windows
        shards.append(shard_of(r, genes))
        r.close(); b.close()
    merged = merge_by_gene(shards)
    del shards
    t["somatic_s"] = time.perf_counter() - t0
    stats["somatic_windows"] = som_windows
    stats["somatic_tsv_rows"] = merged["tsv"].count(b"\n") - 1
    t0 = time.perf_counter()
    f = ctx.filter(merged["tsv"], peptidome)        # the peptidome handle: its keys go to the GPU as they are
    t["filter_s"] = time.perf_counter() - t0
    stats.update(filter_rows=f.rows, filter_kept=f.kept, filter_removed=f.removed, filter_groups=f.groups)
    t["total_s"] = sum(v for k, v in t.items() if k != "generate_s")
    out = {"config": "E: normal + build_reference -l %d + somatic + filter, %d transcripts, %d gene chunks, one MI355X" % (L, a.transcripts, a.chunks),
           "wall_s": t, "stats": stats}
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
