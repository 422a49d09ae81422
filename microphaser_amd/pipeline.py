"""BASELINE.json config E as a product driver, one process per GPU: `normal` + `build_reference` + `somatic` + `filter` on ONE
exome whose genes are dealt to the ranks by cost.

  rank r:  normal(genes_r) -> FASTA_r -> build_reference(FASTA_r) -> sorted distinct peptide keys_r          (all on GPU r)
  all:     all-gather of the key arrays (torch.distributed: RCCL over xGMI on GPUs, gloo on CPU) -> mp_peptides_union
           = the patient's normal peptidome (reference: peptides::build's HashSet, src/peptides.rs:148-186) - the one
           collective of the path
  rank r:  somatic(genes_r) -> shard (streams + per-gene offsets)
  rank 0:  shards gathered as tensors -> merged into GTF order -> filter against the peptidome (src/peptides.rs:221-709)

`somatic` / `normal` themselves need no collective: genes are independent (src/microphasing.rs:895-942).
"""
import numpy as np

from . import MODE_NORMAL, MODE_SOMATIC, STREAM_FASTA
from .shard import gather_shards, merge_by_gene, shard_of, union_keys


def normal_peptidome_keys(ctx, ds, local_genes, peptide_len):
    """This rank's share of the normal peptidome: `normal` on its genes, every window translated and de-duplicated on the GPU."""
    nb = ds.batch_genes(local_genes, window_len=3 * peptide_len, mode=MODE_NORMAL)
    nb.run()
    nres = nb.results(STREAM_FASTA)                # build_reference reads the FASTA only: the TSV text is not produced
    pep = ctx.peptidome(nres.fasta, peptide_len)   # keys only (no translated FASTA text)
    return pep.keys_np, nres


def somatic_shard(ds, local_genes, global_genes, peptide_len):
    sb = ds.batch_genes(local_genes, window_len=3 * peptide_len, mode=MODE_SOMATIC)
    sb.run()
    return shard_of(sb.results(), global_genes)


def config_e_rank(ctx, ds, local_genes, global_genes, peptide_len=9, dist=None, device="cpu"):
    """One rank of config E. ds holds (at least) this rank's genes; local_genes are their ordinals in ds, global_genes their
    ordinals in the whole exome (the merge key). Returns (merged somatic streams, peptidome, filter result) on rank 0 and
    (None, peptidome, None) elsewhere."""
    keys, _ = normal_peptidome_keys(ctx, ds, local_genes, peptide_len)
    peptidome = union_keys(ctx, keys, peptide_len, dist, device)          # the same on every rank
    shard = somatic_shard(ds, local_genes, global_genes, peptide_len)
    shards = gather_shards(shard, dist, dst=0, device=device)
    if shards is None:
        return None, peptidome, None
    merged = merge_by_gene(shards)
    filtered = ctx.filter(merged["tsv"], peptidome) if merged["tsv"] else None
    return merged, peptidome, filtered
