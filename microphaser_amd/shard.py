"""Gene sharding across ranks (one process per GPU) and the ordered merge of the per-rank outputs.

Genes are independent units of the phasing path (reference: phase_gene is called once per gene and shares no
state across genes, src/microphasing.rs:1963-1979), so `somatic` needs no data-path collective: every rank phases
a contiguous range of genes and the three output streams are concatenated in gene order. The only communication is
the final gather of the (already formatted) streams, done with torch.distributed (RCCL on GPUs, gloo on CPU).
"""


def shard_range(n_genes, rank, world):
    """Contiguous gene range [lo, hi) of `rank`; ranges of ranks 0..world-1 tile [0, n_genes) in order."""
    base, extra = divmod(n_genes, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def merge_streams(parts):
    """parts: list of dict(fasta=bytes, normal_fasta=bytes, tsv=bytes) in rank (= gene) order.
    FASTA streams concatenate; the TSV header is written once, by the first shard that emitted a record
    (the reference's csv writer emits it with the first record only, src/common.rs:350-373)."""
    fasta = b"".join(p["fasta"] for p in parts)
    normal = b"".join(p["normal_fasta"] for p in parts)
    tsv = b""
    for p in parts:
        t = p["tsv"]
        if not t:
            continue
        tsv += t if not tsv else t.split(b"\n", 1)[1]
    return dict(fasta=fasta, normal_fasta=normal, tsv=tsv)


def gather_streams(local, dist=None, dst=0):
    """Gather every rank's streams on `dst` (list in rank order) - the only exchange step of `somatic`."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local, out, dst=dst)
    return out
