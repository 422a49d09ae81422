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


def union_keys(local_keys, dist=None, device="cpu"):
    """Peptidome union, the one real exchange step of the path (config E): every rank contributes its sorted
    distinct u64 peptide keys, all ranks end up with the sorted distinct union. Variable-length all-gather = exchange the
    counts, pad to the maximum, all_gather (RCCL over xGMI on GPUs; ~10 MB per rank, so a single direct all-gather),
    then one merge-unique."""
    import torch
    t = torch.tensor(sorted(set(local_keys)), dtype=torch.int64, device=device)  # keys < 2^60: safe as int64
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return t.tolist()
    world = dist.get_world_size()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    m = int(max(c.item() for c in counts))
    padded = torch.full((max(m, 1),), -1, dtype=torch.int64, device=device)
    padded[: t.numel()] = t
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded)
    allk = torch.cat([b[: int(c.item())] for b, c in zip(bufs, counts)])
    return torch.unique(allk, sorted=True).tolist()
