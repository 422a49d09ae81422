"""Gene sharding across ranks (one process per GPU), the ordered merge of the per-rank outputs and the peptidome exchange.

Genes are independent units of the phasing path (reference: phase_gene is called once per gene and shares no state across
genes, src/microphasing.rs:895-942, :1963-1979), so `somatic` / `normal` need no data-path collective: the genes of ONE exome
are dealt to the ranks by estimated cost (longest processing time first on CDS_nt x depth), every rank phases
its own genes, and the output streams are merged back into GTF order - the reference's output order - from the per-gene byte
offsets the library reports (mp_results_gene_offsets). The one real exchange step of the path is the peptidome union of
`build_reference` (config E): sorted distinct u64 keys, all-gathered as tensors (RCCL over xGMI on GPUs, gloo on CPU) and
merged by the library (mp_peptides_union).
"""
import heapq


def shard_range(n_genes, rank, world):
    """Contiguous gene range [lo, hi) of `rank`; ranges of ranks 0..world-1 tile [0, n_genes) in order."""
    base, extra = divmod(n_genes, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def lpt_partition(costs, world):
    """Greedy longest-processing-time-first: genes in descending cost order, each to the least loaded rank.
    Returns `world` ascending gene lists (deterministic: ties broken by gene ordinal / rank)."""
    heap = [(0, r) for r in range(world)]
    heapq.heapify(heap)
    parts = [[] for _ in range(world)]
    for g in sorted(range(len(costs)), key=lambda g: (-costs[g], g)):
        load, r = heapq.heappop(heap)
        parts[r].append(g)
        heapq.heappush(heap, (load + costs[g], r))
    for p in parts:
        p.sort()
    return parts


def shard_of(results, genes):
    """dict describing one rank's output: its gene list, the three streams and the per-gene offsets in each of them."""
    return dict(genes=list(genes), fasta=results.fasta, normal_fasta=results.normal_fasta, tsv=results.tsv,
                off=[results.gene_offsets(k) for k in range(3)], windows=results.windows)


def merge_streams(parts):
    """parts: list of dict(fasta=bytes, normal_fasta=bytes, tsv=bytes) in gene order (contiguous ranges).
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


def merge_by_gene(shards):
    """shards: list of shard_of() dicts whose gene lists partition the exome. Returns the three streams in global gene order
    (what a single-GPU run writes): gene g's bytes are cut out of its shard with the shard's per-gene offsets; the TSV header line
    is kept once."""
    owner = {}
    for s in shards:
        for k, g in enumerate(s["genes"]):
            owner[g] = (s, k)
    out = {"fasta": [], "normal_fasta": [], "tsv": []}
    header = b""
    for s in shards:
        if s["tsv"] and not header:
            header = s["tsv"][: s["off"][2][0]]
    views = {id(s): {name: memoryview(s[name]) for name in out} for s in shards}   # slices without copies; b"".join copies once
    run_s, run_k0, run_k1 = None, 0, 0     # a run of consecutive genes of one shard is one slice per stream

    def flush():
        if run_s is None:
            return
        for which, name in enumerate(("fasta", "normal_fasta", "tsv")):
            off = run_s["off"][which]
            if off:
                out[name].append(views[id(run_s)][name][off[run_k0]: off[run_k1]])

    for g in sorted(owner):
        s, k = owner[g]
        if s is run_s and k == run_k1:
            run_k1 = k + 1
        else:
            flush()
            run_s, run_k0, run_k1 = s, k, k + 1
    flush()
    tsv_body = b"".join(out["tsv"])
    return dict(fasta=b"".join(out["fasta"]), normal_fasta=b"".join(out["normal_fasta"]), tsv=(header + tsv_body) if tsv_body else b"",
                windows=sum(s.get("windows", 0) for s in shards))


def _bytes_tensor(torch, data, device):
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8) if data else torch.zeros(0, dtype=torch.uint8)
    return t.to(device)


def gather_shards(local, dist=None, dst=0, device="cpu"):
    """Gather every rank's shard_of() dict on `dst` (list in rank order; None elsewhere): the byte streams and offset tables travel as
    uint8 / int64 tensors, not as pickled Python objects. Only `dst` merges, so only `dst` receives: the field sizes are all-gathered
    (a few integers per rank), then every other rank SENDS its fields to `dst` point to point at their exact sizes - no padding, and no
    rank holds world x max-size bytes of FASTA / TSV it will never read (an all_gather of config E's streams is gigabytes per rank)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    fields = [("fasta", "u8"), ("normal_fasta", "u8"), ("tsv", "u8"), ("genes", "i64"), ("off0", "i64"), ("off1", "i64"), ("off2", "i64")]
    loc = {"fasta": local["fasta"], "normal_fasta": local["normal_fasta"], "tsv": local["tsv"], "genes": local["genes"],
           "off0": local["off"][0], "off1": local["off"][1], "off2": local["off"][2]}
    sizes = torch.tensor([len(loc[n]) for n, _ in fields] + [int(local.get("windows", 0))], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = [s.cpu().tolist() for s in all_sizes]
    if rank != dst:
        for name, kind in fields:
            if len(loc[name]) == 0:
                continue   # (nothing to send: dst knows the size)
            t = _bytes_tensor(torch, loc[name], device) if kind == "u8" else torch.tensor(loc[name], dtype=torch.int64, device=device)
            dist.send(t, dst=dst)
            del t
        return None
    out = []
    for r in range(world):
        if r == dst:
            out.append(dict(genes=list(local["genes"]), fasta=local["fasta"], normal_fasta=local["normal_fasta"], tsv=local["tsv"],
                            off=[list(o) for o in local["off"]], windows=int(local.get("windows", 0))))
            continue
        g = {}
        for i, (name, kind) in enumerate(fields):
            n = int(all_sizes[r][i])
            buf = torch.empty(n, dtype=torch.uint8 if kind == "u8" else torch.int64, device=device)
            if n:
                dist.recv(buf, src=r)
            g[name] = buf.cpu()
            del buf
        out.append(dict(genes=g["genes"].tolist(), fasta=bytes(g["fasta"].numpy()), normal_fasta=bytes(g["normal_fasta"].numpy()),
                        tsv=bytes(g["tsv"].numpy()), off=[g["off0"].tolist(), g["off1"].tolist(), g["off2"].tolist()],
                        windows=int(all_sizes[r][len(fields)])))
    return out


def allgather_keys(local_keys, dist=None, device="cpu"):
    """The exchange step of the multi-GPU build_reference: every rank contributes its sorted distinct u64 peptide keys (a numpy
    uint64 array or a torch int64 tensor - keys are < 2^60), every rank gets the list of all ranks' arrays (numpy uint64).
    Variable-length all-gather = exchange the counts, pad to the maximum, one all_gather. Size at config E: the 20000-transcript
    exome's peptidome is 70 M distinct 9-mers = 560 MB of keys (measured, profiles/r04o_config_e_20k_one_gpu.json); every distinct key
    is on at least one rank, so the ranks' arrays sum to >= 560 MB - >= 70 MB per rank at N = 8 - and every rank receives that sum.
    xGMI is point to point (7 links x ~153 GB/s per GPU): ~0.6 GB per rank is a few milliseconds for RCCL's all-gather, small
    beside the sort / unique that produced the keys. (Unmeasured on a multi-GPU node.)"""
    import numpy as np
    import torch
    if isinstance(local_keys, torch.Tensor):
        t = local_keys.to(device=device, dtype=torch.int64)
    else:
        t = torch.from_numpy(np.ascontiguousarray(local_keys, dtype=np.uint64).view(np.int64)).to(device)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [t.cpu().numpy().view(np.uint64)]
    world = dist.get_world_size()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    m = max(1, max(int(c.item()) for c in counts))
    padded = torch.zeros(m, dtype=torch.int64, device=device)
    padded[: t.numel()] = t
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded)
    return [b[: int(c.item())].cpu().numpy().view(np.uint64) for b, c in zip(bufs, counts)]


def union_keys(ctx, local_keys, peptide_len, dist=None, device="cpu"):
    """All-gather the ranks' key arrays and merge them in the library (mp_peptides_union): the peptidome of the whole exome, the
    same on every rank (Peptides: .keys, .binary)."""
    return ctx.peptides_union(allgather_keys(local_keys, dist, device), peptide_len)


def gather_streams(local, dist=None, dst=0):
    """(kept for contiguous-range shards) Gather every rank's streams on `dst` as Python objects."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local, out, dst=dst)
    return out
