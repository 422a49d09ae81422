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


def cost_ranges(costs, n_chunks):
    """Contiguous gene ranges [lo, hi) of about equal estimated cost (at most n_chunks of them, none empty), in gene order."""
    n, total = len(costs), sum(costs)
    bounds, acc, k = [0], 0, 1
    for g, c in enumerate(costs):
        acc += c
        if k < n_chunks and acc * n_chunks >= total * k and g + 1 < n:
            bounds.append(g + 1)
            k += 1
    bounds.append(n)
    bounds = sorted(set(bounds))
    return [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1) if bounds[i + 1] > bounds[i]]


def phase_chunked(ds, n_chunks=4, device=0, window_len=27, contexts=None):
    """`somatic` on ONE GPU in gene chunks whose legs overlap: while chunk i is phased, downloaded and consumed (host threads + the
    download DMA), chunk i + 1 is planned, packed and uploaded on a second context of the same GPU (host threads + the upload DMA) -
    the transfers and the GPU pass of one chunk hide behind the host work of the other (chunked, overlapped
    transfers). Genes are independent (src/microphasing.rs:895-942), so the chunks' streams concatenate into the single-batch
    output (TSV header once: shard.merge_streams). Returns (list of Results in gene order, windows); the caller reads or merges
    the streams (the Results copy their text out of the library only when asked)."""
    import ctypes
    import queue
    import threading
    from . import Batch, Context, lib
    ranges = cost_ranges(ds.gene_costs(), n_chunks)
    own = contexts is None
    ctxs = [Context(device), Context(device)] if own else list(contexts)
    free = [threading.Semaphore(1) for _ in ctxs]
    ready = queue.Queue()
    errors = []

    def producer():
        try:
            for i, (lo, hi) in enumerate(ranges):
                c = i % len(ctxs)
                free[c].acquire()   # the context holds one batch at a time: wait until its previous chunk has been consumed
                if errors:
                    break
                h = ctypes.c_void_p()
                ctxs[c]._check(lib().mp_batch_create(ctxs[c]._h, ds._h, MODE_SOMATIC, window_len, lo, hi, ctypes.byref(h)))
                ready.put((c, Batch(ctxs[c], h, ds)))
        except Exception as e:   # noqa: BLE001 - handed to the caller's thread
            errors.append(e)
        finally:
            ready.put(None)

    t = threading.Thread(target=producer, name="mp-chunk-planner")
    t.start()
    results, windows = [], 0
    try:
        while True:
            item = ready.get()
            if item is None:
                break
            c, b = item
            try:
                b.run()
                r = b.results()
                results.append(r)
                windows += r.windows
            except Exception as e:   # noqa: BLE001
                errors.append(e)
            finally:
                b.close()
                free[c].release()
    finally:
        for s in free:   # (a producer blocked on a context after an error must be able to leave)
            s.release()
        t.join()
        if own:
            for c in ctxs:
                c.close()
    if errors:
        raise errors[0]
    return results, windows
