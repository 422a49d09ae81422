#!/usr/bin/env python3
"""Benchmark of the phasing hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU)

Metric (BASELINE.json): peptide-windows/s, `somatic`, 9-mer (27 nt), on the synthetic 20k-transcript
whole exome (30x, ~5 variant sites per window). A "step" is one pass of the
hot path (K1 pileup bits -> K2 window replay -> K3 window sequences + ids) over the batch, with
the packed inputs already resident in HBM and the results left in HBM.

Multi-GPU (BASELINE.json configs[2]: ONE 20k-transcript exome sharded across the GPUs): genes are
independent units, so every rank takes the genes a cost-weighted partition gives it (longest
processing time first on CDS_nt x depth), generates only those (the generator draws
every gene from its own random stream, so the N ranks hold disjoint parts of the SAME exome) and
phases them with no data-path collective. value = windows of the whole exome / slowest rank's
time -> "scaling": "strong". N=1 runs the same exome on one GPU.

The JSON line also carries
  roofline                 - the dominant unit's algorithmic bytes / its HIP-event time vs the 8 TB/s HBM peak (a unit = one kernel, or the
                             three window kernels that run concurrently, priced together over the phase's wall time)
  roofline_units, pass_hbm_frac, valu_issue - the same for every unit, for the whole pass, and the pass's VALU issue rate vs the chip's
  cpu_baseline             - the CPU oracle (single thread, like the reference) on a bounded sample of the workload
  cpu_baseline_all_cores   - the same oracle with the sample's genes sharded over all host cores
  end_to_end               - plan+pack+H2D, pass, D2H+consume wall times of rank 0 and the in-memory end-to-end rate
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (seed, transcripts, depth, variant spacing)
    "B": (1001, 1000, 30.0, 5.4),
    "C": (2020, 20000, 30.0, 5.4),
    "D": (5005, 500, 500.0, 1.35),
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_PER_S = 256 * 4 * 0.5 * 2.4e9   # wave64 VALU instructions per second: 256 CUs x 4 SIMD-32, 2 cycles each, 2.4 GHz (same guide)


def effective_cores():
    """Host cores this process may use: the affinity mask, capped by the cgroup CPU quota (a GPU box gives one GPU's job a share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg_name, sample_transcripts, threads=1):
    """The CPU oracle on a bounded sample of the workload (same generator, seed and parameters, fewer transcripts)."""
    seed, _n, depth, spacing = CONFIGS[cfg_name]
    cli = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
    r = subprocess.run([cli, "synth", "--seed", str(seed), "--transcripts", str(sample_transcripts), "--depth", str(depth),
                        "--spacing", str(spacing), "--gene-streams", "--threads", str(threads)], capture_output=True, text=True)
    if r.returncode != 0:
        return None
    st = json.loads(r.stdout)
    return {
        "value": st["windows"] / st["phase_seconds"], "unit": "peptide-windows/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
        "sample": "%d-transcript exome from the same generator and parameters (seed %d, %gx, spacing %g nt): %d windows in %.2f s on %d thread(s); "
                  "phasing only, inputs in memory (ingest excluded); the Rust reference cannot be built here, the CPU oracle (C++ -O2) stands in - "
                  "single-threaded like the reference, or with the genes sharded over the host threads"
                  % (sample_transcripts, seed, depth, spacing, st["windows"], st["phase_seconds"], threads),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C", choices=sorted(CONFIGS))
    ap.add_argument("--transcripts", type=int, default=0, help="override the transcript count (debugging)")
    ap.add_argument("--cpu-sample", type=int, default=1000, help="transcripts in the single-thread CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-sample-all", type=int, default=4000, help="transcripts in the all-cores CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-consume", action="store_true", help="skip the end-to-end D2H + consumer leg")
    ap.add_argument("--e2e-chunks", type=int, default=8, help="gene chunks of the overlapped end-to-end leg (end_to_end.chunked; <= 1 skips it)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)

    import torch
    dist = None
    # MP_BENCH_REHEARSE=1: run the N-rank code path on a box with ONE GPU (all ranks on device 0, gloo for the two reductions) -
    # a functional rehearsal of the launch contract, never a measurement (the line says so)
    rehearse = os.environ.get("MP_BENCH_REHEARSE") == "1"
    device_index = 0 if rehearse else local_rank
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(device_index)
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import microphaser_amd as m
    import __graft_entry__ as entry
    if rank == 0 and not os.path.exists(m.LIB_PATH):
        entry.build()
    if dist is not None:
        dist.barrier()

    seed, n_tx, depth, spacing = CONFIGS[args.config]
    if args.transcripts:
        n_tx = args.transcripts
    n_cores = effective_cores()
    if world > 1:   # the ranks of one node share its host cores (planner / consumer threads)
        os.environ.setdefault("MP_THREADS", str(max(1, n_cores // world)))
    ctx = m.Context(device_index)
    t0 = time.perf_counter()
    # one exome for all ranks: cost-weighted partition of its genes, every rank materialises only its own
    my_genes = None
    if world > 1:
        from microphaser_amd.shard import lpt_partition
        costs = ctx.synth_gene_costs(seed, n_tx, depth, spacing)
        parts = lpt_partition(costs, world)
        my_genes = parts[rank]
        my_cost, total_cost = sum(costs[g] for g in my_genes), sum(costs)
    ds = ctx.synth(seed, n_tx, depth, spacing, gene_streams=True, keep=my_genes)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    batch = ds.batch(window_len=27)   # plan + pack + H2D of this rank's genes: inputs are resident in HBM from here on
    t_plan = time.perf_counter() - t0

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    st = None
    for _ in range(args.warmup):
        st = batch.run()
    sync_all()
    t0 = time.perf_counter()
    k1 = k2s = k2a = k2l = k2w = k3 = k3b = k2win = 0.0
    for _ in range(args.steps):
        st = batch.run()          # returns after the launch stream has drained (results in HBM)
        k1 += st.k1_ms
        k2s += st.k2seq_ms
        k2a += st.k2a_ms
        k2l += st.k2l_ms
        k2w += st.k2w_ms
        k2win += st.k2win_ms
        k3 += st.k3_ms
        k3b += st.k3b_ms
    sync_all()
    elapsed = time.perf_counter() - t0

    # the unit count: main-ORF print_haplotypes calls the reference makes = what the consumer actually walks
    t0 = time.perf_counter()
    windows = None
    emitted = None
    if not args.no_consume:
        res = batch.results()
        windows = res.windows
    t_consume = time.perf_counter() - t0
    if not args.no_consume:
        emitted = max(0, res.tsv.count(b"\n") - 1)   # TSV rows = emitted haplotypes (outside the timed leg: this copies the text into Python)
    if windows is None:
        windows = st.n_windows_planned

    if dist is not None:
        red_dev = "cpu" if rehearse else "cuda"
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wsum = torch.tensor([float(windows)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(wsum, op=dist.ReduceOp.SUM)
        elapsed_max = float(tmax.item())
        total_windows = float(wsum.item())
        # every rank's own time and window count, as the process group reports them (a scaling curve is only as good as its slowest rank)
        mine = torch.tensor([elapsed / max(1, args.steps) * 1e3, float(windows)], dtype=torch.float64, device=red_dev)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        rank_info = {"world_size_reported_by_process_group": dist.get_world_size(), "backend": dist.get_backend(),
                     "pass_ms_per_rank": [float(x[0].item()) for x in per_rank], "windows_per_rank": [int(x[1].item()) for x in per_rank]}
    else:
        elapsed_max, total_windows = elapsed, float(windows)
        rank_info = None

    if rank == 0:
        steps = max(1, args.steps)
        ms_per_step = elapsed_max / steps * 1e3
        value = total_windows * steps / elapsed_max
        kern = {"k1_pileup_bits": (k1 / steps, st.bytes_k1), "k2_window_replay": (k2s / steps, st.bytes_k2seq),
                "k2a_admission": (k2a / steps, st.bytes_k2a), "k2l_window_lanes": (k2l / steps, st.bytes_k2l),
                "k2w_window_rows": (k2w / steps, st.bytes_k2w),
                "k3_window_seq": (k3 / steps, st.bytes_k3), "k3b_haplotype_ids": (k3b / steps, st.bytes_k3b)}
        # Roofline units: kernels that run one after the other on the launch stream, except the window phase - k2l (two launches) and k2w are
        # launched side by side on separate streams, so their HIP-event intervals overlap (each spans most of the phase); the phase is
        # priced as ONE unit: the three launches' bytes over the phase's wall time (first start to last end, HIP events)
        k3_name = "k3_window_seq (sequences + records + SHA-1 ids: list A | flags: list B | + stop scan: list C | general walk: list D, concurrent)"
        kern[k3_name] = kern.pop("k3_window_seq")
        units = {"k1_pileup_bits": kern["k1_pileup_bits"] + (["k1_pileup_bits"],),
                 "k2a_admission": kern["k2a_admission"] + (["k2a_admission"],),
                 "k2_window_phase (k2l_window_lanes<6> | <8> | <16> | k2w_window_rows, concurrent)":
                     (k2win / steps, st.bytes_k2l + st.bytes_k2w, ["k2l_window_lanes", "k2w_window_rows"]),
                 k3_name: kern[k3_name] + (["k3_window_seq"],)}
        if kern["k3b_haplotype_ids"][0] > 0:   # (`normal` mode only: somatic ids are hashed inside K3)
            units["k3b_haplotype_ids"] = kern["k3b_haplotype_ids"] + (["k3b_haplotype_ids"],)
        else:
            del kern["k3b_haplotype_ids"]
        if k2win <= 0:   # no window-parallel work in this batch: the sequential replay is the K2 unit
            del units["k2_window_phase (k2l_window_lanes<6> | <8> | <16> | k2w_window_rows, concurrent)"]
            units["k2_window_replay"] = kern["k2_window_replay"] + (["k2_window_replay"],)
        dom = max(units, key=lambda k: units[k][0])
        dom_ms, dom_bytes, dom_kernels = units[dom]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM bytes per launch and instruction counts from the PMC counters: they need separate rocprofv3 --pmc passes of this very
        # command, so they come from the committed summary of those passes (tools/pmc_summary.py), only when it is for this workload
        traffic, traffic_src, traffic_note = None, None, None
        valu = {}
        pass_pmc = None
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_config%s.json" % args.config)), reverse=True):
            try:
                pt = json.load(open(path))
            except (OSError, ValueError):
                continue
            if args.transcripts or world != 1:
                break
            traffic_src = os.path.relpath(path, ROOT)
            pt = {k: e for k, e in pt.items() if not k.startswith("k0_")}   # (the layout kernels run once per batch, at upload: not part of a pass)
            pass_pmc = {"fetch_raw": sum(e.get("fetch_bytes_raw", 0.0) for e in pt.values()), "write": sum(e.get("write_bytes", 0.0) for e in pt.values())}
            hit = [k for k in pt if any(k.split("<")[0].startswith(x) for x in dom_kernels) and "fetch_bytes_raw" in pt[k]]   # (k2w_window_rows[_multi|_deep])
            if hit:
                traffic = sum(pt[k]["fetch_bytes_raw"] + pt[k].get("write_bytes", 0.0) for k in hit)
                traffic_note = ("FETCH_SIZE x 1024 + WRITE_SIZE x 1024 per launch, summed over %s; with the gfx950 correction for wide streaming reads "
                                "(FETCH_SIZE counts them at half, MI355X_MICROARCH.md HBM section) the read side is at most %.3g bytes"
                                % (", ".join(hit), sum(pt[k].get("fetch_bytes_x2", 0.0) for k in hit)))
            for k, e in pt.items():
                if "SQ_INSTS_VALU" in e:
                    valu[k.split("<")[0]] = valu.get(k.split("<")[0], 0.0) + e["SQ_INSTS_VALU"]
            break
        t_pass = elapsed_max / steps
        # Second byte accounting: SURVEY.md 8(d)'s COMPULSORY traffic of the whole pass, every byte counted once, with this run's counts -
        # per read 172 B in (16 B core + 51 B packed bases + 101 B qualities + 4 B cigar) and 2 x 32 B K1 -> K2 hand-over (+ 16 B per extra
        # mask word), 16 B per variant, 1 B per CDS nt (= nt-offset steps), 16 B per window, 16 B per distinct haplotype, 54 B per emitted
        # haplotype. The per-kernel figures above count each kernel's own inputs + outputs (planner-made structures and inter-kernel
        # hand-overs included, hand-overs twice); this one prices only what the problem itself has to move.
        n_emit = emitted if emitted is not None else int(st.n_ids)
        compulsory = (172 * st.n_reads + (64 + 32 * (st.mask_words - 1)) * st.n_reads + 16 * st.n_variants + st.n_steps
                      + 16 * st.n_windows_device + 16 * st.n_groups + 54 * n_emit)
        out = {
            "metric": "peptide-windows/s (somatic, 9-mer)", "value": value, "unit": "peptide-windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "ONE synthetic %d-transcript exome, %gx, SNV every %g nt, 101-nt reads (SURVEY 8d config %s, seed %d, per-gene random streams)%s; "
                                   "step = K1 + K2 (k2a admission, k2l one lane per window, k2w one wave per window for the widest / deepest ones; sequential k2 replay for the "
                                   "segments that need it) + K3 (window sequences, records and their SHA-1 ids) over the HBM-resident batch"
                                   % (n_tx, depth, spacing, args.config, seed,
                                      "" if world == 1 else ", its genes dealt to the %d GPUs by estimated cost (LPT on CDS nt x depth)" % world),
                       "windows_total": int(total_windows), "windows_rank0": int(windows), "reads_rank0": int(st.n_reads), "variants_rank0": int(st.n_variants),
                       "transcripts_rank0": int(st.n_transcripts), "window_len": 27,
                       "windows_counted_by": ("the consumer (main-ORF print_haplotypes calls of the reference, = the oracle's count)" if not args.no_consume else
                                              "the PLANNER (--no-consume: includes the speculative windows past a stop codon, ~13 % more than the reference makes - "
                                              "a kernel-timing run, its `value` is not the metric)"), "sharding": "genes (LPT on cost), no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": dom_ms, "rocprof_kernels": dom_kernels,
                         "compulsory_bytes": int(compulsory), "compulsory_frac": compulsory / t_pass / 1e9 / HBM_PEAK_GBS,
                         "compulsory_note": "whole pass, SURVEY 8(d) formula with this run's counts (%d reads, %d variants, %d steps, %d windows, %d haplotype groups, "
                                            "%d emitted) over ms_per_step, against the 8 TB/s peak" % (st.n_reads, st.n_variants, st.n_steps, st.n_windows_device, st.n_groups, n_emit),
                         "pmc_over_compulsory": None if not pass_pmc else {
                             "raw": (pass_pmc["fetch_raw"] + pass_pmc["write"]) / compulsory, "with_fetch_x2": (2 * pass_pmc["fetch_raw"] + pass_pmc["write"]) / compulsory,
                             "write_bytes_per_pass": pass_pmc["write"], "fetch_bytes_raw_per_pass": pass_pmc["fetch_raw"], "source": traffic_src,
                             "note": "FETCH_SIZE + WRITE_SIZE of every kernel of one pass (committed PMC summary) over the compulsory bytes; FETCH_SIZE raw and "
                                     "with the gfx950 x2 correction for wide streaming reads (MI355X_MICROARCH.md, HBM)"},
                         "limiter": "instruction issue / latency, not HBM (integer and bitset work: see DESIGN.md section 4 and profiles/)"},
            "roofline_units": {k: {"ms": v[0], "algorithmic_bytes": int(v[1]), "hbm_frac": (v[1] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS if v[0] > 0 else 0.0)}
                               for k, v in units.items()},
            "pass_hbm_frac": sum(v[1] for v in units.values()) / t_pass / 1e9 / HBM_PEAK_GBS,
            "valu_issue": None if not valu else {
                "wave_instructions_per_pass": int(sum(valu.values())), "source": traffic_src,
                "peak_per_s": VALU_PEAK_PER_S, "pass_frac": sum(valu.values()) / t_pass / VALU_PEAK_PER_S,
                "measured_int_peak_per_s": 661e9, "pass_frac_of_measured_int_peak": sum(valu.values()) / t_pass / 661e9,
                "measured_peak_source": "profiles/r05d_issue_rate.txt (tools/ubench/issue_rate.hip: integer add / xor / shift with every wave slot occupied)",
                "note": "SQ_INSTS_VALU of every kernel of one pass (committed PMC summary) over the live pass time, against 256 CUs x 4 SIMDs x "
                        "one wave64 VALU instruction per 2 cycles x 2.4 GHz: the pass is integer / bitset work bound by instruction issue and latency"},
            "kernels_ms": {k: v[0] for k, v in kern.items()},
            "kernels_note": "k2l_window_lanes (its three launches) and k2w_window_rows run side by side on separate streams after k2a (and k2_window_replay beside k2a): "
                            "their intervals overlap; k2_window_phase_ms is the wall time of that phase",
            "k2_window_phase_ms": k2win / steps,
            "kernels_algorithmic_bytes": {k: int(v[1]) for k, v in kern.items()},
            "kernels_hbm_frac": {k: (v[1] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS if v[0] > 0 else 0.0) for k, v in kern.items()},
            "rows_per_lane": int(st.rows_per_lane), "mask_words": int(st.mask_words),
            "replay": {"steps_window_parallel": int(st.n_steps_w), "steps_sequential": int(st.n_steps_seq), "admission_entries": int(st.n_adm),
                       "windows_lane_kernel": int(st.n_windows_lane), "windows_wave_kernel": int(st.n_windows_wave),
                       "groups": int(st.n_groups), "groups_k3": int(st.n_groups_k3), "groups_k3_list_a": int(st.n_groups_k3a), "groups_k3_list_c": int(st.n_groups_k3c), "groups_k3_list_d": int(st.n_groups_k3d), "ids_hashed": int(st.n_ids),
                       "emitted_haplotypes": emitted},
            "hbm_resident_bytes": int(st.hbm_bytes),
            "end_to_end": {"generate_s": t_gen, "plan_pack_h2d_s": t_plan, "pass_s": t_pass, "d2h_consume_s": None if args.no_consume else t_consume,
                           "windows_per_s": None if args.no_consume else windows / (t_plan + t_pass + t_consume),
                           "host_threads": int(os.environ.get("MP_THREADS", "0")) or min(32, n_cores),
                           "host_peak_rss_gb": round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1048576.0, 1),
                           "note": "rank 0, in memory: decoded inputs -> plan + pack + H2D -> one pass -> D2H + consumer (FASTA / TSV text); "
                                   "windows_per_s is this rank's windows over the sum of the three legs; not part of `value`"},
        }
        if world == 1 and not args.no_consume and args.e2e_chunks > 1:
            # the same in-memory end to end in gene chunks whose legs overlap (microphaser_amd/pipeline.py phase_chunked): chunk i + 1 is
            # planned and uploaded on a second context of the GPU while chunk i is phased, downloaded and consumed
            try:
                from microphaser_amd.pipeline import phase_chunked
                res.close()
                batch.close()
                ctx2 = m.Context(device_index)
                t0 = time.perf_counter()
                parts, w_chunked = phase_chunked(ds, n_chunks=args.e2e_chunks, contexts=[ctx, ctx2], window_len=27)
                t_chunked = time.perf_counter() - t0
                out["end_to_end"]["chunked"] = {"chunks": len(parts), "wall_s": t_chunked, "windows_per_s": w_chunked / t_chunked, "windows": int(w_chunked),
                                                "same_window_count_as_single_batch": bool(w_chunked == windows),
                                                "note": "decoded inputs -> text streams in host memory, the chunks' plan + H2D overlapped with the previous chunk's pass + D2H + "
                                                        "consumer (two contexts on the one GPU; byte-identical to the single batch: tests/test_gpu_parity.py). The legs are bound by "
                                                        "the host cores, which both chunks share: the overlap buys only the DMA and GPU time (4 chunks measured slower than "
                                                        "the single batch, 8 chunks 7 % faster)"}
                for r in parts:
                    r.close()
                ctx2.close()
            except Exception as e:   # noqa: BLE001 - a side measurement must not cost the bench line
                out["end_to_end"]["chunked"] = {"error": str(e)[:300]}
        if world > 1:
            out["partition"] = {"rank0_cost_share": my_cost / total_cost, "ideal_share": 1.0 / world}
            out["ranks"] = rank_info
        if rehearse:
            out["rehearsal"] = "all ranks on one GPU, gloo reductions: functional check of the N-rank path, not a measurement"
        if world == 1 and args.cpu_sample > 0:
            cb = cpu_baseline(args.config, args.cpu_sample)
            if cb:
                out["cpu_baseline"] = cb
                out["vs_cpu_baseline"] = value / cb["value"]
        if world == 1 and args.cpu_sample_all > 0 and n_cores > 1:
            cb = cpu_baseline(args.config, args.cpu_sample_all, threads=n_cores)
            if cb:
                out["cpu_baseline_all_cores"] = cb
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
