#!/usr/bin/env python3
"""Benchmark of the phasing hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU)

Metric (BASELINE.json): peptide-windows/s, `somatic`, 9-mer (27 nt), on the synthetic 20k-transcript
whole exome (30x, ~5 variant sites per window). A "step" is one pass of the
hot path (K1 pileup bits -> K2 window replay -> K3 window sequences) over the batch, with the packed
inputs already resident in HBM and the results left in HBM. Genes are independent units: every rank
phases its own 20k-transcript exome (seed 2020 + rank) with no data-path collective -> weak scaling.

The JSON line also carries
  roofline      - the dominant kernel's algorithmic bytes / its HIP-event time vs the 8 TB/s HBM peak
  cpu_baseline  - the CPU oracle (single thread, like the reference) on a bounded sample of the workload
  end_to_end    - plan+pack, H2D, D2H+consume wall times of this rank (host side, outside `value`)
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


def cpu_baseline(cfg_name, sample_transcripts):
    seed, _n, depth, spacing = CONFIGS[cfg_name]
    cli = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
    r = subprocess.run([cli, "synth", "--seed", str(seed), "--transcripts", str(sample_transcripts), "--depth", str(depth),
                        "--spacing", str(spacing)], capture_output=True, text=True)
    if r.returncode != 0:
        return None
    st = json.loads(r.stdout)
    return {
        "value": st["windows"] / st["phase_seconds"], "unit": "peptide-windows/s", "cores": 1, "kind": "port",
        "sample": "%d-transcript exome from the same generator and parameters (seed %d, %gx, spacing %g nt): %d windows in %.2f s; "
                  "phasing only, inputs in memory; the Rust reference cannot be built here, the CPU oracle (C++ -O2, "
                  "single thread like the reference) stands in" % (sample_transcripts, seed, depth, spacing, st["windows"], st["phase_seconds"]),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C", choices=sorted(CONFIGS))
    ap.add_argument("--transcripts", type=int, default=0, help="override the transcript count (debugging)")
    ap.add_argument("--cpu-sample", type=int, default=1000, help="transcripts in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-consume", action="store_true", help="skip the end-to-end D2H + consumer leg")
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
    ctx = m.Context(device_index)
    t0 = time.perf_counter()
    ds = ctx.synth(seed + rank, n_tx, depth, spacing)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    batch = ds.batch(window_len=27)   # plan + pack + H2D: inputs are resident in HBM from here on
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
    k1 = k2s = k2a = k2l = k2w = k3 = k3b = 0.0
    for _ in range(args.steps):
        st = batch.run()          # returns after the launch stream has drained (results in HBM)
        k1 += st.k1_ms
        k2s += st.k2seq_ms
        k2a += st.k2a_ms
        k2l += st.k2l_ms
        k2w += st.k2w_ms
        k3 += st.k3_ms
        k3b += st.k3b_ms
    sync_all()
    elapsed = time.perf_counter() - t0

    # the unit count: main-ORF print_haplotypes calls the reference makes = what the consumer actually walks
    t0 = time.perf_counter()
    windows = None
    if not args.no_consume:
        res = batch.results()
        windows = res.windows
    t_consume = time.perf_counter() - t0
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
    else:
        elapsed_max, total_windows = elapsed, float(windows)

    if rank == 0:
        steps = max(1, args.steps)
        ms_per_step = elapsed_max / steps * 1e3
        value = total_windows * steps / elapsed_max
        kern = {"k1_pileup_bits": (k1 / steps, st.bytes_k1), "k2_window_replay": (k2s / steps, st.bytes_k2seq),
                "k2a_admission": (k2a / steps, st.bytes_k2a), "k2l_window_lanes": (k2l / steps, st.bytes_k2l),
                "k2w_window_rows": (k2w / steps, st.bytes_k2w),
                "k3_window_seq": (k3 / steps, st.bytes_k3), "k3b_haplotype_ids": (k3b / steps, st.bytes_k3b)}
        dom = max(kern, key=lambda k: kern[k][0])
        dom_ms, dom_bytes = kern[dom]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM bytes per launch from the PMC counters: they need separate rocprofv3 --pmc passes of this very command, so
        # they come from the committed summary of those passes (tools/pmc_traffic.py), only when it is for this workload
        traffic, traffic_src = None, None
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
            try:
                pt = json.load(open(path))
            except (OSError, ValueError):
                continue
            if pt.get("config") == args.config and not args.transcripts and dom in pt.get("kernels", {}):
                traffic, traffic_src = pt["kernels"][dom]["hbm_bytes"], os.path.relpath(path, ROOT)
                break
        out = {
            "metric": "peptide-windows/s (somatic, 9-mer)", "value": value, "unit": "peptide-windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "synthetic %d-transcript exome per GPU, %gx, SNV every %g nt, 101-nt reads (SURVEY 8d config %s, seed %d+rank); "
                                   "step = K1 + K2 (k2a admission, k2l one lane per window, k2w one wave per window for the wide ones; sequential k2 replay for the segments that need it) + K3 + K3b over the HBM-resident batch" % (n_tx, depth, spacing, args.config, seed),
                       "windows_per_gpu": int(windows), "reads_per_gpu": int(st.n_reads), "variants_per_gpu": int(st.n_variants),
                       "transcripts_per_gpu": int(st.n_transcripts), "window_len": 27, "sharding": "genes, no collective"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": dom_ms},
            "kernels_ms": {k: v[0] for k, v in kern.items()},
            "kernels_algorithmic_bytes": {k: int(v[1]) for k, v in kern.items()},
            "rows_per_lane": int(st.rows_per_lane), "mask_words": int(st.mask_words),
            "replay": {"steps_window_parallel": int(st.n_steps_w), "steps_sequential": int(st.n_steps_seq), "admission_entries": int(st.n_adm),
                       "windows_lane_kernel": int(st.n_windows_lane), "windows_wave_kernel": int(st.n_windows_wave)},
            "hbm_resident_bytes": int(st.hbm_bytes),
            "end_to_end": {"generate_s": t_gen, "plan_pack_h2d_s": t_plan, "d2h_consume_s": None if args.no_consume else t_consume,
                           "host_peak_rss_gb": round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1048576.0, 1),
                           "note": "host legs of rank 0 (planner and consumer shard genes over host threads); not part of `value`"},
        }
        if rehearse:
            out["rehearsal"] = "all ranks on one GPU, gloo reductions: functional check of the N-rank path, not a measurement"
        if world == 1 and args.cpu_sample > 0:
            cb = cpu_baseline(args.config, args.cpu_sample)
            if cb:
                out["cpu_baseline"] = cb
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
