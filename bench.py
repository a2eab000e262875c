#!/usr/bin/env python3
"""bench.py -- genome bases p-scored per second on synthetic hg38 (BASELINE.json configs[1]):
50 M paired fragments, treatment only, default ChIP-seq mode (-p 0.01).

One "step" = one pass of the whole hot path over the whole genome: fragment events already
resident in HBM -> tile-bucketed endpoint records -> LDS difference arrays / prefix sums ->
run-length pileup -> lambda -> log-normal -log10 p -> peak sweep -> peak list on the host.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frags F] [--qval]
  N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
         chromosomes are sharded over ranks (LPT); the only exchanges are the fragLen
         fixed-point sums (and the p-value table with --qval) -> "scaling": "strong".

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from genrich_amd import synth  # noqa: E402
from genrich_amd.dist import Collectives, lpt_partition  # noqa: E402
from genrich_amd.lib import GxParams, Genrich, minus_log10f  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(ev_all, lens, n_chrom_sample, qval):
    """Single-threaded CPU restatement (oracle/, kind "port") on the first chromosomes only:
    events in memory -> peaks, the same span the GPU step covers."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import backends as B

    sel = ev_all[ev_all["chrom"] < n_chrom_sample]
    sub_lens = lens[:n_chrom_sample]
    o = B.Oracle(B.make_params(pq=0.05 if qval else 0.01, qval=qval))
    o.set_chroms(sub_lens)
    t0 = time.perf_counter()
    o.sample_begin(0, None)
    o.push_events(sel)
    o.sample_end()
    o.sample_no_control()
    o.pvalues()
    o.find_peaks()
    dt = time.perf_counter() - t0
    bases = float(sum(sub_lens))
    o.close()
    return dict(value=bases / dt / 1e9, unit="Gbases/s", cores=1, kind="port",
                sample=f"hg38 chr1-chr{n_chrom_sample} ({bases/1e6:.0f} Mbp, {len(sel)} fragments), "
                       f"events in memory -> peaks, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frags", type=int, default=50_000_000)
    ap.add_argument("--qval", action="store_true", help="-q 0.05 instead of -p 0.01")
    ap.add_argument("--control", action="store_true",
                    help="add a 50M-fragment uniform control (configs[2] shape; not the headline metric)")
    ap.add_argument("--lean", action="store_true",
                    help="do not materialise the pileup floats of the intervals (gx_set_keep_pileups(0))")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-chroms", type=int, default=12)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0 and world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    # GX_BENCH_BACKEND=gloo is a validation mode for a box with fewer GPUs than ranks: ranks share the
    # devices round-robin and the (tiny) collectives go through host memory; it is labelled in `config`
    backend = os.environ.get("GX_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local >= ndev:
        sys.exit(f"rank {rank}: no GPU {local} on this node ({ndev} visible)")
    local_dev = local % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where collective payloads live
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    lens = synth.HG38_LENS
    G = int(sum(lens))
    ev_all = synth.make_fragments(lens, args.frags, seed=1)
    owner = lpt_partition(lens, world)
    owned = np.array([o == rank for o in owner], dtype=np.uint8)
    mine = ev_all[owned[ev_all["chrom"]].astype(bool)] if world > 1 else ev_all
    d_ev = torch.from_numpy(mine.view(np.uint32).reshape(-1, 4).copy()).to(dev)
    d_ct = None
    if args.control:
        ct_all = synth.make_fragments(lens, args.frags, seed=2, uniform_only=True)
        ct = ct_all[owned[ct_all["chrom"]].astype(bool)] if world > 1 else ct_all
        d_ct = torch.from_numpy(ct.view(np.uint32).reshape(-1, 4).copy()).to(dev)
    torch.cuda.synchronize()

    params = GxParams(minus_log10f(0.05 if args.qval else 0.01), int(args.qval), 200.0, 0, 100, local_dev, 0)
    gx = Genrich(params)
    gx.set_chroms(lens)
    # By default the whole interval table (end, treatment pileup, p) is materialised, as the reference holds it.
    # --lean drops the pileup floats, which only the -f / -k emitters read (what the command-line host does
    # when neither option is given): ~3 % faster, reported as such in `config`.
    gx.set_keep_pileups(not args.lean)
    if world > 1:
        coll = Collectives(device=cdev)
        gx.set_owned(owned)
        gx.set_collectives(rank, world, coll.allreduce_i64, coll.allgather_tab)

    def step():
        gx.reset()
        gx.sample_begin(0, None)
        gx.push_events_device(d_ev.data_ptr(), d_ev.shape[0])
        gx.sample_end()
        if d_ct is not None:
            gx.sample_begin(1, None)
            gx.push_events_device(d_ct.data_ptr(), d_ct.shape[0])
            gx.sample_end()
        else:
            gx.sample_no_control()
        gx.pvalues()
        return gx.find_peaks()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    phase_acc = {}
    for _ in range(args.steps):
        res = step()
        for name, ms in gx.phase_times():
            phase_acc.setdefault(name, []).append(ms)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        npk = torch.tensor([res[0]], dtype=torch.int64, device=cdev)
        dist.all_reduce(npk)
        n_peaks = int(npk.item())
    else:
        n_peaks = res[0]

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        phases = {k: float(np.mean(v)) for k, v in phase_acc.items()}
        # dominant kernel: the tile kernel (LDS difference array + prefix sum + RLE emit).
        # Algorithmic bytes of the stage it implements, SURVEY.md 8(d): clear + scan of the dense
        # per-base array (8 B/base), its endpoint records (16 B/event) and the RLE it writes
        # (8 B/interval).  Rank 0's share of the genome when sharded.
        g0 = float(sum(l for l, o in zip(lens, owner) if o == 0))
        e0 = float(d_ev.shape[0])
        iv0 = float(gx.interval_total()) if hasattr(gx, "interval_total") else 0.0
        alg_bytes = 8.0 * g0 + 16.0 * e0 + 8.0 * iv0
        t_tile = phases.get("t.tile", 0.0) * 1e-3
        achieved = alg_bytes / t_tile / 1e9 if t_tile > 0 else 0.0
        # HBM bytes actually moved by one k_tile launch: PMC counters from a separate rocprofv3 run
        # (profiles/r01_traffic.json says how they were collected and corrected); single-GPU workload only
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if world == 1 and args.frags == 50_000_000 and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("k_tile", {}).get("hbm_bytes_per_launch")
        out = {
            "metric": "genome bases p-scored/sec, hg38 50M frags",
            "value": G / (dt / args.steps) / 1e9,
            "unit": "Gbases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32 pileup (1/120 units) + f64 p-values",
            "data": "synthetic",
            "config": {
                "workload": f"hg38 25 contigs ({G} bp), {args.frags} paired fragments, treatment only, "
                            + ("-q 0.05" if args.qval else "-p 0.01")
                            + (" + 50M-fragment control (configs[2] shape)" if args.control else " (BASELINE.json configs[1])"),
                "parallelism": f"chromosome-sharded x{world}"
                               + ("" if backend == "nccl" or world == 1 else f" ({backend} validation mode, {ndev} GPU(s))"),
                "peaks": n_peaks,
                "pileup_floats_kept": not args.lean,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_tile",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes": alg_bytes,
                "launch_ms": t_tile * 1e3,
                "traffic_gbs": (traffic / t_tile / 1e9) if (traffic and t_tile > 0) else None,
                "traffic_frac_of_peak": (traffic / t_tile / 1e9 / HBM_PEAK_GBS) if (traffic and t_tile > 0) else None,
                "note": "algorithmic bytes = 8 B/base (clear + scan of the dense difference array) + 16 B/event "
                        "+ 8 B/interval, SURVEY 8(d); k_tile keeps that array in LDS, so its real HBM traffic "
                        "(`traffic`, PMC; `traffic_gbs` = traffic / launch time, `traffic_frac_of_peak` = that over 8 TB/s) "
                        "is ~25x smaller and `frac` can exceed 1: the kernel is VALU-issue / LDS-latency bound, not HBM-bound",
            },
            "phases_ms": phases,
            "whole_path_hbm_frac": ((8.0 * G + 16.0 * 2 * args.frags + 52.0 * iv0) / (dt / args.steps) / 1e9)
            / HBM_PEAK_GBS if world == 1 else None,
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(ev_all, lens, args.cpu_chroms, args.qval)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
