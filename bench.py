#!/usr/bin/env python3
"""bench.py -- genome bases p-scored per second on synthetic hg38 (BASELINE.json).

One "step" = one pass of the whole hot path over the whole genome: fragment events already resident
in HBM -> tile-bucketed endpoint records -> LDS difference arrays / prefix sums -> run-length pileup
-> lambda (-> control scaling) -> log-normal -log10 p (-> Fisher over replicates) (-> BH q) -> peak
sweep -> peak list on the host.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5]
    --config 2  hg38, 50 M paired fragments, treatment only, -p 0.01          (configs[1], the headline; default)
    --config 3  + a 50 M-fragment uniform control, -q 0.05                    (configs[2])
    --config 4  ATAC -j -d 100 cut-site intervals + -s multimapping weights   (configs[3])
    --config 5  3 replicates, Fisher-combined p, global q (-q 0.05)           (configs[4])
  N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
         chromosomes are sharded over ranks (LPT); the only exchanges are the fragLen fixed-point sums
         and (with -q) the p-value table, done by the library itself with RCCL on device buffers.
         The genome is fixed, per-GPU work shrinks with N -> "scaling": "strong".

Prints ONE JSON line (rank 0).  Besides the contract's fields:
  gate          the same workload (or its first chromosomes, see cpu_baseline.sample) through the CPU oracle
                and through the HIP path: narrowpeak_diff (differing narrowPeak lines; must be 0), max_abs_dp /
                max_abs_dq over every interval
  roofline      bound "hbm" for the dominant kernel: achieved = HBM bytes it moves (PMC counters of a separate
                rocprofv3 run of this very build, profiles/) / its mean duration measured here (HIP events);
                frac = achieved / 8 TB/s <= 1 by construction; `algorithmic_bytes` = what the sparse formulation
                must move; `whole_step` the same for the whole path; `issue` = the SQ-counter issue model
  h2d / e2e     PCIe upload of the events from pinned memory, and a step that starts from pinned host memory
  cpu_baseline  the oracle (kind "port"), one core, events in memory -> peaks, on the sample named
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from genrich_amd import synth  # noqa: E402
from genrich_amd.dist import Collectives, lpt_partition  # noqa: E402
from genrich_amd.lib import GxParams, Genrich, minus_log10f, rccl_unique_id  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec

CONFIGS = {
    2: dict(name="configs[1]", desc="treatment only, -p 0.01", qval=False, control=False, atac=False, multimap=False, reps=1,
            gate_chroms=25),
    3: dict(name="configs[2]", desc="treatment + 50M-fragment uniform control, -q 0.05", qval=True, control=True, atac=False,
            multimap=False, reps=1, gate_chroms=8),
    4: dict(name="configs[3]", desc="ATAC -j -d 100 cut-site intervals, 10 % of fragments multimapped (-s weights 1/k), -p 0.01",
            qval=False, control=False, atac=True, multimap=True, reps=1, gate_chroms=12),
    5: dict(name="configs[4]", desc="3 replicates, Fisher-combined p, global -q 0.05", qval=True, control=False, atac=False,
            multimap=False, reps=3, gate_chroms=6),
}


def source_hash():
    """Identifies the build the profile files under profiles/ belong to."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "genrich_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".cpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def build_workload(cfg, frags, lens):
    """[(treatment events, control events or None)] per replicate, SURVEY.md 8(d)."""
    reps = []
    seeds = [1, 3, 5, 7, 9][:cfg["reps"]]
    for sd in seeds:
        # with a control the treatment needs towers whose q survives the genome-wide correction
        tv = synth.make_fragments(lens, frags, seed=sd, tower_every=50_000_000 if cfg["control"] else 5_000_000)
        if cfg["multimap"]:
            tv = synth.add_multimap(tv, lens, 0.10, seed=sd + 10)
        if cfg["atac"]:
            tv = synth.atac_events(tv, lens, d=100)
        cv = synth.make_fragments(lens, frags, seed=sd + 1, uniform_only=True) if cfg["control"] else None
        reps.append((tv, cv))
    return reps


def subset(reps, n_chrom):
    out = []
    for tv, cv in reps:
        out.append((tv[tv["chrom"] < n_chrom], None if cv is None else cv[cv["chrom"] < n_chrom]))
    return out


def run_backend(be, lens, reps, peaks_to=None):
    be.set_chroms(lens)
    t0 = time.perf_counter()
    for tv, cv in reps:
        be.sample_begin(0, None)
        be.push_events(tv)
        be.sample_end()
        if cv is not None:
            be.sample_begin(1, None)
            be.push_events(cv)
            be.sample_end()
        else:
            be.sample_no_control()
        be.pvalues()
    if peaks_to:
        be.find_peaks_to(peaks_to[0], None, peaks_to[1])  # the oracle's own narrowPeak emitter
    else:
        be.find_peaks()
    return time.perf_counter() - t0


def gate_and_cpu_baseline(cfg, lens, reps, n_chrom, qval, device):
    """The oracle (CPU restatement, one core) and the HIP path on the same events: narrowPeak text diff,
    max |dp| / |dq| over all intervals, and the oracle's rate as the CPU baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import backends as B

    sub_lens = lens[:n_chrom]
    sub = subset(reps, n_chrom)
    names = synth.HG38_NAMES[:n_chrom]
    par = B.make_params(pq=0.05 if qval else 0.01, qval=qval)
    td = tempfile.mkdtemp()
    po, ph = os.path.join(td, "o.narrowPeak"), os.path.join(td, "h.narrowPeak")
    o = B.Oracle(par)
    dt = run_backend(o, sub_lens, sub, peaks_to=(po, names))
    par.device = device
    h = Genrich(par)
    run_backend(h, sub_lens, sub)
    h.write_narrowpeak(names, ph)
    want, got = open(po, "rb").read(), open(ph, "rb").read()
    os.remove(po)
    os.remove(ph)
    os.rmdir(td)
    if got == want:
        ndiff = 0
    else:
        a, b = set(got.split(b"\n")), set(want.split(b"\n"))
        ndiff = max(1, len(a ^ b))
    dp = dq = 0.0
    nbits = 0
    n_iv = 0
    ends_equal = True
    for c in range(n_chrom):
        eo, co = o.get_intervals(-1, c)
        eh, ch = h.get_intervals(-1, c, piles=False)
        if len(eo) != len(eh) or not np.array_equal(eo, eh):
            ends_equal = False
            continue
        n_iv += len(eo)
        for k in ("p",) + (("q",) if qval else ()):
            a, b = co[k].astype(np.float64), ch[k].astype(np.float64)
            fin = np.abs(a) < 1e30
            if fin.any():
                d = float(np.max(np.abs(a[fin] - b[fin])))
                if k == "p":
                    dp = max(dp, d)
                else:
                    dq = max(dq, d)
            nbits += int((co[k].view(np.uint32) != ch[k].view(np.uint32)).sum())
    bases = float(sum(sub_lens)) * len(sub)
    n_ev = int(sum(len(t) + (0 if c is None else len(c)) for t, c in sub))
    gate = dict(narrowpeak_diff=ndiff, peaks_oracle=int(o.n_peaks), peaks_hip=int(h.n_peaks), interval_ends_equal=ends_equal,
                intervals_compared=n_iv, max_abs_dp=dp, max_abs_dq=dq if qval else None, pq_values_differing_in_bits=nbits,
                passed=bool(ndiff == 0 and ends_equal and dp <= 1e-5 and dq <= 1e-5))
    what = "the whole workload" if n_chrom == len(lens) else f"hg38 chr1-chr{n_chrom} of the workload"
    cpu = dict(value=bases / dt / 1e9, unit="Gbases/s", cores=1, kind="port",
               sample=f"{what} ({sum(sub_lens)/1e6:.0f} Mbp x {len(sub)} replicate(s), {n_ev} events), events in memory -> "
                      f"peaks, {dt:.1f} s")
    o.close()
    return gate, cpu


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frags", type=int, default=50_000_000)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--qval", action="store_true", help="-q 0.05 instead of -p 0.01 (on top of --config)")
    ap.add_argument("--control", action="store_true", help="add a uniform control (on top of --config)")
    ap.add_argument("--lean", action="store_true",
                    help="do not materialise the pileup floats of the intervals (gx_set_keep_pileups(0))")
    ap.add_argument("--no-cpu", action="store_true", help="skip the gate + cpu_baseline leg")
    ap.add_argument("--cpu-chroms", type=int, default=0, help="gate / cpu_baseline on the first K chromosomes (0: per config)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the H2D / end-to-end-from-pinned figures")
    args = ap.parse_args()
    # The JSON line must be the only thing on stdout: libraries below Python (RCCL prints a version banner
    # through C stdio, flushed at exit) write to file descriptor 1, so the real stdout is put aside and
    # descriptor 1 joins stderr until the line is written.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    cfg = dict(CONFIGS[args.config])
    cfg["qval"] = cfg["qval"] or args.qval
    cfg["control"] = cfg["control"] or args.control

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
    # devices round-robin and the (tiny) collectives go through host callbacks; it is labelled in `config`
    backend = os.environ.get("GX_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local >= ndev:
        sys.exit(f"rank {rank}: no GPU {local} on this node ({ndev} visible)")
    local_dev = local % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where torch's own collective payloads live
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    lens = synth.HG38_LENS
    G = int(sum(lens))
    reps_all = build_workload(cfg, args.frags, lens)
    owner = lpt_partition(lens, world)
    owned = np.array([o == rank for o in owner], dtype=np.uint8)

    def mine(ev):
        if ev is None:
            return None
        return ev[owned[ev["chrom"]].astype(bool)] if world > 1 else ev

    def to_dev(ev):
        return None if ev is None else torch.from_numpy(ev.view(np.uint32).reshape(-1, 4).copy()).to(dev)

    d_reps = [(to_dev(mine(tv)), to_dev(mine(cv))) for tv, cv in reps_all]
    torch.cuda.synchronize()

    params = GxParams(minus_log10f(0.05 if cfg["qval"] else 0.01), int(cfg["qval"]), 200.0, 0, 100, local_dev, 0)
    gx = Genrich(params)
    gx.set_chroms(lens)
    # By default the whole interval table (end, treatment pileup, p) is materialised, as the reference holds it.
    # --lean drops the pileup floats, which only the -f / -k emitters read: reported as such in `config`.
    gx.set_keep_pileups(not args.lean)
    coll_kind = "none"
    force_rccl = world == 1 and os.environ.get("GX_BENCH_FORCE_RCCL") == "1"   # exercise the RCCL path with one rank
    if force_rccl:
        os.environ["GX_FORCE_COLL"] = "1"
        gx.set_rccl(0, 1, rccl_unique_id())
        coll_kind = "RCCL inside the library, one-rank communicator (exercise mode)"
    if world > 1:
        gx.set_owned(owned)
        if backend == "nccl":
            # the library's own RCCL communicator; torch.distributed only carries the 128-byte id (outside the timed region)
            ok = 1
            try:
                box = [rccl_unique_id() if rank == 0 else None]
            except Exception as e:  # noqa: BLE001
                box, ok = [None], 0
                print(f"rank {rank}: {e}", file=sys.stderr)
            dist.broadcast_object_list(box, src=0)
            if box[0] is None:
                ok = 0
            else:
                try:
                    gx.set_rccl(rank, world, box[0])
                except Exception as e:  # noqa: BLE001
                    ok = 0
                    print(f"rank {rank}: {e}", file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                coll_kind = "RCCL inside the library (device buffers, library stream)"
            else:
                # some rank could not open the library's communicator: all ranks take the callback route together
                coll = Collectives(device=cdev)
                gx.set_collectives(rank, world, coll.allreduce_i64, coll.allgather_tab)
                coll_kind = "host callbacks over torch.distributed/nccl (the library's own communicator failed)"
        else:
            coll = Collectives(device=cdev)
            gx.set_collectives(rank, world, coll.allreduce_i64, coll.allgather_tab)
            coll_kind = f"host callbacks over torch.distributed/{backend} (validation mode)"

    def step(dreps=d_reps):
        gx.reset()
        for d_tv, d_cv in dreps:
            gx.sample_begin(0, None)
            gx.push_events_device(d_tv.data_ptr(), d_tv.shape[0])
            gx.sample_end()
            if d_cv is not None:
                gx.sample_begin(1, None)
                gx.push_events_device(d_cv.data_ptr(), d_cv.shape[0])
                gx.sample_end()
            else:
                gx.sample_no_control()
            gx.pvalues()
        return gx.find_peaks()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def collect(acc):
        seen = {}
        for name, ms in gx.phase_times():
            seen[name] = seen.get(name, 0.0) + ms
        for name, ms in seen.items():
            acc.setdefault(name, []).append(ms)

    # Inside the timed region only the tile stage is bracketed by HIP events (the roofline's live duration): an event
    # record costs the stream a ~5 us bubble, so the other phases are timed in two extra, untimed steps afterwards.
    gx.set_phase_timing(1)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    phase_acc = {}
    for _ in range(args.steps):
        res = step()
        collect(phase_acc)
    barrier()
    dt = time.perf_counter() - t0
    gx.set_phase_timing(2)
    all_phases = {}
    for _ in range(2):
        step()
        collect(all_phases)
    gx.set_phase_timing(1)
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
        step_s = dt / args.steps
        phases = {k: float(np.mean(v)) for k, v in all_phases.items()}       # untimed steps, every phase
        phases.update({k: float(np.mean(v)) for k, v in phase_acc.items()})  # the tile stage: live, timed region
        n_rep = len(reps_all)
        iv0 = float(gx.interval_total())
        ev_n = float(sum(d_tv.shape[0] + (0 if d_cv is None else d_cv.shape[0]) for d_tv, d_cv in d_reps))
        # ---- roofline of the dominant kernel (k_tile: LDS difference slices -> run-length pileup) -------------
        # compulsory HBM bytes of the sparse formulation, per launch: 2 B per endpoint record in (two per event),
        # a 48 B descriptor per tile, 8 B per interval out + 8 B of per-tile counts
        n_tiles = sum((l + 4095) // 4096 for l, o in zip(lens, owner) if o == 0)
        launches = n_rep * (2 if cfg["control"] else 1)   # k_tile runs once per sample (its narrow + wide launch pair)
        # (the library's phase timers cover the last replicate of a step and gx_find_peaks)
        tile_ms = (phases.get("t.tile", 0.0) + phases.get("c.tile", 0.0)) / (2 if cfg["control"] else 1)
        prof = None
        ppath = os.path.join(ROOT, "profiles", f"r02_counters_config{args.config}.json")
        if os.path.exists(ppath):
            prof = json.load(open(ppath))
            if prof.get("source_hash") != source_hash() or world != 1 or args.frags != 50_000_000 or args.qval or args.control:
                prof = None  # counters of another build / another workload are not this run's
        # (the tile stage = k_tile_fast, one launch per sample; k_tile only exists in -E runs)
        traffic = (sum(prof["kernels"].get(k, {}).get("hbm_bytes_per_step", 0.0) for k in ("k_tile_fast", "k_tile")) / launches
                   if prof else None)
        alg_tile = (2.0 * 2.0 * ev_n + 8.0 * (iv0 if launches == 1 else 2.0 * ev_n)) / launches + 56.0 * n_tiles
        used = traffic if traffic else alg_tile
        achieved = used / (tile_ms * 1e-3) / 1e9 if tile_ms > 0 else 0.0
        # whole step: events in + final interval table (end, p[, pileup]) + sweep masks out
        alg_step = 16.0 * ev_n + (12.0 if not args.lean else 8.0) * iv0 + 3.0 * iv0 / 8.0
        roof = {
            "bound": "hbm", "kernel": "k_tile_fast", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "launch_ms": tile_ms,
            "algorithmic_bytes": alg_tile,
            "traffic_over_algorithmic": (traffic / alg_tile) if traffic else None,
            "whole_step": {
                "ms": step_s * 1e3,
                "algorithmic_bytes": alg_step,
                "traffic": prof["whole_step"]["hbm_bytes_per_step"] if prof else None,
                "frac_of_peak": ((prof["whole_step"]["hbm_bytes_per_step"] if prof else alg_step) / step_s / 1e9) / HBM_PEAK_GBS,
                "traffic_over_algorithmic": (prof["whole_step"]["hbm_bytes_per_step"] / alg_step) if prof else None,
            },
            "issue": prof.get("issue") if prof else None,
            "dense_model": {"bytes": 8.0 * G + 16.0 * ev_n + 52.0 * iv0,
                            "note": "SURVEY 8(d)'s dense int32-array model; the array lives in LDS here, so this is not HBM traffic"},
            "note": "achieved = HBM bytes the tile stage moves per sample (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE of this build, "
                    "profiles/r02_counters_config*.json; the sparse formulation's compulsory bytes when no counter file matches "
                    "this build) / its mean duration (HIP events on the library's stream); frac <= 1 by construction",
        }
        out = {
            "metric": "genome bases p-scored/sec, hg38 50M frags",
            "value": n_rep * G / step_s / 1e9,
            "unit": "Gbases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_s * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32 pileup (1/120 units) + f64 p-values",
            "data": "synthetic",
            "config": {
                "workload": f"hg38 25 contigs ({G} bp), {args.frags} paired fragments x {n_rep} replicate(s), {cfg['desc']} "
                            f"(BASELINE.json {cfg['name']})",
                "parallelism": f"chromosome-sharded x{world}"
                               + ("" if backend == "nccl" or world == 1 else f" ({backend} validation mode, {ndev} GPU(s))"),
                "collectives": coll_kind,
                "peaks": n_peaks,
                "intervals": int(iv0),
                "events_per_step": int(ev_n),
                "pileup_floats_kept": not args.lean,
                "source_hash": source_hash(),
            },
            "roofline": roof,
            "phases_ms": phases,
            "phases_note": "t.tile / c.tile: HIP events inside the timed region; the other phases: two extra untimed steps "
                           "(an event record costs the stream ~5 us, so the timed steps carry only the tile stage's pair)",
        }
        if world == 1 and not args.no_e2e:
            # PCIe upload of the events from pinned host memory, and a step that starts there (gx_push_events)
            pin = [(torch.from_numpy(tv.view(np.uint32).reshape(-1, 4)).pin_memory(),
                    None if cv is None else torch.from_numpy(cv.view(np.uint32).reshape(-1, 4)).pin_memory()) for tv, cv in reps_all]
            nbytes = sum(t.numel() * 4 + (0 if c is None else c.numel() * 4) for t, c in pin)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(3):
                e0.record()
                for (t, c), (dtv, dcv) in zip(pin, d_reps):
                    dtv.copy_(t, non_blocking=True)
                    if c is not None:
                        dcv.copy_(c, non_blocking=True)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            h2d_ms = float(np.median(ts))

            def step_from_host():
                gx.reset()
                for t, c in pin:
                    gx.sample_begin(0, None)
                    gx.push_events_ptr(t.data_ptr(), t.shape[0], pinned=True)
                    gx.sample_end()
                    if c is not None:
                        gx.sample_begin(1, None)
                        gx.push_events_ptr(c.data_ptr(), c.shape[0], pinned=True)
                        gx.sample_end()
                    else:
                        gx.sample_no_control()
                    gx.pvalues()
                return gx.find_peaks()

            step_from_host()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                step_from_host()
            torch.cuda.synchronize()
            e2e_ms = (time.perf_counter() - t1) / 3 * 1e3
            out["h2d"] = {"ms": h2d_ms, "bytes": nbytes, "gbs": nbytes / h2d_ms / 1e6,
                          "note": "events from pinned host memory to HBM (outside the timed region of `value`)"}
            out["e2e_from_pinned"] = {"ms_per_step": e2e_ms, "value": n_rep * G / (e2e_ms * 1e-3) / 1e9, "unit": "Gbases/s",
                                      "note": "gx_push_events_pinned: 64 MiB pieces uploaded on a side stream, the first kernel "
                                              "(k_sort1) starts on the pieces that have arrived -> peak list on the host"}
        if not args.no_cpu and world == 1:
            k = args.cpu_chroms or cfg["gate_chroms"]
            gate, cpu = gate_and_cpu_baseline(cfg, lens, reps_all, min(k, len(lens)), cfg["qval"], local_dev)
            out["gate"] = gate
            out["cpu_baseline"] = cpu
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
