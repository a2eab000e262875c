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
  roofline      bound "hbm" for the dominant kernel: achieved = its ALGORITHMIC bytes per launch / its mean duration
                measured here (HIP events); frac = achieved / 8 TB/s; `traffic` = the HBM bytes it moves (PMC counters of
                a separate rocprofv3 run of this very build, profiles/), `traffic_frac` the same fraction on those;
                `whole_step` both for the whole path; `issue` = the second bound (VALU issue) from the SQ counters
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
from genrich_amd.lib import GxParams, Genrich, minus_log10f, pack_events, rccl_unique_id  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec

CONFIGS = {
    2: dict(name="configs[1]", desc="treatment only, -p 0.01", qval=False, control=False, atac=False, multimap=False, reps=1,
            gate_chroms=25),
    3: dict(name="configs[2]", desc="treatment (peaks every 200 kb) + 50M-fragment uniform control, -q 0.05", qval=True, control=True, atac=False,
            multimap=False, reps=1, gate_chroms=8),
    4: dict(name="configs[3]", desc="ATAC -j -d 100 cut-site intervals, 10 % of fragments multimapped (-s weights 1/k), -p 0.01",
            qval=False, control=False, atac=True, multimap=True, reps=1, gate_chroms=12),
    5: dict(name="configs[4]", desc="3 replicates, Fisher-combined p, global -q 0.05", qval=True, control=False, atac=False,
            multimap=False, reps=3, gate_chroms=6),
    # the headline stream the way the reference's README runs its flagship command (README.md:463-465): -e chrM,chrY -E <N-gap
    # regions> (synth.excluded_regions: 777 merged regions, 92 Mbp, one touching position 0, one reaching a chromosome's end)
    "2E": dict(name="configs[1] with -e chrY,chrM -E <777 regions>", desc="treatment only, -p 0.01, -e chrY,chrM, -E 777 excluded regions (92 Mbp)",
               qval=False, control=False, atac=False, multimap=False, reps=1, gate_chroms=8, excl=True),
}
for _c in CONFIGS.values():
    _c.setdefault("excl", False)


def exclusions(cfg, lens):
    """(skip, beds) of a config: what gx_set_chroms takes besides the lengths."""
    if not cfg["excl"]:
        return None, None
    skip = [False] * len(lens)
    skip[23] = skip[24] = True   # chrY, chrM
    return skip, synth.excluded_regions(lens, n=800, seed=7, skip=(23, 24))


def source_hash():
    """Identifies the build the profile files under profiles/ belong to."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "genrich_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".cpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


PATH_BITS = ((1, "fused"), (16, "pairs"), (128, "frac_pairs"), (2, "loose_sweep"), (4, "fell_back"), (8, "pt_grew"), (32, "dense_bh"),
             (64, "range_bh"), (1024, "merge_p"), (2048, "pack_hist"), (8192, "lazy_q"), (16384, "late_loose"), (32768, "q_loose"))


def decode_path(flags):
    """gx_path_info's bits by name (include/genrich_amd.h GX_PATH_*: k_sbtile, one record per fragment, ... with a weight class,
    the sweep on the tile stage's loose slots, a sample sent back to the general chain, page tables grown, the dense /
    range-partitioned BH exchange), joined with '+': which device path a context took."""
    return "+".join(n for b, n in PATH_BITS if flags & b) or "general"


def sig(x, n=4):
    """n significant digits (the short per-config summary at the end of the line)"""
    return None if x is None else float(f"{float(x):.{n}g}")


_PLAIN = {}   # (fragments, seed) -> the plain config-2 stream of that seed: four of the default run's workloads start from seed 1


def plain_fragments(lens, frags, seed):
    key = (int(frags), int(seed))
    if key not in _PLAIN:
        _PLAIN[key] = synth.make_fragments(lens, frags, seed=seed)
    return _PLAIN[key]


def build_workload(cfg, frags, lens):
    """[(treatment events, control events or None)] per replicate, SURVEY.md 8(d)."""
    reps = []
    seeds = [1, 3, 5, 7, 9][:cfg["reps"]]
    for sd in seeds:
        # with a control the treatment needs towers whose q survives the genome-wide correction
        # (and peaks of ~1,000 fragments every 200 kb instead of ~250 every 50 kb, so that > 10^4 peaks stay significant
        # at -q 0.05, not only the towers: round 2's stream left the q-mode sweep 33 peaks to work on)
        tv = (synth.make_fragments(lens, frags, seed=sd, peak_every=200_000, tower_every=50_000_000) if cfg["control"]
              else plain_fragments(lens, frags, sd))
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


def run_backend(be, lens, reps, peaks_to=None, skip=None, beds=None):
    be.set_chroms(lens, skip, beds)
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


def gate_and_cpu_baseline(cfg, lens, reps, n_chrom, qval, device, timed_path=None):
    """The oracle (CPU restatement, one core) and the HIP path on the same events: narrowPeak text diff,
    max |dp| / |dq| over all intervals, and the oracle's rate as the CPU baseline.  The gate's context is given what the
    timed context was given (the -s hint of a multimapped workload), its device path is reported next to the timed one,
    and the gate fails when the two differ: what is compared with the oracle is the path that was timed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import backends as B

    sub_lens = lens[:n_chrom]
    sub = subset(reps, n_chrom)
    names = synth.HG38_NAMES[:n_chrom]
    par = B.make_params(pq=0.05 if qval else 0.01, qval=qval)
    td = tempfile.mkdtemp()
    po, ph = os.path.join(td, "o.narrowPeak"), os.path.join(td, "h.narrowPeak")
    skip, beds = exclusions(cfg, lens)
    if skip is not None:
        skip, beds = skip[:n_chrom], beds[:n_chrom]
    o = B.Oracle(par)
    dt = run_backend(o, sub_lens, sub, peaks_to=(po, names), skip=skip, beds=beds)
    par.device = device
    h = Genrich(par)
    if cfg["multimap"]:
        h.expect_fractional(True)   # (what genrich-amd -s tells the library, and what the timed context was told)
    run_backend(h, sub_lens, sub, skip=skip, beds=beds)
    if cfg["multimap"]:
        # (the timed steps run on a context that has SEEN the fractions -- its warm-up steps: lambda comes with the sample's end and the
        # loose slots are swept late, k_loose_late; the gate's context learns the same way, and its second run is what is compared)
        h.reset()
        run_backend(h, sub_lens, sub, skip=skip, beds=beds)
    gate_flags = h.path_info()
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
        if skip is not None and skip[c]:
            continue
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
    gate_path = decode_path(gate_flags)
    # (the BH exchanges and "the page tables grew" belong to N ranks / to a pile-up, not to the choice of kernels)
    kernels = lambda p: {x for x in p.split("+") if x in ("fused", "pairs", "frac_pairs", "loose_sweep", "fell_back", "merge_p", "pack_hist", "lazy_q", "late_loose", "q_loose")}  # noqa: E731
    same = timed_path is None or kernels(gate_path) == kernels(timed_path)
    gate = dict(narrowpeak_diff=ndiff, peaks_oracle=int(o.n_peaks), peaks_hip=int(h.n_peaks), interval_ends_equal=ends_equal,
                intervals_compared=n_iv, max_abs_dp=dp, max_abs_dq=dq if qval else None, pq_values_differing_in_bits=nbits,
                device_path=gate_path, same_path_as_timed=bool(same),
                passed=bool(ndiff == 0 and ends_equal and dp <= 1e-5 and dq <= 1e-5 and same))
    what = "the whole workload" if n_chrom == len(lens) else f"hg38 chr1-chr{n_chrom} of the workload"
    cpu = dict(value=bases / dt / 1e9, unit="Gbases/s", cores=1, kind="port",
               sample=f"{what} ({sum(sub_lens)/1e6:.0f} Mbp x {len(sub)} replicate(s), {n_ev} events), events in memory -> "
                      f"peaks, {dt:.1f} s")
    o.close()
    return gate, cpu


def load_profile(config, frags, world, plain):
    """The rocprofv3 counters of THIS build for this config (tools/profile_round.sh + tools/make_counters_json.py):
    accepted only when the hash of the kernel sources matches and the workload is the profiled one."""
    if world != 1 or frags != 50_000_000 or not plain:
        return None
    import glob
    for ppath in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_counters_config{config}.json")), reverse=True):
        prof = json.load(open(ppath))
        if prof.get("source_hash") == source_hash():
            prof["_path"] = os.path.relpath(ppath, ROOT)
            return prof
    return None


def issue_roof(prof, kname, launches_per_step, launch_ms):
    """The second bound of a kernel that is not waiting for HBM: instruction issue.  A wavefront's VALU instruction
    occupies its SIMD for 4 cycles (64 lanes on 16), so the kernel cannot finish before
        VALU wave-instructions x 4 / (256 CUs x 4 SIMDs x 2.4 GHz)
    -- instruction counts from the SQ counters of this build's rocprofv3 profile (per launch), the duration measured
    live; valu_frac = that floor / the measured duration.  SALU instructions issue from the same wavefronts' streams
    (one instruction per wavefront and cycle) and are listed beside it."""
    if not prof or not prof.get("issue") or prof["issue"].get("kernel") != kname or launch_ms <= 0:
        return None
    out = {k: v for k, v in prof["issue"].items() if k != "raw"}   # (no counter dumps in the line: profiles/ has them)
    raw = prof["issue"].get("raw", {})
    valu = raw.get("SQ_INSTS_VALU", 0.0) / max(1.0, launches_per_step)
    salu = raw.get("SQ_INSTS_SALU", 0.0) / max(1.0, launches_per_step)
    n_simd, clock = 256 * 4, 2.4e9
    floor_ms = valu * 4.0 / (n_simd * clock) * 1e3
    out.update({"bound": "valu-issue", "valu_wave_insts_per_launch": valu, "salu_wave_insts_per_launch": salu,
                "salu_per_valu": (salu / valu) if valu else None, "simds": n_simd, "clock_ghz": clock / 1e9,
                "valu_floor_ms": floor_ms, "launch_ms": launch_ms, "valu_frac": floor_ms / launch_ms})
    return out


def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count()


def reference_e2e(lens, reps, max_frags=1_500_000):
    """SAM text -> narrowPeak through the REFERENCE binary built by oracle/Makefile (oracle/_ref/Genrich; it travels
    with the repo), one thread, on a bounded sample: BASELINE.md section 4's end-to-end CPU figure."""
    import subprocess
    ref = os.path.join(ROOT, "oracle", "_ref", "Genrich")
    if not os.path.exists(ref):
        return None
    sel = [18, 19, 20, 21]  # chr19..chr22 (220 Mbp): a genome the reference's per-base arrays fill in seconds
    tv = reps[0][0]
    tv = tv[np.isin(tv["chrom"], sel)][:max_frags].copy()
    remap = {c: i for i, c in enumerate(sel)}
    tv["chrom"] = np.array([remap[c] for c in tv["chrom"]], dtype=np.uint32)
    names = [synth.HG38_NAMES[c] for c in sel]
    slens = [lens[c] for c in sel]
    td = tempfile.mkdtemp()
    sam, outp = os.path.join(td, "t.sam"), os.path.join(td, "o.narrowPeak")
    synth.write_sam(sam, names, slens, tv)
    nrec = 2 * len(tv)
    t0 = time.perf_counter()
    rc = subprocess.call([ref, "-t", sam, "-o", outp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    # the same SAM text through THIS repo's host program up to the events (--events-only: no device work), with its
    # default number of decoder threads: the ingest rate SURVEY 8 row f3 is about
    host = None
    hb = os.path.join(ROOT, "genrich_amd", "genrich-amd")
    if os.path.exists(hb):
        t1 = time.perf_counter()
        hrc = subprocess.call([hb, "--events-only", "-t", sam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        hdt = time.perf_counter() - t1
        if hrc == 0:
            ncpu = os.cpu_count() or 1
            n = min(16, ncpu)
            pools = {"inflate": n, "decode": n, "state": n} if 3 * n + 2 <= ncpu or n <= 2 else \
                    {"inflate": max(2, n // 4), "decode": max(2, n // 2), "state": max(1, n // 2)}
            # (genrich-amd --events-only on the same SAM text: parse, pair, weight -> events; no device work)
            host = {"records_per_s": nrec / hdt, "seconds": hdt, "threads": pools}
    for f in (sam, outp):
        if os.path.exists(f):
            os.remove(f)
    os.rmdir(td)
    if rc != 0:
        return None
    return {"kind": "reference", "host_ingest": host, "seconds": dt, "sam_records": nrec, "records_per_s": nrec / dt,
            "gbases_per_s": sum(slens) / dt / 1e9,
            "sample": f"oracle/_ref/Genrich -t (SAM text, {nrec} records = {len(tv)} fragments of the workload on chr19-chr22, "
                      f"{sum(slens)/1e6:.0f} Mbp) -> narrowPeak, one thread, {dt:.1f} s"}


def e2e_cli(lens, frags, n_frags=10_000_000):
    """North_star's drop-in surface at size: SAM text of the headline stream's first `n_frags` fragments (2 x n_frags
    records, the whole hg38 table) -> narrowPeak through `genrich-amd` (this repo's host program: threaded ingest ->
    events -> the device path -> text), wall clock with process start and HIP initialisation; the REFERENCE binary
    (oracle/_ref/Genrich, one thread) on the very same file beside it, and whether the two narrowPeak files are the
    same bytes.  Outside the timed region of `value`."""
    import shutil
    import subprocess
    hb = os.path.join(ROOT, "genrich_amd", "genrich-amd")
    tool = os.path.join(ROOT, "tools", "events_to_sam")
    ref = os.path.join(ROOT, "oracle", "_ref", "Genrich")
    if not os.path.exists(tool) and os.path.exists(tool + ".c"):
        subprocess.call(["gcc", "-O2", "-o", tool, tool + ".c"])
    if not (os.path.exists(hb) and os.path.exists(tool)):
        return {"error": "genrich-amd / tools/events_to_sam not built"}
    td = tempfile.mkdtemp()
    try:
        if shutil.disk_usage(td).free < 4 << 30:
            return {"error": "less than 4 GB free for the SAM text"}
        ev = plain_fragments(lens, frags, 1)[:n_frags]
        evp, chp, sam = os.path.join(td, "ev.bin"), os.path.join(td, "chroms.txt"), os.path.join(td, "t.sam")
        ev.tofile(evp)
        open(chp, "w").write("".join(f"{n} {l}\n" for n, l in zip(synth.HG38_NAMES, lens)))
        subprocess.check_call([tool, evp, chp, sam])
        os.remove(evp)
        nrec, nbytes = 2 * len(ev), os.path.getsize(sam)
        ncpu = os.cpu_count() or 1
        out = {"sam_records": nrec, "sam_bytes": nbytes, "host_cores": ncpu,
               "sample": f"SAM text, {nrec} records = the first {len(ev)} fragments of the headline stream on all 25 hg38 contigs, "
                         "-p 0.01 defaults -> narrowPeak"}
        runs = []
        oh = os.path.join(td, "h.narrowPeak")
        for _ in range(2):   # (the first run also pages the file and the HIP runtime in)
            t0 = time.perf_counter()
            rc = subprocess.call([hb, "-t", sam, "-o", oh], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            runs.append(time.perf_counter() - t0)
            if rc != 0:
                return dict(out, error=f"genrich-amd exited with {rc}")
        dt = min(runs)
        out["genrich_amd"] = {"seconds": dt, "seconds_first_run": runs[0], "records_per_s": nrec / dt, "gbases_per_s": sum(lens) / dt / 1e9,
                              "threads": "default"}   # (process start + HIP initialisation + threaded ingest + device path + text)
        if os.path.exists(ref):
            orf = os.path.join(td, "r.narrowPeak")
            t0 = time.perf_counter()
            try:
                rc = subprocess.call([ref, "-t", sam, "-o", orf], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
            except subprocess.TimeoutExpired:
                rc = -1
            rdt = time.perf_counter() - t0
            if rc == 0:
                same = open(orf, "rb").read() == open(oh, "rb").read()
                out["reference"] = {"seconds": rdt, "records_per_s": nrec / rdt, "gbases_per_s": sum(lens) / rdt / 1e9, "threads": 1,
                                    "kind": "reference"}   # (oracle/_ref/Genrich -t on the same file)
                out["narrowpeak_identical_to_reference"] = bool(same)
                out["speedup_vs_reference"] = rdt / dt
            else:
                out["reference"] = {"error": f"exit {rc} after {rdt:.0f} s"}
        return out
    finally:
        shutil.rmtree(td, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frags", type=int, default=50_000_000)
    ap.add_argument("--config", type=lambda v: v if v in CONFIGS else int(v), default=2, choices=list(CONFIGS))
    ap.add_argument("--qval", action="store_true", help="-q 0.05 instead of -p 0.01 (on top of --config)")
    ap.add_argument("--control", action="store_true", help="add a uniform control (on top of --config)")
    ap.add_argument("--lean", action="store_true",
                    help="do not materialise the pileup floats of the intervals (gx_set_keep_pileups(0))")
    ap.add_argument("--no-cpu", action="store_true", help="skip the gate + cpu_baseline leg (and the other configs)")
    ap.add_argument("--cpu-chroms", type=int, default=0, help="gate / cpu_baseline on the first K chromosomes (0: per config)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the H2D / end-to-end-from-pinned figures")
    ap.add_argument("--no-materialised", action="store_true", help="skip the second timed loop with the tight interval table written (profiling runs)")
    ap.add_argument("--no-others", action="store_true", help="headline config only (default: configs 3, 4, 5 ride along, 3 steps each)")
    args = ap.parse_args()
    # The JSON line must be the only thing on stdout: libraries below Python (RCCL prints a version banner
    # through C stdio, flushed at exit) write to file descriptor 1, so the real stdout is put aside and
    # descriptor 1 joins stderr until the line is written.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

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
    env = dict(torch=torch, dist=dist, rank=rank, world=world, local_dev=local_dev, dev=dev, cdev=cdev, backend=backend, ndev=ndev)

    cfg = dict(CONFIGS[args.config])
    plain = not (args.qval or args.control)
    if args.qval and not cfg["qval"]:
        cfg["desc"] = cfg["desc"].replace("-p 0.01", "-q 0.05")
    if args.control and not cfg["control"]:
        cfg["desc"] = cfg["desc"].replace("treatment only", "treatment + 50M-fragment uniform control")
    cfg["qval"] = cfg["qval"] or args.qval
    cfg["control"] = cfg["control"] or args.control
    out = bench_one(args.config, cfg, args, env, steps=args.steps, warmup=args.warmup, plain=plain,
                    want_e2e=not args.no_e2e, want_cpu=not args.no_cpu, headline=True)
    if rank == 0 and world == 1 and args.config == 2 and plain and not args.no_cpu and not args.no_others:
        # BASELINE.json's other GPU configs in the same line (the driver runs bench.py once, with defaults)
        detail, brief = {}, {}
        # "2q": the headline workload with -q 0.05 -- north_star's "p + q scan" of one 50 M-fragment sample
        cfg2q = dict(CONFIGS[2], qval=True, gate_chroms=8, name="configs[1] with -q 0.05", profile_as="2q",
                     desc=CONFIGS[2]["desc"].replace("-p 0.01", "-q 0.05"))
        for c, ccfg, cplain in (("2q", cfg2q, False), ("2E", dict(CONFIGS["2E"]), True), (3, dict(CONFIGS[3]), True), (4, dict(CONFIGS[4]), True),
                                (5, dict(CONFIGS[5]), True)):
            try:
                r = bench_one(2 if c == "2q" else c, ccfg, args, env, steps=5, warmup=3, plain=cplain, want_e2e=False, want_cpu=True,
                              headline=False)
                d = {k: r[k] for k in ("ms_per_step", "value", "gate", "phases_ms") if k in r}
                d["workload"] = r["config"]["workload"]
                d["whole_step"] = r["roofline"]["whole_step"]
                d["dominant"] = {k: r["roofline"].get(k) for k in ("kernel", "frac", "frac_is", "achieved", "launch_ms", "traffic", "traffic_frac", "algorithmic_bytes")}
                d["cpu_baseline"] = r.get("cpu_baseline")
                detail[str(c)] = d
                g, w = r.get("gate", {}), r["roofline"]["whole_step"]
                brief[str(c)] = {"ms_per_step": sig(r["ms_per_step"]), "value": sig(r["value"]), "gate_passed": g.get("passed"),
                                 "narrowpeak_diff": g.get("narrowpeak_diff"), "pq_bits_differing": g.get("pq_values_differing_in_bits"),
                                 "path": r["config"]["device_path"], "dominant": r["roofline"]["kernel"],
                                 "dominant_ms": sig(r["roofline"]["launch_ms"]), "dominant_frac": sig(r["roofline"]["frac"]),
                                 "traffic_over_alg": sig(w.get("traffic_over_algorithmic"))}
            except Exception as e:  # noqa: BLE001  (the headline must still be printed)
                brief[str(c)] = {"error": repr(e)[:120]}
        out["other_configs_detail"] = detail
        # north_star's sentence -- the p + q scan of one 50 M-fragment sample -- and the materialised step as flat keys, here and
        # (the driver's record keeps the scalars of `config`) there
        q2 = brief.get("2q", {})
        flat = {"p_plus_q_ms_per_step": q2.get("ms_per_step"), "p_plus_q_value": q2.get("value"), "p_plus_q_gate_passed": q2.get("gate_passed"),
                "excluded_regions_ms_per_step": brief.get("2E", {}).get("ms_per_step")}
        out.update(flat)
        out["config"].update(flat)
        others_brief = brief
    else:
        others_brief = None
    if rank == 0 and world == 1 and args.config == 2 and plain and not args.no_cpu and not args.no_e2e:
        try:
            out["e2e_cli"] = e2e_cli(synth.HG38_LENS, args.frags)
        except Exception as e:  # noqa: BLE001
            out["e2e_cli"] = {"error": repr(e)}
    if rank == 0:
        if others_brief is not None:
            out["other_configs"] = others_brief   # LAST in the line, short: the tail of a truncated record still holds every config
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if world > 1:
        dist.destroy_process_group()


def bench_one(config, cfg, args, env, steps, warmup, plain, want_e2e, want_cpu, headline):
    torch, dist = env["torch"], env["dist"]
    rank, world, local_dev, dev, cdev, backend, ndev = (env[k] for k in ("rank", "world", "local_dev", "dev", "cdev", "backend", "ndev"))
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
    skip, beds = exclusions(cfg, lens)
    gx.set_chroms(lens, skip, beds)
    # By default the whole interval table (end, treatment pileup, p) is materialised, as the reference holds it.
    # --lean drops the pileup floats, which only the -f / -k emitters read: reported as such in `config`.
    gx.set_keep_pileups(not args.lean)
    if cfg["multimap"]:
        gx.expect_fractional(True)   # (genrich-amd -s does: pair records with a weight class from the first sample on)
    coll_kind = "none"
    force_rccl = world == 1 and os.environ.get("GX_BENCH_FORCE_RCCL") == "1"   # exercise the RCCL path with one rank
    if force_rccl:
        # (the library reads its switches once, in gx_create: the environment is too late here -- ADVICE r5)
        gx.set_knob("GX_FORCE_COLL", 1)
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
                gx.set_collectives(rank, world, coll.allreduce_i64)
                coll_kind = "host callbacks over torch.distributed/nccl (the library's own communicator failed)"
        else:
            coll = Collectives(device=cdev)
            gx.set_collectives(rank, world, coll.allreduce_i64)
            coll_kind = f"host callbacks over torch.distributed/{backend} (validation mode)"
    rccl_nranks = gx.rccl_nranks()

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

    # The roofline's kernel = the longest kernel of this build's rocprofv3 profile of this config (k_sort1 at config 2);
    # without a matching profile, the tile stage.  Inside the timed region only THAT kernel's phase is bracketed by HIP
    # events on the library's stream (an event record costs the stream a ~5 us bubble); all phases are timed in two extra,
    # untimed steps afterwards.
    # (the headline with -q has a profile of its own: tools/profile_round.sh <tag> 2q)
    prof = load_profile(cfg.get("profile_as", config), args.frags, world, plain or "profile_as" in cfg)
    dom = prof.get("dominant", {}).get("kernel") if prof else None
    # (the library phase that brackets it -- gx_set_phase_filter -- comes from the profile's own marker trace: the library's roctx
    # ranges, GX_ROCTX; without a profile of this build the tile stage is what is timed)
    dom_phase = prof["kernels"].get(dom, {}).get("phase") if prof and dom else None
    if not dom_phase:
        dom, dom_phase = None, "tile"
    gx.set_phase_filter(dom_phase)
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    phase_acc = {}
    for _ in range(steps):
        res = step()
        collect(phase_acc)
    barrier()
    dt = time.perf_counter() - t0
    path_flags = gx.path_info()   # (after the timed steps: which path THEY took, and whether they wrote pileup floats)
    if force_rccl and cfg["qval"] and not path_flags & (32 | 64):
        raise RuntimeError("GX_BENCH_FORCE_RCCL: the BH exchange did not go through the collectives (path flags %#x)" % path_flags)
    # The same step with the tight interval table MATERIALISED (GX_NO_LOOSE: lambda only after the tile stage, then
    # k_pack_pval writes (end, p) and the sweep's masks, as every run with -q / a control / a further replicate / -f / -k
    # does): what the default step of a single -p sample leaves out because the sweep reads (end, V) where the tile
    # stage put them.  Only for the headline, one rank.
    mat_ms = None
    if headline and world == 1 and not cfg["qval"] and not cfg["control"] and cfg["reps"] == 1 and not args.no_materialised:
        gx.set_knob("GX_NO_LOOSE", 1)
        try:
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            mat_ms = (time.perf_counter() - t1) / steps * 1e3
            mat_flags = gx.path_info()
        finally:
            gx.set_knob("GX_NO_LOOSE", 0)
        step()
    # The same step with the events in their 8-byte form (gx_event8, gx_push_events_packed: half of the step's largest stream),
    # resident in HBM like the 16-byte ones of `value`.  Only for the headline, one rank.
    packed_ms = packed_same = None
    if headline and world == 1 and not args.no_materialised:
        peaks16 = gx.get_peaks().tobytes()
        p8 = [(pack_events(tv), None if cv is None else pack_events(cv)) for tv, cv in reps_all]
        if all(len(t[1]) == 0 and (c is None or len(c[1]) == 0) for t, c in p8):
            d8 = [(torch.from_numpy(t[0].view(np.uint32).reshape(-1, 2).copy()).to(dev),
                   None if c is None else torch.from_numpy(c[0].view(np.uint32).reshape(-1, 2).copy()).to(dev)) for t, c in p8]

            def step8():
                gx.reset()
                for d_tv, d_cv in d8:
                    gx.sample_begin(0, None)
                    gx.push_events_packed(d_tv.data_ptr(), where=2, n=d_tv.shape[0])
                    gx.sample_end()
                    if d_cv is not None:
                        gx.sample_begin(1, None)
                        gx.push_events_packed(d_cv.data_ptr(), where=2, n=d_cv.shape[0])
                        gx.sample_end()
                    else:
                        gx.sample_no_control()
                    gx.pvalues()
                return gx.find_peaks()

            for _ in range(2):
                step8()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                step8()
            torch.cuda.synchronize()
            packed_ms = (time.perf_counter() - t1) / steps * 1e3
            packed_same = bool(gx.path_info() & 512) and gx.get_peaks().tobytes() == peaks16
            del d8
            step()
    gx.set_phase_timing(2)
    all_phases = {}
    for _ in range(2):
        step()
        collect(all_phases)
    gx.set_phase_timing(0)
    peaks_local = gx.get_peaks()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        npk = torch.tensor([res[0]], dtype=torch.int64, device=cdev)
        dist.all_reduce(npk)
        n_peaks = int(npk.item())
        nr = torch.tensor([rccl_nranks], dtype=torch.int64, device=cdev)
        dist.all_reduce(nr, op=dist.ReduceOp.MIN)
        rccl_nranks = int(nr.item())
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(peaks_local, gathered, dst=0)
    else:
        n_peaks = res[0]
        gathered = [peaks_local]

    out = None
    if rank == 0:
        step_s = dt / steps
        phases = {k: float(np.mean(v)) for k, v in all_phases.items()}       # untimed steps, every phase
        phases.update({k: float(np.mean(v)) for k, v in phase_acc.items()})  # the roofline kernel's phase: live, timed region
        n_rep = len(reps_all)
        iv0 = float(gx.interval_total())
        ev_n = float(sum(d_tv.shape[0] + (0 if d_cv is None else d_cv.shape[0]) for d_tv, d_cv in d_reps))
        n_tiles = sum((l + 4095) // 4096 for l, o in zip(lens, owner) if o == 0)
        launches = n_rep * (2 if cfg["control"] else 1)   # sort / tile kernels run once per sample
        # (the library's phase timers cover the last replicate of a step and gx_find_peaks)
        per_sample = dom_phase in ("sort1", "tile", "bucket")
        live_ms = sum(phases.get(pfx + dom_phase, 0.0) for pfx in (("t.", "c.") if per_sample else ("",)))
        if per_sample and cfg["control"]:
            live_ms /= 2
        kname = dom or ("k_sbtile" if path_flags & 1 else "k_tile" if cfg["excl"] else "k_tile_fast")
        kprof = prof["kernels"].get(kname) if prof else None
        # (a sample's tile stage = k_sbtile's first launch + its second, usually idle one over the listed bins: ONE launch
        # here, as the phase timer brackets both -- the profile counts the instances of the template apart)
        klaunch = float(launches) if per_sample or not kprof else max(1.0, kprof.get("launches_per_step", 1.0))
        traffic = kprof["hbm_bytes_per_step"] / klaunch if kprof and kprof.get("hbm_bytes_per_step") else None
        # compulsory HBM bytes of the sparse formulation, per launch:
        #   k_sort1: 16 B per event in, 4 B per endpoint key out (two per event)
        #   tile stage: 4 B per key in (k_sbtile reads level 1's pages; k_tile_fast 2 B offsets), 8 B per interval out,
        #               56 B of descriptors / counts per tile
        ev_launch = ev_n / launches
        if not per_sample:
            alg_k = None  # (a merge / Fisher / BH kernel: only the counter figure is quoted)
        elif dom_phase == "sort1":
            alg_k = 16.0 * ev_launch + 8.0 * ev_launch
        else:
            # (pair mode: one 4-byte record per fragment; else two 4-byte keys (k_sbtile) or two 2-byte offsets (k_tile_fast))
            key_bytes = 4.0 if path_flags & 16 else (8.0 if path_flags & 1 else 4.0)
            alg_k = key_bytes * ev_launch + 8.0 * (iv0 if launches == 1 else 2.0 * ev_launch) + 56.0 * n_tiles
        # `achieved` / `frac`: ALGORITHMIC bytes of the launch over its live duration (the contract's definition).  The same with
        # the PMC traffic of the profile: `traffic_gbs` / `traffic_frac` (round 4's line quoted that one as `frac`).  A merge /
        # Fisher / BH kernel has no closed form of its compulsory bytes here: the counter figure stands in, and says so.
        used = alg_k if alg_k else (traffic or 0.0)
        achieved = used / (live_ms * 1e-3) / 1e9 if live_ms > 0 else 0.0
        traffic_gbs = (traffic / (live_ms * 1e-3) / 1e9) if traffic and live_ms > 0 else None
        # whole step: events in + final interval table (end, p[, pileup]) + sweep masks out; the loose-slot sweep of a
        # single sample with -p leaves (end, V) in the tile stage's slots and makes no second table
        loose = bool(path_flags & 2)
        alg_step = 16.0 * ev_n + (8.0 if loose else (12.0 if not args.lean else 8.0)) * iv0 + (2.0 if loose else 3.0) * iv0 / 8.0
        whole_traffic = prof["whole_step"]["hbm_bytes_per_step"] if prof else None
        roof = {
            "bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "frac_is": "algorithmic bytes" if alg_k else "PMC traffic (no closed form)",
            "traffic": traffic, "traffic_gbs": traffic_gbs, "traffic_frac": (traffic_gbs / HBM_PEAK_GBS) if traffic_gbs else None,
            "launch_ms": live_ms / (1.0 if per_sample else klaunch),
            "algorithmic_bytes": alg_k,
            "traffic_over_algorithmic": (traffic / alg_k) if traffic and alg_k else None,
            "profile": prof["_path"] if prof else None,
            "whole_step": {
                "ms": step_s * 1e3,
                "algorithmic_bytes": alg_step,
                "traffic": whole_traffic,
                "frac_of_peak": (alg_step / step_s / 1e9) / HBM_PEAK_GBS,
                "traffic_frac_of_peak": ((whole_traffic / step_s / 1e9) / HBM_PEAK_GBS) if whole_traffic else None,
                "traffic_over_algorithmic": (whole_traffic / alg_step) if whole_traffic else None,
            },
            "issue": issue_roof(prof, kname, launches if per_sample else klaunch, live_ms / (1.0 if per_sample else klaunch)),
            "dense_model_bytes": 8.0 * G + 16.0 * ev_n + 52.0 * iv0,   # SURVEY 8(d)'s dense int32-array model (the array lives in LDS here)
        }
        qdesc = cfg["desc"]
        out = {
            "metric": "genome bases p-scored/sec, hg38 50M frags",
            "value": n_rep * G / step_s / 1e9,
            "unit": "Gbases/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": step_s * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32 pileup (1/120 units) + f64 p-values",
            "data": "synthetic",
            "config": {
                "workload": f"hg38 25 contigs ({G} bp), {args.frags} paired fragments x {n_rep} replicate(s), {qdesc} "
                            f"(BASELINE.json {cfg['name']}{'' if plain else ', modified by --qval / --control'})",
                "parallelism": f"chromosome-sharded x{world}"
                               + ("" if backend == "nccl" or world == 1 else f" ({backend} validation mode, {ndev} GPU(s))"),
                "collectives": coll_kind,
                "rccl_nranks": rccl_nranks,
                "device_path": decode_path(path_flags),
                "peaks": n_peaks,
                "intervals": int(iv0),
                "events_per_step": int(ev_n),
                # what the timed step wrote to HBM besides the loose (end, V) slots of the tile stage and the sweep's masks
                # (pileup floats: gx_path_info bit 8 -- made on request only)
                "tables_written_in_step": ("none: the sweep walks the loose (end, V) slots, p = table p(V)" if loose else
                                           "tight (end, p[, q]) interval table"
                                           + (" + pileup floats" if path_flags & 256 else "")),
                "source_hash": source_hash(),
            },
            "roofline": roof,
            # the same step with the tight (end, p) table written (k_pack_pval) and the sweep on it
            "materialised": ({"ms_per_step": mat_ms, "value": n_rep * G / (mat_ms * 1e-3) / 1e9, "unit": "Gbases/s",
                              "sweep_on_loose_slots": bool(mat_flags & 2)} if mat_ms else None),
            "materialised_ms_per_step": mat_ms,
            # the same step on 8-byte events (gx_event8) resident in HBM; same_peaks: read in place and the 16-byte step's peak bytes
            "packed_events": ({"ms_per_step": packed_ms, "value": n_rep * G / (packed_ms * 1e-3) / 1e9, "unit": "Gbases/s",
                               "same_peaks": packed_same} if packed_ms else None),
            # (the roofline kernel's phase: HIP events inside the timed region; the others: two extra untimed steps)
            "phases_ms": phases,
        }
        out["config"]["materialised_ms_per_step"] = mat_ms
        out["config"]["packed_events_ms_per_step"] = packed_ms
        if world == 1 and want_e2e:
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
            # (events from pinned host memory to HBM, outside the timed region of `value`; and gx_push_events_pinned -> peaks)
            out["h2d"] = {"ms": h2d_ms, "bytes": nbytes, "gbs": nbytes / h2d_ms / 1e6}
            out["e2e_from_pinned"] = {"ms_per_step": e2e_ms, "value": n_rep * G / (e2e_ms * 1e-3) / 1e9, "unit": "Gbases/s"}
            # ... and from pinned 8-byte events (half the bytes over PCIe)
            p8 = [(pack_events(tv), None if cv is None else pack_events(cv)) for tv, cv in reps_all]
            if all(len(t[1]) == 0 and (c is None or len(c[1]) == 0) for t, c in p8):
                pin8 = [(torch.from_numpy(t[0].view(np.uint32).reshape(-1, 2)).pin_memory(),
                         None if c is None else torch.from_numpy(c[0].view(np.uint32).reshape(-1, 2)).pin_memory()) for t, c in p8]

                def step_from_host8():
                    gx.reset()
                    for t, c in pin8:
                        gx.sample_begin(0, None)
                        gx.push_events_packed(t.data_ptr(), where=1, n=t.shape[0])
                        gx.sample_end()
                        if c is not None:
                            gx.sample_begin(1, None)
                            gx.push_events_packed(c.data_ptr(), where=1, n=c.shape[0])
                            gx.sample_end()
                        else:
                            gx.sample_no_control()
                        gx.pvalues()
                    return gx.find_peaks()

                step_from_host8()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    step_from_host8()
                torch.cuda.synchronize()
                e8 = (time.perf_counter() - t1) / 3 * 1e3
                out["e2e_from_pinned_packed"] = {"ms_per_step": e8, "value": n_rep * G / (e8 * 1e-3) / 1e9, "unit": "Gbases/s"}
                del pin8
            del pin
    # the timed context and its device arrays go before the gate's (and the next config's) are made
    gx.close()
    del d_reps
    torch.cuda.empty_cache()
    if rank == 0 and want_cpu:
        if world == 1:
            k = args.cpu_chroms or cfg["gate_chroms"]
            gate, cpu = gate_and_cpu_baseline(cfg, lens, reps_all, min(k, len(lens)), cfg["qval"], local_dev,
                                              timed_path=out["config"]["device_path"])
        else:
            # N ranks: the ranks' peak lists, merged in chromosome order, against the oracle's list for the whole workload
            gate, cpu = gate_merged_peaks(cfg, lens, reps_all, gathered, cfg["qval"])
        model, ncpu = cpu_info()
        cpu["cpu_model"], cpu["nproc"] = model, ncpu
        if headline:
            cpu["end_to_end"] = reference_e2e(lens, reps_all)
        out["gate"] = gate
        out["cpu_baseline"] = cpu
    return out


def gate_merged_peaks(cfg, lens, reps, gathered, qval):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import backends as B

    par = B.make_params(pq=0.05 if qval else 0.01, qval=qval)
    o = B.Oracle(par)
    dt = run_backend(o, lens, reps)
    want = o.get_peaks()
    got = np.concatenate([g for g in gathered if g is not None and len(g)]) if any(len(g) for g in gathered) else want[:0]
    got = got[np.lexsort((got["start"], got["chrom"]))]
    same = len(got) == len(want) and got.tobytes() == want.tobytes()
    ndiff = 0 if same else max(1, abs(len(got) - len(want)) + int(sum(1 for a, b in zip(got, want) if a.tobytes() != b.tobytes())))
    bases = float(sum(lens)) * len(reps)
    n_ev = int(sum(len(t) + (0 if c is None else len(c)) for t, c in reps))
    # (the ranks' peak records gathered to rank 0, merged in chromosome order and compared field by field -- chrom, start, end,
    # summit, AUC / p / q bits -- with the oracle's list for the whole workload: the narrowPeak text is a function of these)
    gate = dict(narrowpeak_diff=ndiff, peaks_oracle=int(len(want)), peaks_hip=int(len(got)), passed=bool(same))
    cpu = dict(value=bases / dt / 1e9, unit="Gbases/s", cores=1, kind="port",
               sample=f"the whole workload ({sum(lens)/1e6:.0f} Mbp x {len(reps)} replicate(s), {n_ev} events), events in memory -> "
                      f"peaks, {dt:.1f} s")
    o.close()
    return gate, cpu


if __name__ == "__main__":
    main()
