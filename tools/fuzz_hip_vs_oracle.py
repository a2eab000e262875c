"""GPU box: the HIP library against the CPU oracle on random runs.
usage: fuzz_hip_vs_oracle.py SEED0 SEED1 [--mid]

default: the generator of tests/test_hip_parity.py::test_random_runs_against_oracle (small runs)
--mid:   1-4 chromosomes of 0.2-6 Mbases, 0.1-0.9 M fragments (deep towers, multimapping, control)
--extreme: thresholds at and beyond their ends (-p / -q of 1, 0.999, 1e-30, 1e-300; -a 0, 1e6; -g 0,
         100000; -l 100000) on top of whichever generator is chosen
--x50:   the default generator (skipped chromosomes, -E regions, replicates, -p / -q, -a / -l / -g) with
         chromosomes and samples 50 times larger

--paths: mid-size runs with the device paths of round 4 forced at random -- few-tile bins (GX_SBSHIFT 1..3: rounds, heavy
         tiles, the second launch), half-size bins with the 128-key level 1, pair records off, fractional pairs off or
         announced (gx_expect_fractional), every sample pushed in 1-5 pieces; --seconds N stops after N seconds

Inputs that saturate the reference's int16 difference array are compared like any other: the oracle
drops alignments as the reference does, and so does the library (gx_saturate.h)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import backends as B  # noqa: E402
import test_hip_parity as T  # noqa: E402
synth = T.synth


def mid_case(seed):
    rng = np.random.default_rng(seed)
    nch = int(rng.integers(1, 5))
    lens = [int(x) for x in rng.integers(200_000, 6_000_000, nch)]
    tr = synth.make_fragments(lens, int(rng.integers(100_000, 900_000)), seed=seed,
                              frac_peak=float(rng.choice([0.05, 0.2, 0.4])), frac_tower=float(rng.choice([0.0, 0.05, 0.3])))
    if rng.random() < 0.3:
        tr = synth.add_multimap(tr, lens, 0.2, seed=seed + 1)
    ct = None
    if rng.random() < 0.5:
        ct = synth.make_fragments(lens, int(rng.integers(100_000, 900_000)), seed=seed + 2, uniform_only=True)
    qval = bool(rng.random() < 0.5)
    params = B.make_params(pq=0.05 if qval else 0.01, qval=qval, min_auc=float(rng.choice([20, 200])))
    return dict(lens=lens, replicates=[dict(save=None, treat=tr, ctrl=ct)]), params


def run_paths(seed):
    """one mid-size run with random path knobs; returns what assert_same_run needs"""
    rng = np.random.default_rng(seed + 12345)
    case, params = mid_case(seed)
    if rng.random() < 0.5:  # fractional weights more often than mid_case has them
        for rep in case["replicates"]:
            rep["treat"] = synth.add_multimap(rep["treat"][rep["treat"]["count"] == 1], case["lens"], float(rng.choice([0.1, 0.5])), seed=seed + 5)
    if rng.random() < 0.35:  # -E regions (round 6: on the fused tile stage), some touching position 0 / the chromosome's end
        beds = []
        for L in case["lens"]:
            regs = []
            for _ in range(int(rng.integers(0, 6))):
                s = int(rng.integers(0, max(1, L - 5)))
                regs.append((s, min(L, s + int(rng.choice([1, 50, 3000, 40_000, 400_000])))))
            if rng.random() < 0.3:
                regs.append((0, int(rng.integers(1, 9000))))
            if rng.random() < 0.3:
                regs.append((L - int(rng.integers(1, 9000)), L))
            regs.sort()
            merged = []
            for s, e in regs:
                if merged and s <= merged[-1][1]:
                    merged[-1][1] = max(merged[-1][1], e)
                else:
                    merged.append([s, e])
            beds.append([v for r in merged for v in r])
        case["beds"] = beds
    knobs = {}
    if rng.random() < 0.6:
        knobs["GX_SBSHIFT"] = str(int(rng.integers(1, 4)))
    if rng.random() < 0.3:
        knobs["GX_FORCE_HALF_BINS"] = "1"
    if rng.random() < 0.15:
        knobs["GX_NO_PAIRS"] = "1"
    if rng.random() < 0.15:
        knobs["GX_NO_FRAC_PAIRS"] = "1"
    # (round 6's alternatives: q of every interval by k_qlookup, BH's histogram by insertion, the control merge by a workgroup per
    # tile / leaving both pileups)
    for k, pr in (("GX_NO_LAZY_Q", 0.25), ("GX_NO_PACK_HIST", 0.2), ("GX_MERGE_WG", 0.15), ("GX_NO_MERGE_P", 0.15), ("GX_NO_LATE_LOOSE", 0.2),
                  ("GX_NO_Q_LOOSE", 0.25)):
        if rng.random() < pr:
            knobs[k] = "1"
    for k in PATH_KNOBS:
        os.environ.pop(k, None)
    os.environ.update(knobs)
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = T.hip_backend(params)
    if rng.random() < 0.5:
        h.expect_fractional(True)
    pieces = int(rng.integers(1, 6))
    whole = h.push_events
    packed = rng.random() < 0.5   # the pieces as 8-byte events where they fit (gx_push_events_packed)
    from genrich_amd.lib import pack_events

    def in_pieces(ev):
        cuts = sorted(int(x) for x in rng.integers(0, len(ev) + 1, pieces - 1))
        for a, b in zip([0] + cuts, cuts + [len(ev)]):
            if packed:
                p8, rest = pack_events(ev[a:b])
                if len(p8):
                    h.push_events_packed(p8)
                if len(rest):
                    whole(rest)
            else:
                whole(ev[a:b])
    h.push_events = in_pieces
    sh = B.run_case(h, case)
    T.assert_same_run(o, h, so, sh, case)
    flags = h.path_info()
    again = rng.random() < 0.4   # the same case once more on the context (what it has learned -- fractions seen -- takes other paths: k_loose_late)
    if again:
        h.reset()
        sh = B.run_case(h, case)
        T.assert_same_run(o, h, so, sh, case)
        flags |= h.path_info() << 16
    knobs = dict(knobs, beds=bool(case.get("beds")), packed=packed, again=again)
    return knobs, pieces, flags


def describe(case):
    """what a failing run looked like (chromosome lengths, skipped ones, -E regions, samples)"""
    reps = [dict(n=len(r["treat"]), frac=int((r["treat"]["count"] > 1).sum()), ctrl=None if r["ctrl"] is None else len(r["ctrl"]))
            for r in case["replicates"]]
    return dict(lens=case["lens"], skip=case.get("skip"), beds=case.get("beds"), reps=reps)


PATH_KNOBS = ("GX_SBSHIFT", "GX_FORCE_HALF_BINS", "GX_NO_PAIRS", "GX_NO_FRAC_PAIRS", "GX_NO_LAZY_Q", "GX_NO_PACK_HIST", "GX_MERGE_WG", "GX_NO_MERGE_P", "GX_NO_LATE_LOOSE", "GX_NO_Q_LOOSE")
mid = "--mid" in sys.argv
paths = "--paths" in sys.argv
import time  # noqa: E402
t_end = time.time() + float(sys.argv[sys.argv.index("--seconds") + 1]) if "--seconds" in sys.argv else None
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    if t_end and time.time() > t_end:
        print("time is up at seed", seed)
        break
    if paths:
        try:
            knobs, pieces, flags = run_paths(seed)
            print("seed", seed, knobs, "pieces", pieces, "path flags", flags, flush=True)
        except Exception as ex:  # noqa: BLE001
            bad += 1
            print("seed", seed, type(ex).__name__, str(ex)[:300], {k: os.environ.get(k) for k in PATH_KNOBS}, flush=True)
            print("   ", describe(mid_case(seed)[0]), flush=True)
        continue
    case, params = mid_case(seed) if mid else T._random_case(seed, 50 if "--x50" in sys.argv else 1)
    if "--extreme" in sys.argv:
        r = np.random.default_rng(seed + 77)
        qv = bool(r.random() < 0.5)
        params = B.make_params(pq=float(r.choice([1, 0.999, 1e-30, 1e-300, 0.05])), qval=qv,
                               min_auc=float(r.choice([0, 0.001, 1e6, 20])), min_len=int(r.choice([0, 100000])),
                               max_gap=int(r.choice([0, 100, 100000])))
    try:
        o, h, so, sh = T.run_both(case, params)
        T.assert_same_run(o, h, so, sh, case)
    except Exception as ex:  # noqa: BLE001
        bad += 1
        print("seed", seed, type(ex).__name__, str(ex)[:200], flush=True)
        print("   ", describe(case), flush=True)
print("done, failures:", bad)
