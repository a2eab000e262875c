"""One GPU playing one rank of an N-rank run: what a rank's share of the step costs.

   python tools/emulate_ranks.py [--config 2|3|4|5] [N | N:rank] ...        (default config 2; 1 2 4 8, rank 0)

Every exchange of the library is an all-reduce in the callback mode (sums of words, or of disjoint regions of a zeroed
buffer = concatenation, gx_api.hip: coll_concat / coll_alltoallv), so ONE callback stands for all of them: it adds what
the OTHER ranks would contribute, call by call, and lambda, the factor, the splitters and the q-values -- and with them
every kernel's work -- are the real run's.  What the other ranks contribute to call k can depend on the results of the
calls before it (the closed form of fragLen -> lambda -> the BH tables -> the splitters -> the records sent), so the
contributions are found by iteration: every rank's share is run with the others' words of the previous round until a
round changes nothing (a handful of rounds).  Then the chosen rank is timed.  The callbacks' host round trips (a copy
down, a synchronisation, a copy up per exchange; the whole exchange buffer for the all-to-all) stand in for RCCL in
the wall time; the phases' device times are printed next to it, `bh.xfer` = the exchange buffers' trips to the host."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import bench  # noqa: E402  (the workloads of BASELINE.json's configs)
from genrich_amd import synth  # noqa: E402
from genrich_amd.dist import lpt_partition  # noqa: E402
from genrich_amd.lib import GxParams, Genrich, minus_log10f  # noqa: E402

args = sys.argv[1:]
config = 2
if args and args[0] == "--config":
    config = int(args[1])
    args = args[2:]
cfg = bench.CONFIGS[config]
lens = synth.HG38_LENS
reps_all = bench.build_workload(cfg, 50_000_000, lens)
print(f"config {config}: {sum(len(t) + (0 if c is None else len(c)) for t, c in reps_all)} events", flush=True)
worlds = args or ["1", "2", "4", "8"]
libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]


def params():
    return GxParams(minus_log10f(0.05 if cfg["qval"] else 0.01), 1 if cfg["qval"] else 0, 200.0, 0, 100, 0, 0)


def to_dev(ev):
    return torch.from_numpy(ev.view(np.uint32).reshape(-1, 4).copy()).cuda()


def share(owned):
    sel = lambda ev: None if ev is None else to_dev(ev[owned[ev["chrom"]].astype(bool)])  # noqa: E731
    return [(sel(t), sel(c)) for t, c in reps_all]


def step(gx, dreps):
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


class Replay:
    """The all-reduce callback of one rank: record its own contribution, add the others' (by call index)."""

    def __init__(self, others=None):
        self.others = others
        self.red = []
        self.ri = 0

    def begin(self):
        self.red, self.ri = [], 0

    def allreduce(self, buf, n, _user):
        a = np.ctypeslib.as_array(buf, shape=(n,))
        self.red.append(a.copy())
        if self.others is not None and self.ri < len(self.others) and len(self.others[self.ri]) == n:
            a += self.others[self.ri]
        self.ri += 1
        return 0


def context(owned, rp, rank, world):
    gx = Genrich(params())
    gx.set_chroms(lens)
    gx.set_owned(owned)
    if cfg["multimap"]:
        gx.expect_fractional(True)   # (as genrich-amd -s and bench.py: fractional pair records from the first sample on)
    if world > 1:
        gx.set_collectives(rank, world, rp.allreduce)
    return gx


def others_sum(recs, rank):
    """per call: the sum over the other ranks of the words they handed to the all-reduce (None before the first round)"""
    if any(r is None for r in recs):
        return None
    calls = min(len(r) for r in recs)
    out = []
    for k in range(calls):
        n = len(recs[rank][k]) if k < len(recs[rank]) else -1
        if any(len(recs[r][k]) != n for r in range(len(recs))):
            break  # (the ranks are not yet in step at this call: a later round)
        out.append(sum(recs[r][k] for r in range(len(recs)) if r != rank))
    return out


def same(a, b):
    return a is not None and b is not None and len(a) == len(b) and all(len(x) == len(y) and np.array_equal(x, y) for x, y in zip(a, b))


for spec in worlds:
    world, rank = (int(x) for x in (spec.split(":") + ["0"])[:2])
    owner = lpt_partition(lens, world)
    owneds = [np.array([o == r for o in owner], dtype=np.uint8) for r in range(world)]
    red = [None] * world
    if world > 1:
        for rnd in range(12):
            new = [None] * world
            for r in range(world):
                rp = Replay(others_sum(red, r))
                gx = context(owneds[r], rp, r, world)
                d = share(owneds[r])
                rp.begin()
                try:
                    step(gx, d)
                except RuntimeError:  # (a round with stale words of the others may fail an internal check: the next one has better ones)
                    pass
                new[r] = rp.red
                gx.close()
                del d
            done = all(same(a, b) for a, b in zip(red, new))
            red = new
            if done:
                break
        print(f"  ({world} ranks: the others' words settled after {rnd + 1} rounds, {len(red[0])} exchanges per step)", flush=True)
    rp = Replay(others_sum(red, rank) if world > 1 else None)
    d_reps = share(owneds[rank])
    gx = context(owneds[rank], rp, rank, world)

    def one():
        rp.begin()
        return step(gx, d_reps)

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    gx.set_phase_timing(2)
    one()
    ph = {}
    for name, ms in gx.phase_times():
        ph[name] = ph.get(name, 0.0) + ms
    n_ev = sum(t.shape[0] + (0 if c is None else c.shape[0]) for t, c in d_reps)
    print(spec, "events", n_ev, "peaks", one()[0], f"{dt*1e3:.3f} ms", {k: round(v, 3) for k, v in ph.items()},
          "sum", round(sum(ph.values()), 3), flush=True)
    gx.close()
    del d_reps
