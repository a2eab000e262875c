"""One GPU playing one rank of an N-rank run: what a rank's share of the step costs.

   python tools/emulate_ranks.py [--config 2|3|4|5] [N | N:rank] ...        (default config 2; 1 2 4 8, rank 0)

The collectives are host callbacks that add / append what the OTHER ranks would contribute, so lambda, the factor and
the q-values -- and with them the sweep's work -- are the real run's; the callbacks' host round trips stand in for
RCCL.  What the other ranks contribute is measured first, in two passes over every rank's share:
  pass 1  the all-reduces (fragLen / ctrlFrag: exact fixed-point parts + flags) only depend on a rank's own events:
          every rank's words are recorded, call by call;
  pass 2  with the true sums replayed, every rank's part of the BH table (-q: the all-gather of {p bits, bp}) is
          recorded -- it depends on lambda, hence on pass 1.
Then the chosen rank is timed with both kinds of callback replaying the others' parts."""
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
    """Collective callbacks of one rank: record its own contributions, add / append the others' (by call index)."""

    def __init__(self, red_others=None, gat_others=None):
        self.red_others, self.gat_others = red_others, gat_others
        self.red, self.gat = [], []
        self.ri = self.gi = 0

    def begin(self):
        self.red, self.gat, self.ri, self.gi = [], [], 0, 0

    def allreduce(self, buf, n, _user):
        self.red.append([int(buf[i]) for i in range(n)])
        if self.red_others is not None:
            for i in range(n):
                buf[i] += self.red_others[self.ri][i]
        self.ri += 1
        return 0

    def allgather(self, local, n_local, out, n_out, _user):
        mine = C.string_at(local, n_local * 16) if n_local else b""
        self.gat.append(mine)
        cat = mine + (self.gat_others[self.gi] if self.gat_others is not None else b"")
        self.gi += 1
        mem = libc.malloc(max(16, len(cat)))
        C.memmove(mem, cat, len(cat))
        out[0] = mem
        n_out[0] = len(cat) // 16
        return 0


def context(owned, rp, rank, world):
    gx = Genrich(params())
    gx.set_chroms(lens)
    gx.set_owned(owned)
    if world > 1:
        gx.set_collectives(rank, world, rp.allreduce, rp.allgather)
    return gx


def others_sum(recs, rank):
    """per call: the sum over the other ranks of the words they handed to the all-reduce"""
    calls = len(recs[0])
    return [[sum(recs[r][k][i] for r in range(len(recs)) if r != rank) for i in range(len(recs[0][k]))] for k in range(calls)]


def others_cat(recs, rank):
    calls = len(recs[0])
    return [b"".join(recs[r][k] for r in range(len(recs)) if r != rank) for k in range(calls)]


for spec in worlds:
    world, rank = (int(x) for x in (spec.split(":") + ["0"])[:2])
    owner = lpt_partition(lens, world)
    owneds = [np.array([o == r for o in owner], dtype=np.uint8) for r in range(world)]
    red, gat = [None] * world, [None] * world
    if world > 1:
        for r in range(world):  # pass 1: the all-reduce words
            rp = Replay()
            gx = context(owneds[r], rp, r, world)
            d = share(owneds[r])
            rp.begin()
            step(gx, d)
            red[r] = rp.red
            gx.close()
            del d
        if cfg["qval"]:
            for r in range(world):  # pass 2: the BH tables, under the true lambda
                rp = Replay(others_sum(red, r))
                gx = context(owneds[r], rp, r, world)
                d = share(owneds[r])
                rp.begin()
                step(gx, d)
                gat[r] = rp.gat
                gx.close()
                del d
    rp = Replay(others_sum(red, rank) if world > 1 else None, others_cat(gat, rank) if world > 1 and cfg["qval"] else None)
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
