"""One GPU playing one rank of an N-rank run: what a rank's share of the step costs.

   python tools/emulate_ranks.py [--config 2|4] [N | N:rank] ...        (default config 2; 1 2 4 8, rank 0)

The fragLen all-reduce is a host callback that adds the other ranks' contribution, so lambda -- and with it the
sweep's work -- is the real run's; the callback's host round trip stands in for the RCCL all-reduce.  The other
ranks' contributions are measured first: every rank's share goes through the library once with a callback that
only records the three words the rank hands to the all-reduce (exact fixed-point parts of fragLen + flags).
Configs with -q (3, 5) also exchange the BH table and are not emulated here."""
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
assert not cfg["qval"] and not cfg["control"] and cfg["reps"] == 1, "configs 2 and 4 only"
lens = synth.HG38_LENS
ev_all = bench.build_workload(cfg, 50_000_000, lens)[0][0]
print(f"config {config}: {len(ev_all)} events", flush=True)
worlds = args or ["1", "2", "4", "8"]


def context(owned, callback, rank, world):
    gx = Genrich(GxParams(minus_log10f(0.01), 0, 200.0, 0, 100, 0, 0))
    gx.set_chroms(lens)
    gx.set_owned(owned)
    if world > 1:
        gx.set_collectives(rank, world, callback, lambda *a: 1)
    return gx


for spec in worlds:
    world, rank = (int(x) for x in (spec.split(":") + ["0"])[:2])
    owner = lpt_partition(lens, world)
    shares = []
    for r in range(world):
        owned = np.array([o == r for o in owner], dtype=np.uint8)
        shares.append((owned, ev_all[owned[ev_all["chrom"]].astype(bool)]))
    # what every rank contributes to the all-reduce
    contrib = []
    if world > 1:
        for r, (owned, mine) in enumerate(shares):
            got = []

            def record(buf, n, _user, got=got):
                got.append([int(buf[i]) for i in range(3)])
                return 0

            gx = context(owned, record, r, world)
            d = torch.from_numpy(mine.view(np.uint32).reshape(-1, 4).copy()).cuda()
            gx.reset(); gx.sample_begin(0, None); gx.push_events_device(d.data_ptr(), d.shape[0]); gx.sample_end()
            contrib.append(got[-1])
            gx.close()
            del d
    owned, mine = shares[rank]
    others = [sum(c[i] for r, c in enumerate(contrib) if r != rank) for i in range(3)] if world > 1 else [0, 0, 0]

    def allreduce(buf, n, _user, others=others):
        for i in range(3):
            buf[i] += others[i]
        return 0

    d_ev = torch.from_numpy(mine.view(np.uint32).reshape(-1, 4).copy()).cuda()
    gx = context(owned, allreduce, rank, world)

    def step():
        gx.reset(); gx.sample_begin(0, None); gx.push_events_device(d_ev.data_ptr(), d_ev.shape[0]); gx.sample_end()
        gx.sample_no_control(); gx.pvalues()
        return gx.find_peaks()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    gx.set_phase_timing(2)
    step()
    ph = dict(gx.phase_times())
    print(spec, "events", len(mine), "peaks", step()[0], f"{dt*1e3:.3f} ms", {k: round(v, 3) for k, v in ph.items()},
          "sum", round(sum(ph.values()), 3), flush=True)
    gx.close()
    del d_ev
