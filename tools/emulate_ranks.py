"""One GPU playing one rank of an N-rank run: what a rank's share of the step costs.  The fragLen all-reduce is
   a host callback that adds the other ranks' (known) contribution, so lambda -- and with it the sweep's work --
   is the real run's; the callback's host round trip stands in for the RCCL all-reduce.
   python tools/emulate_ranks.py [N | N:rank] ...        (default 1 2 4 8, rank 0)"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from genrich_amd import synth
from genrich_amd.dist import lpt_partition
from genrich_amd.lib import GxParams, Genrich, minus_log10f
lens = synth.HG38_LENS
ev_all = synth.make_fragments(lens, 50_000_000, seed=1)
worlds = [a for a in sys.argv[1:]] or ["1", "2", "4", "8"]
for spec in worlds:
    world, rank = (int(x) for x in (spec.split(":") + ["0"])[:2])
    owner = lpt_partition(lens, world)
    owned = np.array([o == rank for o in owner], dtype=np.uint8)
    mine = ev_all[owned[ev_all["chrom"]].astype(bool)]
    d_ev = torch.from_numpy(mine.view(np.uint32).reshape(-1, 4).copy()).cuda()
    gx = Genrich(GxParams(minus_log10f(0.01), 0, 200.0, 0, 100, 0, 0))
    gx.set_chroms(lens)
    gx.set_owned(owned)
    if world > 1:
        clamp = np.minimum(ev_all["end"].astype(np.int64), np.asarray(lens, dtype=np.int64)[ev_all["chrom"]])
        cov = clamp - ev_all["start"].astype(np.int64)
        others = int(cov.sum() - cov[owned[ev_all["chrom"]].astype(bool)].sum())
        def allreduce(buf, n, _user, others=others):
            buf[0] += others
            return 0
        gx.set_collectives(rank, world, allreduce, lambda *a: 1)
    def step():
        gx.reset(); gx.sample_begin(0, None); gx.push_events_device(d_ev.data_ptr(), d_ev.shape[0]); gx.sample_end(); gx.sample_no_control(); gx.pvalues(); return gx.find_peaks()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    gx.set_phase_timing(2); step()
    ph = dict(gx.phase_times())
    print(spec, "peaks", step()[0], f"{dt*1e3:.3f} ms", {k: round(v, 3) for k, v in ph.items()}, "sum", round(sum(ph.values()), 3), flush=True)
