"""One GPU playing rank 0 of an N-rank run (no collectives): what a rank's share of the step costs."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from genrich_amd import synth
from genrich_amd.dist import lpt_partition
from genrich_amd.lib import GxParams, Genrich, minus_log10f
lens = synth.HG38_LENS
ev_all = synth.make_fragments(lens, 50_000_000, seed=1)
worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
for world in worlds:
    owner = lpt_partition(lens, world)
    owned = np.array([o == 0 for o in owner], dtype=np.uint8)
    mine = ev_all[owned[ev_all["chrom"]].astype(bool)]
    d_ev = torch.from_numpy(mine.view(np.uint32).reshape(-1, 4).copy()).cuda()
    gx = Genrich(GxParams(minus_log10f(0.01), 0, 200.0, 0, 100, 0, 0))
    gx.set_chroms(lens)
    gx.set_owned(owned)
    def step():
        gx.reset(); gx.sample_begin(0, None); gx.push_events_device(d_ev.data_ptr(), d_ev.shape[0]); gx.sample_end(); gx.sample_no_control(); gx.pvalues(); return gx.find_peaks()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    gx.set_phase_timing(2); step()
    ph = dict(gx.phase_times())
    print(world, f"{dt*1e3:.3f} ms", {k: round(v, 3) for k, v in ph.items()}, "sum", round(sum(ph.values()), 3), flush=True)
