"""Host-side floor of one step: a tiny genome, so GPU work is negligible and what remains is API overhead."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from genrich_amd import synth
from genrich_amd.lib import GxParams, Genrich, minus_log10f
lens = [2_000_000, 1_000_000]
ev = synth.make_fragments(lens, 20_000, seed=1)
d_ev = torch.from_numpy(ev.view(np.uint32).reshape(-1, 4).copy()).cuda()
gx = Genrich(GxParams(minus_log10f(0.01), 0, 200.0, 0, 100, 0, 0))
gx.set_chroms(lens)
names = ["reset", "sample_begin", "push", "sample_end", "no_control", "pvalues", "find_peaks"]
acc = {k: 0.0 for k in names}
def step(rec):
    ts = [time.perf_counter()]
    gx.reset(); ts.append(time.perf_counter())
    gx.sample_begin(0, None); ts.append(time.perf_counter())
    gx.push_events_device(d_ev.data_ptr(), d_ev.shape[0]); ts.append(time.perf_counter())
    gx.sample_end(); ts.append(time.perf_counter())
    gx.sample_no_control(); ts.append(time.perf_counter())
    gx.pvalues(); ts.append(time.perf_counter())
    gx.find_peaks(); ts.append(time.perf_counter())
    if rec:
        for k, a, b in zip(names, ts[:-1], ts[1:]): acc[k] += b - a
for _ in range(5): step(False)
N = 200
t0 = time.perf_counter()
for _ in range(N): step(True)
dt = (time.perf_counter() - t0) / N
print(f"step {dt*1e6:.0f} us", {k: round(v / N * 1e6) for k, v in acc.items()})
