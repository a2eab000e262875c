"""GPU diagnostic: where do device p-values differ from the host's, and by how much in double?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import backends as B
import synth
import genrich_amd
from genrich_amd.lib import selftest_host

h = genrich_amd.Genrich(B.make_params())
rng = np.random.default_rng(2)
n = 400_000
expt = (rng.integers(0, 400_000, n) / 120.0).astype(np.float32)
ctrl = np.where(rng.random(n) < 0.5, rng.random(n) * 7.5, rng.random(n) * 300).astype(np.float32)
ctrl[:10] = [0, -1, 7, 7.0000005, 1e-30, 6.9999995, 3, 3, 3, 3]
expt[:10] = [5, 5, 0, 1, 1, 1, 0, 1e6, 3e38, 1e-3]
got, dd, nrisky = h.selftest2(1, expt, ctrl)
hw, hd = selftest_host(1, expt, ctrl)
bad = np.flatnonzero(got.view(np.uint32) != hw.view(np.uint32))
print("risky", nrisky, "of", n, "; mismatching floats:", len(bad))
ok = (hd > 0) & (hd < 1e30)
rel = np.zeros(n); rel[ok] = np.abs(dd[ok] - hd[ok]) / hd[ok]
print("max rel double diff", rel.max(), "at", int(rel.argmax()), expt[rel.argmax()], ctrl[rel.argmax()], dd[rel.argmax()], hd[rel.argmax()])
for q in (50, 90, 99, 99.9, 99.99):
    print(" percentile", q, np.percentile(rel[ok], q))
order = np.argsort(-rel)[:15]
for i in order:
    print(f"  expt {expt[i]!r} ctrl {ctrl[i]!r} dev {dd[i]!r} host {hd[i]!r} rel {rel[i]:.3g} float dev {got[i]!r} host {hw[i]!r}")
print("mismatches:")
for i in bad[:20]:
    print(f"  expt {expt[i]!r} ctrl {ctrl[i]!r} dev {dd[i]!r} host {hd[i]!r} rel {rel[i]:.3g} float dev {got[i]!r} host {hw[i]!r}")

# the control case that failed
lens = [300_000, 9_000_000]
bg = synth.make_fragments(lens[:1], 4000, seed=3)
deep = np.array([(0, 100_000, 100_150, 1)] * 1500 + [(1, 8_000_000, 8_000_999, 1)] * 20_001, dtype=B.EVENT_DTYPE)
tr = np.concatenate([bg, deep])
case = dict(lens=lens, replicates=[dict(save=None, treat=tr, ctrl=bg)])
par = B.make_params(pq=0.01, min_auc=20.0)
o = B.Oracle(par); so = B.run_case(o, case)
g = genrich_amd.Genrich(par); sg = B.run_case(g, case)
print(so, sg)
for c in range(2):
    eo, co = o.get_intervals(-1, c)
    eh, ch = g.get_intervals(-1, c)
    bad = np.flatnonzero(co["p"].view(np.uint32) != ch["p"].view(np.uint32))
    print("chrom", c, "intervals", len(eo), "p mismatches", len(bad))
    for i in bad[:12]:
        print(f"   end {eo[i]} expt {co['expt'][i]!r} ctrl {co['ctrl'][i]!r} p oracle {co['p'][i]!r} hip {ch['p'][i]!r}")
    if len(bad):
        e, c_ = co["expt"][bad], co["ctrl"][bad]
        hw, hd = selftest_host(1, e, c_)
        dw, dd2, nr = g.selftest2(1, e, c_)
        print("   re-evaluated: host==oracle", np.array_equal(hw.view(np.uint32), co["p"][bad].view(np.uint32)),
              "device selftest==oracle", int((dw.view(np.uint32) == co["p"][bad].view(np.uint32)).sum()), "of", len(bad), "risky", nr)
