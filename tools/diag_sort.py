import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import backends as B, synth, genrich_amd
LENS = synth.HG38_LENS
ev = synth.make_fragments(LENS, 50_000_000, seed=1)
want = int((ev["end"].astype(np.int64) - ev["start"].astype(np.int64)).sum())
nvalid_e = int((ev["end"] < np.asarray(LENS, dtype=np.int64)[ev["chrom"]]).sum())
print("events", len(ev), "ends with a record", nvalid_e, "fragLen", want, flush=True)
gx = genrich_amd.Genrich(B.make_params(pq=0.01)); gx.set_chroms(LENS)
for it in range(8):
    try:
        gx.reset(); gx.sample_begin(0, None); gx.push_events(ev if it % 2 == 0 else ev[::-1].copy())
        frag, _, _ = gx.sample_end(); lam = gx.sample_no_control(); gx.pvalues(); n = gx.find_peaks()
        print("run", it, "ok fragLen", frag, frag == want, "peaks", n[0], flush=True)
    except Exception as e:
        print("run", it, "FAILED", e, flush=True)
