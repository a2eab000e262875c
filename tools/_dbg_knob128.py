import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import backends as B, synth
import genrich_amd
lens = [400_000, 123_457, 16_384, 4_097, 5]
for seed, n, pile in ((5, 150_000, 0), (11, 90_000, 0), (5, 150_000, 1)):
    ev = synth.make_fragments(lens[:4], n, seed, peak_every=20_000, tower_every=150_000)
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=None)])
    par = B.make_params(pq=0.01, min_auc=20.0)
    o = B.Oracle(par); B.run_case(o, case)
    h = genrich_amd.Genrich(par); B.run_case(h, case)
    same = o.get_peaks().tobytes() == h.get_peaks().tobytes()
    ivok = True
    for c in range(len(lens)):
        eo, co = o.get_intervals(-1, c); eh, ch = h.get_intervals(-1, c)
        ivok &= np.array_equal(eo, eh) and np.array_equal(co["p"].view(np.uint32), ch["p"].view(np.uint32))
    print(os.environ.get("GENRICH_AMD_LIB", "default"), "seed", seed, "n", n, "flags", h.path_info(), "peaks same", same, "intervals same", ivok, flush=True)
