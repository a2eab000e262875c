"""GPU: the margin behind RISK_B (gx_math.h), swept over the REACHABLE domain instead of sampled.

calcPval (Genrich.c:1628-1653) is double math rounded once to float.  The device evaluates the same IEEE operations
around OCML's log / exp / log1p, the host (and the reference) around glibc's; a result is only handed to the host for
re-evaluation when it lies within RISK_B = 2^-38 (relative) of a float rounding boundary.  That is sound as long as the
two doubles never differ by more than RISK_B relative.  This sweep evaluates, on the device and with the host build of
the same routines:
  A  no control:  every exact pileup V in [0, 2^18) (1/120 units: everything the table p(V) holds) x a few hundred lambda
  B  a control:   the 256 x 256 table of whole (treatment, control) pileups x a grid of (factor, lambda)
and reports max |device - host| / |host| of the doubles in units of RISK_B, the number of results flagged risky, and the
number of floats that differ after the library's own re-evaluation (must be 0).

  python tools/sweep_risk_margin.py [OUT.txt]          (on the GPU box; ~1 min)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import backends as B  # noqa: E402
import genrich_amd  # noqa: E402
from genrich_amd.lib import selftest_host  # noqa: E402

RISK_B = 2.0 ** -38


def sweep(h, expt, ctrl, tag, out):
    t0 = time.time()
    got, dd, nrisky = h.selftest2(1, expt, ctrl)
    hw, hd = selftest_host(1, expt, ctrl)
    ok = (hd != 0) & np.isfinite(hd) & (np.abs(hd) < 1e300) & (expt > 0) & (ctrl > 0)
    rel = np.zeros(len(expt))
    rel[ok] = np.abs(dd[ok] - hd[ok]) / np.abs(hd[ok])
    # only values that are not zero as a float can change a float (z^2 / 2 <= 103: gx_math.h)
    live = ok & (hw != 0) & (np.abs(hw) < 3e38)
    worst = int(np.argmax(np.where(live, rel, 0)))
    nbad = int((got.view(np.uint32) != hw.view(np.uint32)).sum())
    line = (f"{tag}: {len(expt)} pairs, {int(live.sum())} with a non-zero finite float result; max |dev - host| / |host| = "
            f"{rel[live].max() if live.any() else 0:.3e} = {(rel[live].max() if live.any() else 0) / RISK_B:.4f} x RISK_B "
            f"(at expt {expt[worst]!r} ctrl {ctrl[worst]!r}: dev {dd[worst]!r} host {hd[worst]!r}); p99.99 "
            f"{np.percentile(rel[live], 99.99) if live.any() else 0:.2e}; flagged risky {nrisky} ({nrisky / max(1, len(expt)):.2e}); "
            f"floats differing after re-evaluation: {nbad}; {time.time() - t0:.1f} s")
    print(line, flush=True)
    out.append(line)
    return (rel[live].max() if live.any() else 0.0), nbad


def main():
    out = []
    h = genrich_amd.Genrich(B.make_params())
    o = B.Oracle.lib()
    V = np.arange(1 << 18, dtype=np.int64)
    ng = B.C.c_int(0)
    pile = np.array([o.gxo_getval(int(v), B.C.byref(ng)) for v in V], dtype=np.float32)   # getVal of every table entry
    # lambda: what fragLen / genomeLen gives for 10^5 .. 10^10 covered bases on 10^4 .. 3 x 10^9 bp, on both sides of the
    # mu > 7 switch of calcPval, plus the values next to it
    lam = np.unique(np.concatenate([np.geomspace(1e-4, 2000.0, 280), np.linspace(6.5, 7.5, 21), [7.0, np.nextafter(np.float32(7), np.float32(8))],
                                    [3.221889, 19.950832, 0.25, 1.0, 100.0]]).astype(np.float32))
    worst = 0.0
    bad = 0
    for i in range(0, len(lam), 16):
        ls = lam[i:i + 16]
        e = np.tile(pile, len(ls))
        c = np.repeat(ls, len(pile))
        w, b = sweep(h, e, c, f"A lambda[{i}:{i + len(ls)}] ({ls[0]:.4g} .. {ls[-1]:.4g}) x V in [0, 2^18)", out)
        worst, bad = max(worst, w), bad + b
    # B: whole pileups 0 .. 255 against max(factor * c, lambda) (savePileupCtrl 2107-2109) for c in 0 .. 255
    fac = np.unique(np.concatenate([np.geomspace(0.05, 40.0, 24), [1.0, 1.000008]]).astype(np.float32))
    lamB = np.array([0.1, 0.5, 1.0, 3.221889, 6.9999995, 7.0000005, 20.0], dtype=np.float32)
    tv = np.arange(256, dtype=np.float32)
    for f in fac:
        es, cs = [], []
        for lb in lamB:
            net = np.maximum((f * tv).astype(np.float32), lb)   # float product, as the reference's
            es.append(np.repeat(tv, 256))
            cs.append(np.tile(net, 256))
        w, b = sweep(h, np.concatenate(es), np.concatenate(cs), f"B factor {f:.6g} x {len(lamB)} lambda x 256 x 256", out)
        worst, bad = max(worst, w), bad + b
    tail = (f"OVERALL: max double-level difference {worst:.3e} = {worst / RISK_B:.4f} x RISK_B (RISK_B = 2^-38 = {RISK_B:.3e}); "
            f"floats differing after re-evaluation: {bad}")
    print(tail)
    out.append(tail)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write("\n".join(out) + "\n")
    sys.exit(0 if bad == 0 and worst < RISK_B else 1)


if __name__ == "__main__":
    main()
