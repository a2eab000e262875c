"""CPU: the host-side saturation filter of the library (gx_filter_saturation, gx_saturate.h) takes the
same decisions as the oracle's restatement of saveInterval's int16 checks (Genrich.c:2558-2573), which
tools/fuzz_oracle_vs_reference.py --saturate pins against the reference binary."""
import ctypes as C

import numpy as np
import pytest

import backends as B
from genrich_amd import lib as L
from test_hip_parity import _saturating_case
import synth


def _oracle(ev, lens):
    o = B.Oracle(B.make_params(pq=0.01, min_auc=20.0))
    o.set_chroms(lens)
    o.sample_begin(0, None)
    o.push_events(ev)
    frag = o.sample_end()[0]
    o.lib.gxo_skipped_overflow.restype = C.c_uint64
    o.lib.gxo_skipped_overflow.argtypes = [C.c_void_p]
    skipped = o.lib.gxo_skipped_overflow(o.ctx)
    o.sample_no_control()
    o.pvalues()
    o.find_peaks()
    return frag, skipped, [o.get_intervals(-1, c) for c in range(len(lens))], o.get_peaks()


@pytest.mark.parametrize("seed", range(4))
def test_filter_takes_the_oracles_decisions(seed):
    lens, ev = _saturating_case(seed, frac=bool(seed % 2))
    keep, dropped = L.filter_saturation(ev, lens)
    f1, s1, iv1, pk1 = _oracle(ev, lens)
    f2, s2, iv2, pk2 = _oracle(ev[keep == 1], lens)
    assert dropped == s1 > 0 and s2 == 0  # what is left saturates nothing
    assert f1 == f2 and np.array_equal(pk1, pk2)
    for a, b in zip(iv1, iv2):
        assert np.array_equal(a[0], b[0])
        assert np.array_equal(a[1]["expt"].view(np.uint32), b[1]["expt"].view(np.uint32))


def test_filter_keeps_everything_on_ordinary_input():
    lens = [300_000, 100_000]
    ev = synth.make_fragments(lens, 50_000, seed=9, frac_tower=0.3)
    keep, dropped = L.filter_saturation(ev, lens)
    assert dropped == 0 and keep.all()
