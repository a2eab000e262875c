"""genrich_amd -- MI355X-native Genrich hot path (events -> pileup -> p -> q -> peaks).

The product is the C-ABI shared library `libgenrich_amd.so` (include/genrich_amd.h) built
from genrich_amd/csrc/*.hip; this package is the thin Python host mirror used by the tests
and bench.py.  It never falls back to a CPU implementation: loading fails loudly when the
HIP library is missing.
"""
from .lib import GxParams, Genrich, EVENT_DTYPE, PEAK_DTYPE, load_library, minus_log10f  # noqa: F401
