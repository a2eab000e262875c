"""Synthetic fragment streams live in the package (bench.py uses them too)."""
from genrich_amd.synth import *  # noqa: F401,F403
from genrich_amd.synth import EVENT_DTYPE, HG38_LENS, HG38_NAMES  # noqa: F401
