"""Multi-GPU host logic: chromosome sharding and the three tiny genome-wide exchanges
(SURVEY.md section 8e).  One process per GPU; `torch.distributed` (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests) carries the collectives.  The hot path itself
never needs a bulk exchange: chromosomes are independent (Genrich.c:2172, 1729, 987).
"""
from __future__ import annotations

import ctypes as C

import numpy as np


def lpt_partition(lens, world):
    """Longest-processing-time bin packing of chromosomes by length.
    Returns owner[i] in [0, world) for every chromosome (deterministic on every rank)."""
    order = sorted(range(len(lens)), key=lambda i: (-int(lens[i]), i))
    load = [0] * world
    owner = [0] * len(lens)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += int(lens[i])
    return owner


class Collectives:
    """ctypes callbacks for gx_set_collectives built on torch.distributed."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.device = device if device is not None else "cpu"

    # int (*)(int64_t* buf, size_t n, void* user): sum over ranks, in place
    def allreduce_i64(self, buf, n, _user):
        try:
            a = np.ctypeslib.as_array(buf, shape=(n,))
            t = self.torch.from_numpy(a.copy()).to(self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            a[:] = t.cpu().numpy()
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("allreduce callback failed:", e)
            return 1


def merge_peaks(per_rank_peaks):
    """Peaks of all ranks -> chromosome-table order then position (the order in which the
    reference numbers them, Genrich.c:986, 925)."""
    allp = np.concatenate([p for p in per_rank_peaks if len(p)]) if any(len(p) for p in per_rank_peaks) \
        else per_rank_peaks[0][:0]
    order = np.lexsort((allp["start"], allp["chrom"]))
    return allp[order]
