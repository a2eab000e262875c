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

    # int (*)(const void* local, size_t n_local, void** out, size_t* n_out, void* user)
    # records are 16 bytes {u32 key, u32 pad, u64 bp}; *out is malloc'd (freed by the library)
    def allgather_tab(self, local, n_local, out, n_out, _user):
        try:
            torch, dist = self.torch, self.dist
            world = dist.get_world_size()
            cnt = torch.tensor([n_local], dtype=torch.int64, device=self.device)
            cnts = [torch.zeros_like(cnt) for _ in range(world)]
            dist.all_gather(cnts, cnt)
            cnts = [int(c.item()) for c in cnts]
            mx = max(cnts + [1])
            mine = np.zeros((mx, 2), dtype=np.int64)
            if n_local:
                src = np.frombuffer(C.string_at(local, n_local * 16), dtype=np.int64).reshape(-1, 2)
                mine[:n_local] = src
            t = torch.from_numpy(mine).to(self.device)
            parts = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            cat = np.concatenate([p.cpu().numpy()[:c] for p, c in zip(parts, cnts)], axis=0)
            total = cat.shape[0]
            libc = C.CDLL(None)
            libc.malloc.restype = C.c_void_p
            libc.malloc.argtypes = [C.c_size_t]
            mem = libc.malloc(max(16, total * 16))
            C.memmove(mem, cat.ctypes.data, total * 16)
            out[0] = mem
            n_out[0] = total
            return 0
        except Exception as e:
            print("allgather callback failed:", e)
            return 1


def merge_peaks(per_rank_peaks):
    """Peaks of all ranks -> chromosome-table order then position (the order in which the
    reference numbers them, Genrich.c:986, 925)."""
    allp = np.concatenate([p for p in per_rank_peaks if len(p)]) if any(len(p) for p in per_rank_peaks) \
        else per_rank_peaks[0][:0]
    order = np.lexsort((allp["start"], allp["chrom"]))
    return allp[order]
