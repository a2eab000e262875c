"""Loader for the committed golden fixtures (tests/golden/<case>/, produced by
tests/golden/make_golden.py from the reference binary)."""
from __future__ import annotations

import gzip
import json
import os

import numpy as np

from backends import EVENT_DTYPE, make_params

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names():
    return sorted(d for d in os.listdir(GOLDEN) if os.path.exists(os.path.join(GOLDEN, d, "case.json")))


def read_gz(case, fn):
    p = os.path.join(GOLDEN, case, fn + ".gz")
    if not os.path.exists(p):
        return None
    with gzip.open(p, "rb") as f:
        return f.read()


def load_case(name):
    """-> (meta, case dict for backends.run_case, params, names)"""
    meta = json.load(open(os.path.join(GOLDEN, name, "case.json")))
    names = [c["name"] for c in meta["chroms"]]
    idx = {n: i for i, n in enumerate(names)}
    nrep = len(meta["replicates"])
    rows = {(r, k): [] for r in range(nrep) for k in "EC"}
    for line in read_gz(name, "events.bed").decode().splitlines():
        c, s, e, nm = line.split("\t")
        _, cnt, kind, smp = nm.rsplit("_", 3)
        rows[(int(smp), kind)].append((idx[c], int(s), int(e), int(cnt)))
    reps = []
    for r, rm in enumerate(meta["replicates"]):
        reps.append(dict(
            save=rm["save"],
            treat=np.array(rows[(r, "E")], dtype=EVENT_DTYPE),
            ctrl=np.array(rows[(r, "C")], dtype=EVENT_DTYPE) if rm["control"] == "file" else None,
        ))
    a = meta["args"]

    def opt(flag, default, conv=float):
        return conv(a[a.index(flag) + 1]) if flag in a else default

    qval = "-q" in a
    params = make_params(
        pq=opt("-q", 0.0) if qval else opt("-p", 0.01), qval=qval,
        min_auc=opt("-a", 200.0), min_len=opt("-l", 0, int), max_gap=opt("-g", 100, int),
        genome_len=opt("-L", 0, int))
    case = dict(
        lens=[c["len"] for c in meta["chroms"]],
        skip=[c["skip"] for c in meta["chroms"]],
        beds=[c["bed"] for c in meta["chroms"]],
        replicates=reps,
    )
    return meta, case, params, names
