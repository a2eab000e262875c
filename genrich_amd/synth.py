"""Deterministic synthetic fragment streams shaped like the workloads of BASELINE.json.

Used by the tests, by tests/golden/make_golden.py (which also renders them as SAM text
for the reference binary) and by bench.py.  Pure numpy; the same seed always gives the
same stream.
"""
from __future__ import annotations

import numpy as np

# hg38 primary contigs, chr1..22, X, Y, M
HG38_LENS = [
    248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973,
    145138636, 138394717, 133797422, 135086622, 133275309, 114364328, 107043718,
    101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983,
    50818468, 156040895, 57227415, 16569,
]
HG38_NAMES = [f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY", "chrM"]

EVENT_DTYPE = np.dtype(
    [("chrom", "<u4"), ("start", "<u4"), ("end", "<u4"), ("count", "<u4")]
)


def make_fragments(
    lens,
    n_frag,
    seed,
    peak_every=50_000,
    tower_every=5_000_000,
    frac_peak=0.31,
    frac_tower=0.02,
    uniform_only=False,
    min_len=100,
):
    """Fragments (chrom, start, end) on chromosomes of the given lengths.

    Shape follows SURVEY.md 8(d): chromosome chosen in proportion to its length;
    fragment length = min_len + U[0,100) + U[0,100) (triangular); 67 % uniform positions,
    31 % within +-150 bp of a peak centre every `peak_every` bp, 2 % in towers every
    `tower_every` bp.  Returns a structured array with count = 1.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.asarray(lens, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(lens)])
    g = rng.integers(0, cum[-1], size=n_frag, dtype=np.int64)
    chrom = np.searchsorted(cum, g, side="right") - 1
    pos = g - cum[chrom]
    flen = min_len + rng.integers(0, 100, size=n_frag) + rng.integers(0, 100, size=n_frag)
    if not uniform_only:
        u = rng.random(n_frag)
        jitter = rng.integers(-150, 151, size=n_frag)
        is_peak = u < frac_peak
        is_tower = (u >= frac_peak) & (u < frac_peak + frac_tower)
        centre_p = (pos // peak_every) * peak_every + peak_every // 2
        centre_t = (pos // tower_every) * tower_every + tower_every // 2
        pos = np.where(is_peak, centre_p + jitter - flen // 2, pos)
        pos = np.where(is_tower, centre_t + jitter // 3 - flen // 2, pos)
    clen = lens[chrom]
    flen = np.minimum(flen, clen)  # tiny chromosomes
    pos = np.clip(pos, 0, clen - flen)
    ev = np.empty(n_frag, dtype=EVENT_DTYPE)
    ev["chrom"] = chrom
    ev["start"] = pos
    ev["end"] = pos + flen
    ev["count"] = 1
    return ev


def add_multimap(ev, lens, frac, seed):
    """Replace a fraction of fragments by k equally-scored copies (k in 2,3,4,5,6,8,10) at
    random loci with weight 1/k (the -s fractional-pileup path, Genrich.c:3122-3176)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.asarray(lens, dtype=np.int64)
    n = len(ev)
    pick = rng.random(n) < frac
    keep = ev[~pick]
    multi = ev[pick]
    ks = rng.choice(np.array([2, 3, 4, 5, 6, 8, 10]), size=len(multi))
    rep = np.repeat(np.arange(len(multi)), ks)
    out = multi[rep].copy()
    out["count"] = np.repeat(ks, ks)
    flen = (out["end"] - out["start"]).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(lens)])
    g = rng.integers(0, cum[-1], size=len(out), dtype=np.int64)
    chrom = np.searchsorted(cum, g, side="right") - 1
    pos = g - cum[chrom]
    clen = lens[chrom]
    flen = np.minimum(flen, clen)
    pos = np.clip(pos, 0, clen - flen)
    # the first copy of each read stays where it was
    first = np.concatenate([[True], rep[1:] != rep[:-1]])
    out["chrom"] = np.where(first, out["chrom"], chrom)
    out["start"] = np.where(first, out["start"], pos)
    out["end"] = np.where(first, out["end"], pos + flen)
    fix = out["end"] > lens[out["chrom"]]
    out["end"] = np.where(fix, lens[out["chrom"]], out["end"])
    return np.concatenate([keep, out])


def excluded_regions(lens, n=800, seed=7, skip=()):
    """A deterministic -E file for a genome: `n` N-gap-like regions (most a few kilobases, some hundreds of kilobases, a few
    centromere-sized), one that touches position 0 of the first chromosome and one that reaches the end of the second --
    the two edge cases of saveXBed / savePileupExpt's bedPos walk (Genrich.c:1144-1206, 2185-2263) -- none on the chromosomes
    listed in `skip` (-e).  Returns, per chromosome, the merged and sorted coordinates [s0, e0, s1, e1, ...] as saveXBed
    leaves them (what gx_set_chroms takes)."""
    rng = np.random.default_rng(seed)
    lens = [int(x) for x in lens]
    regs = [[] for _ in lens]
    ok = [i for i, L in enumerate(lens) if i not in set(skip) and L > 200_000]
    w = np.array([lens[i] for i in ok], dtype=np.float64)
    where = rng.choice(len(ok), size=n, p=w / w.sum())
    kind = rng.random(n)
    for c, k in zip(where, kind):
        c = ok[int(c)]
        size = int(rng.integers(1_000, 50_000)) if k < 0.85 else int(rng.integers(50_000, 500_000)) if k < 0.98 else int(rng.integers(1_000_000, 3_000_000))
        size = min(size, lens[c] // 4)
        s = int(rng.integers(0, lens[c] - size))
        regs[c].append((s, s + size))
    if ok:
        regs[ok[0]].append((0, 10_000))                                   # touches position 0
        last = ok[1] if len(ok) > 1 else ok[0]
        regs[last].append((lens[last] - 60_000, lens[last]))              # reaches the chromosome's end
    out = []
    for r in regs:
        r.sort()
        merged = []
        for a, b in r:
            if merged and a <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], b)
            else:
                merged.append([a, b])
        out.append([v for m in merged for v in m])
    return out


def atac_events(ev, lens, d=100, adj=True):
    """ATAC-seq mode geometry (saveFragAtac, Genrich.c:2728-2749; -d halves 5796-5797):
    each fragment becomes one or two cut-site intervals, clamped like saveInterval."""
    lens = np.asarray(lens, dtype=np.int64)
    len3 = int(np.float32(d) / np.float32(2.0) + np.float32(0.5))
    len5 = d // 2
    s = ev["start"].astype(np.int64)
    e = ev["end"].astype(np.int64)
    if adj:
        s = s + 5
        e = e - 5
    one = (s + len3) >= (e - len3)
    clen = lens[ev["chrom"]]

    def clamp(a, b, c):
        a = np.maximum(a, 0)
        b = np.minimum(b, c)
        return a, b

    a1, b1 = clamp(s - len5, np.where(one, e + len5, s + len3), clen)
    a2, b2 = clamp(e - len3, e + len5, clen)
    first = np.empty(len(ev), dtype=EVENT_DTYPE)
    first["chrom"], first["start"], first["end"], first["count"] = ev["chrom"], a1, b1, ev["count"]
    second = np.empty(int((~one).sum()), dtype=EVENT_DTYPE)
    m = ~one
    second["chrom"], second["start"], second["end"], second["count"] = (
        ev["chrom"][m], a2[m], b2[m], ev["count"][m])
    out = np.concatenate([first, second])
    return out[out["start"] < lens[out["chrom"]]]


def write_sam(path, names, lens, ev, read_len=50, name_prefix="r"):
    """Render fragments as queryname-grouped paired SAM records (flags 99/147, +256 for the
    2nd..kth alignment of a multimapped read; consecutive events with count = k > 1 are the
    k alignments of one read, as add_multimap lays them out)."""
    with open(path, "w") as f:
        f.write("@HD\tVN:1.0\tSO:queryname\n")
        for n, l in zip(names, lens):
            f.write(f"@SQ\tSN:{n}\tLN:{l}\n")
        i = 0
        rid = 0
        n = len(ev)
        while i < n:
            k = int(ev["count"][i])
            for a in range(k):
                c, s, e = int(ev["chrom"][i + a]), int(ev["start"][i + a]), int(ev["end"][i + a])
                rl = min(read_len, e - s)
                sec = 256 if a else 0
                p1, p2 = s + 1, e - rl + 1
                nm = f"{name_prefix}{rid}"
                f.write(f"{nm}\t{99 + sec}\t{names[c]}\t{p1}\t30\t{rl}M\t=\t{p2}\t{e - s}\t*\t*\tAS:i:0\n")
                f.write(f"{nm}\t{147 + sec}\t{names[c]}\t{p2}\t30\t{rl}M\t=\t{p1}\t{-(e - s)}\t*\t*\tAS:i:0\n")
            i += k
            rid += 1


def write_sam_mixed(path, names, lens, ev, seed, read_len=50, name_prefix="m", frac_single=0.3, bam=False):
    """Queryname-grouped SAM (or BAM when bam=True) with a mix of proper pairs and unpaired
    alignments (forward / reverse), varying MAPQ, a few unmapped, supplementary and secondary
    records, soft clips and deletions in the CIGAR: exercises -y / -w / -x / -m and the record
    filters of readSAM / parseBAM (Genrich.c:4531-4559, 4891-4911)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    recs = []  # (qname, flag, chrom, pos0, mapq, cigar ops [(n, op)], pnext0, tlen, AS or None)
    for i in range(len(ev)):
        c, s, e = int(ev["chrom"][i]), int(ev["start"][i]), int(ev["end"][i])
        nm = f"{name_prefix}{i}"
        rl = min(read_len, e - s)
        mapq = int(rng.integers(0, 45))
        u = rng.random()
        if u < frac_single:
            rev = rng.random() < 0.5
            ops = [(rl, "M")]
            if rng.random() < 0.3 and rl > 20:
                ops = [(5, "S"), (rl - 15, "M"), (3, "D"), (10, "M")]
            pos = s if not rev else max(0, e - rl)
            flag = (16 if rev else 0) | (64 if rng.random() < 0.5 else 128 if rng.random() < 0.5 else 0)
            if flag & 0xC0:
                flag |= 1 | 8  # paired in sequencing, mate unmapped
            recs.append((nm, flag, c, pos, mapq, ops, -1, 0, int(rng.integers(-30, 1))))
        else:
            p1, p2 = s, e - rl
            recs.append((nm, 99, c, p1, mapq, [(rl, "M")], p2, e - s, 0))
            recs.append((nm, 147, c, p2, mapq, [(rl, "M")], p1, -(e - s), 0))
        if rng.random() < 0.03:
            recs.append((nm + "u", 4, 0, 0, 0, [], -1, 0, None))        # unmapped
        if rng.random() < 0.03:
            recs.append((nm + "s", 2048, c, s, 30, [(rl, "M")], -1, 0, 0))  # supplementary
    if not bam:
        with open(path, "w") as f:
            f.write("@HD\tVN:1.0\tSO:queryname\n")
            for n, l in zip(names, lens):
                f.write(f"@SQ\tSN:{n}\tLN:{l}\n")
            for nm, flag, c, pos, mapq, ops, pn, tlen, sc in recs:
                cigar = "".join(f"{n}{o}" for n, o in ops) or "*"
                rn = "*" if flag & 4 else names[c]
                extra = f"\tNM:i:0\tAS:i:{sc}" if sc is not None else ""
                f.write(f"{nm}\t{flag}\t{rn}\t{0 if flag & 4 else pos + 1}\t{mapq}\t{cigar}\t"
                        f"{'=' if pn >= 0 else '*'}\t{pn + 1}\t{tlen}\t*\t*{extra}\n")
        return
    import gzip
    import struct
    raw = bytearray()
    text = "@HD\tVN:1.0\tSO:queryname\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in zip(names, lens))
    raw += b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(names))
    for n, l in zip(names, lens):
        raw += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    opcode = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "P": 6, "=": 7, "X": 8}
    for nm, flag, c, pos, mapq, ops, pn, tlen, sc in recs:
        lseq = sum(n for n, o in ops if o in "MIS=X")
        cig = b"".join(struct.pack("<I", (n << 4) | opcode[o]) for n, o in ops)
        aux = b""
        if sc is not None:
            aux = b"NMC\0" + (b"ASc" + struct.pack("<b", sc) if -128 <= sc < 128 else b"ASi" + struct.pack("<i", sc))
        body = struct.pack("<iiBBHHHiiii", -1 if flag & 4 else c, -1 if flag & 4 else pos, len(nm) + 1, mapq, 0,
                           len(ops), flag, lseq, c if pn >= 0 else -1, pn, tlen)
        body += nm.encode() + b"\0" + cig + b"\0" * ((lseq + 1) // 2) + b"\xff" * lseq + aux
        raw += struct.pack("<i", len(body)) + body
    with gzip.GzipFile(path, "wb", mtime=0) as g:
        g.write(bytes(raw))


def write_sam_dups(path, names, lens, ev, seed, read_len=50, name_prefix="d", bam=False, quirks=0.0):
    """Queryname-grouped SAM / BAM for -r (PCR-duplicate removal, Genrich.c:3267-4042): proper
    pairs, singletons (mate unmapped), discordant pairs (both mates aligned, not as a proper
    pair, possibly on two chromosomes) and multi-mapped pairs with secondary alignments; about a
    quarter of the templates reuse the coordinates of an earlier one (the duplicates); base
    qualities vary per read (they decide which copy is kept) and a few reads carry none.
    quirks > 0: that share of the records comes in forms that are valid but unusual and that the
    reference treats in its own way -- SAM without optional fields (QUAL is then the last token of the
    line and keeps its line feed, loadFields 4350), SAM / BAM without SEQ (calcDistBAM 4694 takes l_seq = 0
    at face value, calcDist falls back to the CIGAR) and soft-clipped CIGARs."""
    rng = np.random.Generator(np.random.PCG64(seed))
    qrng = np.random.Generator(np.random.PCG64(seed + 9999))  # (its own stream: quirks = 0 writes what it always wrote)
    recs = []  # (qname, flag, chrom, pos0, mapq, rl, rnext chrom or -1, pnext0, tlen, AS, qual array or None)
    used, used_dc = [], []

    def quals(rl):
        if rng.random() < 0.05:
            return None
        return rng.integers(int(rng.integers(2, 30)), 41, rl).astype(np.uint8)

    for i in range(len(ev)):
        c, s, e = int(ev["chrom"][i]), int(ev["start"][i]), int(ev["end"][i])
        if used and rng.random() < 0.25:
            c, s, e = used[int(rng.integers(0, len(used)))]
        used.append((c, s, e))
        nm = f"{name_prefix}{i}"
        rl = min(read_len, e - s)
        kind = rng.choice(["pair", "pair", "pair", "single", "discord", "multi"])
        mapq = 30
        if kind == "pair" or kind == "multi":
            recs.append((nm, 99, c, s, mapq, rl, c, e - rl, e - s, -2, quals(rl)))
            recs.append((nm, 147, c, e - rl, mapq, rl, c, s, -(e - s), -3, quals(rl)))
            if kind == "multi":  # a secondary pair elsewhere on the chromosome, slightly worse score
                d = int(rng.integers(300, 2000))
                s2 = s + d if e + d < lens[c] else max(0, s - d)
                e2 = s2 + (e - s)
                recs.append((nm, 99 | 256, c, s2, mapq, rl, c, e2 - rl, e - s, -4, None))
                recs.append((nm, 147 | 256, c, e2 - rl, mapq, rl, c, s2, -(e - s), -4, None))
        elif kind == "single":
            rev = rng.random() < 0.5
            first = rng.random() < 0.5
            flag = 1 | 8 | (64 if first else 128) | (16 if rev else 0)
            recs.append((nm, flag, c, e - rl if rev else s, mapq, rl, -1, -1, 0, -1, quals(rl)))
        else:  # discordant: both mates aligned on their own
            if used_dc and rng.random() < 0.3:  # an earlier discordant template again, sometimes with the mates swapped
                c1, p1, r1rev, c2, p2, r2rev = used_dc[int(rng.integers(0, len(used_dc)))]
                if rng.random() < 0.5:
                    c1, p1, r1rev, c2, p2, r2rev = c2, p2, r2rev, c1, p1, r1rev
            else:
                c2 = int(rng.integers(0, len(lens)))
                p2 = int(rng.integers(0, max(1, lens[c2] - read_len)))
                r1rev, r2rev = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
                c1, p1 = c, (e - rl if r1rev else s)
            used_dc.append((c1, p1, r1rev, c2, p2, r2rev))
            rl = min(read_len, lens[c1] - p1, lens[c2] - p2)
            recs.append((nm, 1 | 64 | (16 if r1rev else 0) | (32 if r2rev else 0), c1, p1, mapq, rl, c2, p2, 0, -1, quals(rl)))
            recs.append((nm, 1 | 128 | (16 if r2rev else 0) | (32 if r1rev else 0), c2, p2, mapq, rl, c1, p1, 0, -2, quals(rl)))
    if not bam:
        with open(path, "w") as f:
            f.write("@HD\tVN:1.0\tSO:queryname\n")
            for n, l in zip(names, lens):
                f.write(f"@SQ\tSN:{n}\tLN:{l}\n")
            for nm, flag, c, pos, mapq, rl, rn, pn, tlen, sc, q in recs:
                rnext = "*" if rn < 0 else ("=" if rn == c else names[rn])
                qs = "*" if q is None else "".join(chr(33 + int(v)) for v in q)
                seq, cigar, tags = "A" * rl, f"{rl}M", f"\tNM:i:0\tAS:i:{sc}"
                if quirks:
                    u = qrng.random(3)
                    if u[0] < quirks:
                        tags = ""
                    if u[1] < quirks:
                        seq = "*"
                    if u[2] < quirks and rl > 12:
                        cigar = f"4S{rl - 4}M"
                f.write(f"{nm}\t{flag}\t{names[c]}\t{pos + 1}\t{mapq}\t{cigar}\t{rnext}\t{pn + 1}\t{tlen}\t"
                        f"{seq}\t{qs}{tags}\n")
        return
    import gzip
    import struct
    raw = bytearray()
    text = "@HD\tVN:1.0\tSO:queryname\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in zip(names, lens))
    raw += b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(names))
    for n, l in zip(names, lens):
        raw += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    for nm, flag, c, pos, mapq, rl, rn, pn, tlen, sc, q in recs:
        cig = struct.pack("<I", (rl << 4) | 0)
        aux = b"NMC\0" + b"ASc" + struct.pack("<b", sc)
        ncig, lseq = 1, rl
        if quirks:
            u = qrng.random(3)
            if u[0] < quirks:
                aux = b""
            if u[2] < quirks and rl > 12:
                cig, ncig = struct.pack("<II", (4 << 4) | 4, ((rl - 4) << 4) | 0), 2
            if u[1] < quirks:
                lseq = 0
        body = struct.pack("<iiBBHHHiiii", c, pos, len(nm) + 1, mapq, 0, ncig, flag, lseq, rn, pn, tlen)
        qb = b"\xff" * lseq if q is None else bytes(q.tolist())[:lseq]
        body += nm.encode() + b"\0" + cig + b"\x11" * ((lseq + 1) // 2) + qb + aux
        raw += struct.pack("<i", len(body)) + body
    with gzip.GzipFile(path, "wb", mtime=0) as g:
        g.write(bytes(raw))


# ---- -r at size: 10^6 .. 10^7 alignments, vectorised (write_sam_dups above walks a Python loop per record) ----------
# 28-byte records, the input of tools/records_to_sam (which renders them as SAM text)
REC_DTYPE = np.dtype([("tmpl", "<u4"), ("flag", "<u2"), ("chrom", "<i2"), ("pos", "<i4"), ("rnext", "<i2"), ("rl", "u1"),
                      ("qual", "u1"), ("pnext", "<i4"), ("tlen", "<i4"), ("AS", "i1"), ("mapq", "u1"), ("pad", "<u2")])


def make_dups_records(lens, n_templates, seed, dup_frac=0.2, read_len=50):
    """Queryname-grouped alignment records for -r (PCR duplicates, Genrich.c:3267-4042) at size: proper pairs (55 %), proper
    pairs with a secondary pair elsewhere (10 %: multi-alignment sets), singletons with the mate unmapped (15 %), discordant
    pairs (20 %, possibly on two chromosomes).  `dup_frac` of the templates take the coordinates of an earlier template --
    pairs and singletons from one pool (a singleton can duplicate an end of a kept pair: checkAndAdd 3514), discordant
    pairs from their own, with the mates swapped half of the time (findDupsDc looks both orders up).  Every read has its own
    base quality (one Phred value on all bases; 5 % have none): the quality sums decide which copy of a set is kept."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = int(n_templates)
    lens = np.asarray(lens, dtype=np.int64)
    kind = rng.choice(4, n, p=[0.55, 0.10, 0.15, 0.20])          # 0 pair, 1 pair + secondary pair, 2 singleton, 3 discordant
    chrom = rng.choice(len(lens), n, p=lens / lens.sum())
    flen = 100 + rng.integers(0, 100, n) + rng.integers(0, 100, n)
    start = (rng.random(n) * np.maximum(1, lens[chrom] - flen - 2100)).astype(np.int64)
    # a quarter of the templates within ~+-150 bp of a centre every 100 kb (peaks for the caller to find once the copies are gone)
    near = rng.random(n) < 0.25
    centre = (rng.integers(0, 1 << 30, n) % np.maximum(1, (lens[chrom] - 4000) // 100_000)) * 100_000 + 50_000
    start = np.where(near, np.clip(centre + rng.normal(0, 70, n).astype(np.int64) - flen // 2, 0, lens[chrom] - flen - 2100), start)
    end = start + flen
    dup = rng.random(n) < dup_frac
    pick = rng.random(n)

    def sources(members):
        """For the templates `members` (indices, ascending): the member whose coordinates each one ends up with."""
        k = np.arange(len(members))
        src = np.where(dup[members] & (k > 0), (pick[members] * k).astype(np.int64), k)
        for _ in range(40):   # (pointer jumping: a copy of a copy of ...)
            nxt = src[src]
            if np.array_equal(nxt, src):
                break
            src = nxt
        return members[src]

    pool = np.flatnonzero(kind != 3)
    src = np.arange(n)
    src[pool] = sources(pool)
    chrom, start, end = chrom[src], start[src], end[src]      # (discordant templates: src is the identity so far)
    # discordant pairs: R1 at one end of the template's fragment, R2 anywhere
    dc = np.flatnonzero(kind == 3)
    r1rev, r2rev = rng.random(n) < 0.5, rng.random(n) < 0.5
    c2 = rng.choice(len(lens), n, p=lens / lens.sum())
    p2 = (rng.random(n) * np.maximum(1, lens[c2] - read_len - 1)).astype(np.int64)
    c1 = chrom.copy()
    p1 = np.where(r1rev, end - read_len, start)
    sdc = sources(dc)
    swap = np.zeros(n, dtype=bool)
    swap[dc] = (sdc != dc) & (rng.random(len(dc)) < 0.5)
    for a in (c1, p1, r1rev, c2, p2, r2rev):
        a[dc] = a[sdc]
    c1s, p1s, r1s = np.where(swap, c2, c1), np.where(swap, p2, p1), np.where(swap, r2rev, r1rev)
    c2s, p2s, r2s = np.where(swap, c1, c2), np.where(swap, p1, p2), np.where(swap, r1rev, r2rev)
    # singletons
    srev, sfirst = rng.random(n) < 0.5, rng.random(n) < 0.5
    # secondary pairs of kind 1
    d = rng.integers(300, 2000, n)
    s2 = start + d
    e2 = s2 + (end - start)
    qual = rng.integers(2, 41, (n, 2)).astype(np.uint8)
    qual[rng.random((n, 2)) < 0.05] = 0xFF
    per = np.array([2, 4, 1, 2])[kind]
    first = np.concatenate([[0], np.cumsum(per)])
    recs = np.zeros(int(first[-1]), dtype=REC_DTYPE)
    recs["rl"] = read_len
    recs["mapq"] = 30
    recs["tmpl"] = np.repeat(np.arange(n, dtype=np.uint32), per)

    def put(sel, slot, flag, c, pos, rn, pn, tlen, AS, q):
        at = first[:-1][sel] + slot
        for k, v in (("flag", flag), ("chrom", c), ("pos", pos), ("rnext", rn), ("pnext", pn), ("tlen", tlen), ("AS", AS), ("qual", q)):
            recs[k][at] = v[sel] if isinstance(v, np.ndarray) else v

    pr = kind <= 1
    tl = end - start
    put(pr, 0, 99, chrom, start, chrom, end - read_len, tl, -2, qual[:, 0])
    put(pr, 1, 147, chrom, end - read_len, chrom, start, -tl, -3, qual[:, 1])
    mu = kind == 1
    put(mu, 2, 99 | 256, chrom, s2, chrom, e2 - read_len, tl, -4, 0xFF)
    put(mu, 3, 147 | 256, chrom, e2 - read_len, chrom, s2, -tl, -4, 0xFF)
    sn = kind == 2
    put(sn, 0, 1 | 8 | np.where(sfirst, 64, 128) | np.where(srev, 16, 0), chrom, np.where(srev, end - read_len, start), -1, -1, 0, -1, qual[:, 0])
    dd = kind == 3
    put(dd, 0, 1 | 64 | np.where(r1s, 16, 0) | np.where(r2s, 32, 0), c1s, p1s, c2s, p2s, 0, -1, qual[:, 0])
    put(dd, 1, 1 | 128 | np.where(r2s, 16, 0) | np.where(r1s, 32, 0), c2s, p2s, c1s, p1s, 0, -2, qual[:, 1])
    return recs
