"""A BAM that looks like bowtie2 output (queryname-grouped proper pairs, 2 x 100 bases with sequence,
qualities and the usual aux tags), BGZF-compressed: input for timing the host program's ingest.
usage: make_bench_bam.py OUT.bam N_PAIRS [SEED]"""
import struct
import sys
import zlib

import numpy as np


def bgzf(data, block=65_280, level=6):
    out = bytearray()
    for off in list(range(0, len(data), block)) + [None]:
        chunk = b"" if off is None else bytes(data[off:off + block])
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        payload = c.compress(chunk) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(payload) + 8 - 1) + payload
        out += struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


def main():
    path, n = sys.argv[1], int(sys.argv[2])
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    names = [f"chr{i + 1}" for i in range(24)]
    lens = [int(x) for x in rng.integers(40_000_000, 250_000_000, 24)]
    text = "@HD\tVN:1.0\tSO:queryname\n" + "".join(f"@SQ\tSN:{a}\tLN:{b}\n" for a, b in zip(names, lens))
    text += "@PG\tID:bowtie2\tPN:bowtie2\tVN:2.4.1\n"
    raw = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(names)))
    for a, b in zip(names, lens):
        raw += struct.pack("<i", len(a) + 1) + a.encode() + b"\0" + struct.pack("<i", b)
    rl = 100
    cig = struct.pack("<I", (rl << 4) | 0)
    chrom = rng.integers(0, 24, n)
    flen = rng.integers(110, 600, n)
    u = rng.random(n)
    seqs = rng.integers(0, 256, (n, 2, rl // 2), dtype=np.uint8) & 0x33  # nibbles of 1/2/... -> A/C-ish codes
    seqs |= 0x11
    quals = np.clip(rng.normal(34, 6, (n, 2, rl)), 2, 41).astype(np.uint8)
    mapq = rng.choice([0, 1, 23, 40, 42, 42, 42], n)
    for i in range(n):
        c = int(chrom[i])
        s = int(u[i] * (lens[c] - 1000))
        e = s + int(flen[i])
        nm = b"SRR1234567.%d\0" % (i + 1)
        for k, (flag, pos, pn, tl) in enumerate(((99, s, e - rl, e - s), (147, e - rl, s, -(e - s)))):
            aux = (b"ASc" + struct.pack("<b", -int(i % 20)) + b"XSc" + struct.pack("<b", -30) + b"XNC\0XMC\1XOC\0XGC\0NMC\1"
                   b"YSc" + struct.pack("<b", -5) + b"YTZCP\0" + b"MDZ57A42\0")
            body = struct.pack("<iiBBHHHiiii", c, pos, len(nm), int(mapq[i]), 4681, 1, flag, rl, c, pn, tl)
            body += nm + cig + seqs[i, k].tobytes() + quals[i, k].tobytes() + aux
            raw += struct.pack("<i", len(body)) + body
    open(path, "wb").write(bgzf(raw))


main()
