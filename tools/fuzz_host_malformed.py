"""Build container only (needs oracle/_ref/Genrich): damaged SAM / BAM inputs.  The host program
(--events-only) and the unmodified reference must agree on success / failure, on the `Error!` line and,
when both succeed, on the -b event stream.  usage: fuzz_host_malformed.py SEED0 SEED1"""
import sys, os, subprocess, random, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from genrich_amd import synth
REF = os.path.join(ROOT, 'oracle', '_ref', 'Genrich'); BIN = os.path.join(ROOT, 'genrich_amd', 'genrich-amd')
N2 = ["chr1", "chr2", "chrM"]


def mutate_sam(text, rng):
    lines = text.split("\n")
    body = [i for i, l in enumerate(lines) if l and not l.startswith("@")]
    head = [i for i, l in enumerate(lines) if l.startswith("@")]
    kind = rng.choice(["field", "flag", "pos", "cigar", "rname", "drop_sq", "dup_sq", "so", "trunc", "mapq", "tag", "empty_line", "seq"])
    if kind == "trunc":
        cut = rng.randint(1, len(text) - 1)
        return text[:cut], kind
    if kind == "drop_sq" and head:
        sq = [i for i in head if lines[i].startswith("@SQ")]
        if sq: del lines[rng.choice(sq)]
        return "\n".join(lines), kind
    if kind == "dup_sq":
        sq = [i for i in head if lines[i].startswith("@SQ")]
        if sq:
            f = lines[rng.choice(sq)].split("\t"); f[2] = "LN:%d" % rng.randint(1, 99999)
            lines.insert(rng.choice(sq), "\t".join(f))
        return "\n".join(lines), kind
    if kind == "so":
        lines[0] = rng.choice(["@HD\tVN:1.0\tSO:coordinate", "@HD\tVN:1.0", "@HD", "@CO\thello", "@HD\tSO:queryname\tVN:1"])
        return "\n".join(lines), kind
    if not body:
        return text, "none"
    i = rng.choice(body)
    f = lines[i].split("\t")
    if kind == "field":
        k = rng.randrange(len(f))
        if rng.random() < 0.5: del f[k:]
        else: f.pop(k)
    elif kind == "flag":
        f[1] = rng.choice(["4", "77", "141", "2048", "256", "x", "", "65536", "-1", "99x"])
    elif kind == "pos":
        f[3] = rng.choice(["0", "-5", "abc", "", "999999999", "12x", "1e3"])
    elif kind == "cigar":
        f[5] = rng.choice(["*", "10M5", "M", "50Z", "20M10I20M", "0M", "30S20M", "50M50M", "10H40M", ""])
    elif kind == "rname":
        f[2] = rng.choice(["*", "chrZ", "", "chr1 ", "CHR1"])
    elif kind == "mapq":
        f[4] = rng.choice(["255", "-1", "x", "", "300"])
    elif kind == "tag" and len(f) > 11:
        f[rng.randrange(11, len(f))] = rng.choice(["AS:i:x", "AS:f:1.5", "AS:i:", "AS", "XS:i:3", "AS:Z:7", "AS:i:-2147483648"])
    elif kind == "seq" and len(f) > 10:
        f[9] = rng.choice(["ACGT", "", "*", "A" * 50]); f[10] = rng.choice(["*", "IIII", ""])
    elif kind == "empty_line":
        lines.insert(i, "")
        return "\n".join(lines), kind
    lines[i] = "\t".join(f)
    return "\n".join(lines), kind


def mutate_bam(raw, rng):
    kind = rng.choice(["trunc", "flip", "flip_head", "len"])
    b = bytearray(raw)
    if kind == "trunc":
        return bytes(b[:rng.randint(1, len(b) - 1)]), kind
    if kind == "flip_head":
        k = rng.randrange(min(len(b), 200)); b[k] ^= 1 << rng.randrange(8)
        return bytes(b), kind
    k = rng.randrange(len(b)); b[k] ^= 1 << rng.randrange(8)
    return bytes(b), kind


def first_error(txt):
    for l in txt.splitlines():
        if l.startswith("Error!"):
            return l
    return ""


def one(seed):
    rng = random.Random(seed)
    L = [rng.randint(20_000, 60_000), rng.randint(10_000, 40_000), rng.randint(2_000, 8_000)]
    d = f"/tmp/fuzz/m{seed}"; os.makedirs(d, exist_ok=True)
    ev = synth.make_fragments(L, rng.randint(30, 300), seed=seed)
    bam = rng.random() < 0.4
    p = f"{d}/t.{'bam' if bam else 'sam'}"
    writer = rng.choice(["mixed", "dups"])
    (synth.write_sam_mixed if writer == "mixed" else synth.write_sam_dups)(p, N2, L, ev, seed, name_prefix="t_", bam=bam)
    if bam:
        data, kind = mutate_bam(gzip.decompress(open(p, "rb").read()), rng)
        open(p, "wb").write(gzip.compress(data))
    else:
        data, kind = mutate_sam(open(p).read(), rng)
        open(p, "w").write(data)
    args = ["-t", p] + rng.choice([[], ["-y"], ["-r"], ["-y", "-r"], ["-x"]])
    r = subprocess.run([REF] + args + ["-b", f"{d}/ref.bed", "-o", f"{d}/ref.np"], capture_output=True, text=True, errors="replace")
    h = subprocess.run([BIN, "--events-only"] + args + ["-b", f"{d}/hip.bed"], capture_output=True, text=True, errors="replace")
    if r.returncode not in (0, 1):      # the reference crashed (signal): nothing to compare with
        subprocess.run(["rm", "-rf", d]); return None
    re_, he_ = first_error(r.stderr), first_error(h.stderr)
    # the reference goes on into the statistics, which the events-only host does not run
    late = ("no analyzable fragments" in re_) or ("Experimental sample" in re_) or ("peak" in re_.lower()) or ("No analyzable" in re_)
    msg = None
    if late:
        pass   # (the events-only host has no statistics to fail in and goes on)
    elif (r.returncode != 0) != (h.returncode != 0):
        msg = f"rc ref={r.returncode} host={h.returncode}\n  REF: {re_}\n  HOST: {he_}"
    elif r.returncode != 0 and re_ != he_:
        msg = f"messages differ\n  REF: {re_}\n  HOST: {he_}"
    if msg is None and h.returncode == 0 and os.path.exists(f"{d}/ref.bed"):
        if open(f"{d}/ref.bed", "rb").read() != open(f"{d}/hip.bed", "rb").read(): msg = "-b differs"
    if msg:
        return f"seed {seed} ({'bam' if bam else 'sam'}, {kind}, {' '.join(args[2:])}): {msg}"
    subprocess.run(["rm", "-rf", d]); return None


bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    m = one(seed)
    if m: print(m); bad += 1
print("done, failures:", bad)
