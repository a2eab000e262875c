"""The host program genrich_amd/genrich-amd (C++): option handling, SAM ingest and interval
geometry are checked on the CPU against the reference's own -b event lists (golden fixtures);
the full command line (SAM in -> narrowPeak / -f / -k out) is checked on the GPU."""
import gzip
import importlib.util
import os
import subprocess

import pytest

import golden_cases as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G.GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return {c["name"]: c for c in mg.cases()}, mg


def _write_inputs(case, mg, tmp):
    """Regenerate the synthetic SAM inputs of a golden case (deterministic) under the same
    path prefix the fixture was made with, so that the -k header lines match byte for byte."""
    import synth
    os.makedirs(tmp, exist_ok=True)
    tf, cf = [], []
    for r, rep in enumerate(case["reps"]):
        names, lens, ev = rep["t"]
        mixed = case.get("mixed")
        ext = "bam" if mixed and mixed["bam"] else "sam"
        p = os.path.join(tmp, f"t{r}.{ext}")
        mg.write_input(p, names, lens, ev, mixed, 0, f"t{r}_")
        tf.append(p)
        if rep["c"] is None:
            cf.append(None)
        elif rep["c"] == "null":
            cf.append("null")
        else:
            names, lens, ev = rep["c"]
            p = os.path.join(tmp, f"c{r}.{ext}")
            mg.write_input(p, names, lens, ev, mixed, 1, f"c{r}_")
            cf.append(p)
    args = ["-t", ",".join(tf)]
    if any(c is not None for c in cf):
        args += ["-c", ",".join(c if c else "null" for c in cf)]
    if case.get("bed"):
        bp = os.path.join(tmp, "x.bed")
        with open(bp, "w") as f:
            for c, s, e in case["bed"]:
                f.write(f"{c}\t{s}\t{e}\n")
        args += ["-E", bp]
    return args + case["args"]


def _binary():
    from genrich_amd import build
    build.build()
    return build.HOST_BIN


@pytest.mark.parametrize("name", G.case_names())
def test_cli_event_stream_matches_reference(name, tmp_path):
    cases, mg = _cases()
    case = cases[name]
    args = _write_inputs(case, mg, str(tmp_path / "in"))
    bed = str(tmp_path / "events.bed")
    a = [x for x in args if x != "-X"]
    dups = str(tmp_path / "dups.txt")
    if "-r" in a:  # -R: the log of removed PCR duplicates (file names in its '#' lines differ by directory)
        a += ["-R", dups, "-v"]
    int16 = G.read_gz(name, "out.int16")  # saveInterval's "skipped due to overflow / underflow" lines (Genrich.c:2558-2573)
    if int16 is not None:
        a += ["-v"]
    res = subprocess.run([_binary(), "--events-only", "-b", bed] + a, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert open(bed, "rb").read() == G.read_gz(name, "events.bed")
    if int16 is not None:
        assert [l for l in res.stderr.splitlines() if "skipped due to" in l] == int16.decode().splitlines()
    if "-r" in a:
        strip = lambda txt: [l if not l.startswith("#") else l.rsplit("/", 1)[0].rsplit(" ", 1)[0] for l in txt.splitlines()]
        assert strip(open(dups).read()) == strip(G.read_gz(name, "out.dups").decode())
        meta, _, _, _ = G.load_case(name)
        got = [l.strip() for l in res.stderr.splitlines() if "aln sets:" in l or "duplicates:" in l]
        assert got == meta["ref_dups"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", G.case_names())
def test_cli_outputs_byte_identical(name):
    """Whole command line on the GPU box: the same files the reference wrote."""
    cases, mg = _cases()
    case = cases[name]
    meta, _, _, _ = G.load_case(name)
    tmp = meta["tmp_prefix"].rstrip("/")  # same input paths as when the fixture was made
    args = _write_inputs(case, mg, tmp)
    out = os.path.join(tmp, "cli_out")
    cmd = [_binary(), "-v", "-f", out + ".log", "-k", out + ".pile", "-b", out + ".bed"] + args
    if "-r" in args:
        cmd += ["-R", out + ".dups"]
    if "-X" not in case["args"]:
        cmd += ["-o", out + ".narrowPeak"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert open(out + ".bed", "rb").read() == G.read_gz(name, "events.bed")
    assert open(out + ".pile", "rb").read() == G.read_gz(name, "out.pile")
    assert open(out + ".log", "rb").read() == G.read_gz(name, "out.log")
    if "-X" not in case["args"]:
        assert open(out + ".narrowPeak", "rb").read() == G.read_gz(name, "out.narrowPeak")
    if "-r" in args:
        assert open(out + ".dups", "rb").read() == G.read_gz(name, "out.dups")
    lam = [f"{v:f}" for v in meta["ref_lambda"]]
    assert [l.split(": ")[1] for l in res.stderr.splitlines() if "Background pileup value" in l] == lam
    int16 = G.read_gz(name, "out.int16")  # the reads saveInterval dropped at its int16 limits, read by read (2558-2573)
    if int16 is not None:
        assert [l for l in res.stderr.splitlines() if "skipped due to" in l] == int16.decode().splitlines()
        assert "16-bit counters" not in res.stderr  # (nothing was left for the library's own replay to drop)
    if meta["ref_peaks"]:
        assert f"Peaks identified: {meta['ref_peaks'][0][0]} ({meta['ref_peaks'][0][1]}bp)" in res.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dups_pairs", "dups_y", "dups_x_bam", "quirks_sam", "quirks_bam"])
def test_cli_dups_on_the_device_match_the_reference_log(name):
    """-r with the membership half of findDups on the device (gx_dups_first): the -R log, the -b event list and the
    duplicate counts of the reference, and the report says the device path ran; the host-only path gives the same."""
    cases, mg = _cases()
    case = cases[name]
    meta, _, _, _ = G.load_case(name)
    tmp = meta["tmp_prefix"].rstrip("/")
    args = _write_inputs(case, mg, tmp)
    got = {}
    for mode in ("device", "host"):
        out = os.path.join(tmp, "dup_" + mode)
        env = dict(os.environ, GENRICH_DUPS_REPORT="1")
        if mode == "host":
            env["GENRICH_DUPS_HOST"] = "1"
        res = subprocess.run([_binary(), "-v", "-b", out + ".bed", "-R", out + ".dups", "-o", out + ".narrowPeak"] + args,
                             capture_output=True, text=True, env=env)
        assert res.returncode == 0, res.stderr
        assert open(out + ".dups", "rb").read() == G.read_gz(name, "out.dups"), mode
        assert open(out + ".bed", "rb").read() == G.read_gz(name, "events.bed"), mode
        assert open(out + ".narrowPeak", "rb").read() == G.read_gz(name, "out.narrowPeak"), mode
        rep = [l for l in res.stderr.splitlines() if l.startswith("[dups] device:")]
        if mode == "device":
            assert rep and all(int(l.split()[2]) > 0 for l in rep), res.stderr[-400:]
        else:
            assert not rep
        got[mode] = [l.strip() for l in res.stderr.splitlines() if "duplicates:" in l or "aln sets:" in l]
    assert got["device"] == got["host"] == meta["ref_dups"]   # the reference's counts, by either route


def _p_runs():
    import json
    out = []
    for name in G.case_names():
        meta = json.load(open(os.path.join(G.GOLDEN, name, "case.json")))
        for k, pr in enumerate(meta.get("p_runs", [])):
            out.append((name, k, pr))
    return out


@pytest.mark.parametrize("name,k,pr", _p_runs())
def test_cli_peaks_from_log(name, k, pr, tmp_path):
    """-P: re-calling peaks from the reference's own -f log must reproduce the reference's -P output."""
    log = str(tmp_path / "in.log")
    with open(log, "wb") as f:
        f.write(G.read_gz(name, "out.log"))
    args = [_binary(), "-P", "-f", log, "-o", str(tmp_path / "np")] + pr["args"]
    if pr["bed"]:
        bp = str(tmp_path / "x.bed")
        with open(bp, "w") as f:
            for c, s, e in pr["bed"]:
                f.write(f"{c}\t{s}\t{e}\n")
        args += ["-E", bp]
    res = subprocess.run(args, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert open(tmp_path / "np", "rb").read() == G.read_gz(name, f"out.P{k}.narrowPeak")


def _bgzf(data, block=30_000):
    """BGZF (SAM spec 4.1): independent gzip members of < 64 KiB with a 'BC' extra field, plus the
    empty end-of-file member."""
    import struct
    import zlib
    out = bytearray()
    for off in list(range(0, len(data), block)) + [None]:
        chunk = b"" if off is None else data[off:off + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = c.compress(chunk) + c.flush()
        bsize = 18 + len(payload) + 8 - 1
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + payload
        out += struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


@pytest.mark.parametrize("block", [37, 4099])
def test_cli_bgzf_records_straddling_blocks(block, tmp_path):
    """BAM records are parsed in place when they lie inside one inflated block and copied when they
    straddle blocks: with 37-byte blocks every record (and most 4-byte length fields) is split."""
    cases, mg = _cases()
    name = "dups_x_bam"
    args = _write_inputs(cases[name], mg, str(tmp_path / "in"))
    for i, a in enumerate(args):
        for p in (a.split(",") if i and args[i - 1] in ("-t", "-c") else []):
            if p != "null":
                raw = gzip.decompress(open(p, "rb").read())
                open(p, "wb").write(_bgzf(raw, block=block))
    a = [x for x in args if x != "-X"]
    bed = str(tmp_path / "events.bed")
    res = subprocess.run([_binary(), "--events-only", "--threads", "3", "-b", bed] + a, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert open(bed, "rb").read() == G.read_gz(name, "events.bed")


@pytest.mark.parametrize("name", ["unpaired_bam_atac", "dups_x_bam", "ctrl_q"])
def test_cli_bgzf_input_parallel_inflate(name, tmp_path):
    """BGZF input (what samtools / bgzip write) is inflated by a thread pool; the event stream must be
    the one the single-threaded zlib path (and the reference) produce."""
    cases, mg = _cases()
    case = cases[name]
    args = _write_inputs(case, mg, str(tmp_path / "in"))
    for i, a in enumerate(args):  # re-compress every input file as BGZF
        for p in (a.split(",") if i and args[i - 1] in ("-t", "-c") else []):
            if p != "null":
                raw = open(p, "rb").read()
                if p.endswith(".bam"):
                    raw = gzip.decompress(raw)
                open(p, "wb").write(_bgzf(raw))
    a = [x for x in args if x != "-X"]
    for threads in ("4", "1"):
        bed = str(tmp_path / f"events{threads}.bed")
        res = subprocess.run([_binary(), "--events-only", "--threads", threads, "-b", bed] + a, capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        assert open(bed, "rb").read() == G.read_gz(name, "events.bed")


def test_cli_bgzf_corrupt_block_is_an_error(tmp_path):
    cases, mg = _cases()
    args = _write_inputs(cases["basic"], mg, str(tmp_path / "in"))
    p = args[1]
    z = bytearray(_bgzf(open(p, "rb").read()))
    z[len(z) // 2] ^= 0x5A
    open(p, "wb").write(bytes(z))
    res = subprocess.run([_binary(), "--events-only", "--threads", "4", "-b", str(tmp_path / "e.bed")] + args,
                         capture_output=True, text=True)
    assert res.returncode != 0 and "BGZF" in res.stderr


@pytest.mark.parametrize("which", ["-t", "-c"])
def test_cli_reads_a_sam_stream_from_stdin(which, tmp_path):
    """`samtools view -h ... | Genrich -t - ...` (openRead 5135: '-' is stdin).  The host needs every
    header before anything goes to the device, so the part of the stream its header pre-scan consumed
    is replayed to the real pass; the event stream must be the one the same data gives as a file."""
    cases, mg = _cases()
    args = _write_inputs(cases["ctrl_q"], mg, str(tmp_path / "in"))
    a = [x for x in args if x != "-X"]
    path = a[a.index(which) + 1]
    assert "," not in path and path.endswith(".sam")
    piped = list(a)
    piped[piped.index(which) + 1] = "-"
    bed = str(tmp_path / "events.bed")
    with open(path, "rb") as f:
        res = subprocess.run([_binary(), "--events-only", "-b", bed] + piped, stdin=f, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert open(bed, "rb").read() == G.read_gz("ctrl_q", "events.bed")


def test_cli_refuses_compressed_or_empty_stdin(tmp_path):
    """openRead 5145-5160: an empty input 'cannot be opened'; gzip data on stdin is an error of its own."""
    cases, mg = _cases()
    args = _write_inputs(cases["basic"], mg, str(tmp_path / "in"))
    z = gzip.compress(open(args[1], "rb").read())
    res = subprocess.run([_binary(), "--events-only", "-t", "-"], input=z, capture_output=True)
    assert res.returncode != 0 and b"Cannot pipe in gzip-compressed file" in res.stderr
    res = subprocess.run([_binary(), "--events-only", "-t", "-"], input=b"", capture_output=True)
    assert res.returncode != 0 and b"cannot open file for reading" in res.stderr
    empty = tmp_path / "empty.sam"
    empty.write_bytes(b"")
    res = subprocess.run([_binary(), "--events-only", "-t", str(empty)], capture_output=True)
    assert res.returncode != 0 and b"cannot open file for reading" in res.stderr
    # checkBAM 5107-5125: a compressed input that ends inside the "BAM\1" magic (or holds nothing at all)
    for content in (b"", b"BA"):
        for wrap in (gzip.compress, _bgzf):
            z = tmp_path / "short.gz"
            z.write_bytes(wrap(content))
            res = subprocess.run([_binary(), "--events-only", "--threads", "2", "-t", str(z)], capture_output=True)
            assert res.returncode != 0 and b"Error! : cannot open file for reading" in res.stderr


def test_cli_damaged_sam_as_the_reference(tmp_path):
    """Small SAM inputs with a field dropped or emptied, a non-numeric value, a broken CIGAR or tag, a changed
    or misplaced header line, a truncation ..., BAM inputs truncated or with one bit flipped, and what the
    reference made of each (its exit status,
    its `Error!` line, its -b stream): tests/golden/cli_errors.json.gz, written by make_cli_errors.py from the
    reference binary.  The host program must cut the lines up, convert and complain in the same order."""
    import json
    cases = json.loads(gzip.open(os.path.join(G.GOLDEN, "cli_errors.json.gz")).read())
    assert len(cases) >= 80
    import base64
    bed = tmp_path / "e.bed"
    for k, c in enumerate(cases):
        if "bam" in c:
            sam = tmp_path / "t.bam"
            sam.write_bytes(base64.b64decode(c["bam"]))
        else:
            sam = tmp_path / "t.sam"
            sam.write_text(c["sam"])
        if bed.exists():
            bed.unlink()
        res = subprocess.run([_binary(), "--events-only", "-t", str(sam), "-b", str(bed)] + c["args"],
                             capture_output=True, text=True, errors="replace")
        what = f"case {k} ({c['kind']}, {' '.join(c['args'])})"
        err = next((l for l in res.stderr.splitlines() if l.startswith("Error!")), "")
        if c["rc"]:
            assert res.returncode != 0 and err == c["error"], f"{what}: {err!r} instead of {c['error']!r}"
        else:
            assert res.returncode == 0, f"{what}: {err}"
            if c["events"] is not None:
                assert bed.read_text(errors="replace") == c["events"], what


def test_cli_several_bed_files(tmp_path):
    """-E takes a comma-separated list (loadBED 5187-5238); the names are walked with strtok_r because every
    line of a file is cut up with strtok."""
    cases, mg = _cases()
    args = _write_inputs(cases["basic"], mg, str(tmp_path / "in"))
    a, b = tmp_path / "a.bed", tmp_path / "b.bed"
    a.write_text("chrA\t100\t900\tname\t0\t+\nchrA\t5000\t5100\n")
    b.write_text("chrA\t20000\t20500\n")
    bed = tmp_path / "e.bed"
    res = subprocess.run([_binary(), "--events-only", "-b", str(bed), "-E", f"{a},{b}"] + [x for x in args if x != "-X"],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert open(bed, "rb").read() == G.read_gz("basic", "events.bed")


def test_cli_bam_reference_name_without_nul(tmp_path):
    """A BAM whose reference name lacks its NUL terminator is "Cannot parse BAM file" in the header
    pre-scan too (it used to reach saveChrom as an unterminated C string)."""
    import struct
    hdr = b"@HD\tVN:1.0\tSO:queryname\n"
    for gz in (gzip.compress, _bgzf):
        raw = b"BAM\1" + struct.pack("<i", len(hdr)) + hdr + struct.pack("<i", 1) + struct.pack("<i", 4) + b"chrA" + struct.pack("<i", 1000)
        p = str(tmp_path / f"bad_{gz.__name__}.bam")
        open(p, "wb").write(gz(raw))
        res = subprocess.run([_binary(), "--events-only", "-t", p, "-b", str(tmp_path / "e.bed")], capture_output=True, text=True)
        assert res.returncode == 1 and "Cannot parse BAM file" in res.stderr, res.stderr


def test_cli_bgzf_trailing_garbage_is_end_of_stream(tmp_path):
    """Bytes after the last BGZF member that do not begin another gzip member end the stream quietly, as
    with zlib's reader (through which the reference reads): same events, exit status 0."""
    cases, mg = _cases()
    name = "dups_x_bam"
    args = _write_inputs(cases[name], mg, str(tmp_path / "in"))
    for i, a in enumerate(args):
        for p in (a.split(",") if i and args[i - 1] in ("-t", "-c") else []):
            if p != "null":
                raw = gzip.decompress(open(p, "rb").read())
                open(p, "wb").write(_bgzf(raw) + b"trailing bytes that are no gzip member")
    a = [x for x in args if x != "-X"]
    bed = str(tmp_path / "events.bed")
    res = subprocess.run([_binary(), "--events-only", "--threads", "3", "-b", bed] + a, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert open(bed, "rb").read() == G.read_gz(name, "events.bed")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ctrl_q", "reps3", "multimap", "bedx", "basic", "ctrl_only_chrom"])
def test_cli_two_contexts_one_gpu_equal_one(name):
    """--devices 0,0: two library contexts on the one GPU of the test box, chromosomes sharded between them,
    the collectives through the host program's in-process callbacks (RCCL wants distinct devices).  All
    outputs must be the reference's bytes, as with one context."""
    cases, mg = _cases()
    case = cases[name]
    meta, _, _, _ = G.load_case(name)
    tmp = meta["tmp_prefix"].rstrip("/")  # same input paths as when the fixture was made (the -k header names them)
    args = _write_inputs(case, mg, tmp)
    out = os.path.join(tmp, "cli2_out")
    cmd = [_binary(), "--devices", "0,0", "-f", out + ".log", "-k", out + ".pile"] + args
    if "-X" not in case["args"]:
        cmd += ["-o", out + ".narrowPeak"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert open(out + ".pile", "rb").read() == G.read_gz(name, "out.pile")
    assert open(out + ".log", "rb").read() == G.read_gz(name, "out.log")
    if "-X" not in case["args"]:
        assert open(out + ".narrowPeak", "rb").read() == G.read_gz(name, "out.narrowPeak")


# ---- parallel record decoding: decoder threads over batches of records, the state on one thread ----
@pytest.mark.parametrize("name", ["basic", "multimap", "dups_pairs", "dups_x_bam", "quirks_sam", "quirks_bam",
                                  "unpaired_x", "unpaired_bam_atac", "ctrl_q", "reps3", "saturate16"])
@pytest.mark.parametrize("threads,batch,chunk", [("1", None, None), ("4", None, None), ("4", "150", "3"), ("3", "1", "1"),
                                                 ("4", None, "serial")])
def test_cli_parallel_decoding_is_the_sequential_stream(name, threads, batch, chunk, tmp_path):
    """The -b event stream, the -R log and everything -v prints must not depend on how many threads decode the
    records, where the batches are cut (GENRICH_BATCH_BYTES: batches of a line or two, so that read-name groups,
    pairs and multimapping sets straddle them) or how the read-name groups are dealt to the threads that run the
    state machine (GENRICH_CHUNK_RECS: a chunk per group; GENRICH_SERIAL_STATE: one thread for all of them).  The
    golden event stream is the reference's."""
    cases, mg = _cases()
    args = _write_inputs(cases[name], mg, str(tmp_path / "in"))
    a = [x for x in args if x != "-X"] + ["-v"]
    bed, dups = str(tmp_path / "events.bed"), str(tmp_path / "dups.txt")
    if "-r" in a:
        a += ["-R", dups]
    env = dict(os.environ)
    if batch:
        env["GENRICH_BATCH_BYTES"] = batch
    if chunk == "serial":
        env["GENRICH_SERIAL_STATE"] = "1"
    elif chunk:
        env["GENRICH_CHUNK_RECS"] = chunk
    res = subprocess.run([_binary(), "--events-only", "--threads", threads, "-b", bed] + a, capture_output=True, text=True, env=env)
    assert res.returncode == 0, res.stderr
    assert open(bed, "rb").read() == G.read_gz(name, "events.bed")
    ref = subprocess.run([_binary(), "--events-only", "--threads", "1", "-b", bed + ".1"] + [x if x != dups else dups + ".1" for x in a],
                         capture_output=True, text=True)
    assert res.stderr == ref.stderr
    if "-r" in a:
        assert open(dups).read() == open(dups + ".1").read()


@pytest.mark.parametrize("threads", ["1", "4"])
@pytest.mark.parametrize("name", ["saturate16_frac", "saturate16_atac"])
def test_cli_int16_decisions_more(name, threads, tmp_path):
    """saveInterval's int16 checks (Genrich.c:2558-2573), read by read, beyond the `saturate16` case:
    saturate16_frac -- a third of the piled-up reads multimapped: the int16 part of the reference's (cov, eighths, sixths,
    tenths) counters reaches 32,767 / -32,768 in between whole numbers, and which alignments are dropped depends on the
    exact fractional state when each one arrives (73,054 are);
    saturate16_atac -- ATAC cut sites (-j -d 40 -y): two intervals per fragment, of which one may be dropped alone.
    The reference's -b list (hash and line count) and its warnings, in order (tests/golden/<name>/, made by
    make_golden.py's make_int16_extra)."""
    import hashlib
    _, mg = _cases()
    spec = mg.int16_extra()[name]
    sam = str(tmp_path / "t0.sam")
    mg.write_input(sam, ["chrA"], [60_000], spec["ev"](), spec["mixed"], 0, "t0_")
    bed = str(tmp_path / "events.bed")
    res = subprocess.run([_binary(), "--events-only", "--threads", threads, "-v", "-b", bed, "-t", sam] + spec["args"],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    want_hash, want_lines = open(os.path.join(G.GOLDEN, name, "events.bed.sha256")).read().split()
    got = open(bed, "rb").read()
    assert got.count(b"\n") == int(want_lines)
    assert hashlib.sha256(got).hexdigest() == want_hash
    want = gzip.open(os.path.join(G.GOLDEN, name, "out.int16.gz"), "rt").read().splitlines()
    assert len(want) > 1000
    assert [l for l in res.stderr.splitlines() if "skipped due to" in l] == want


def test_cli_parallel_decoding_keeps_warnings_and_errors_in_file_order(tmp_path):
    """A record that cannot be decoded is found by a decoder thread, possibly long before the records ahead of it have
    gone through the run's state: its error must still come after their warnings, and be the reference's."""
    sq = "@HD\tVN:1.0\tSO:queryname\n@SQ\tSN:chr1\tLN:100000\n"
    rec = lambda q, flag, pos, pn, tl, cigar="50M": f"{q}\t{flag}\tchr1\t{pos}\t30\t{cigar}\t=\t{pn}\t{tl}\t*\t*\tAS:i:0\n"
    body = ""
    for i in range(40):
        body += rec(f"r{i}", 99, 100 + 10 * i, 300 + 10 * i, 250) + rec(f"r{i}", 147, 300 + 10 * i, 100 + 10 * i, -250)
    # a read with more alignments than the reference's buffer holds: a -v warning (Genrich.c:4169-4171)
    many = "".join(rec("crowd", 256 | 0x10 * (k % 2), 5000 + 7 * k, 0, 0) for k in range(140))
    bad = rec("broken", 99, 9000, 9100, 150, cigar="50Q")  # unknown CIGAR op: an error
    tail = "".join(rec(f"t{i}", 99, 20000 + i, 20100 + i, 150) for i in range(30))
    sam = tmp_path / "t.sam"
    sam.write_text(sq + body + many + body.replace("r", "s").replace("chs1", "chr1") + bad + tail)
    outs = []
    for threads, batch, chunk in (("1", None, None), ("4", "120", "1"), ("3", "1", "2"), ("8", None, None)):
        env = dict(os.environ)
        if batch:
            env["GENRICH_BATCH_BYTES"] = batch
        if chunk:
            env["GENRICH_CHUNK_RECS"] = chunk
        res = subprocess.run([_binary(), "--events-only", "--threads", threads, "-v", "-y", "-t", str(sam), "-b", str(tmp_path / "e.bed")],
                             capture_output=True, text=True, env=env)
        assert res.returncode != 0
        outs.append(res.stderr)
    lines = outs[0].splitlines()
    assert lines[-1] == "Error! 'Q': unknown Op in CIGAR"
    assert any("more than 128 alignments" in l for l in lines[:-1])
    assert all(o == outs[0] for o in outs[1:])


def test_cli_line_longer_than_the_reference_buffer_is_cut_the_same_way(tmp_path):
    """readSAM reads a line in pieces of at most 65,519 characters (its 65,520-byte buffer): a longer record loses
    its tail and the piece behind it is a record of its own -- here an error, `poorly formatted`, exactly as the
    reference binary reports it for this input (checked against oracle/_ref/Genrich when this test was written).  The
    decoders that cut a mapped file into lines themselves (cutSpan) must cut at the same places, whatever the batches."""
    sq = "@HD\tVN:1.0\tSO:queryname\n@SQ\tSN:chr1\tLN:1000000\n"
    rec = lambda q, flag, pos, pn, tl, seq="*", qual="*", cig="50M": f"{q}\t{flag}\tchr1\t{pos}\t30\t{cig}\t=\t{pn}\t{tl}\t{seq}\t{qual}\tAS:i:0\n"
    body = "".join(rec(f"r{i}", 99, 100 + 10 * i, 300 + 10 * i, 250) + rec(f"r{i}", 147, 300 + 10 * i, 100 + 10 * i, -250) for i in range(50))
    L = 70000
    long1 = rec("long", 99, 5000, 5300, 300 + L, seq="A" * L, qual="I" * L, cig=f"{L}M") + rec("long", 147, 5300, 5000, -(300 + L))
    tail = "".join(rec(f"t{i}", 99, 20000 + i, 20100 + i, 150) for i in range(20))
    sam = tmp_path / "t.sam"
    sam.write_text(sq + body + long1 + tail)
    beds = []
    for threads, batch in (("1", None), ("4", "100"), ("8", None), ("3", "1")):
        env = dict(os.environ)
        if batch:
            env["GENRICH_BATCH_BYTES"] = batch
            env["GENRICH_CHUNK_RECS"] = "1"
        bed = tmp_path / f"e{threads}{batch}.bed"
        res = subprocess.run([_binary(), "--events-only", "--threads", threads, "-t", str(sam), "-b", str(bed)],
                             capture_output=True, text=True, env=env)
        assert res.returncode != 0
        assert res.stderr.splitlines()[-1] == "Error! long: poorly formatted SAM/BAM record"
        beds.append(bed.read_text())
    # (49: the last pair before the bad record is still waiting for the next read name when the error ends the run)
    assert beds[0].count("\n") == 49 and all(b == beds[0] for b in beds[1:])


def test_cli_parallel_state_keeps_the_warning_count_and_the_float_sum_in_file_order(tmp_path):
    """Two things that are summed ACROSS read-name groups and are not just integers: the count that suppresses -v
    warnings after the first 128 (saveInterval 2530-2543; the tail line `(another N warning messages suppressed)`), and
    the double behind the -x average length, to which a multimapping pair adds length / count (processPair 3150) -- a
    float sum, so the workers hand their terms to the state's owner, which adds them in file order."""
    L = 30000
    sq = f"@HD\tVN:1.0\tSO:queryname\n@SQ\tSN:chr1\tLN:{L}\n"
    rec = lambda q, flag, pos, pn, tl, ln=50: f"{q}\t{flag}\tchr1\t{pos}\t30\t{ln}M\t=\t{pn}\t{tl}\t*\t*\tAS:i:0\n"
    body = ""
    for i in range(300):  # pairs whose second mate runs past the end of the reference: a counted warning each
        body += rec(f"e{i}", 99, L - 300 - i, L - 40, 340 + i) + rec(f"e{i}", 147, L - 40, L - 300 - i, -(340 + i), ln=60 + i % 7)
    for i in range(200):  # pairs with two or three equally good alignments: totalLen += length / count
        k = 2 + i % 2
        for j in range(k):
            p = 1000 + 37 * i + 3000 * j
            body += rec(f"m{i}", 99 if j == 0 else 355, p, p + 150 + i % 11, 200 + i % 11) + \
                    rec(f"m{i}", 147 if j == 0 else 403, p + 150 + i % 11, p, -(200 + i % 11))
    for i in range(150):  # unpaired alignments, extended to the average length by -x
        body += rec(f"u{i}", 0 if i % 2 else 16, 5000 + 31 * i, 0, 0)
    sam = tmp_path / "t.sam"
    sam.write_text(sq + body)
    outs = []
    for threads, batch, chunk in (("1", None, None), ("4", None, None), ("4", "200", "1"), ("3", "1", "5")):
        env = dict(os.environ)
        if batch:
            env["GENRICH_BATCH_BYTES"] = batch
        if chunk:
            env["GENRICH_CHUNK_RECS"] = chunk
        bed = tmp_path / f"e{threads}{batch}.bed"
        res = subprocess.run([_binary(), "--events-only", "--threads", threads, "-v", "-x", "-s", "20", "-t", str(sam), "-b", str(bed)],
                             capture_output=True, text=True, env=env)
        assert res.returncode == 0, res.stderr
        outs.append((res.stderr, bed.read_text()))
    err = outs[0][0]
    assert err.count("prevented from extending past") == 128 and "warning messages suppressed" in err
    assert "extended to" in err
    assert all(o == outs[0] for o in outs[1:])


def test_cli_long_runs_of_skipped_records_neither_end_a_group_nor_change_the_counts(tmp_path):
    """Records that never reach the state machine (unmapped, low MAPQ) do not end a read-name group: a pair whose mates
    are 200 unmapped records apart is still a pair (record() keeps the name until a different one gets that far).  The
    thread that cuts the stream into chunks of whole groups counts long runs of such records itself and leaves them out
    of the chunks; the -v accounting and the events must be the sequential run's (the numbers below are the reference
    binary's for this input)."""
    import random
    random.seed(3)
    sq = "@HD\tVN:1.0\tSO:queryname\n@SQ\tSN:chr1\tLN:1000000\n"
    rec = lambda q, flag, pos, pn, tl, mapq=30: f"{q}\t{flag}\tchr1\t{pos}\t{mapq}\t50M\t=\t{pn}\t{tl}\t*\t*\tAS:i:0\n"
    unm = lambda q, flag: f"{q}\t{flag}\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII\n"
    out = [unm(f"u{i}", 77) + unm(f"u{i}", 141) for i in range(500)]
    for i in range(300):
        out.append(rec(f"r{i}", 99, 1000 + 10 * i, 1200 + 10 * i, 250) + rec(f"r{i}", 147, 1200 + 10 * i, 1000 + 10 * i, -250))
        if i % 50 == 10:
            out += [unm(f"x{i}_{k}", 77) + unm(f"x{i}_{k}", 141) for k in range(random.choice([10, 40, 100, 300]))]
        if i % 70 == 5:
            out += [rec(f"q{i}_{k}", 99, 5000 + k, 5200 + k, 250, mapq=0) + rec(f"q{i}_{k}", 147, 5200 + k, 5000 + k, -250, mapq=0)
                    for k in range(120)]
    out.append(rec("split", 99, 70000, 70200, 250))
    out += [unm(f"y{k}", 77) for k in range(200)]
    out.append(rec("split", 147, 70200, 70000, -250))
    out += [unm(f"z{i}", 77) for i in range(1000)]
    sam = tmp_path / "t.sam"
    sam.write_text(sq + "".join(out))
    outs = []
    for threads, batch, chunk in (("1", None, None), ("4", None, None), ("4", "300", "1"), ("3", "1", "2"), ("8", "5000", "50")):
        env = dict(os.environ)
        if batch:
            env["GENRICH_BATCH_BYTES"] = batch
        if chunk:
            env["GENRICH_CHUNK_RECS"] = chunk
        bed = tmp_path / f"e{threads}{batch}.bed"
        res = subprocess.run([_binary(), "--events-only", "--threads", threads, "-v", "-m", "10", "-t", str(sam), "-b", str(bed)],
                             capture_output=True, text=True, env=env)
        assert res.returncode == 0, res.stderr
        outs.append((res.stderr, bed.read_text()))
    err = outs[0][0]
    nums = {l.split(":")[0].strip(): int(l.split(":")[1]) for l in err.splitlines() if ":" in l and l.split(":")[1].strip().isdigit()}
    assert nums["SAM records analyzed"] == 5002 and nums["Unmapped"] == 3200 and nums["MAPQ < 10"] == 1200
    assert nums["Paired alignments"] == 602 and nums["Full fragments"] == 301
    assert outs[0][1].count("\n") == 301 and "split_1_E_0" in outs[0][1]
    assert all(o == outs[0] for o in outs[1:])
