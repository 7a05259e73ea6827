"""The library's FASTA/FASTQ(.gz) reader must yield what the reference's reader (kseq over gzFile, src/Reads.cpp) yields.
CPU only: parsing needs no device.  Pinned two ways: a pure-Python statement of kseq's record grammar, and the compiled
reference itself (mapping a file through its own reader == mapping our parsed records from memory)."""
import gzip
import os

import numpy as np
import pytest

from conftest import have_ref


def kseq_python(data: bytes):
    """kseq_read (lib/bwa/kseq.h:176-215), restated; returns [(name, seq, qual)] and stops at the first malformed record"""
    recs, i, n = [], 0, len(data)

    def line(j):                      # -> (text without newline / trailing CR as kseq trims it, next index, hit_eof)
        k = data.find(b"\n", j)
        if k < 0:
            return data[j:], n, True
        return data[j:k], k + 1, False
    last = 0
    while True:
        if not last:
            while i < n and data[i:i + 1] not in (b">", b"@"):
                i += 1
            if i >= n:
                break
            i += 1
        j = i
        while j < n and data[j:j + 1] not in (b" ", b"\t", b"\n", b"\v", b"\f", b"\r"):
            j += 1
        name = data[i:j]
        if j >= n and not name:
            break
        delim = data[j:j + 1]
        i = j + 1
        if delim != b"\n" and j < n:
            _, i, _ = line(i)
        seq = b""
        last = 0
        c = b""
        while i < n:
            c = data[i:i + 1]
            i += 1
            if c in (b">", b"+", b"@"):
                break
            if c == b"\n":
                c = b""
                continue
            rest, i, _ = line(i)
            seq += c + rest
            if len(seq) > 1 and seq.endswith(b"\r"):
                seq = seq[:-1]
            c = b""
        if c in (b">", b"@"):
            last = 1
        if c != b"+":
            recs.append((name, seq, b""))
            if i >= n and not last:
                break
            continue
        if i >= n:
            break
        _, i, eof = line(i)
        if eof:
            break
        qual = b""
        while len(qual) < len(seq) and i < n:
            rest, i, _ = line(i)
            qual += rest
            if len(qual) > 1 and qual.endswith(b"\r"):
                qual = qual[:-1]
        if len(qual) != len(seq):
            break
        recs.append((name, seq, qual))
    return recs


CASES = {
    "fasta_multiline": b">r1 some comment\nACGT\nAC\n\nGG\n>r2\tx\nTTTT\n>empty\n>r3\nA\n",
    "fasta_crlf": b">r1 c\r\nACGT\r\nGG\r\n>r2\r\nTT\r\n",
    "fastq": b"@q1 desc\nACGTAC\n+\nIIIIII\n@q2\nAC\nGT\n+q2\nII\nII\n",
    "fastq_at_in_qual": b"@q1\nACGT\n+\n@III\n@q2\nAA\n+\n>>\n",
    "mixed": b"junk before\n>r1\nACGT\n@q1\nAC\n+\nII\n>r2\nGG",
    "no_trailing_newline": b">r1\nACGT",
    "truncated_quality": b"@q1\nACGT\n+\nIIII\n@q2\nACGT\n+\nII\n",
    # starts like a FASTA (the mapped, multi-threaded parser takes it) but holds a FASTQ record: handed to the sequential parser
    "fasta_then_fastq": b">r1\nACGT\n@q1\nAC\n+\nII\n>r2\nGG",
    "fasta_plus_line": b">r1\nACGT\n+\nIIII\n>r2\nGG\n",
    "fasta_gt_inside_header": b">r1 a>b c\nAC>GT\nAA\n>r2\n\n\nTT\n",
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("gz", [False, True])
def test_reader_matches_kseq_grammar(tmp_path, name, gz):
    import lordfast_amd as la
    data = CASES[name]
    path = str(tmp_path / (name + (".gz" if gz else ".txt")))
    with (gzip.open(path, "wb") if gz else open(path, "wb")) as fh:
        fh.write(data)
    got = [(n, s, q) for names, seqs, quals in la.read_file(path) for n, s, q in zip(names, seqs, quals)]
    assert got == kseq_python(data)
    # batching does not change the record stream
    got2 = [(n, s, q) for names, seqs, quals in la.read_file(path, batch_reads=1) for n, s, q in zip(names, seqs, quals)]
    assert got2 == got


def test_reader_vs_reference_reader(tmp_path, golden_dir, golden_reads):
    """the compiled reference reading the file itself == the compiled reference fed with OUR parse of that file"""
    if not have_ref():
        pytest.skip("needs oracle/_ref/liblfref.so")
    import lordfast_amd as la
    from oracle import pyoracle as po
    names, seqs = golden_reads
    names, seqs = names[:24], seqs[:24]
    rng = np.random.default_rng(3)
    path = str(tmp_path / "reads.fq.gz")
    with gzip.open(path, "wb") as fh:
        for i, (n, s) in enumerate(zip(names, seqs)):
            if i % 3 == 0:                              # FASTA record, wrapped lines, a comment
                fh.write(b">" + n + b" comment text\n" + b"\n".join(s[k:k + 70] for k in range(0, len(s), 70)) + b"\n")
            else:                                       # FASTQ record (qualities without '@' / '+' line starts are not required by kseq)
                q = bytes(rng.integers(35, 74, size=len(s)).astype(np.uint8))
                fh.write(b"@" + n + b"\n" + s + b"\n+\n" + q + b"\n")
    batches = la.read_file(path)
    pn = [x for b in batches for x in b[0]]
    ps = [x for b in batches for x in b[1]]
    pq = [x if x else b"*" for b in batches for x in b[2]]
    assert pn == list(names) and ps == list(seqs)
    ref = po.Ref()
    fa = os.path.join(golden_dir, "genome.fa")
    if not os.path.exists(fa + ".cache"):
        ref.index_build(fa)                              # the reference's loader wants its 12-mer cache file
    ref.load(fa)
    ref.set_params(po.default_params(threads=1), "reader-test")
    sam_file, _ = ref.map_file(path, header=False)
    sam_mem, _ = ref.map_mem(pn, ps, [q if q != b"*" else b"" for q in pq])
    assert sam_file == sam_mem


def test_mapped_fasta_reader_many_threads_equals_sequential(tmp_path, monkeypatch):
    """a plain FASTA larger than the multi-thread threshold (pieces cut at record boundaries, wrapped lines, comments, CRLF,
    empty records): the mapped parser, the sequential parser and the kseq grammar agree, also under read / base limits"""
    import lordfast_amd as la
    rng = np.random.default_rng(11)
    parts = []
    total = 0
    i = 0
    while total < 9_000_000:
        n = int(rng.integers(0, 60000))
        s = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=n))
        w = int(rng.choice([60, 70, 80, 10**9]))
        eol = b"\r\n" if i % 7 == 3 else b"\n"
        hdr = b">read%d" % i + (b" comment with > and @ signs" if i % 3 == 0 else b"") + eol
        body = eol.join(s[k:k + w] for k in range(0, max(1, n), w)) + eol if n else b""
        parts.append(hdr + body + (b"\n" if i % 11 == 0 else b""))
        total += len(parts[-1]); i += 1
    data = b"".join(parts)
    path = str(tmp_path / "big.fa")
    with open(path, "wb") as fh:
        fh.write(data)
    exp = kseq_python(data)
    for kw in (dict(), dict(batch_reads=37), dict(batch_bases=1_500_000)):
        got = [(n, s, q) for names, seqs, quals in la.read_file(path, **kw) for n, s, q in zip(names, seqs, quals)]
        assert got == exp, kw
    monkeypatch.setenv("LF_READER_SEQUENTIAL", "1")
    got = [(n, s, q) for names, seqs, quals in la.read_file(path) for n, s, q in zip(names, seqs, quals)]
    assert got == exp


def _bgzf(data: bytes, block: int = 0xff00) -> bytes:
    """what bgzip writes: gzip members with a 'BC' extra subfield holding the member's size - 1, then the 28-byte end marker"""
    import struct
    import zlib
    out = []
    for a in list(range(0, len(data), block)) + [None]:
        chunk = data[a:a + block] if a is not None else b""
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        bsize = 12 + 6 + len(body) + 8
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
                   + body + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out)


def _big_fastq(total_bytes: int, seed: int, crlf_every: int = 0):
    rng = np.random.default_rng(seed)
    parts, recs, total, i = [], [], 0, 0
    qa = np.frombuffer(bytes(range(33, 74)), dtype=np.uint8)
    while total < total_bytes:
        n = int(rng.integers(1, 40000))
        s = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=n))
        q = bytearray(rng.choice(qa, size=n).tobytes())
        if i % 5 == 0:
            q[0] = ord("@")                              # a quality line that starts like a header
        if i % 7 == 0:
            q[0] = ord("+")
        q = bytes(q)
        eol = b"\r\n" if (crlf_every and i % crlf_every == 0) else b"\n"
        name = b"q%d" % i
        parts.append(b"@" + name + (b" a comment @ with signs" if i % 3 == 0 else b"") + eol + s + eol + b"+" + (name if i % 4 == 0 else b"") + eol + q + eol
                     + (b"\n" if i % 13 == 0 else b""))
        recs.append((name, s, q))
        total += len(parts[-1]); i += 1
    return b"".join(parts), recs


@pytest.mark.parametrize("container", ["plain", "bgzf", "gz", "gz_multi_member"])
def test_window_parser_fastq_all_containers(tmp_path, container):
    """four-line FASTQ larger than the multi-thread threshold: the window parser (pieces cut at record starts -- a quality line
    may begin with '@' or '+'), the BGZF block inflater and the one-stream gzip path all give kseq's records, also under
    read-only / base-only batch limits"""
    import lordfast_amd as la
    data, recs = _big_fastq(12_000_000, 5, crlf_every=9)
    assert kseq_python(data[:400_000])[:5] == recs[:5]            # the generator and the grammar statement agree
    path = str(tmp_path / ("reads.fq" + ("" if container == "plain" else ".gz")))
    with open(path, "wb") as fh:
        if container == "plain":
            fh.write(data)
        elif container == "bgzf":
            fh.write(_bgzf(data))
        elif container == "gz":
            fh.write(gzip.compress(data, 1))
        else:
            cut = [0, 1_000_003, 5_000_001, len(data)]
            fh.write(b"".join(gzip.compress(data[a:b], 1) for a, b in zip(cut, cut[1:])))
    for kw in (dict(), dict(batch_reads=41), dict(batch_bases=2_000_000)):
        batches = la.read_file(path, **kw)
        got = [(n, s, q) for names, seqs, quals in batches for n, s, q in zip(names, seqs, quals)]
        assert got == recs, (container, kw)
        if "batch_reads" in kw:
            assert all(len(b[0]) == 41 for b in batches[:-1])
        if "batch_bases" in kw:
            assert all(sum(len(x) for x in b[1]) >= 2_000_000 for b in batches[:-1])


def test_bgzf_file_with_a_plain_gzip_member_appended(tmp_path):
    """`cat a.bgz b.gz`: the file is taken for BGZF from its first member; when a member that is not a BGZF block turns up the rest of the
    file goes through the one-stream inflater (gzread-based readers -- the reference's kseq -- read such files); record boundaries need
    not coincide with member boundaries"""
    import lordfast_amd as la
    data, recs = _big_fastq(5_000_000, 11)
    cut = 3_000_001
    bg = _bgzf(data[:cut])
    bg = bg[:-28]                                        # (without the empty end-of-file block, as `cat` of a truncated stream would leave it)
    path = str(tmp_path / "cat.fq.gz")
    with open(path, "wb") as fh:
        fh.write(bg + gzip.compress(data[cut:], 1))
    for kw in (dict(), dict(batch_reads=53)):
        got = [(n, s, q) for names, seqs, quals in la.read_file(path, **kw) for n, s, q in zip(names, seqs, quals)]
        assert got == recs, kw


def test_window_parser_hands_wrapped_fastq_to_the_sequential_parser(tmp_path):
    """a big FASTQ whose later records are wrapped over several lines: the batches before them come from the window parser,
    the rest from the sequential one -- one record stream, kseq's"""
    import lordfast_amd as la
    data, recs = _big_fastq(6_000_000, 8)
    rng = np.random.default_rng(1)
    extra = []
    for i in range(40):
        n = int(rng.integers(100, 3000))
        s = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n))
        q = bytes(rng.integers(35, 60, size=n).astype(np.uint8))                 # no '@' / '+' / '>' in these qualities
        w = 70
        extra.append(b"@w%d\n" % i + b"\n".join(s[k:k + w] for k in range(0, n, w)) + b"\n+\n" + b"\n".join(q[k:k + w] for k in range(0, n, w)) + b"\n")
        recs.append((b"w%d" % i, s, q))
    data += b"".join(extra)
    assert kseq_python(data) == recs
    for gz in (False, True):
        path = str(tmp_path / ("mixed.fq" + (".gz" if gz else "")))
        with open(path, "wb") as fh:
            fh.write(_bgzf(data) if gz else data)
        got = [(n, s, q) for names, seqs, quals in la.read_file(path, batch_reads=100) for n, s, q in zip(names, seqs, quals)]
        assert got == recs, gz


def test_window_parser_reads_only_limit_parses_a_batch_not_the_file(tmp_path):
    """a batch limited by reads only (lf_reads_next(max_reads = N, max_bases = 0)) must cost O(batch), not O(rest of the file):
    300 batches of a 60 MB FASTA in well under the time one pass per batch would take"""
    import time
    import lordfast_amd as la
    rng = np.random.default_rng(2)
    body = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=200))
    n = 300_000
    data = b"".join(b">s%d\n" % i + body + b"\n" for i in range(n))
    path = str(tmp_path / "short.fa")
    with open(path, "wb") as fh:
        fh.write(data)
    t0 = time.time()
    batches = la.read_file(path, batch_reads=1000)
    dt = time.time() - t0
    assert len(batches) == 300 and all(len(b[0]) == 1000 for b in batches)
    assert batches[-1][0][-1] == b"s%d" % (n - 1) and batches[17][1][5] == body
    assert dt < 20, dt           # quadratic behaviour (every batch parses the rest of the file) takes minutes here
