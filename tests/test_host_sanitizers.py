"""CPU sanitizer runs of the threaded HOST code of libkmap_hip (round-3 verdict item: none existed).  tests/host_san/Makefile compiles
kmap_amd/csrc/host_io.hip (FASTA / gz reader, occurrence-CSV formatter + pwrite pool) and kmap_amd/csrc/host_pool.h (the
conversion pool of counts.hip's table fetch) host-only, once with -fsanitize=address,undefined and once with -fsanitize=thread, into
a small driver; any sanitizer report fails the run (non-zero exit, text on stderr).  The FASTA results are also compared with a
Python restatement of the reference's array contract (kmer_count.py:244-347) on ragged / empty / CRLF / lowercase / over-long-line
inputs and with the golden arrays of tests/test.fa.  GPU AddressSanitizer is not available on this pool: CPU only."""
import gzip
import os
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent
SAN = HERE / "host_san"
_CODE = {c: v for v, c in enumerate("ACGT")}
_CODE.update({c.lower(): v for c, v in list(_CODE.items())})


@pytest.fixture(scope="module")
def drivers():
    if shutil.which("/opt/rocm/bin/hipcc") is None:
        pytest.skip("hipcc not available")
    r = subprocess.run(["make", "-s", "-C", str(SAN)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return {"asan": SAN / "build" / "driver_asan", "tsan": SAN / "build" / "driver_tsan"}


def _run(exe, *args, env=None):
    e = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
             TSAN_OPTIONS="halt_on_error=1")
    e.update(env or {})
    r = subprocess.run([str(exe), *[str(a) for a in args]], capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.returncode, r.stderr[-3000:])
    return r.stdout


def _model(text):
    """the reference's arrays for a FASTA text (kmer_count.py:244-347 through Bio.SeqIO: a record starts at a line beginning with
    '>', its sequence is the following lines with white space removed; A/C/G/T in either case -> 0..3, anything else -> 255; one
    255 after every record; borders [start, end))"""
    seq, borders, cur = [], [], None
    for line in text.split("\n"):
        if line.startswith(">"):
            if cur is not None:
                borders.append((cur, len(seq)))
                seq.append(255)
            cur = len(seq)
        elif cur is not None:
            seq.extend(_CODE.get(ch, 255) for ch in line if ch not in " \t\r\v\f")
    if cur is not None:
        borders.append((cur, len(seq)))
        seq.append(255)
    return np.array(seq, np.uint8), np.array(borders, np.int64).reshape(-1, 2)


def _fasta(exe, path, tmp_path):
    out = _run(exe, "fasta", path, tmp_path / "s.bin", tmp_path / "b.bin")
    nb, ns = (int(t) for t in out.split())
    seq = np.fromfile(tmp_path / "s.bin", np.uint8)
    borders = np.fromfile(tmp_path / "b.bin", np.int64).reshape(-1, 2)
    assert len(seq) == nb and len(borders) == ns
    return seq, borders


CASES = {
    "plain": ">r1\nACGTACGT\n>r2 some description\nTTGACA\nGGG\n",
    "no_trailing_newline": ">r1\nACGT\n>r2\nGGCC",
    "crlf": ">r1\r\nACGTAC\r\nGT\r\n>r2\r\nNNAC\r\n",
    "lowercase_and_iupac": ">r1\nacgtnRYacgt\n>r2\nAcGt-*\n",
    "empty_records": ">e1\n>e2\n\n>r\nAC\n>e3\n",
    "text_before_first_header": "junk line\nACGT\n>r1\nGATTACA\n",
    "gt_inside_a_line": ">r1\nAC>GT\nA>C\n",
    "blank_lines_and_spaces": ">r1\n\nAC GT\n\t\nTT\n\n>r2\n \n",
    "empty_file": "",
    "only_newlines": "\n\n\n",
    "over_long_line": ">long\n" + "ACGTTGCA" * 700_000 + "\n>tail\nAC\n",       # one 5.6 MB line: crosses the reader's 4 MiB buffer
    "long_header": ">" + "h" * 5_000_000 + "\nACGT\n",
}


# the plain-file path cuts the mapped file into ranges behind newlines and walks them on several threads (count in kmap_fasta_open,
# encode into the caller's arrays in kmap_fasta_read); KMAP_FASTA_MIN_CHUNK lets a small file be cut as finely as a 1.6-GB one
FA_ENVS = [{"KMAP_IO_THREADS": "1"}, {"KMAP_IO_THREADS": "4", "KMAP_FASTA_MIN_CHUNK": "1"},
           {"KMAP_IO_THREADS": "8", "KMAP_FASTA_MIN_CHUNK": "64"}, {"KMAP_IO_THREADS": "3", "KMAP_FASTA_MIN_CHUNK": "1048576"}]


def _fasta_env(exe, path, tmp_path, env):
    out = _run(exe, "fasta", path, tmp_path / "s.bin", tmp_path / "b.bin", env=env)
    nb, ns = (int(t) for t in out.split())
    seq = np.fromfile(tmp_path / "s.bin", np.uint8)
    borders = np.fromfile(tmp_path / "b.bin", np.int64).reshape(-1, 2)
    assert len(seq) == nb and len(borders) == ns
    return seq, borders


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_fasta_reader_edge_cases(drivers, tmp_path, san):
    for name, text in CASES.items():
        long_case = name in ("over_long_line", "long_header")
        p = tmp_path / f"{name}.fa"
        p.write_bytes(text.encode())
        want_s, want_b = _model(text)
        for env in (FA_ENVS[3:] if long_case else FA_ENVS):        # the multi-MB cases once (three threads, 1-MiB ranges)
            seq, borders = _fasta_env(drivers[san], p, tmp_path, env)
            np.testing.assert_array_equal(seq, want_s, err_msg=f"{name} {env}")
            np.testing.assert_array_equal(borders, want_b, err_msg=f"{name} {env}")
        if name in ("plain", "crlf", "over_long_line") and not (san == "tsan" and long_case):   # the same through gzip (one thread)
            pz = tmp_path / f"{name}.fa.gz"
            with gzip.open(pz, "wb") as fh:
                fh.write(text.encode())
            seq_z, borders_z = _fasta(drivers[san], pz, tmp_path)
            np.testing.assert_array_equal(seq_z, want_s, err_msg=name + ".gz")
            np.testing.assert_array_equal(borders_z, want_b, err_msg=name + ".gz")


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_fasta_reader_random_texts(drivers, tmp_path, san):
    """random line soup (headers, empty lines, white space of every kind, '>' inside lines, LF / CRLF, with and without a final
    newline) through every range / thread setting: the ranges' encodings must concatenate to the model's arrays"""
    import random
    rng = random.Random(5 if san == "asan" else 6)
    alpha = "ACGTacgtNn>> \t\r\v\fxyz-*"
    for it in range(40):
        lines = []
        for _ in range(rng.randint(0, 60)):
            kind = rng.random()
            if kind < 0.25:
                lines.append(">" + "".join(rng.choice("hdr >x") for _ in range(rng.randint(0, 8))))
            elif kind < 0.35:
                lines.append("")
            else:
                lines.append("".join(rng.choice(alpha) for _ in range(rng.randint(0, 30))))
        sep = rng.choice(["\n", "\n", "\r\n"])
        text = sep.join(lines) + (sep if rng.random() < 0.7 else "")
        p = tmp_path / f"r{it}.fa"
        p.write_bytes(text.encode())
        want_s, want_b = _model(text)
        env = FA_ENVS[it % 3]
        seq, borders = _fasta_env(drivers[san], p, tmp_path, env)
        np.testing.assert_array_equal(seq, want_s, err_msg=f"{it} {env} {text!r}")
        np.testing.assert_array_equal(borders, want_b, err_msg=f"{it} {env} {text!r}")


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_fasta_reader_gzip_ring(drivers, tmp_path, san):
    """a gzip stream of ~19 MB of text: the inflating thread fills its ring of 4-MiB buffers several times over while the calling
    thread encodes (records and header lines cut by the buffer ends); the same arrays as the mapped plain file and the model"""
    rng = np.random.default_rng(3)
    n, L = 120_000, 150
    seqs = np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, size=(n, L))]
    text = "".join(f">read{i} len={L}\n{row.tobytes().decode()}\n" for i, row in enumerate(seqs))
    plain, pz = tmp_path / "ring.fa", tmp_path / "ring.fa.gz"
    plain.write_bytes(text.encode())
    with gzip.open(pz, "wb", compresslevel=1) as fh:
        fh.write(text.encode())
    want_s, want_b = _model(text)
    seq_z, borders_z = _fasta(drivers[san], pz, tmp_path)
    np.testing.assert_array_equal(seq_z, want_s)
    np.testing.assert_array_equal(borders_z, want_b)
    seq_p, borders_p = _fasta_env(drivers[san], plain, tmp_path, {"KMAP_IO_THREADS": "8", "KMAP_FASTA_MIN_CHUNK": "1048576"})
    np.testing.assert_array_equal(seq_p, want_s)
    np.testing.assert_array_equal(borders_p, want_b)
    # a stream cut off in the middle: an error from the inflating thread, reported by the calling one; nothing leaks, nothing hangs
    cut = tmp_path / "cut.fa.gz"
    cut.write_bytes(pz.read_bytes()[: pz.stat().st_size // 2])
    e = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1", TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([str(drivers[san]), "fasta", str(cut), "/dev/null", "/dev/null"], capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 2 and "read error" in r.stderr and "Sanitizer" not in r.stderr, r.stderr[-2000:]


def test_fasta_reader_golden_test_fa(drivers, tmp_path):
    """tests/test.fa of the reference (golden G1: 45 979 bytes, 1002 reads) through the sanitized reader, plain and gzipped"""
    g = np.load(HERE / "golden" / "scan_testfa.npz")          # seq / borders as the reference's preproc wrote them
    seq, borders = _fasta(drivers["asan"], HERE / "golden" / "test.fa", tmp_path)
    assert len(seq) == 45_979 and len(borders) == 1002
    want_s, want_b = _model((HERE / "golden" / "test.fa").read_text())
    np.testing.assert_array_equal(seq, want_s)
    np.testing.assert_array_equal(borders, want_b)
    np.testing.assert_array_equal(seq, g["seq"])
    np.testing.assert_array_equal(borders, g["borders"])
    pz = tmp_path / "t.fa.gz"
    with gzip.open(pz, "wb") as fh:
        fh.write((HERE / "golden" / "test.fa").read_bytes())
    seq_z, _ = _fasta(drivers["asan"], pz, tmp_path)
    np.testing.assert_array_equal(seq_z, seq)
    missing = subprocess.run([str(drivers["asan"]), "fasta", str(tmp_path / "nope.fa"), "/dev/null", "/dev/null"], capture_output=True, text=True)
    assert missing.returncode == 2 and "cannot open" in missing.stderr and "Sanitizer" not in missing.stderr


@pytest.mark.parametrize("san,threads", [("asan", "16"), ("tsan", "8"), ("tsan", "1")])
def test_csv_writer_pool(drivers, tmp_path, san, threads):
    """formatter threads + pwrite pool: int32 and byte-sized hit counts give the same file, whatever the thread count; the run is
    clean under the sanitizer (TSan: the chunk cursors, buffers and the write-status flag are shared between the pools)"""
    a, b = tmp_path / f"a_{san}_{threads}.csv", tmp_path / f"b_{san}_{threads}.csv"
    rows = int(_run(drivers[san], "csv", a, b, 300_000, 11, env={"KMAP_IO_THREADS": threads}).strip())
    da = a.read_bytes()
    assert da == b.read_bytes() and da.count(b"\n") == rows + 1
    ref = tmp_path / "ref.csv"
    _run(drivers["asan"], "csv", ref, tmp_path / "ref8.csv", 300_000, 11, env={"KMAP_IO_THREADS": "3"})
    assert ref.read_bytes() == da
    f = da.split(b"\n")[1].split(b";")
    assert len(f) == 5 and f[0].isdigit() and f[-1].isdigit()


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_conversion_pool(drivers, san):
    """u32 -> i64 into an UNALIGNED destination (the memory-mapped pickle view of TableSaver), u32 -> u64, u32 -> u32, empty input"""
    assert int(_run(drivers[san], "pool", 2_500_000).strip()) == 2_500_000
    assert int(_run(drivers[san], "pool", 1000).strip()) == 1000


@pytest.mark.parametrize("san,threads", [("asan", "16"), ("tsan", "8"), ("tsan", "1")])
def test_f2_line_writer_and_cell_medians(drivers, tmp_path, san, threads):
    """kmap_write_f2_tsv_line (formatter threads, one writer) and kmap_cell_medians_i32 under the sanitizers: the line equals
    Python's f"{x:.2f}" join, the medians equal numpy's per cell"""
    rng = np.random.default_rng(11)
    vals = np.concatenate([rng.integers(-300, 300, 150_000) / 2.0, rng.normal(size=50_000) * 37,
                           [0.0, -0.0, 0.005, 0.015, 1e15, -1e15, 1e16, 1e300, float("inf"), float("-inf"), float("nan"), 2.0 ** 52 + 0.5]])
    (tmp_path / "v.f64").write_bytes(vals.astype(np.float64).tobytes())
    out = _run(drivers[san], "f2", tmp_path / "v.f64", tmp_path / "o.tsv", tmp_path / "m.f64", env={"KMAP_IO_THREADS": threads})
    n, n_cells, n_pos = (int(t) for t in out.split())
    assert n == len(vals)
    assert (tmp_path / "o.tsv").read_text() == "pair\n" + "\t".join(f"{x:.2f}" for x in vals.tolist()) + "\n\n"
    packed = np.fromfile(tmp_path / "m.f64", np.float64)
    hits, pos, med = packed[:n_cells].astype(np.int64), packed[n_cells:n_cells + n_pos], packed[n_cells + n_pos:]
    offs = np.concatenate([[0], np.cumsum(hits)])
    want = np.array([np.median(pos[offs[r]:offs[r + 1]]) if hits[r] else np.nan for r in range(n_cells)])
    np.testing.assert_array_equal(med, want)
