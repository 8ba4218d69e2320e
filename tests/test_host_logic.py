"""CPU tests of the host-side logic and of the C-ABI library surface (no GPU compute calls)."""
import ctypes
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"


@pytest.fixture(scope="module")
def built_lib():
    from kmap_amd import _ffi
    if not _ffi.LIB_PATH.exists():
        from kmap_amd.build import build
        build()
    return _ffi.LIB_PATH


def test_library_exports_every_declared_symbol(built_lib):
    """Every function declared in include/kmap_hip.h is exported by libkmap_hip.so and bound in _ffi."""
    from kmap_amd import _ffi
    header = (ROOT / "include" / "kmap_hip.h").read_text()
    declared = set(re.findall(r"\b(kmap_[A-Za-z0-9_]+)\s*\(", header))
    lib = ctypes.CDLL(str(built_lib))
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"declared but not exported: {missing}"
    unbound = sorted(declared - set(_ffi.exported_symbols()))
    assert not unbound, f"declared but not bound in _ffi: {unbound}"
    assert set(_ffi.exported_symbols()) <= declared
    _ffi.lib()
    assert _ffi.lib().kmap_version() >= 1


def test_no_fallback_when_library_missing(tmp_path, monkeypatch):
    from kmap_amd import _ffi
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_ffi.KmapError):
        _ffi.lib()


def test_product_never_imports_oracle():
    for py in (ROOT / "kmap_amd").rglob("*.py"):
        assert "oracle" not in py.read_text(), f"{py} references the oracle"


def test_dtype_rules_and_scalar_helpers(golden):
    import kmap_amd.kmer_count as K
    assert K.get_hash_dtype(15) == np.uint32 and K.get_hash_dtype(16) == np.uint64 and K.get_hash_dtype(31) == np.uint64
    assert K.get_cnt_dtype(15) == np.int32 and K.get_cnt_dtype(16) == np.int64
    with pytest.raises(Exception):
        K.get_hash_dtype(32)
    assert K.get_invalid_hash(np.uint32) == 0xFFFFFFFF
    g = golden("ops.npz")
    for s, h, rc in zip(g["kmer_strs"], g["kmer_hashes"], g["kmer_rc_hashes"]):
        s = str(s)
        assert int(K.kmer2hash(s)) == int(h) and K.hash2kmer(h, len(s)) == s
        assert int(K.revcom_hash(h, len(s))) == int(rc)
        assert K.hash2kmer(rc, len(s)) == K.reverse_complement(s)
    assert K.arr2dna(K.dna2arr("ACGTNacgt")) == "ACGTNNNNNN"           # lower case is not mapped (callers upper-case)


def test_fasta_encoding_matches_reference_arrays(golden, tmp_path):
    import kmap_amd.kmer_count as K
    s = golden("scan_testfa.npz")
    arr, borders = K.encode_fasta(str(GOLD / "test.fa"))
    np.testing.assert_array_equal(arr, s["seq"])
    np.testing.assert_array_equal(borders, s["borders"])
    # multi-line records, lower case, N, blank lines, gz
    fa = tmp_path / "m.fa"
    fa.write_text(">r1 desc\nacgt\nNNAC\n\n>r2\nTTTT\n>empty\n>r3\nAC GT\n")
    a2, b2 = K.encode_fasta(str(fa))
    assert K.arr2dna(a2) == "ACGTNNACN" + "TTTTN" + "N" + "ACGTN"
    np.testing.assert_array_equal(b2, [[0, 8], [9, 13], [14, 14], [15, 19]])
    import gzip
    with gzip.open(tmp_path / "m.fa.gz", "wt") as fh:
        fh.write(fa.read_text())
    a3, b3 = K.encode_fasta(str(tmp_path / "m.fa.gz"))
    np.testing.assert_array_equal(a2, a3)
    np.testing.assert_array_equal(b2, b3)
    # native parser == record-by-record Python encoder, incl. CRLF, text before the first header, no trailing newline
    fb = tmp_path / "w.fa"
    fb.write_bytes(b"junk before\r\n>a\r\nACGT\r\nnnAC\r\n>b x y\r\n\r\n>c\r\nTTGA")
    for f in (GOLD / "test.fa", fa, fb):
        an, bn = K.encode_fasta(str(f))
        ap, bp = K.encode_fasta_py(str(f))
        np.testing.assert_array_equal(an, ap)
        np.testing.assert_array_equal(bn, bp)
    with pytest.raises(ValueError):
        K.encode_fasta(str(tmp_path / "missing.fa"))
    # a .gz that ends in the middle of its stream is an error, not a shorter input (zlib hands out what it could inflate and then
    # reports an ordinary end: only gzerror knows)
    whole = (tmp_path / "m.fa.gz").read_bytes()
    (tmp_path / "cut.fa.gz").write_bytes(whole[: len(whole) - 9])
    with pytest.raises(ValueError, match="truncated"):
        K.encode_fasta(str(tmp_path / "cut.fa.gz"))


def test_fasta_encoder_is_the_same_for_every_range_and_thread_count(golden, tmp_path, monkeypatch):
    """the shipped library's plain-file path: the mapped file cut into ranges behind newlines, counted and encoded on several threads
    -- whatever the cut and the thread count, the arrays are those of the one-range walk and of the record-by-record Python encoder"""
    import kmap_amd.kmer_count as K
    s = golden("scan_testfa.npz")
    big = tmp_path / "big.fa"                       # 40 copies of the reference's test.fa with CRLF records and stray text mixed in
    text = (GOLD / "test.fa").read_bytes()
    big.write_bytes(b"ignored\n" + b"".join(text + (b">crlf\r\nAC GT\r\nnn\r\n" if i % 3 == 0 else b"") for i in range(40)))
    monkeypatch.setenv("KMAP_IO_THREADS", "1")
    monkeypatch.setenv("KMAP_FASTA_MIN_CHUNK", str(1 << 40))
    want, want_big = K.encode_fasta(str(GOLD / "test.fa")), K.encode_fasta(str(big))
    np.testing.assert_array_equal(want[0], s["seq"])
    ap, bp = K.encode_fasta_py(str(big))
    np.testing.assert_array_equal(want_big[0], ap)
    np.testing.assert_array_equal(want_big[1], bp)
    for threads, chunk in (("2", "1"), ("8", "1"), ("8", "100"), ("16", "4096"), ("3", "65536")):
        monkeypatch.setenv("KMAP_IO_THREADS", threads)
        monkeypatch.setenv("KMAP_FASTA_MIN_CHUNK", chunk)
        for f, w in ((GOLD / "test.fa", want), (big, want_big)):
            a, b = K.encode_fasta(str(f))
            np.testing.assert_array_equal(a, w[0], err_msg=f"{f.name} threads {threads} chunk {chunk}")
            np.testing.assert_array_equal(b, w[1], err_msg=f"{f.name} threads {threads} chunk {chunk}")


def test_merge_consensus_seqs_golden(golden):
    from kmap_amd.motif_discovery import merge_consensus_seqs
    g = golden("ops.npz")
    assert merge_consensus_seqs([str(x) for x in g["mcs_in"]]) == [str(x) for x in g["mcs_out"]] == ["CGTACGT", "CTAGGGG"]
    cands = [ln.split(",")[2] for ln in (GOLD / "scan_testfa" / "candidate_conseq.csv").read_text().splitlines()[1:]]
    assert merge_consensus_seqs(cands) == (GOLD / "scan_testfa" / "final_conseq.txt").read_text().splitlines()
    assert merge_consensus_seqs([]) == []


def test_motif_def_table_and_config(golden, tmp_path):
    import kmap_amd.kmer_count as K
    from kmap_amd._toml import dump_toml, load_toml
    cfg = K.read_default_config_file()
    ref_cfg = load_toml(GOLD / "scan_testfa" / "config.toml")
    for sec in ref_cfg:                                               # same sections and keys as the reference's config
        assert set(ref_cfg[sec]) - {"input_fasta_file"} <= set(cfg[sec]) | {"input_fasta_file"}
    assert cfg["kmer_count"] == {"min_k": 6, "max_k": 16, "revcom_mode": True}
    assert cfg["visualization"]["n_max_iter"] == 2500 and cfg["motif_discovery"]["n_total_sample"] == 5000
    dump_toml(cfg, tmp_path / "c.toml")
    assert load_toml(tmp_path / "c.toml") == cfg
    table = K.gen_motif_def_dict(cfg)
    g = golden("ops.npz")
    for k, cut in zip(g["mdef_k"], g["mdef_cutoff"]):
        np.testing.assert_allclose(table[int(k)].ratio_cutoff, cut, rtol=1e-12, equal_nan=True)
    assert table[8].max_ham_dist == 2 and table[14].max_ham_dist == 5


def test_cli_verbs_and_options():
    from click.testing import CliRunner
    from kmap_amd.cli import cli
    r = CliRunner().invoke(cli, ["--help"])
    assert r.exit_code == 0
    for verb in ("preproc", "scan_motif", "visualize_kmers"):
        assert verb in r.output
        h = CliRunner().invoke(cli, [verb, "--help"])
        assert h.exit_code == 0 and "--res_dir" in h.output
    assert "--fasta_file" in CliRunner().invoke(cli, ["preproc", "--help"]).output


def test_row_partition():
    from kmap_amd.distributed import row_partition
    for n, w in ((10, 3), (50000, 8), (7, 8), (0, 2), (200000, 8)):
        parts = [row_partition(n, w, r) for r in range(w)]
        assert parts[0][0] == 0 and sum(p[1] for p in parts) == n
        for (a0, an), (b0, _) in zip(parts, parts[1:]):
            assert a0 + an == b0
        assert max(p[1] for p in parts) - min(p[1] for p in parts) <= 1


def test_lut_expression_matches_matrix_expression(golden):
    """The LUT of all integer sums equals the reference's matrix expression evaluated entry-wise (same numpy)."""
    from kmap_amd.visualization import hd_prob_lut, sigmoid
    e = golden("embed_ops.npz")
    k, n_nb = int(e["kmer_len"]), int(e["n_nb"])
    S = e["S"]
    T = sigmoid(S, 16.0, change_point=k / 2, scale_factor=0.2 * k - 0.2)
    p_mat = np.exp(-T / 0.5).astype("float32")
    sums = np.rint(S.astype(np.float64) * n_nb * n_nb).astype(np.int64)
    np.testing.assert_array_equal(hd_prob_lut(k, n_nb, n_nb * n_nb * k)[sums], p_mat)


def test_hamdist_pitch_rule():
    """kmap_hamdist_pitch (pure host code in the C-ABI library): 256-byte rows below one 4-KiB column block, then a multiple
    of 4 KiB with an odd number of 4-KiB chunks per row (what the tiled Hamming kernel's XCD-affine chunk ownership needs)."""
    from kmap_amd.hamdist import pitch_for
    assert pitch_for(1) == 256 and pitch_for(300) == 512 and pitch_for(4095) == 4096
    for n in (4096, 4097, 8192, 8193, 50_000, 65_536, 70_704, 100_000, 141_424, 200_000):
        ld = pitch_for(n)
        assert ld >= n and ld % 4096 == 0 and (ld // 4096) % 2 == 1 and ld - n < 2 * 4096
    assert pitch_for(50_000) == 53_248


def test_dist_context_can_be_disabled(monkeypatch):
    """bench.py's rank 0 runs the C3 pipeline on its own while the other ranks wait: KMAP_DIST_DISABLE=1 keeps the verbs out of the
    process group (found by a 2-rank rehearsal: rank 0's scan_motif joined collectives nobody else entered)"""
    from kmap_amd.visualization import _dist_context
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("KMAP_DIST_DISABLE", "1")
    assert _dist_context() == (None, 0, False)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.delenv("KMAP_DIST_DISABLE")
    assert _dist_context() == (None, 0, False)


def test_locate_pickled_array_and_mapped_load(tmp_path):
    """load_array_pickle maps the payload of a default-protocol ndarray pickle (what preproc writes, reference kmer_count.py:333)
    and falls back to pickle.load for everything else; either way the values are the pickled ones."""
    import pickle
    from kmap_amd.kmer_count import load_array_pickle, locate_pickled_array
    rng = np.random.default_rng(5)
    arrays = [rng.integers(0, 256, 3_000_001, dtype=np.uint8), rng.integers(0, 1 << 40, (300_000, 2), dtype=np.int64),
              rng.integers(0, 1 << 32, 1_000_003, dtype=np.uint32)]
    for proto in (2, 3, 4, 5, None):
        for a in arrays:
            p = tmp_path / "a.pkl"
            with open(p, "wb") as fh:
                pickle.dump(a, fh) if proto is None else pickle.dump(a, fh, protocol=proto)
            loc = locate_pickled_array(p, 1 << 20)
            if proto in (3, 4, None):
                off, dt, shape = loc
                assert dt == a.dtype and shape == a.shape
                raw = p.read_bytes()[off:off + a.nbytes]
                assert raw == a.tobytes()
            else:
                assert loc is None                      # latin-1 text payload (2) / _frombuffer layout (5): ordinary unpickling
            b = load_array_pickle(p, 1 << 20)
            assert b.dtype == a.dtype and b.shape == a.shape and (b == a).all()
            if loc is not None and loc[0] % a.dtype.itemsize == 0:
                assert not b.flags.writeable and not b.flags.owndata      # a view of the mapping, not a copy
    # anything that is not ONE plain C-ordered array is left to pickle
    for obj in ([arrays[0], arrays[0]], np.asfortranarray(arrays[0][:3_000_000].reshape(2000, 1500)), {"a": arrays[0]},
                arrays[0][:1000]):
        p = tmp_path / "o.pkl"
        with open(p, "wb") as fh:
            pickle.dump(obj, fh)
        assert locate_pickled_array(p, 1 << 20) is None
        back = load_array_pickle(p, 1 << 20)
        assert type(back) is type(obj)
    # small files never take the mapped path (default threshold 64 MiB)
    with open(tmp_path / "s.pkl", "wb") as fh:
        pickle.dump(arrays[0], fh)
    assert load_array_pickle(tmp_path / "s.pkl").flags.owndata or load_array_pickle(tmp_path / "s.pkl").flags.writeable


def test_hashes2kmers_matches_scalar():
    from kmap_amd.kmer_count import hash2kmer, hashes2kmers
    for k in (1, 6, 8, 15, 16, 20, 31):
        h = np.random.default_rng(k).integers(0, 4 ** k, 500, dtype=np.uint64)
        a, b = hashes2kmers(h, k), np.array([hash2kmer(x, k) for x in h])
        assert a.dtype == b.dtype and (a == b).all()


@pytest.mark.parametrize("narrow", [False, True])
def test_occurrence_csv_writer_matches_python_formatting(tmp_path, narrow):
    """kmap_write_occurrence_csv{,_u8} (host code of the C-ABI library): rows `seq_ind;loc,loc;...;seq_len` for reads with a hit
    (reference motif_discovery.py:1396-1419), numbers on both sides of the 4- and 8-digit fast paths of the formatter"""
    import ctypes as C
    from kmap_amd import _ffi
    rng = np.random.default_rng(11)
    n_seq, n_cons = 150_000, 3
    read_len = rng.choice([7, 150, 9999, 10_000, 123_456, 99_999_999, 100_000_000, 3_000_000_000], n_seq).astype(np.int64)
    hits, pos = [], []
    for c in range(n_cons):
        h = rng.choice([0, 0, 0, 1, 2, 5, 20], n_seq).astype(np.int32)
        h[rng.integers(0, n_seq, 50)] = 0
        p = rng.choice([0, 9, 10, 99, 100, 999, 1000, 9999, 10_000, 10_001, 99_999, 12_345_678, 99_999_999, 100_000_000,
                        2_147_483_647], int(h.sum())).astype(np.int32)
        hits.append(h.astype(np.uint8) if narrow else h)
        pos.append(p)
    header = "seq_ind;" + ";".join(f"motif_{i}_ACGT" for i in range(n_cons)) + ";seq_len"
    out = tmp_path / "o.csv"
    rows = _ffi.i64(0)
    fn = _ffi.lib().kmap_write_occurrence_csv_u8 if narrow else _ffi.lib().kmap_write_occurrence_csv
    _ffi.check(fn(str(out).encode(), header.encode(), n_seq, n_cons, (C.c_void_p * n_cons)(*[h.ctypes.data for h in hits]),
                  (C.c_void_p * n_cons)(*[p.ctypes.data for p in pos]), _ffi.ptr(read_len), C.byref(rows)))
    offs = [np.concatenate([[0], np.cumsum(h, dtype=np.int64)]) for h in hits]
    want = [header]
    for i in np.nonzero(np.sum([h > 0 for h in hits], axis=0))[0]:
        cells = [",".join(str(v) for v in pos[c][offs[c][i]:offs[c][i + 1]]) for c in range(n_cons)]
        want.append(f"{i};" + ";".join(cells) + f";{read_len[i]}")
    assert rows.value == len(want) - 1
    assert out.read_text() == "\n".join(want) + "\n"


def test_cli_fast_exit_leaves_complete_output(tmp_path):
    """`kmap preproc` (no GPU needed) through the interpreter's normal shutdown (the default) and, with KMAP_FAST_EXIT=1, through the
    opt-in fast exit (os._exit after the flush): same exit code, same complete stdout through a pipe, byte-identical files; a failing
    run keeps click's exit code either way; with a tool preloaded (ROCPROFILER_* in the environment) the fast exit is not taken"""
    import os
    import subprocess
    import sys
    outs = {}
    for name, env in (("fast", {"KMAP_FAST_EXIT": "1"}), ("normal", {})):
        res = tmp_path / name
        e = dict(os.environ, PYTHONPATH=str(ROOT) + os.pathsep + os.environ.get("PYTHONPATH", ""), **env)
        e.pop("RANK", None)
        e.pop("WORLD_SIZE", None)
        r = subprocess.run([sys.executable, "-m", "kmap_amd", "preproc", "--fasta_file", str(GOLD / "test.fa"), "--res_dir", str(res)],
                           capture_output=True, text=True, env=e, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.rstrip().endswith("generated.")                    # the last line the verb prints made it through the pipe
        outs[name] = {f.name: f.read_bytes() for f in sorted(res.iterdir()) if f.name != "config.toml"}   # config.toml names its res_dir
        bad = subprocess.run([sys.executable, "-m", "kmap_amd", "preproc"], capture_output=True, text=True, env=e, timeout=300)
        assert bad.returncode == 2 and "Missing option" in bad.stderr
    assert outs["fast"].keys() == outs["normal"].keys() and len(outs["fast"]) == 3
    assert outs["fast"] == outs["normal"]
    # which way out was taken: an atexit hook (registered through sitecustomize-free -c code) runs only on the normal path
    code = ("import atexit, sys; atexit.register(lambda: sys.stderr.write('ATEXIT_RAN\\n')); import kmap_amd; "
            f"sys.argv = ['kmap', 'preproc', '--fasta_file', {str(GOLD / 'test.fa')!r}, '--res_dir', {str(tmp_path / 'x')!r}]; kmap_amd.main()")
    for env, ran in (({}, True), ({"KMAP_FAST_EXIT": "1"}, False), ({"KMAP_FAST_EXIT": "1", "ROCPROFILER_TEST": "1"}, True)):
        e = dict(os.environ, PYTHONPATH=str(ROOT) + os.pathsep + os.environ.get("PYTHONPATH", ""), **env)
        for k_ in ("RANK", "WORLD_SIZE", "LD_PRELOAD"):
            e.pop(k_, None)
        if "ROCPROFILER_TEST" not in env:
            e = {k_: v for k_, v in e.items() if not (k_.startswith("ROCPROFILER_") or k_ == "ROCP_TOOL_LIBRARIES")}
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("ATEXIT_RAN" in r.stderr) == ran, (env, r.stderr[-500:])


def test_co_occurrence_distance_lines_equal_python_formatting(tmp_path):
    """write_co_occurence_dist_arr formats natively (kmap_write_f2_tsv_line): the file must be what the reference's
    "\\t".join(f"{n:.2f}" ...) writes -- multiples of 0.5 (the values a run produces: the fast path), arbitrary doubles, ties of the
    rounding, huge / tiny / signed-zero / non-finite values; lists and arrays; an empty pair is skipped"""
    from kmap_amd.kmer_count import reverse_complement
    from kmap_amd.reports import write_co_occurence_dist_arr
    rng = np.random.default_rng(1)
    special = [0.0, -0.0, 0.005, 0.015, 0.025, -0.005, 0.125, 0.375, 999999.995, 1e15, -1e15, 1e16, 2.5e15, 2.0 ** 52 + 0.5, 1e300, -1e300,
               1e-320, float("inf"), float("-inf"), float("nan")]
    vals = np.concatenate([rng.integers(-300, 300, 200_000) / 2.0, rng.normal(size=70_000) * 100, rng.normal(size=1000) * 1e-3, special])
    conseqs = ["ACGT", "GGCC", "TTAA"]
    d = {(0, 1): vals, (0, 2): np.array([]), (1, 2): [1.0, 2.5, -3.25]}
    names = [f"m{i}_{s}_{reverse_complement(s)}" for i, s in enumerate(conseqs)]
    want = "".join(names[i] + "-" + names[j] + "\n" + "\t".join(f"{n:.2f}" for n in v) + "\n" for (i, j), v in d.items() if len(v))
    out = tmp_path / "dist.tsv"
    write_co_occurence_dist_arr(out, d, conseqs)
    assert out.read_text() == want


def _motif_table_by_pandas(path, p_value_cutoff=1e-10):
    """the reference's reader, verbatim in behaviour (kmer_count.py:719-740): pandas.read_csv + iterrows + scipy.stats.norm.ppf"""
    import pandas as pd
    from scipy.stats import norm
    out = {}
    for _, row in pd.read_csv(path).iterrows():
        k = int(row["kmer_len"])
        out[k] = (row["p_uniform"], int(row["max_ham_dist"]), row["ratio_mu"], row["ratio_std"],
                  norm.ppf(1 - p_value_cutoff, loc=row["ratio_mu"], scale=row["ratio_std"]))
    return out


def _same_float(a, b):
    a, b = float(a), float(b)
    return (a != a and b != b) or a.hex() == b.hex()


def test_motif_table_without_pandas_equals_the_reference_reader(tmp_path):
    """init_motif_def_dict reads tables of plain short decimals without importing pandas / scipy.stats (0.85 s of a fresh process):
    bit for bit the pandas + scipy.stats result on the packaged table, on the table `preproc` writes from it, and on random tables of
    that kind; anything pandas' parser could read differently (long numbers, exponents, odd NA spellings, spaces) is left to pandas"""
    import random
    import kmap_amd.kmer_count as K
    default = K._pkg_file(K.FileNameDict["default_motif_def_file"])
    written = tmp_path / "motif_def_table.csv"                       # as _preproc writes it: str() of every field, cut-offs included
    tab = K.init_motif_def_dict(default)
    written.write_text(K.MotifDef.get_field_names() + "\n" + "".join(str(tab[k]) + "\n" for k in sorted(k for k in tab if isinstance(k, int))))
    for path, cut in ((default, 1e-10), (written, 1e-10), (default, 0.05)):
        assert K._read_motif_table_plain(path) is not None
        got, want = K.init_motif_def_dict(path, p_value_cutoff=cut), _motif_table_by_pandas(path, cut)
        assert sorted(k for k in got if isinstance(k, int)) == sorted(want)
        for k, w in want.items():
            g = got[k]
            assert g.max_ham_dist == w[1] and all(_same_float(a, b) for a, b in zip((g.p_uniform, g.ratio_mu, g.ratio_std, g.ratio_cutoff),
                                                                                    (w[0], w[2], w[3], w[4]))), (path, k, g, w)
    rng = random.Random(3)
    n_plain = 0
    for it in range(300):
        lines = ["kmer_len,max_ham_dist,p_uniform,ratio_mu,ratio_std"]
        risky = it % 3 == 0
        for k in range(3, 3 + rng.randint(1, 12)):
            def num():
                r = rng.random()
                if r < 0.15:
                    return rng.choice(["", "nan"])
                if risky and r < 0.4:                       # what only pandas may read: many digits, exponents, other NA words, spaces
                    return rng.choice(["0.%017d" % rng.randint(1, 10 ** 17 - 1), "1.5e-3", "NA", " 0.5", "0.00000000468636788280", "1e-5"])
                d = rng.randint(1, 14)
                return "0." + "0" * rng.randint(0, 14 - d) + str(rng.randint(1, 10 ** d - 1)) if rng.random() < 0.7 else \
                    str(rng.randint(0, 999)) + "." + str(rng.randint(0, 10 ** 8))
            lines.append(f"{k},{rng.randint(0, 8)},{'0.' + str(rng.randint(1, 10 ** 9))},{num()},{num()}")
        p = tmp_path / f"t{it}.csv"
        p.write_text("\n".join(lines) + "\n")
        plain = K._read_motif_table_plain(p)
        n_plain += plain is not None
        try:
            want = _motif_table_by_pandas(p)
        except Exception:                                   # noqa: BLE001 -- a table pandas itself rejects (e.g. " 0.5" as text)
            assert plain is None
            continue
        got = K.init_motif_def_dict(p)
        for k, w in want.items():
            g = got[k]
            assert g.max_ham_dist == w[1] and all(_same_float(a, b) for a, b in zip((g.p_uniform, g.ratio_mu, g.ratio_std, g.ratio_cutoff),
                                                                                    (w[0], w[2], w[3], w[4]))), (it, k, g, w, lines)
    assert n_plain >= 150                                   # the plain reader did take the tables meant for it
    # quoted fields are pandas' business: the plain reader declines, the result is still the pandas one
    q = tmp_path / "quoted.csv"
    q.write_text('kmer_len,max_ham_dist,p_uniform,ratio_mu,ratio_std\n8,2,"0.0042",3.5,"0.25"\n9,2,0.0011,"3.25",0.5\n')
    assert K._read_motif_table_plain(q) is None
    got, want = K.init_motif_def_dict(q), _motif_table_by_pandas(q)
    for k, w in want.items():
        g = got[k]
        assert all(_same_float(a, b) for a, b in zip((g.p_uniform, g.ratio_mu, g.ratio_std, g.ratio_cutoff), (w[0], w[2], w[3], w[4])))


def test_norm_functions_equal_scipy_stats():
    """norm_ppf / norm_logsf (scipy.special behind them) against scipy.stats.norm for scalars: bit-identical, NaN where scipy says NaN"""
    from scipy.stats import norm
    import kmap_amd.kmer_count as K
    rng = np.random.default_rng(0)
    for _ in range(20_000):
        mu, sd = rng.normal() * 3, abs(rng.normal()) * 2 + 1e-3
        x = mu + rng.normal() * sd * rng.choice([0.1, 1, 5, 20, 60])
        p = rng.random() * rng.choice([1, 1e-3, 1e-6, 1e-10])
        assert _same_float(K.norm_logsf(x, mu, sd), norm.logsf(x, loc=mu, scale=sd))
        assert _same_float(K.norm_ppf(1 - p, mu, sd), norm.ppf(1 - p, loc=mu, scale=sd))
    nan, inf = float("nan"), float("inf")
    for q, mu, sd in ((0.5, 1.0, 0.0), (0.5, 1.0, -1.0), (0.5, nan, 1.0), (0.5, 1.0, nan), (0.0, 1.0, 2.0), (1.0, 1.0, 2.0), (1.5, 0.0, 1.0),
                      (-0.1, 0.0, 1.0), (1 - 1e-10, 1.0, 0.05)):
        assert _same_float(K.norm_ppf(q, mu, sd), norm.ppf(q, loc=mu, scale=sd)), (q, mu, sd)
    for x, mu, sd in ((1.0, 0.0, 0.0), (1.0, 0.0, -2.0), (nan, 0.0, 1.0), (1.0, nan, 1.0), (1.0, 0.0, nan), (inf, 0.0, 1.0), (-inf, 0.0, 1.0),
                      (40.0, 1.0, 0.05), (1.3, 1.0, 0.05)):
        assert _same_float(K.norm_logsf(x, mu, sd), norm.logsf(x, loc=mu, scale=sd)), (x, mu, sd)


def test_fasta_file_changed_between_open_and_read_is_an_error(tmp_path):
    """kmap_fasta_open counts the mapped file, kmap_fasta_read encodes it into arrays of exactly that size: a file rewritten in between
    (more sequence bytes / other header lines than counted) must end in an error, never in a write past the caller's arrays"""
    import ctypes as C
    from kmap_amd import _ffi
    lib = _ffi.lib()
    text = b"".join(b">r%d\nACGTACGTAC\n" % i for i in range(20000))
    for new in (b"A" * len(text), b">x\n" * (len(text) // 3) + b"\n" * (len(text) % 3), text.replace(b">r19999\n", b"ACGTACG\n")):
        assert len(new) == len(text)
        p = tmp_path / "c.fa"
        p.write_bytes(text)
        h, nb, ns = C.c_void_p(), C.c_int64(0), C.c_int64(0)
        assert lib.kmap_fasta_open(str(p).encode(), C.byref(h), C.byref(nb), C.byref(ns)) == 0
        with open(p, "r+b") as fh:                      # same length, other content: the private mapping shows the new bytes
            fh.write(new)
        guard = 4096
        seq = np.full(nb.value + guard, 77, np.uint8)
        borders = np.full((ns.value + 8, 2), -7, np.int64)
        rc = lib.kmap_fasta_read(h, seq.ctypes.data_as(C.c_void_p), borders.ctypes.data_as(C.c_void_p))
        lib.kmap_fasta_close(h)
        assert rc != 0 and b"not what fasta_open counted" in lib.kmap_last_error()
        assert (seq[nb.value:] == 77).all() and (borders[ns.value:] == -7).all()      # nothing behind the caller's arrays was touched


def test_dump_array_pickle_is_pickle_dump_byte_for_byte(tmp_path):
    """preproc writes its two pickles without the `arr.tobytes()` copy numpy's __reduce__ makes under protocol 4: the same bytes as
    `pickle.dump(arr, fh)` (what the reference writes, kmer_count.py:333,341) for the layouts it takes itself, pickle.dump for the rest;
    the mapped loader finds the payload"""
    import io
    import pickle
    import kmap_amd.kmer_count as K
    rng = np.random.default_rng(0)
    cases = [rng.integers(0, 5, 3_000_000, dtype=np.uint8), rng.integers(0, 1 << 40, (200_000, 2), dtype=np.int64),
             rng.integers(0, 255, 1 << 20, dtype=np.uint8), rng.integers(0, 9, (1 << 20) + 1, dtype=np.uint8)[1:],          # offset view
             rng.integers(0, 255, 1000, dtype=np.uint8), np.zeros(0, np.uint8),                                                 # small: pickle.dump
             np.asfortranarray(rng.integers(0, 9, (2000, 1000), dtype=np.int64)), rng.random(300_000),                           # other layouts / dtypes
             rng.integers(0, 9, 2_000_000, dtype=np.uint8)[::2]]
    for a in cases:
        want, got = io.BytesIO(), io.BytesIO()
        pickle.dump(a, want)
        K.dump_array_pickle(a, got)
        assert got.getvalue() == want.getvalue(), (a.dtype, a.shape)
    m, lab = rng.integers(0, 9, (1200, 1200)).astype(np.int64), rng.integers(0, 3, 1200).astype(np.int64)
    containers = [[8, m, lab], [8, None, rng.integers(0, 3, 200_000).astype(np.int64)], [8, m, m], (m,), [1, 2, "x"],
                  [rng.integers(0, 1 << 16, 300_000).astype(np.uint32), rng.integers(1, 9, 300_000).astype(np.int64),
                   rng.integers(0, 3, 300_000).astype(np.int64), ["ACGTACGT", "GGGTTTAA"]]]            # the shapes scan_motif writes
    for o in containers:
        for proto in (None, 4, 5, 3):
            want, got = io.BytesIO(), io.BytesIO()
            pickle.dump(o, want, **({} if proto is None else {"protocol": proto}))
            K.dump_pickle_nocopy(o, got, protocol=proto)
            assert got.getvalue() == want.getvalue(), (type(o), proto)
    p = tmp_path / "input.bin.pkl"
    with open(p, "wb") as fh:
        K.dump_array_pickle(cases[0], fh)
    assert K.locate_pickled_array(p) is not None
    np.testing.assert_array_equal(K.load_array_pickle(p), cases[0])


def test_bench_line_is_compact_strict_json(tmp_path, monkeypatch):
    """the ONE stdout line of bench.py is <= 4 KB strict JSON with the contract keys, `config`, `roofline`, `cpu_baseline` and
    the rank counts, at N = 1 and at N = 8, whatever the legs put into the record (long prose, 8 per-rank identities, NaN in a
    leg); everything else goes to the detail file, which is strict JSON too"""
    import importlib
    import json
    import sys
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    for world in (1, 8):
        line = {"metric": "hamming_pairs_per_s", "value": 6.1e12 * world, "unit": "pairs/s", "n_gpus": world, "steps": 20, "warmup": 5, "ms_per_step": 0.41,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": "C3 Hamming stage " + "x" * 900, "n_kmers": 50000, "k": 8, "final_conseq": ["CCTACGTA", "ATCGATA"], "conseq_lens": [8, 7],
                           "rows_of_short_consensus_labels": 8733, "rows_per_gpu": 50000 // world, "pairs_per_gpu_per_step": 2.5e9, "parallelism": "row blocks " + "y" * 300,
                           "sample": "s" * 500},
                "roofline": {"bound": "hbm", "achieved": 6144.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.768, "traffic": 2503029091.2, "achievable": {"a": 1},
                             "frac_of_achievable": 0.94, "kernel": "hamdist_tile_kernel" + "k" * 400, "kernel_ms": 0.4069, "algorithmic_bytes": 2500250000,
                             "clock_ramp": "c" * 2000, "stages": {f"stage{i}": {"what": "w" * 300, "frac": float("nan")} for i in range(12)}},
                "cpu_baseline": {"value": 1.0e10, "unit": "pairs/s", "cores": 16, "kind": "port", "cpu": {"model": "AMD EPYC 9575F 64-Core Processor"},
                                 "sample": "z" * 3000, "e2e": {"extrapolated_to_c3": {"total_s": {"all_cores": 1234.5}}}},
                "ranks_seen": world, "distinct_gpus": world, "dist_backend": "nccl", "rccl_version": "2.26.6",
                "ranks": [{"rank": r, "device_name": "AMD Instinct MI355X", "uuid": "u" * 40, "pci_bus": "0000:05:00"} for r in range(world)],
                "e2e": {"k6_9": {"default": {"e2e_s": 4.5, "stages": {f"s{i}": 0.1 for i in range(40)}}, "fast": {"e2e_s": 2.7}}, "workload": "p" * 4000},
                "c5": {"frac": 0.5, "ms_median": 3.9, "what": "q" * 800}, "shard_proxy": {"e2e_predicted_s": 1.3, "stages": {f"p{i}": {"shard_ms": [1.0] * 8} for i in range(20)}},
                "embed_dist": {"ms_per_iteration": 0.9, "seq": {"ms_per_iteration": float("inf")}}, "skipped_legs": ["c4.e2e"], "leg_errors": []}
        text = bench.compact_line(line, f"gpurun_out/bench_detail_n{world}.json")
        assert len(text.encode()) <= 4096 and "\n" not in text
        back = json.loads(text, parse_constant=lambda c: pytest.fail(f"non-finite constant {c} in the line"))
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                    "config", "roofline", "cpu_baseline", "ranks_seen", "distinct_gpus", "rccl_version", "detail"):
            assert key in back, key
        assert back["n_gpus"] == world and back["ranks_seen"] == world and back["config"]["n_kmers"] == 50000
        assert set(back["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes"}
        assert set(back["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample", "model"}
        assert back["e2e_c3_s"] == 4.5 and back["c5_frac"] == 0.5 and back["e2e_predicted_s_g8"] == 1.3 and back["skipped_legs"] == "c4.e2e"
        assert "embed_seq_ms_per_iter" not in back or back["embed_seq_ms_per_iter"] is not None      # the inf of a leg never reaches the line
        rel = bench.write_detail(line, world)
        det = json.loads((tmp_path / rel).read_text(), parse_constant=lambda c: pytest.fail(f"non-finite constant {c} in the detail file"))
        assert det["roofline"]["stages"]["stage0"]["frac"] is None and len(det["ranks"]) == world


def test_bench_budget_skips_legs_that_do_not_fit(monkeypatch, capsys):
    """bench.py's ONE deadline: a leg starts only if its expected cost fits what is left of --time-budget; a skipped leg is named; no
    watchdog thread at N = 1"""
    import importlib
    import sys
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    now = [100.0]
    monkeypatch.setattr(bench.time, "perf_counter", lambda: now[0])
    monkeypatch.setattr(bench, "T_START", 100.0)
    b = bench.Budget(60.0, 10.0, 1, 0, 1)
    assert b.timer is None and abs(b.left() - 60.0) < 1e-9
    assert b.can("a", 50.0) and b.name == "a"
    now[0] = 130.0
    assert not b.can("b", 31.0) and b.skipped == ["b"]
    assert b.can("c", 30.0)
    now[0] = 170.0
    assert not b.can("d", 0.5) and b.skipped == ["b", "d"] and b.left() < 0
    b.done()
    assert "SKIPPED" in capsys.readouterr().err
