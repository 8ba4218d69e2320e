"""Host side of `kmap scan_motif`: the orchestration of the reference's motif_discovery.py
(/root/reference/src/kmap/motif_discovery.py) with every array stage on the GPU.

The sequence array and read borders are uploaded once (DeviceSeq); per k the fused device path does
hash -> per-read dedupe -> histogram count -> revcom merge, Hamming-ball mass, masking and the
occurrence scan without round-tripping hashes through the host.  File contracts (candidate_conseq.csv,
final_conseq*.{txt,csv}, *.motif_occurence.csv, k{k}.pkl, sample_kmers.{pkl,tsv},
sample_kmer_hamdist_mat.pkl) and the caching rules of the reference are kept.  Plot / report branches
(position density, co-occurrence, logos, consensus alignment) are outside this package's scope and are
skipped with a notice.
"""
import ctypes as C
import os
import pickle
import sys
import time
import warnings
from pathlib import Path
from typing import List

import numpy as np

from . import _ffi
from ._ffi import check, ptr
from .hamdist import _convert_to_block_arr, cal_samp_kmer_hamdist_mat  # noqa: F401  (re-exported, reference names)
from .kmer_count import (DeviceCounts, FileNameDict, cal_hamming_dist_head, cal_hamming_dist_tail, dump_pickle_nocopy, encode_fasta,
                         gen_motif_def_dict, get_cnt_dtype, get_hash_dtype, get_revcom_hash_arr, hash2kmer, init_motif_def_dict, norm_logsf,
                         hashes2kmers, kmer2hash, load_array_pickle,
                         mask_ham_ball, revcom_hash, reverse_complement)

# int64 N x N pickle is kept up to this many sampled k-mers (2 GiB); above it scan_motif writes the compact
# hand-off [kmer_len, None, label_arr] and visualize_kmers recomputes the matrix on the device
DENSE_PKL_MAX_N = 16384
# above this many unique k-mers find_motif stops fetching the count arrays every trial: the top_k candidates come from
# the device (largest count, then lowest index) instead of np.argpartition (whose tie order is numpy-specific anyway)
TOPK_DEVICE_MIN = 4_000_000
SAVE_ASYNC_MIN = 4_000_000   # tables above this many unique k-mers are fetched + pickled by a background TableSaver
TOPK_DEVICE_MAX_K = 16       # kmap_counts_topk keeps 16 candidates per thread; a larger top_k takes the host path at any size



def _device_topk(n_uniq, top_k):
    """find_motif's candidates from kmap_counts_topk (largest count, then lowest index) instead of np.argpartition on the
    fetched table (reference motif_discovery.py:661)?  Never under KMAP_EXACT=1 / config general.exact."""
    from . import _policy
    return n_uniq > TOPK_DEVICE_MIN and top_k <= TOPK_DEVICE_MAX_K and not _policy.exact()


STAGE_TIMES = {}   # cumulative wall-clock per stage of the last runs (tools/e2e.py, bench.py report it)


class _stage:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        import time
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        import time
        _ffi.sync()
        STAGE_TIMES[self.name] = STAGE_TIMES.get(self.name, 0.0) + time.perf_counter() - self.t0


def write_lines(str_list: List, outfile):
    with open(outfile, "w+") as fh:
        for line in str_list:
            fh.write(line + "\n")


# ---- device-resident sequence array ----------------------------------------------------------------
class DeviceSeq:
    """The encoded reads resident in HBM as 2-bit codes + invalid bitmask (packed.hip), plus the (n_seq, 2) borders.
    `inval_orig` is the pristine mask, `inval_work` the one find_motif masks; the codes are shared."""

    def __init__(self, seq_np_arr, boarder_mat, _device_arrays=None):
        if _device_arrays is not None:           # from_device(): the uint8 array and the borders already lie in HBM
            raw, self.n, self.borders, self.n_seq, fixed_len = _device_arrays
            self.borders_host = None
            self.read_len = np.full(self.n_seq, fixed_len, np.int64) if fixed_len is not None else None
        else:
            seq = np.ascontiguousarray(seq_np_arr, dtype=np.uint8)
            self.n = len(seq)
            self.borders_host = np.ascontiguousarray(boarder_mat, dtype=np.int64).reshape(-1, 2)
            self.n_seq = len(self.borders_host)
            self.borders = _ffi.DeviceBuffer.from_numpy(self.borders_host)
            self.read_len = (self.borders_host[:, 1] - self.borders_host[:, 0]).astype(np.int64)
            raw = _ffi.DeviceBuffer.from_numpy(seq) if self.n else _ffi.DeviceBuffer(16)
        # the reads scan() results cover: these reads here; ALL reads for a read-sharded DistDeviceSeq (distributed.py)
        self.out_n_seq, self.out_read_len = self.n_seq, self.read_len
        self.groups = int(_ffi.lib().kmap_packed_groups(self.n))
        self.codes = _ffi.DeviceBuffer(self.groups * 4)
        self.inval_orig = _ffi.DeviceBuffer(self.groups * 2)
        self.inval_work = _ffi.DeviceBuffer(self.groups * 2)
        check(_ffi.lib().kmap_pack_reads_dev(raw.ptr, self.n, self.codes.ptr, self.inval_orig.ptr, None))
        # bit planes of the codes (0.25 B / position more): the scans and the masking test all windows bit-sliced on them
        self.planes = _ffi.DeviceBuffer(self.groups * 4)
        check(_ffi.lib().kmap_pack_planes_dev(self.codes.ptr, self.n, self.planes.ptr, None))
        _ffi.sync()
        raw.free()                      # the uint8 array does not stay on the device
        self._layout = False            # not looked at yet (_uniform_layout)
        self.reset()
        self._scan = None
        dev = _ffi.i32(0)
        check(_ffi.lib().kmap_get_device(C.byref(dev)))
        self.device = dev.value         # HIP's current device is per thread: worker threads that fetch hit lists select it
        import threading
        self._lazy_lock, self._lazy_free, self._lazy_all = threading.Lock(), [], []     # scan handles of scan_lazy()

    @classmethod
    def from_device(cls, raw_u8, n, borders_dev, n_seq, fixed_read_len=None):
        """the reads already in HBM (raw_u8: DeviceBuffer with the uint8 array contract, consumed -- freed once packed;
        borders_dev: DeviceBuffer int64[n_seq][2]), e.g. from synth.synth_reads_dev; fixed_read_len: every read's length, if
        the caller knows it (the occurrence CSV needs the lengths on the host)"""
        return cls(None, None, _device_arrays=(raw_u8, int(n), borders_dev, int(n_seq), fixed_read_len))

    def reset(self):
        """restore the unmasked reads (reference motif_discovery.py:263): n/8 bytes"""
        check(_ffi.lib().kmap_memcpy_d2d(self.inval_work.ptr, self.inval_orig.ptr, self.groups * 2, None))

    def count(self, dc: DeviceCounts, k, dedupe, merge_revcom, use_work=True, gather_full=False):
        inval = self.inval_work if use_work else self.inval_orig
        nu = _ffi.i64(0)
        dc._unshard()
        check(_ffi.lib().kmap_counts_run_packed_dev(dc._h, self.codes.ptr, inval.ptr, self.n, self.borders.ptr, self.n_seq, k,
                                                    int(dedupe), int(merge_revcom), C.byref(nu), None))
        dc.k, dc.n_uniq = k, nu.value
        return dc.n_uniq

    def count_range(self, dc: DeviceCounts, k, dedupe, merge_revcom, first_bin, n_bins, use_work=True):
        """positions [first_bin, first_bin + n_bins) -- in key order -- of the table count() would produce, from the windows that
        decide them alone (kmap_counts_run_packed_range_dev, 11 <= k <= 16): a rank's share of a key-space-sharded count"""
        inval = self.inval_work if use_work else self.inval_orig
        nu = _ffi.i64(0)
        dc._unshard()
        check(_ffi.lib().kmap_counts_run_packed_range_dev(dc._h, self.codes.ptr, inval.ptr, self.n, self.borders.ptr, self.n_seq, k,
                                                          int(dedupe), int(merge_revcom), int(first_bin), int(n_bins), C.byref(nu), None))
        dc.k, dc.n_uniq = k, nu.value
        return dc.n_uniq

    def mask(self, k, consensus_kh_arr, max_ham_dist_arr):
        cons = np.ascontiguousarray(consensus_kh_arr, dtype=np.uint64)
        rad = np.ascontiguousarray(max_ham_dist_arr, dtype=np.int32)
        check(_ffi.lib().kmap_mask_hamball_packed_dev(self.codes.ptr, self.inval_work.ptr, self.n, k, ptr(cons), ptr(rad),
                                                      len(cons), self.planes.ptr, None))

    def download(self):
        """the working reads as the reference's uint8 array (masked positions = 255)"""
        out_d = _ffi.DeviceBuffer(max(self.n, 1))
        check(_ffi.lib().kmap_unpack_reads_dev(self.codes.ptr, self.inval_work.ptr, self.n, out_d.ptr, None))
        out = out_d.to_numpy(np.uint8, (self.n,))
        out_d.free()
        return out

    def _uniform_layout(self):
        """(read_len, stride) when read s is [s * stride, s * stride + read_len) -- fixed-length reads -- else None; from the host
        borders (one vectorised comparison, once), or the caller's fixed length for reads that only exist in HBM"""
        if self._layout is False:
            self._layout = None
            bh = self.borders_host
            if bh is not None and len(bh) >= 1 and bh[0, 0] == 0:
                ln = int(bh[0, 1] - bh[0, 0])
                stride = int(bh[1, 0] - bh[0, 0]) if len(bh) > 1 else ln + 1
                if stride >= max(ln, 1):
                    idx = np.arange(len(bh), dtype=np.int64) * stride
                    if np.array_equal(bh[:, 0], idx) and np.array_equal(bh[:, 1], idx + ln):
                        self._layout = (ln, stride)
            elif bh is None and self.read_len is not None and self.n_seq >= 1:
                ln = int(self.read_len[0])
                self._layout = (ln, ln + 1)         # synth_reads_dev: a separator behind every read (verified on the device below)
        return self._layout

    def declare_layout(self, handle):
        """tell a scan handle that these reads are laid out uniformly (kmap_scan_declare_uniform verifies it on the device): its runs on
        these borders then derive them from the read index instead of loading 16 bytes per read"""
        lay = self._uniform_layout()
        if lay is None:
            return False
        ok = _ffi.i32(0)
        check(_ffi.lib().kmap_scan_declare_uniform(handle, self.borders.ptr, self.n_seq, lay[0], lay[1], C.byref(ok), None))
        return bool(ok.value)

    def scan(self, k, consensus_kh, radius, revcom):
        """positions at each read's minimum hit distance (original, unmasked reads):
        returns (hits_per_read int32[n_seq], positions int32[total])."""
        if self._scan is None:
            h = _ffi.vp()
            check(_ffi.lib().kmap_scan_create(C.byref(h)))
            self._scan = h.value
            self.declare_layout(self._scan)
        tot = _ffi.i64(0)
        check(_ffi.lib().kmap_scan_run_packed_dev(self._scan, self.codes.ptr, self.inval_orig.ptr, self.n, self.borders.ptr,
                                                  self.n_seq, k, int(consensus_kh), int(radius), int(revcom), C.byref(tot),
                                                  self.planes.ptr, None))
        hits = np.empty(self.n_seq, np.int32)
        pos = np.empty(tot.value, np.int32)
        check(_ffi.lib().kmap_scan_fetch(self._scan, ptr(hits), None, ptr(pos)))   # per-read minimum distances stay on the device
        return hits, pos

    def scan_lazy(self, k, consensus_kh, radius, revcom):
        """scan() whose hit list stays in HBM, inside its scan handle, until someone asks for it: `ScanHits.n_reads_hit / .total /
        .max_hits` are known at once (what scan_motif's candidate table needs), the two arrays are fetched on first use -- by the
        background CSV writer in scan_motif, off the critical path.  Handles rotate: a fetched (or dropped) ScanHits hands its
        handle back, so a run allocates a handful of result buffers once instead of one set per consensus."""
        with self._lazy_lock:
            h = self._lazy_free.pop() if self._lazy_free else None
        if h is None:
            hv = _ffi.vp()
            check(_ffi.lib().kmap_scan_create(C.byref(hv)))
            h = hv.value
            self.declare_layout(h)
            self._lazy_all.append(h)
        tot, nhit, mx = _ffi.i64(0), _ffi.i64(0), _ffi.i32(0)
        check(_ffi.lib().kmap_scan_run_packed_dev(h, self.codes.ptr, self.inval_orig.ptr, self.n, self.borders.ptr,
                                                  self.n_seq, k, int(consensus_kh), int(radius), int(revcom), C.byref(tot),
                                                  self.planes.ptr, None))
        check(_ffi.lib().kmap_scan_summary(h, C.byref(nhit), C.byref(mx), None))    # returns once the lists are complete
        return ScanHits(self, h, self.n_seq, tot.value, nhit.value, mx.value)

    def _lazy_release(self, h):
        with self._lazy_lock:
            if self._lazy_all is not None:
                self._lazy_free.append(h)

    def close(self):
        if self._scan:
            _ffi.lib().kmap_scan_destroy(self._scan)
            self._scan = None
        with self._lazy_lock:
            handles, self._lazy_all, self._lazy_free = self._lazy_all or [], None, []
        for h in handles:                 # a ScanHits not fetched by now reports that its sequence is closed
            _ffi.lib().kmap_scan_destroy(h)
        for b in (self.codes, self.planes, self.inval_orig, self.inval_work, self.borders):
            b.free()


class ScanHits:
    """One consensus' hit list, resident in HBM (in its scan handle) until first use.  Unpacks like the (hits_per_read, positions)
    pair scan() returns (`hits, pos = scan_hits` fetches); the summary numbers need no fetch."""

    def __init__(self, owner, handle, n_seq, total, n_reads_hit, max_hits):
        import threading
        self._owner, self._handle = owner, handle
        self.n_seq, self.total, self.n_reads_hit, self.max_hits = n_seq, total, n_reads_hit, max_hits
        self._host = None
        self._lock = threading.Lock()

    def host(self):
        with self._lock:
            if self._host is None:
                if self._handle is None:
                    raise RuntimeError("ScanHits: the list was already handed to a CSV writer (host_u8)")
                if self._owner._lazy_all is None:
                    raise RuntimeError("ScanHits: the DeviceSeq was closed before the hit list was fetched")
                hits, pos = np.empty(self.n_seq, np.int32), np.empty(self.total, np.int32)
                check(_ffi.lib().kmap_set_device(self._owner.device))  # this may be a CSV writer thread (fresh threads start on device 0)
                st = _ffi.vp()
                check(_ffi.lib().kmap_stream_create(C.byref(st)))      # own stream: neither waits for nor blocks the launching thread
                try:
                    check(_ffi.lib().kmap_scan_fetch_stream(self._handle, ptr(hits), ptr(pos), st.value))
                finally:
                    _ffi.lib().kmap_stream_destroy(st.value)
                self._owner._lazy_release(self._handle)
                self._owner, self._handle = None, None
                self._host = [hits, pos]
            return self._host

    @property
    def unfetched(self):
        return self._host is None and self._handle is not None

    def host_u8(self):
        """(hits as uint8, positions) for a list with max_hits <= 255, fetched without keeping the int32 pair; the handle goes
        back to its sequence, so this is the list's last use"""
        with self._lock:
            assert self._host is None and self._handle is not None and self.max_hits <= 255
            if self._owner._lazy_all is None:
                raise RuntimeError("ScanHits: the DeviceSeq was closed before the hit list was fetched")
            hits, pos = np.empty(self.n_seq, np.uint8), np.empty(self.total, np.int32)
            check(_ffi.lib().kmap_set_device(self._owner.device))
            st = _ffi.vp()
            check(_ffi.lib().kmap_stream_create(C.byref(st)))
            try:
                check(_ffi.lib().kmap_scan_fetch_stream_u8(self._handle, ptr(hits), ptr(pos), st.value))
            finally:
                _ffi.lib().kmap_stream_destroy(st.value)
            self._owner._lazy_release(self._handle)
            self._owner, self._handle = None, None
            return hits, pos

    def __del__(self):
        try:
            if self._handle is not None:
                self._owner._lazy_release(self._handle)
        except Exception:     # noqa: BLE001 -- interpreter shutdown
            pass

    def __iter__(self):
        return iter(self.host())

    def __getitem__(self, i):
        return self.host()[i]

    def __len__(self):
        return 2


# ---- consensus merging (reference motif_discovery.py:533-591) -------------------------------------
def merge_consensus_seqs(conseq_list: List[str]) -> List[str]:
    """A candidate of length L survives (as its (L-1)-mer relative) only if candidates of length L-1 and L-2
    overlap it (or its reverse complement) up to a one-base shift; everything it covers is then dropped."""

    def covers(long_kmer, short_kmer):
        return short_kmer[:-1] in long_kmer or short_kmer[1:] in long_kmer

    pending = sorted(conseq_list, key=len, reverse=True)
    finals = []
    while pending:
        cur = pending[0]
        rc = reverse_complement(cur)

        def related(s):
            return covers(cur, s) or covers(rc, s)

        one = next((s for s in pending if len(s) == len(cur) - 1 and related(s)), None)
        two = next((s for s in pending if len(s) == len(cur) - 2 and related(s)), None)
        if one and two:
            finals.append(one)
            pending = [s for s in pending if not related(s)]
        else:
            pending = pending[1:]
    return finals


# ---- k{k}.pkl writer for multi-GB tables ----------------------------------------------------------------
class _PickleLayout:
    """write() sink for a dry run of pickle.Pickler(protocol=5): records where the pickler would put the large array payloads
    (it hands them to write() one call each, without copying) and keeps the small pieces, without touching the payload bytes."""

    def __init__(self, big_min=1 << 16):       # the C pickler writes payloads of >= 64 KiB (its frame size target) directly
        self.pos, self.small, self.big, self.big_min = 0, [], [], big_min

    def write(self, data):
        n = memoryview(data).nbytes
        if n >= self.big_min:
            self.big.append((self.pos, n))
        else:
            self.small.append((self.pos, bytes(data)))
        self.pos += n
        return n


class TableSaver:
    """Background save of one k's count table as k{k}.pkl = pickle([k, uniq, cnt]) while the caller goes on counting into
    another handle.  The table is streamed from a DeviceCounts handle that nobody writes any more: a dry run of the real
    pickler over untouched np.empty arrays of the right shape gives the byte layout (where the two array payloads go, and the
    small pieces around them); the payloads are then fetched chunk by chunk -- own stream, pinned staging, conversion to the
    small pieces around them); the file is preallocated, and the payloads then go device -> pinned staging -> pwrite inside ONE native
    call each (kmap_counts_write_range: own stream, the next chunk crossing PCIe while the current one is written, counts widened
    to the reference's int64 on the device) -- no pageable copy and no conversion pass on the host.
    The file is byte for byte what pickle.dump writes, without ever holding the 15 GB of a k = 16 table (C3) in host memory
    (fetching it whole, pickling and freeing it cost 2.5 s; measured alternatives: pwrite from several threads -- no gain, one
    inode lock: tools/probes/file_write_rate.py; filling a memory-mapped file -- 4x slower, page faults through overlayfs).
    join() re-raises a failure on the caller's thread; the handle (the table resident in HBM) stays usable until close()."""
    CHUNK_BYTES = 128 << 20

    def __init__(self, dc, kmer_len, path, device=None):
        import threading
        self.dc, self.k, self.path, self.err = dc, kmer_len, path, None
        self._dev = device
        self._t = threading.Thread(target=self._run)
        self._t.start()

    def _run(self):
        import time
        try:
            t0 = time.perf_counter()
            lib = _ffi.lib()
            if self._dev is not None:
                check(lib.kmap_set_device(self._dev))              # HIP's current device is per thread
            n, dts = self.dc.n_uniq, (get_hash_dtype(self.k), get_cnt_dtype(self.k))
            lay = _PickleLayout()
            pickle.Pickler(lay, protocol=5).dump([self.k, np.empty(n, dts[0]), np.empty(n, dts[1])])   # layout only: pages never touched
            # written under a temporary name and renamed when complete: a killed / failed run must not leave a truncated
            # k{k}.pkl behind that the next run's "already exists" branches would load
            tmp = str(self.path) + ".tmp"
            if [b for _, b in lay.big] != [n * np.dtype(d).itemsize for d in dts]:
                u, c = self.dc.fetch()                             # small table (payloads inside a frame): plain dump
                with open(tmp, "wb") as fh:
                    pickle.dump([self.k, u, c], fh, protocol=5)
                os.replace(tmp, self.path)
                return
            st = _ffi.vp()
            check(lib.kmap_stream_create(C.byref(st)))
            fd = os.open(tmp, os.O_CREAT | os.O_WRONLY | os.O_TRUNC, 0o644)
            try:
                # the size is known: reserve the extents first (one file takes ~10.5 GB/s of buffered writes on the test boxes however
                # many threads write it -- the inode lock --, 12.5 GB/s into preallocated extents: tools/probes/file_write_rate.py)
                try:
                    os.posix_fallocate(fd, 0, lay.pos)
                except OSError:      # a file system without fallocate: plain extending writes
                    pass
                for pos, small in lay.small:
                    os.pwrite(fd, small, pos)
                # the two array payloads: device -> pinned staging -> pwrite, natively (no pageable copy, no conversion pass on the
                # host: the counts are widened on the device), the next chunk crossing PCIe while the current one is written
                for which, (pos, _) in enumerate(lay.big):
                    check(lib.kmap_counts_write_range(self.dc._h, which, 0, n, fd, pos, st.value))
                os.ftruncate(fd, lay.pos)
            finally:
                os.close(fd)
                lib.kmap_stream_destroy(st.value)
            os.replace(tmp, self.path)
            STAGE_TIMES[f"bg_save_k{self.k}"] = time.perf_counter() - t0    # background: overlaps the main thread's stages
        except BaseException as e:   # noqa: BLE001 -- re-raised by join()
            self.err = e
            try:
                os.unlink(str(self.path) + ".tmp")
            except OSError:
                pass

    def join(self):
        self._t.join()
        if self.err is not None:
            raise self.err

    def finished_ok(self):
        return not self._t.is_alive() and self.err is None

    def close(self):
        self._t.join()
        self.dc.close()


# ---- find_motif (reference motif_discovery.py:594-702) -----------------------------------------------
def _wrap_total(total, k):
    """`sum(uniq_kh_cnt_arr)` in the reference accumulates numpy scalars of the count dtype: int32 wrap for k<16."""
    return int(np.array(total, dtype=np.int64).astype(get_cnt_dtype(k)))


def find_motif(seq_np_arr, kmer_len: int, max_ham_dist, p_unif, ratio_mu, ratio_std, ratio_cutoff, top_k=5, n_trial=10,
               merge_revcom_mode=True, rep_mode=False, save_kmer_cnt_flag=True, kmer_cnt_pkl_file: Path = None,
               boarder_pkl_file: Path = None, debug=False, dev_seq: DeviceSeq = None, table_savers: dict = None,
               counts_pool: list = None) -> dict:
    """Greedy motif discovery for one k.  Drop-in signature; `dev_seq` (optional) is an already uploaded
    DeviceSeq whose working copy is masked in place (then seq_np_arr is not touched).
    table_savers (optional dict): for tables above TOPK_DEVICE_MIN unique k-mers the background TableSaver of k{k}.pkl is left
    in table_savers[kmer_len] instead of being joined here -- the caller joins / closes it (scan_motif: at its end, so that the
    multi-GB fetch + pickle overlaps the following k and the occurrence scans, and sample_disp_kmer can re-use the resident table).
    counts_pool (optional list): DeviceCounts handles to re-use and to hand back (their arrays keep their capacity: a fresh handle
    per k re-allocates multi-GB buffers)."""

    def take():
        return counts_pool.pop() if counts_pool else DeviceCounts()

    def give(h):
        if counts_pool is not None:
            counts_pool.append(h)
        else:
            h.close()

    if boarder_pkl_file:
        assert Path(boarder_pkl_file).exists()
    own = dev_seq is None
    if own:
        with open(boarder_pkl_file, "rb") as fh:
            boarder_mat = pickle.load(fh)
        dev_seq = DeviceSeq(seq_np_arr, boarder_mat)
    dc = take()
    saver = None
    detached = False                    # the saver owns a gathered copy, not `first`
    try:
        cached = save_kmer_cnt_flag and kmer_cnt_pkl_file and Path(kmer_cnt_pkl_file).exists()
        if cached:
            with open(Path(kmer_cnt_pkl_file), "rb") as fh:
                k_pkl, uniq_kh_arr, uniq_kh_cnt_arr = pickle.load(fh)
            assert kmer_len == k_pkl
            u = np.ascontiguousarray(uniq_kh_arr, get_hash_dtype(kmer_len))
            c = np.ascontiguousarray(uniq_kh_cnt_arr, get_cnt_dtype(kmer_len))
            dc._unshard()
            check(_ffi.lib().kmap_counts_load(dc._h, ptr(u), ptr(c), len(u), kmer_len))
            dc.k, dc.n_uniq = kmer_len, len(u)
        else:
            dev_seq.count(dc, kmer_len, dedupe=not rep_mode, merge_revcom=merge_revcom_mode, gather_full=True)   # first round: the table k{k}.pkl holds
            uniq_kh_arr, uniq_kh_cnt_arr = None, None
        big = _device_topk(dc.n_uniq, top_k)
        n_total_kmer = _wrap_total(dc.total(), kmer_len)   # first round only (:648)
        first = dc                                          # the first-round table (trial 0 reads it)
        if save_kmer_cnt_flag and kmer_cnt_pkl_file and not Path(kmer_cnt_pkl_file).exists():
            if dc.n_uniq > SAVE_ASYNC_MIN:
                # multi-GB table: fetch + pickle on a background thread / stream; the later rounds count into a second handle
                dev = _ffi.i32(0)
                check(_ffi.lib().kmap_get_device(C.byref(dev)))
                _ffi.sync()
                table = first
                if first._shard is not None:      # sharded table: the saver takes the copy that was gathered on this rank
                    assert first._full is not None, "sharded counts: the rank that saves k{k}.pkl must be the gather target"
                    table, first._full = first._full, None
                    detached = True
                saver = TableSaver(table, kmer_len, kmer_cnt_pkl_file, device=dev.value)
                dc = take()
            else:
                uniq_kh_arr, uniq_kh_cnt_arr = dc.fetch()
                tmp = str(kmer_cnt_pkl_file) + ".tmp"
                try:
                    with open(tmp, "wb") as fh:
                        dump_pickle_nocopy([kmer_len, uniq_kh_arr, uniq_kh_cnt_arr], fh, protocol=4)   # without the arrays' tobytes() copies
                    os.replace(tmp, kmer_cnt_pkl_file)      # never a truncated k{k}.pkl under the cached name
                except BaseException:
                    if os.path.exists(tmp):
                        os.unlink(tmp)
                    raise
        elif not big:
            uniq_kh_arr, uniq_kh_cnt_arr = dc.fetch()

        res = {}
        cur = first
        for i_trial in range(n_trial):
            if top_k > cur.n_uniq:
                if debug:
                    print(f"There are only {cur.n_uniq} kmers, while top_k={top_k}.")
                break
            if big:
                _, cand_kh, _ = cur.topk(top_k)
                cand_kh = cand_kh[::-1]                      # ascending counts, like argpartition's tail
            else:
                if uniq_kh_arr is None:
                    uniq_kh_arr, uniq_kh_cnt_arr = cur.fetch()
                top_k_inds = np.array(np.argpartition(uniq_kh_cnt_arr, -top_k)[-top_k:])   # same numpy call -> same ties
                cand_kh = uniq_kh_arr[top_k_inds]
            if len(cand_kh) == 0:
                break
            hamball_cnt_arr = cur.hamball_mass(cand_kh, max_ham_dist, merge_revcom_mode)
            if debug:
                print(f"{i_trial= }")
            best = int(np.argmax(hamball_cnt_arr))
            consensus_kh = cand_kh[best]
            hamball_proportion = (hamball_cnt_arr[best] + 0.0) / n_total_kmer
            hamball_ratio = hamball_proportion / p_unif
            if not hamball_ratio > ratio_cutoff:
                break
            res[consensus_kh] = (hamball_proportion, hamball_ratio,
                                 norm_logsf(hamball_ratio, loc=ratio_mu, scale=ratio_std) / np.log(10))
            cons = [consensus_kh, revcom_hash(consensus_kh, kmer_len)] if merge_revcom_mode else [consensus_kh]
            dev_seq.mask(kmer_len, np.array(cons), np.array([max_ham_dist] * len(cons)))
            dev_seq.count(dc, kmer_len, dedupe=False, merge_revcom=merge_revcom_mode)   # later rounds: no dedupe (:695)
            cur = dc
            big = _device_topk(dc.n_uniq, top_k)
            uniq_kh_arr, uniq_kh_cnt_arr = None, None
        if saver is not None and table_savers is None:
            saver.join()
        if own:
            seq_np_arr[:] = dev_seq.download()   # the reference mutates its argument
        return res
    finally:
        if saver is None:
            give(dc)
        else:
            if dc is not first:
                give(dc)
            if detached:
                give(first)
            if table_savers is not None:
                table_savers[kmer_len] = saver               # the caller joins (errors surface there) and closes; table stays resident
            else:
                saver.close()
        if own:
            dev_seq.close()


# ---- motif occurrence (reference motif_discovery.py:1396-1477) -------------------------------------------
def scan_hit_lists(dev_seq: DeviceSeq, conseq_list, motif_def_dict, revcom_mode=True):
    """Per consensus the hit list at each read's minimum distance, before any subsampling: ScanHits (resident in HBM) where the
    sequence offers them, else [hits_per_read int32[n_seq], positions int32[sum]] on the host (read-sharded runs)."""
    per = []
    lazy = getattr(dev_seq, "scan_lazy", None)
    for conseq in conseq_list:
        k = len(conseq)
        if lazy is not None:
            per.append(lazy(k, kmer2hash(conseq), motif_def_dict[k].max_ham_dist, revcom_mode))
        else:
            hits, pos = dev_seq.scan(k, kmer2hash(conseq), motif_def_dict[k].max_ham_dist, revcom_mode)
            per.append([hits, pos])
    return per


def needs_draws(per):
    """does any read carry more than 20 hits of a consensus (the reference then draws 20 of them, motif_discovery.py:1466-1469)"""
    return any((r.max_hits > 20) if isinstance(r, ScanHits) else bool((r[0] > 20).any()) for r in per)


def subsample_hit_lists(per):
    """the reference's > 20-hit rule: np.random.choice of 20 positions, reads ascending, then consensus order (its draw order);
    lists without such reads are returned as they are (still in HBM if they were)"""
    if not needs_draws(per):
        return per
    per = [list(r) for r in per]
    big = [(int(r), c) for c, (hits, _) in enumerate(per) for r in np.nonzero(hits > 20)[0]]
    offs = [np.concatenate([[0], np.cumsum(h, dtype=np.int64)]) for h, _ in per]
    keep = [np.ones(len(p), bool) for _, p in per]
    newhits = [h.copy() for h, _ in per]
    for r, c in sorted(big):
        lo, hi = offs[c][r], offs[c][r + 1]
        locs = per[c][1][lo:hi]
        idx = np.random.choice(len(locs), 20, replace=False)        # :1468
        sel = np.zeros(len(locs), bool)
        sel[idx] = True                                             # np.sort(locs[idx]): locs ascending already
        keep[c][lo:hi] = sel
        newhits[c][r] = 20
    return [[newhits[c], per[c][1][keep[c]]] for c in range(len(per))]


def scan_motif_occurence(dev_seq: DeviceSeq, conseq_list, motif_def_dict, revcom_mode=True, subsample=True):
    """Per consensus: (hits_per_read int32[n_seq], positions int32[sum]) -- or a ScanHits that unpacks to that pair -- after the
    reference's > 20-hit random subsample."""
    per = scan_hit_lists(dev_seq, conseq_list, motif_def_dict, revcom_mode)
    return subsample_hit_lists(per) if subsample else per


class _BackgroundCall:
    """A native call (GIL released) on its own thread; join() re-raises its failure on the caller's thread."""

    def __init__(self, fn):
        import threading
        self.err = None

        def run():
            try:
                fn()
            except BaseException as e:   # noqa: BLE001 -- re-raised by join()
                self.err = e
        self._t = threading.Thread(target=run)
        self._t.start()

    def join(self):
        self._t.join()
        if self.err is not None:
            raise self.err

    close = join


def write_occurence_file(per, conseq_list, output_file, n_out, read_len, writers: list = None):
    """the CSV of a (subsampled) hit list; writers (optional list): formatted and written by a background thread appended to the
    list -- the caller joins it -- instead of before returning.  Lists still resident in HBM are fetched by the writing thread,
    with byte-sized hit counts (every count is <= 20 there)."""
    header = "seq_ind;" + ";".join(f"motif_{i}_{c}" for i, c in enumerate(conseq_list)) + ";seq_len"
    n_cons = len(conseq_list)

    def emit():
        narrow = bool(per) and all(isinstance(r, ScanHits) and r.unfetched and r.max_hits <= 255 for r in per)
        host = [r.host_u8() if narrow else tuple(r) for r in per]
        assert all(len(h) == n_out for h, _ in host)
        hits_ptrs = (C.c_void_p * max(n_cons, 1))(*[h.ctypes.data for h, _ in host])
        pos_keep = [np.ascontiguousarray(p, np.int32) if len(p) else np.zeros(1, np.int32) for _, p in host]
        pos_ptrs = (C.c_void_p * max(n_cons, 1))(*[p.ctypes.data for p in pos_keep])
        rows = _ffi.i64(0)
        fn = _ffi.lib().kmap_write_occurrence_csv_u8 if narrow else _ffi.lib().kmap_write_occurrence_csv
        check(fn(str(output_file).encode(), header.encode(), n_out, n_cons, hits_ptrs, pos_ptrs, ptr(read_len), C.byref(rows)))
    if writers is not None:
        writers.append(_BackgroundCall(emit))
    else:
        emit()


def gen_motif_occurence_file(conseq_list: List[str], motif_def_dict: dict, input_fasta_file, output_file, revcom_mode=True,
                             dev_seq: DeviceSeq = None, write=True, writers: list = None):
    """seq_ind;loc,loc;...;seq_len for every read with a hit.  With `dev_seq` the resident read array is
    scanned (it is the encoding of the same FASTA); otherwise the FASTA is encoded and uploaded here.
    write=False: scan only (the ranks of a read-sharded run that do not own the output files).
    writers: see write_occurence_file.  Returns the hit list (entries unpack to (hits_per_read, positions))."""
    own = dev_seq is None
    if own:
        assert Path(input_fasta_file).exists()
        arr, borders = encode_fasta(str(input_fasta_file))
        dev_seq = DeviceSeq(arr, borders)
    try:
        per = scan_motif_occurence(dev_seq, conseq_list, motif_def_dict, revcom_mode, subsample=write)
        if own:
            per = [list(r) for r in per]             # the sequence is closed below: nothing may stay behind in its handles
        if write:
            write_occurence_file(per, conseq_list, output_file, dev_seq.out_n_seq, dev_seq.out_read_len, None if own else writers)
        return per
    finally:
        if own:
            dev_seq.close()


def get_user_motif_occurence_file(input_fasta_file, conseq_list: List[str], max_hamdist_list: List[int], output_file,
                                  revcom_mode=True):
    """Occurrence file for user-given consensuses and radii (reference motif_discovery.py:1480-1507): the default motif
    table with `max_ham_dist` overridden per consensus LENGTH, in list order (a later consensus of the same length wins,
    as in the reference).  Returns the hit list."""
    from .kmer_count import _pkg_file
    assert Path(input_fasta_file).exists()
    motif_def_dict = init_motif_def_dict(_pkg_file(FileNameDict["default_motif_def_file"]))
    for conseq, max_ham_dist in zip(conseq_list, max_hamdist_list):
        motif_def_dict[len(conseq)].max_ham_dist = max_ham_dist
    return gen_motif_occurence_file(conseq_list, motif_def_dict, input_fasta_file, output_file, revcom_mode)


from .reports import (Occurrence, get_motif_seq_num, get_motif_pos_density, get_motif_co_occurence_mat,   # noqa: E402,F401
                      write_co_occurence_mat, write_co_occurence_dist_arr, ex_hamball_kh_arr, cal_cnt_mat, _ex_hamball)


# ---- sampling (reference motif_discovery.py:812-921) -------------------------------------------------------
class _LabelledTable:
    """The counted k-mers of one k resident on the device with a label per entry (csrc/reports.hip: label_kernel).
    Semantics of the reference's labelling (motif_discovery.py:846-883): distance of a k-mer to consensus c = its first
    len(c) bases against c, or -- reverse-complement mode -- its last len(c) bases against rc(c) if that is strictly
    smaller; a distance above c's own radius counts as k; label = first consensus at the minimum, or the noise label
    n_conseq when the minimum exceeds the radius of k; members matched through the reverse complement are re-oriented."""

    def __init__(self, uniq_kh_arr, uniq_kh_cnt_arr, conseq_list, kmer_len, motif_def_dict, revcom_mode, resident=None):
        """resident: a DeviceCounts handle that still holds this k's table (scan_motif keeps the big ones): the k-mers are copied
        device to device (labelling re-orients them in place) and the uint32 counts are read where they lie -- no k{k}.pkl
        load, no upload."""
        self.hd, self.cd = get_hash_dtype(kmer_len), get_cnt_dtype(kmer_len)
        self._bufs = []
        if resident is not None:
            up, cp, nu = _ffi.vp(), _ffi.vp(), _ffi.i64(0)
            check(_ffi.lib().kmap_counts_table_dev(resident._h, C.byref(up), C.byref(cp), C.byref(nu)))
            self.n = nu.value
            self.u_d = self._dev(nbytes=max(self.n, 1) * np.dtype(self.hd).itemsize)
            check(_ffi.lib().kmap_memcpy_d2d(self.u_d.ptr, up.value, self.n * np.dtype(self.hd).itemsize, None))
            self.c_d = _ffi.DeviceView(cp.value, self.n * 4, keep=resident)
            self.cnt64, self.c_dev_dtype = 0, np.uint32                  # device counts are uint32 bins whatever k
        else:
            self.n = len(uniq_kh_arr)
            self.u_d = self._dev(np.ascontiguousarray(uniq_kh_arr, self.hd))
            self.c_d = self._dev(np.ascontiguousarray(uniq_kh_cnt_arr, self.cd))
            self.cnt64, self.c_dev_dtype = int(self.cd == np.int64), self.cd
        self.k, self.n_lab = kmer_len, len(conseq_list) + 1
        if self.n_lab > 64:
            raise ValueError(f"sample_disp_kmer: {self.n_lab - 1} consensus sequences; the device labelling handles at most 63")
        self.lab_d = self._dev(nbytes=max(self.n, 1))
        self.w32_d = self._dev(nbytes=max(self.n, 1) * 4)
        self.excl_d = self._dev(nbytes=(self.n + 1) * 8)
        kh = [int(kmer2hash(s)) for s in conseq_list]
        if revcom_mode:   # consensuses are stored as the smaller of (hash, revcom hash), like the reference asserts (:857)
            assert all(h <= int(revcom_hash(h, len(s))) for h, s in zip(kh, conseq_list))
        cons = np.array(kh, np.uint64)
        lens = np.array([len(s) for s in conseq_list], np.int32)
        rads = np.array([motif_def_dict[len(s)].max_ham_dist for s in conseq_list], np.int32)
        if self.n:
            check(_ffi.lib().kmap_label_kmers_dev(self.u_d.ptr, self.n, kmer_len, len(conseq_list), ptr(cons), ptr(lens), ptr(rads),
                                                  int(motif_def_dict[kmer_len].max_ham_dist), int(bool(revcom_mode)),
                                                  self.lab_d.ptr, None))

    def _dev(self, arr=None, nbytes=None):
        b = _ffi.DeviceBuffer.from_numpy(arr) if arr is not None and len(arr) else _ffi.DeviceBuffer(nbytes or 16)
        self._bufs.append(b)
        return b

    def label_totals(self):
        """(sum of counts, number of members) per label, int64[n_lab] each"""
        wsum, members = np.zeros(self.n_lab, np.int64), np.zeros(self.n_lab, np.int64)
        if self.n:
            check(_ffi.lib().kmap_label_sums_dev(self.lab_d.ptr, self.c_d.ptr, self.cnt64, self.n, self.n_lab, ptr(wsum), ptr(members)))
        return wsum, members

    def member_indices(self, c, m):
        """ascending table indices of label c's m members"""
        idx = np.empty(m, np.int64)
        if m:
            check(_ffi.lib().kmap_label_prefix_dev(self.lab_d.ptr, None, 0, self.n, c, self.w32_d.ptr, self.excl_d.ptr, None))
            check(_ffi.lib().kmap_label_members_dev(self.lab_d.ptr, self.excl_d.ptr, self.n, c, m, ptr(idx)))
        return idx

    def _take(self, buf, dtype, idx):
        out = np.empty(len(idx), dtype)
        check(_ffi.lib().kmap_gather_dev(buf.ptr, np.dtype(dtype).itemsize, ptr(np.ascontiguousarray(idx, np.int64)), len(idx), ptr(out)))
        return out

    def counts_at(self, idx):
        return self._take(self.c_d, self.c_dev_dtype, idx).astype(self.cd)

    def kmers_at(self, idx):
        return self._take(self.u_d, self.hd, idx)

    def labels_at(self, idx):
        return self._take(self.lab_d, np.uint8, idx).astype(np.int64)

    def cdf_pick(self, c, total, n_draw):
        """n_draw inverse-CDF picks over label c's counts: index of the member whose cumulative count interval holds
        floor(u * total), u ~ np.random.random_sample (global legacy stream)."""
        check(_ffi.lib().kmap_label_prefix_dev(self.lab_d.ptr, self.c_d.ptr, self.cnt64, self.n, c, self.w32_d.ptr, self.excl_d.ptr, None))
        u = np.random.random_sample(n_draw) * float(total)
        targets = np.minimum(np.floor(u).astype(np.int64), total - 1)
        hit = np.empty(n_draw, np.int64)
        check(_ffi.lib().kmap_prefix_search_dev(self.excl_d.ptr, self.n, ptr(targets), n_draw, ptr(hit)))
        return hit

    def whole_table(self):
        """(re-oriented k-mers, labels int64) of every entry"""
        return self.u_d.to_numpy(self.hd, (self.n,)), self.lab_d.to_numpy(np.uint8, (self.n,)).astype(np.int64)

    def close(self):
        for b in self._bufs:
            b.free()
        self._bufs = []


def _label_quota(label_weight, n_total_sample, n_motif_kmer):
    """How many of the n_total_sample draws each label gets (reference :893-897): the motif labels share n_motif_kmer in
    proportion to their k-mer mass (rounded half-to-even by np.around), the noise label takes the remainder.  All quantities
    are integer-valued float64, so the order of the sums does not matter."""
    mass = np.asarray(label_weight, np.float64)
    share = mass[:-1] / mass[:-1].sum()
    quota = np.empty(len(mass), np.float64)
    quota[:-1] = np.around(n_motif_kmer * share)
    quota[-1] = n_total_sample - quota[:-1].sum()
    return quota.astype(int)


def sample_disp_kmer(conseq_list: List[str], kmer_len: int, motif_def_dict: dict, kmer_count_dir: Path, n_total_sample=5000,
                     n_motif_kmer=2500, revcom_mode=True, resident=None):
    """Drop-in for the reference's sample_disp_kmer (motif_discovery.py:812-921): labels every counted k-mer of `kmer_len` by
    its nearest consensus and draws a labelled multinomial sample -> (k-mer hashes, sample counts, labels, consensus list).
    The table is labelled on the device whatever its size; the host only makes the random draws, in the reference's order and
    from the same global np.random stream (one np.random.multinomial per label over the members' normalised counts), so the
    sample equals the reference's for a given seed.  Labels with more than TOPK_DEVICE_MIN members are drawn by inverse CDF
    instead (np.random.multinomial walks every category: minutes at 1e9 members) -- a documented deviation in the random
    draws, not in the distribution, for tables the reference cannot process; KMAP_EXACT=1 / config general.exact keeps
    np.random.multinomial at every size."""
    from . import _policy
    conseq_list = [s for s in conseq_list if 2 < len(s) <= kmer_len]
    assert len(conseq_list) > 0
    assert all(len(a) >= len(b) for a, b in zip(conseq_list, conseq_list[1:]))   # longest first, as merge_consensus_seqs emits
    def load_pkl():
        with open(Path(kmer_count_dir) / f"k{kmer_len}.pkl", "rb") as fh:
            k_pkl, u, c = pickle.load(fh)
        assert k_pkl == kmer_len
        return u, c

    # resident: a TableSaver whose DeviceCounts handle still holds the table of k{kmer_len}.pkl (scan_motif keeps the multi-GB
    # ones): label it where it lies instead of reading the file back and uploading it
    if resident is not None:
        uniq_kh_arr = uniq_kh_cnt_arr = None
        tab = _LabelledTable(None, None, conseq_list, kmer_len, motif_def_dict, revcom_mode, resident=resident.dc)
    else:
        uniq_kh_arr, uniq_kh_cnt_arr = load_pkl()
        tab = _LabelledTable(uniq_kh_arr, uniq_kh_cnt_arr, conseq_list, kmer_len, motif_def_dict, revcom_mode)
    try:
        label_weight, label_members = tab.label_totals()
        # the reference compares against the builtin sum() of numpy scalars, i.e. a total wrapped to the count dtype
        n_seq_total = _wrap_total(int(label_weight.sum()), kmer_len)
        if n_total_sample > n_seq_total:
            warnings.warn(f"The number of samples n_sample={n_total_sample} is larger than the original "
                          f"data n_seq={n_seq_total}, process and return original data.")
            kh_all, lab_all = tab.whole_table()
            if uniq_kh_cnt_arr is None:
                resident.join()
                uniq_kh_cnt_arr = load_pkl()[1]
            return kh_all, uniq_kh_cnt_arr, lab_all, conseq_list

        quota = _label_quota(label_weight, n_total_sample, n_motif_kmer)
        picked, picked_cnt = [], []
        for c, n_draw in enumerate(quota):
            m = int(label_members[c])
            if m > TOPK_DEVICE_MIN and not _policy.exact():
                where, times = np.unique(tab.cdf_pick(c, int(label_weight[c]), int(n_draw)), return_counts=True)
            else:
                members = tab.member_indices(c, m)
                w = tab.counts_at(members)
                prob = w / _wrap_total(int(w.sum(dtype=np.int64)), kmer_len)      # builtin sum() of count-dtype scalars
                draw = np.random.multinomial(n_draw, prob, size=1).squeeze()
                where, times = members[draw > 0], draw[draw > 0]
            picked.append(where)
            picked_cnt.append(times)
        picked, picked_cnt = np.concatenate(picked), np.concatenate(picked_cnt)
        return tab.kmers_at(picked), picked_cnt, tab.labels_at(picked), conseq_list
    finally:
        tab.close()


# ---- `kmap scan_motif` (reference motif_discovery.py:187-486) ------------------------------------------------
def _scan_motif(res_dir: str, debug=False):
    """`kmap scan_motif`.  Under `python -m torch.distributed.run --nproc-per-node G -m kmap_amd scan_motif ...` the reads are
    sharded over the G ranks (contiguous read ranges, distributed.make_dist_device_seq): per-read dedupe, masking and the
    occurrence scans are local to a rank, the 4^k-bin histograms are all-reduced (so every rank takes the same find_motif
    decisions) and the scan hits are all-gathered; rank 0 owns every output file and the np.random draws."""
    from .visualization import _dist_context
    dist, rank, owns_group = _dist_context()
    savers = {}     # k -> TableSaver of a multi-GB k{k}.pkl written in the background while the next k is counted
    try:
        _scan_motif_impl(res_dir, debug, dist, rank, savers)
    finally:
        for sv in _flat(savers):     # normally joined inside; after an exception: let the writers end, free the tables
            try:
                sv.close()
            except BaseException:    # noqa: BLE001 -- a writer's failure already surfaced through join(), or the run is failing anyway
                pass
    if dist is not None:            # success path only (a failing rank re-raises and the launcher tears the job down)
        from .distributed import barrier as _dist_barrier
        _dist_barrier(dist)
        if owns_group:
            dist.destroy_process_group()


def _write_co_occurrence_files(out_dir, occ, conseqs):
    """the four co-occurrence data files of scan_motif (reference motif_discovery.py:400-425 writes the same tables): pair counts,
    counts normalised by the two motifs' own read counts (Dice: 2 n_ab / (n_a + n_b)), median hit distance per pair, and the raw
    distances"""
    counts, median_dist, dist_lists = get_motif_co_occurence_mat(occ, len(conseqs), as_arrays=True)
    own = np.diag(counts)
    tables = {"co_occur_mat_file": counts + 0.0, "co_occur_mat_norm_file": 2 * counts / (own[None, :] + own[:, None]),
              "co_occur_dist_mat_file": median_dist}
    for key, table in tables.items():
        write_co_occurence_mat(out_dir / FileNameDict[key], table, conseqs)
    write_co_occurence_dist_arr(out_dir / FileNameDict["co_occur_dist_data_file"], dist_lists, conseqs)


def _flat(savers):
    """background jobs of a scan_motif run: k -> TableSaver, "occurrence" -> [CSV writers]"""
    for v in savers.values():
        yield from (v if isinstance(v, list) else [v])


def _scan_motif_impl(res_dir, debug, dist, rank, savers):
    from ._toml import load_toml
    res = Path(res_dir)
    lead = rank == 0                # owner of the output files

    def exists(path):
        """does the file exist -- rank 0's answer on every rank, so that all ranks take the same branch (the branches below
        contain collectives, and rank 0 creates the files while the others may still be checking)"""
        if dist is None:
            return Path(path).exists()
        import torch
        from .distributed import _coll_device
        t = torch.tensor([int(Path(path).exists())], dtype=torch.int64, device=_coll_device(dist))
        dist.broadcast(t, 0)
        return bool(t.item())

    config_file_path = res / FileNameDict["config_file"]
    motif_def_file_path = res / FileNameDict["motif_def_file"]
    proc_fasta_file_path = res / FileNameDict["processed_fasta_file"]
    assert config_file_path.exists()
    assert motif_def_file_path.exists()
    assert proc_fasta_file_path.exists()

    config_dict = load_toml(config_file_path)
    from . import _policy
    _policy.apply_config(config_dict)          # optional keys general.exact / visualization.embed_mode
    md = config_dict["motif_discovery"]
    min_k, max_k = config_dict["kmer_count"]["min_k"], config_dict["kmer_count"]["max_k"]
    revcom_mode = config_dict["kmer_count"]["revcom_mode"]
    rep_mode = config_dict["general"]["repetitive_mode"]

    # a fresh process pays ~0.1 - 0.2 s for the HIP runtime, the code object and the first host-to-device copy path: start them
    # on a helper thread while this one parses and maps the inputs
    import threading
    dev_now = _ffi.i32(0)

    def _warm():
        try:
            lib = _ffi.lib()
            if lib.kmap_set_device(dev_now.value) == 0:
                b = _ffi.DeviceBuffer.from_numpy(np.zeros(1 << 20, np.uint8))
                g = int(lib.kmap_packed_groups(1 << 20))
                c_, i_ = _ffi.DeviceBuffer(g * 4), _ffi.DeviceBuffer(g * 2)
                lib.kmap_pack_reads_dev(b.ptr, 1 << 20, c_.ptr, i_.ptr, None)
                _ffi.sync()
                for x in (b, c_, i_):
                    x.free()
        except Exception:     # noqa: BLE001 -- warm-up only: any real problem surfaces on the main thread
            pass
    check(_ffi.lib().kmap_get_device(C.byref(dev_now)))
    warm = threading.Thread(target=_warm)
    warm.start()
    boarder_pkl_file = res / FileNameDict["processed_fasta_seqboarder_file"]
    loaded = {}

    def _load():
        try:
            with _stage("load_inputs"):
                # large inputs: a read-only view of the mapped file; a rank of a sharded run touches only its own slice of it
                # (distributed.make_dist_device_seq), so no rank unpickles or pre-faults the whole input
                loaded["seq"] = load_array_pickle(proc_fasta_file_path, populate=dist is None)
                loaded["borders"] = load_array_pickle(boarder_pkl_file, populate=dist is None)
        except BaseException as e:   # noqa: BLE001 -- re-raised on the main thread below
            loaded["error"] = e
    loader = threading.Thread(target=_load)
    loader.start()
    # the motif table goes through pandas.read_csv and scipy.stats.norm like the reference's (kmer_count.py:719-740): ~0.8 s of
    # imports in a fresh process, now beside the mapping of the inputs and the HIP start-up instead of in front of them
    motif_def_dict = gen_motif_def_dict(config_dict, debug=debug)
    loader.join()
    if "error" in loaded:
        warm.join()
        raise loaded["error"]
    seq_np_arr, boarder_mat = loaded["seq"], loaded["borders"]
    n_all_seq = len(boarder_mat)

    def resident(arr):
        if dist is None:
            return DeviceSeq(arr, boarder_mat)
        from .distributed import make_dist_device_seq
        return make_dist_device_seq(arr, boarder_mat, dist)

    # the occurrence scans read the ORIGINAL reads (the reference re-parses the FASTA for them)
    warm.join()
    with _stage("upload"):
        scan_seq = resident(seq_np_arr)
    count_seq = scan_seq
    if md["noise_kmer_file"] != "None":
        assert Path(md["noise_kmer_file"]).exists()
        with open(Path(md["noise_kmer_file"]), "r") as fh:
            noise = [ln.strip() for ln in fh if ln.strip()]
        if noise:
            seq_np_arr = mask_ham_ball(np.array(seq_np_arr), motif_def_dict, noise, [0 for _ in noise])   # private copy
        count_seq = resident(seq_np_arr)

    top_k, n_trial = md["top_k"], md["n_trial"]
    save_kmer_cnt_flag = md["save_kmer_cnt_flag"]
    input_fasta_file = Path(config_dict["general"]["input_fasta_file"])
    candidate_conseq_list = []
    if save_kmer_cnt_flag and lead:
        (res / FileNameDict["kmer_count_dir"]).mkdir(exist_ok=True)

    counts_pool = []          # DeviceCounts handles handed from k to k (closed below)
    candidate_conseq_file = res / FileNameDict["candidate_conseq_file"]
    if exists(candidate_conseq_file):
        print(f"{candidate_conseq_file} already exist, re-use it.")
    else:
        occ_flag = md["store_conseq_occur_info_flag"]
        head = "kmer_len,conseq_hash,conseq,conseq_rc,hamball_proportion,hamball_ratio,log10_p_value"
        if occ_flag:
            head += ",n_motif_reads,n_all_reads,motif_reads_prop,motif_occurrence,motif_occurrence_per_motif_read"
        lines = [head]
        # find_motif of one k depends on nothing but the reads, so the k values go in the order that hides the background work
        # best -- the two largest k first: their count tables (15 GB of k16.pkl, 4 GB of k15.pkl at C3) are then written, to their two
        # files in parallel, while every other k is counted -- and the per-k files, candidate rows and np.random draws (occurrence
        # subsampling) follow in ascending k exactly as the reference emits them
        found, hit_lists, occ_written = {}, {}, set()
        occ_writers = savers.setdefault("occurrence", [])
        n_out, out_read_len = scan_seq.out_n_seq, scan_seq.out_read_len
        if dist is not None:      # tables that stay sharded (distributed.CountShard): only the writer of k{k}.pkl receives all shards
            count_seq.full_table_rank = 0 if save_kmer_cnt_flag else None
        for kmer_len in sorted(range(min_k, max_k + 1), key=lambda k: (k < max_k - 1, k if k < max_k - 1 else -k)):   # max_k, max_k - 1, then ascending
            count_seq.reset()
            d = motif_def_dict[kmer_len]
            kmer_cnt_file = res / FileNameDict["kmer_count_dir"] / f"k{kmer_len}.pkl"
            # sharded: every rank loads a cached k{k}.pkl if rank 0 sees one, otherwise all count and only rank 0 saves
            cached = exists(kmer_cnt_file) if (dist is not None and save_kmer_cnt_flag) else False   # collective: every rank calls it
            save_here = save_kmer_cnt_flag and (dist is None or lead or cached)
            with _stage("find_motif"), _stage(f"find_motif_k{kmer_len}"):
                found[kmer_len] = find_motif(None, kmer_len, d.max_ham_dist, d.p_uniform, d.ratio_mu, d.ratio_std,
                                             d.ratio_cutoff, top_k, n_trial, revcom_mode, rep_mode,
                                             save_kmer_cnt_flag=save_here, kmer_cnt_pkl_file=kmer_cnt_file,
                                             boarder_pkl_file=boarder_pkl_file, debug=debug, dev_seq=count_seq,
                                             table_savers=savers, counts_pool=counts_pool)
            if occ_flag:
                # the k's occurrence scan right away; if no read needs the > 20-hit draw (np.random: must happen in ascending k),
                # its CSV writer starts now as well and works while the other k are counted
                tmp_list = [hash2kmer(kh, kmer_len) for kh in found[kmer_len]]
                with _stage("occurrence_per_k"):
                    hit_lists[kmer_len] = scan_hit_lists(scan_seq, tmp_list, motif_def_dict, revcom_mode)
                    if not lead:                              # only the owner of the files ever fetches a gathered list
                        for r in hit_lists[kmer_len]:
                            getattr(r, "release", lambda: None)()
                    if lead and not needs_draws(hit_lists[kmer_len]):
                        write_occurence_file(hit_lists[kmer_len], tmp_list, res / FileNameDict["kmer_count_dir"] /
                                             f"k{kmer_len}.motif_occurence.csv", n_out, out_read_len, occ_writers)
                        occ_written.add(kmer_len)
        for kmer_len in range(min_k, max_k + 1):
            consensus_kh_dict = found[kmer_len]
            tmp_list = [hash2kmer(kh, kmer_len) for kh in consensus_kh_dict]
            per = None
            if occ_flag:
                per = hit_lists.pop(kmer_len)
                if lead and kmer_len not in occ_written:
                    with _stage("occurrence_per_k"):
                        per = subsample_hit_lists(per)
                        write_occurence_file(per, tmp_list, res / FileNameDict["kmer_count_dir"] /
                                             f"k{kmer_len}.motif_occurence.csv", n_out, out_read_len, occ_writers)
            for i, kmer_seq in enumerate(tmp_list):
                candidate_conseq_list.append(kmer_seq)
                if not lead:
                    continue
                kh = kmer2hash(kmer_seq)
                prop, ratio, log10_p = consensus_kh_dict[get_hash_dtype(kmer_len)(kh)]
                row = (f"{kmer_len},{kh},{kmer_seq},{reverse_complement(kmer_seq)},{prop:0.8f},"
                       f"{ratio:0.4f},{log10_p:0.4f}")
                if occ_flag:
                    n_motif_seq, n_occ = get_motif_seq_num(per, i)
                    row += (f",{n_motif_seq},{n_all_seq},{float(n_motif_seq) / n_all_seq:0.4f},{n_occ},"
                            f"{float(n_occ) / n_motif_seq:0.2f}")
                lines.append(row)
        for h in counts_pool:
            h.close()
        # counting is over: give back the multi-GiB transient tables the library caches between count calls (the shared 4^k-bin
        # histogram: 16 GiB after a k = 16 count, partition keys, hash arrays); small buffers stay for the scans below
        check(_ffi.lib().kmap_scratch_release(1 << 30))
        print(f"kmer counting finished for k={min_k}...{max_k}. Candidate consensus sequences generated.")
        if lead:
            write_lines(lines, candidate_conseq_file)

    final_conseq_file = res / FileNameDict["final_conseq_file"]
    if exists(final_conseq_file):
        with open(final_conseq_file, "r") as fh:
            final_conseq_list = fh.read().splitlines()
        print(f"{final_conseq_file} already exist, re-use it.")
    else:
        final_conseq_list = merge_consensus_seqs(candidate_conseq_list)
        if lead:
            write_lines(final_conseq_list, final_conseq_file)

    # the count tables kept resident for their background k{k}.pkl writers: only the longest final's k is read again (labelled
    # sampling); the others leave HBM as soon as their writer has finished (a failed writer is kept: join() below re-raises)
    keep_k = max((len(c) for c in final_conseq_list), default=None)
    for k_done in [k for k, sv in savers.items() if isinstance(sv, TableSaver) and k != keep_k and sv.finished_ok()]:
        savers.pop(k_done).close()

    final_conseq_info_file = res / FileNameDict["final_conseq_info_file"]
    if not lead:
        pass
    elif final_conseq_info_file.exists():
        print(f"{final_conseq_info_file} already exist, re-use it.")
    else:
        with open(final_conseq_file, "r") as fh:
            final_conseq_list = fh.read().splitlines()
        with open(candidate_conseq_file, "r") as fh:
            cand_lines = fh.read().splitlines()
        cols = cand_lines[0].split(",")
        cols[1], cols[0] = cols[0], "motif_id"
        info = [",".join(cols)]
        motif_ind = 0
        for conseq in final_conseq_list:
            for line in cand_lines:
                if "," + conseq + "," in line:
                    el = line.split(",")
                    el[1], el[0] = el[0], str(motif_ind)
                    motif_ind += 1
                    info.append(",".join(el))
        write_lines(info, final_conseq_info_file)
        print("Final consensus sequences generated.")

    occurence_file = res / FileNameDict["motif_occurence_file"]
    want_occ = md["motif_pos_density_flag"] or md["motif_co_occurence_flag"]
    with _stage("occurrence_final"):
        per_final = scan_motif_occurence(scan_seq, final_conseq_list, motif_def_dict, revcom_mode, subsample=lead)
        if not lead:
            for r in per_final:
                getattr(r, "release", lambda: None)()
        if lead:
            if want_occ:
                per_final = [list(r) for r in per_final]       # the report stages below read the lists on this thread
            write_occurence_file(per_final, final_conseq_list, occurence_file, scan_seq.out_n_seq, scan_seq.out_read_len,
                                 savers.setdefault("occurrence", []))
    if not lead:                    # everything below is host-side reporting / sampling on the hit list and the k{k}.pkl tables
        if count_seq is not scan_seq:
            count_seq.close()
        scan_seq.close()
        return
    occ = None
    if want_occ:
        occ = Occurrence.from_per(per_final, scan_seq.out_read_len)   # the consumers below use the hit list, not the CSV

    # the reference also draws pdf figures in these branches (motif_discovery.py:364-425); only the data files are produced
    if md["motif_pos_density_flag"]:
        with _stage("pos_density"):
            x_step = 0.01
            x_arr = np.arange(0, 1.0 + x_step, x_step)
            rows = [get_motif_pos_density(occ, i, len(conseq), x_step=x_step, x_arr=x_arr)[2]
                    for i, conseq in enumerate(final_conseq_list)]
            with open(res / FileNameDict["motif_pos_density_file"], "wb") as fh:
                pickle.dump([x_arr, np.vstack(rows)], fh)
        print("motif position distribution generated.")

    if md["motif_co_occurence_flag"]:
        co_occur_dir = res / FileNameDict["co_occur_dir"]
        co_occur_dir.mkdir(exist_ok=True)
        co_occur_mat_file = co_occur_dir / FileNameDict["co_occur_mat_file"]
        if co_occur_mat_file.exists():
            print(f"{co_occur_mat_file}, re-use it!")
        else:
            with _stage("co_occurrence"):
                _write_co_occurrence_files(co_occur_dir, occ, final_conseq_list)
        print("motif co-occurence matrix generated.")

    sample_kmer_pkl_file = res / FileNameDict["sample_kmer_pkl_file"]
    if md["sample_kmer_flag"] and not save_kmer_cnt_flag:
        print(f"kmers cannot be sampled when {save_kmer_cnt_flag=}, skip kmer sampling!")
    if sample_kmer_pkl_file.exists():
        print(f"sample kmer file {sample_kmer_pkl_file} exists, skip sampling!")
    elif md["sample_kmer_flag"] and save_kmer_cnt_flag:
        n_total_sample, n_motif_sample = md["n_total_sample"], md["n_motif_sample"]
        kmer_len = max([len(conseq) for conseq in final_conseq_list])   # ValueError if no motif, like the reference
        with _stage("sample_kmers"):
            samp_kh_arr, samp_cnts, samp_label_arr, conseq_list = sample_disp_kmer(
                final_conseq_list, kmer_len, motif_def_dict, kmer_count_dir=res / FileNameDict["kmer_count_dir"],
                n_total_sample=n_total_sample, n_motif_kmer=n_motif_sample, revcom_mode=revcom_mode,
                resident=savers.get(kmer_len))
        with open(sample_kmer_pkl_file, "wb") as fh:
            pickle.dump([samp_kh_arr, samp_cnts, samp_label_arr, conseq_list], fh)
        kmers = hashes2kmers(samp_kh_arr, kmer_len)
        with open(res / FileNameDict["sample_kmer_txt_file"], "w+") as fh:
            fh.write("".join(f"{kmer}\t{label}\n" * cnt for kmer, cnt, label in zip(kmers.tolist(), samp_cnts.tolist(),
                                                                                   samp_label_arr.tolist())))
        print(f"kmers are sampled for visualization. {kmer_len= }, {n_total_sample= }, {n_motif_sample= }")

        label_arr = _convert_to_block_arr(samp_label_arr, samp_cnts)
        if len(label_arr) <= DENSE_PKL_MAX_N:
            with _stage("hamdist_matrix_int64"):
                hamdist_mat = cal_samp_kmer_hamdist_mat(samp_kh_arr, samp_cnts, samp_label_arr, conseq_list, kmer_len,
                                                        uniq_dist_flag=False)
        else:
            hamdist_mat = None   # compact hand-off; visualize_kmers recomputes the matrix on the device
            print(f"N={len(label_arr)} > {DENSE_PKL_MAX_N}: int64 matrix not materialised (compact hand-off).")
        with _stage("write_hamdist_pkl"):
            with open(res / FileNameDict["sample_kmer_hamdist_mat_file"], "wb") as fh:
                dump_pickle_nocopy([kmer_len, hamdist_mat, label_arr], fh)   # 200 MB of int64 at the reference's default size
        print("Hamming distance matrix of sampled kmers are generated.")

    with _stage("join_table_writers"):       # the k{k}.pkl files of the large tables and the occurrence CSVs are complete from here on
        for sv in _flat(savers):
            sv.join()

    if md["gen_hamball_flag"]:
        out_dir_path = res / FileNameDict["hamball_dir"]
        out_dir_path.mkdir(exist_ok=True)
        with _stage("hamming_balls"):
            for i, conseq in enumerate(final_conseq_list):
                output_cntmat_file = out_dir_path / f"cntmat_motif{i}_{conseq}.csv"
                if output_cntmat_file.exists():
                    print(f"motif matrix file {output_cntmat_file} exist, skip generating.")
                    continue
                # the table of the longest finals' k is still in HBM (kept for the labelled sampling): its multi-GB k{k}.pkl is not
                # read back (C3, k = 15: 1.2 s per consensus); shorter finals read their (small) files like the reference
                sv = savers.get(len(conseq))
                _ex_hamball(str(res), conseq, "matrix", str(output_cntmat_file),
                            max_ham_dist=motif_def_dict[len(conseq)].max_ham_dist,
                            resident=sv.dc if isinstance(sv, TableSaver) and dist is None else None)
        print("Motif count matrix extracted.")

    if count_seq is not scan_seq:
        count_seq.close()
    scan_seq.close()
    print("All tasks of scan motif finished.")
