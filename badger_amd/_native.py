"""ctypes binding of libbadger_hip.so (include/badger_hip.h).

The product path: there is NO CPU fallback.  If the HIP library is missing, or no
MI355X is visible, every entry point raises.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbadger_hip.so")

REC_DTYPE = np.dtype([
    ("polyT", "<i4"), ("r1_end", "<i4"), ("bc_start", "<i4"), ("umi_start", "<i4"),
    ("umi_end", "<i4"), ("bc_rank", "<u4"), ("r1_score", "i1"), ("strand", "i1"),
    ("valid", "u1"), ("flags", "u1"), ("reserved", "<u4")])
EDGE_DTYPE = np.dtype([("a", "<u4"), ("b", "<u4"), ("dist", "<u4")])
FLAG_REV = 1
FLAG_RANK_OK = 2
FLAG_BC16 = 4
FLAG_INCOMPLETE = 8
NONE_IDX = 0xFFFFFFFF

E_ARG, E_HIP, E_NOMEM, E_CAPACITY, E_BADBASE, E_FORMAT, E_NOSEQ = -1, -2, -3, -4, -5, -6, -7
SLOTS = 4
STRAND_RULE_DEFAULT, STRAND_RULE_NO_POLYA = 0, 1

EXPORTS = [
    "bdg_init", "bdg_free", "bdg_last_error", "bdg_version", "bdg_device_count", "bdg_selftest_dj_codec",
    "bdg_mem_alloc", "bdg_mem_free", "bdg_mem_to_host", "bdg_mem_from_host", "bdg_set_stream", "bdg_synchronize", "bdg_set_overlap",
    "bdg_profile_enable", "bdg_profile_only", "bdg_profile_reset", "bdg_profile_read",
    "bdg_extract_batch", "bdg_extract_batch_dev", "bdg_extract_status", "bdg_extract_counters", "bdg_extract_set_queue_capacity",
    "bdg_extract_set_strand_rule",
    "bdg_nearest16", "bdg_whitelist_load", "bdg_nearest16_dev", "bdg_nearest16_recs_dev", "bdg_nearest16_set_algo", "bdg_nearest16_index_bytes",
    "bdg_graph_edges", "bdg_graph_edges_dev", "bdg_graph_edges_rows_dev", "bdg_graph_edges_part_dev", "bdg_graph_set_algo", "bdg_graph_status", "bdg_distinct_dev", "bdg_rows_of_dev",
    "bdg_extract_submit", "bdg_extract_collect", "bdg_extract_keep_records", "bdg_kept_records", "bdg_kept_records_to_host", "bdg_keep_observed", "bdg_touched_count_dev",
    "bdg_ingest_open", "bdg_ingest_open_mt", "bdg_ingest_open_ex", "bdg_ingest_next", "bdg_ingest_release", "bdg_ingest_error",
    "bdg_ingest_reads", "bdg_ingest_close", "bdg_format_rows", "bdg_stage1_run",
    "bdg_cluster_dev", "bdg_assign_reads_dev", "bdg_idstore_new", "bdg_idstore_free", "bdg_idstore_count", "bdg_idstore_append",
    "bdg_idstore_get", "bdg_stage1_collect", "bdg_write_assignments", "bdg_import_stage1_tsv", "bdg_host_free",
]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


class IngestChunk(C.Structure):
    """bdg_ingest_chunk (include/badger_hip.h): one chunk of reads in (pinned) host memory"""
    _fields_ = [("id", C.c_uint32), ("n", C.c_uint32), ("bases", C.c_void_p), ("off", C.c_void_p),
                ("total_bytes", C.c_uint64), ("ids", C.c_void_p), ("id_off", C.c_void_p)]


class IngestOpts(C.Structure):
    """bdg_ingest_opts"""
    _fields_ = [("chunk_reads", C.c_uint32), ("ring_chunks", C.c_uint32), ("pinned", C.c_int32), ("threads", C.c_uint32),
                ("segment_bytes", C.c_uint64), ("skip_secondary", C.c_int32), ("reserved", C.c_uint32)]


class Stage1Opts(C.Structure):
    """bdg_stage1_opts"""
    _fields_ = [("umi_len", C.c_uint32), ("threads", C.c_uint32), ("format_threads", C.c_uint32), ("header_every", C.c_uint32),
                ("chunk_reads", C.c_uint32), ("skip_secondary", C.c_int32), ("segment_bytes", C.c_uint64)]


class Stage1Result(C.Structure):
    """bdg_stage1_result"""
    _fields_ = [("reads", C.c_uint64), ("barcodes", C.c_uint64), ("polyt", C.c_uint64), ("r1", C.c_uint64),
                ("first_polyt", C.c_uint64), ("first_r1", C.c_uint64), ("bad_read", C.c_uint64),
                ("chunks", C.c_uint64), ("out_bytes", C.c_uint64), ("seconds_total", C.c_double),
                ("seconds_wait_parse", C.c_double), ("seconds_submit", C.c_double), ("seconds_wait_gpu", C.c_double),
                ("seconds_wait_format", C.c_double), ("seconds_format", C.c_double), ("seconds_write", C.c_double)]


class BadgerHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libbadger_hip error %d: %s" % (code, msg))
        self.code = code


_LIB = None
PRELOAD_TORCH = os.environ.get("BADGER_AMD_PRELOAD_TORCH", "1") != "0"


def load():
    """dlopen the in-tree HIP library and declare its prototypes.  No GPU needed for this."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C badger_amd/csrc` "
                          "(or `python -c 'import __graft_entry__ as g; g.build()'`)" % LIB_PATH)
    # PyTorch-ROCm bundles its own HIP runtime and two HIP runtimes cannot both open the device: whichever
    # initialises second sees no GPU.  If torch can end up in this process, load it first so that the library's HIP
    # symbols bind to the runtime torch uses (stand-alone C/C++ hosts just link /opt/rocm's).
    # A process that will never import torch (the stage-1 command line) sets PRELOAD_TORCH = False before the first call
    # and starts more than a second sooner.
    if PRELOAD_TORCH and "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    L.bdg_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.bdg_free.argtypes = [vp]
    L.bdg_free.restype = None
    L.bdg_last_error.argtypes = [vp]
    L.bdg_last_error.restype = C.c_char_p
    L.bdg_version.restype = C.c_char_p
    L.bdg_device_count.restype = C.c_int
    L.bdg_selftest_dj_codec.argtypes = [C.c_uint64, C.c_uint32]
    L.bdg_mem_alloc.argtypes = [vp, u64, C.POINTER(vp)]
    L.bdg_mem_free.argtypes = [vp, vp]
    L.bdg_mem_to_host.argtypes = [vp, vp, vp, u64]
    L.bdg_mem_from_host.argtypes = [vp, vp, vp, u64]
    L.bdg_set_stream.argtypes = [vp, vp]
    L.bdg_synchronize.argtypes = [vp]
    L.bdg_set_overlap.argtypes = [vp, C.c_int]
    L.bdg_profile_enable.argtypes = [vp, C.c_int]
    L.bdg_profile_only.argtypes = [vp, C.c_char_p]
    L.bdg_profile_reset.argtypes = [vp]
    L.bdg_profile_read.argtypes = [vp, C.POINTER(KernelTime), C.c_int]
    L.bdg_extract_batch.argtypes = [vp, vp, vp, u32, u32, vp]
    L.bdg_extract_batch_dev.argtypes = [vp, vp, vp, u32, u64, u32, vp]
    L.bdg_extract_status.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.bdg_extract_counters.argtypes = [vp, C.POINTER(u64)]
    L.bdg_extract_set_queue_capacity.argtypes = [vp, u64]
    L.bdg_extract_set_strand_rule.argtypes = [vp, C.c_int]
    L.bdg_nearest16.argtypes = [vp, vp, u32, vp, u32, u32, vp, vp, vp]
    L.bdg_whitelist_load.argtypes = [vp, vp, u32]
    L.bdg_nearest16_dev.argtypes = [vp, vp, u32, u32, vp, vp, vp]
    L.bdg_nearest16_recs_dev.argtypes = [vp, vp, u32, u32, vp, vp, vp]
    L.bdg_nearest16_set_algo.argtypes = [vp, C.c_int]
    L.bdg_nearest16_index_bytes.argtypes = [vp]
    L.bdg_nearest16_index_bytes.restype = C.c_uint64
    L.bdg_graph_edges.argtypes = [vp, vp, u32, u32, i32, vp, u64, C.POINTER(u64)]
    L.bdg_graph_edges_dev.argtypes = [vp, vp, u32, u32, i32, vp, u64, vp]
    L.bdg_graph_edges_rows_dev.argtypes = [vp, vp, u32, u32, u32, u32, i32, vp, u64, vp]
    L.bdg_graph_edges_part_dev.argtypes = [vp, vp, u32, u32, u32, u32, i32, vp, u64, vp]
    L.bdg_graph_set_algo.argtypes = [vp, C.c_int]
    L.bdg_graph_status.argtypes = [vp]
    L.bdg_distinct_dev.argtypes = [vp, vp, u32, vp, vp, vp, vp]
    L.bdg_rows_of_dev.argtypes = [vp, vp, u32, vp, u64, u32, vp]
    L.bdg_extract_submit.argtypes = [vp, u32, vp, vp, u32, u32]
    L.bdg_extract_collect.argtypes = [vp, u32, vp]
    L.bdg_extract_keep_records.argtypes = [vp, C.c_int]
    L.bdg_kept_records.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.bdg_kept_records_to_host.argtypes = [vp, vp, u64]
    L.bdg_keep_observed.argtypes = [vp, vp, vp, u64]
    L.bdg_touched_count_dev.argtypes = [vp, vp, vp, u64, u32, vp, u32, C.POINTER(u64)]
    L.bdg_ingest_open.argtypes = [C.c_char_p, u32, u32, C.c_int, C.POINTER(vp)]
    L.bdg_ingest_open_mt.argtypes = [C.c_char_p, u32, u32, C.c_int, u32, C.POINTER(vp)]
    L.bdg_ingest_open_ex.argtypes = [C.c_char_p, C.POINTER(IngestOpts), C.POINTER(vp)]
    L.bdg_ingest_reads.argtypes = [vp]
    L.bdg_ingest_reads.restype = C.c_uint64
    L.bdg_stage1_run.argtypes = [C.POINTER(vp), u32, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(Stage1Opts), C.POINTER(Stage1Result)]
    L.bdg_cluster_dev.argtypes = [vp, vp, vp, u64, u32, vp]
    L.bdg_assign_reads_dev.argtypes = [vp, vp, u64, vp, u32, vp, vp, vp, vp]
    L.bdg_idstore_new.restype = vp
    L.bdg_idstore_free.argtypes = [vp]
    L.bdg_idstore_free.restype = None
    L.bdg_idstore_count.argtypes = [vp]
    L.bdg_idstore_count.restype = C.c_uint64
    L.bdg_idstore_append.argtypes = [vp, vp, vp, u64]
    L.bdg_idstore_get.argtypes = [vp, u64, C.POINTER(vp), C.POINTER(u32)]
    L.bdg_stage1_collect.argtypes = [vp, C.c_char_p, C.POINTER(Stage1Opts), vp, C.POINTER(Stage1Result)]
    L.bdg_write_assignments.argtypes = [vp, vp, vp, u64, C.c_char_p]
    L.bdg_import_stage1_tsv.argtypes = [C.c_char_p, u32, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.bdg_host_free.argtypes = [vp]
    L.bdg_host_free.restype = None
    L.bdg_ingest_next.argtypes = [vp, C.POINTER(IngestChunk)]
    L.bdg_ingest_release.argtypes = [vp, u32]
    L.bdg_ingest_error.argtypes = [vp]
    L.bdg_ingest_error.restype = C.c_char_p
    L.bdg_ingest_close.argtypes = [vp]
    L.bdg_ingest_close.restype = None
    L.bdg_format_rows.argtypes = [C.POINTER(IngestChunk), vp, vp, u64, C.POINTER(u64)]
    L.bdg_format_rows.restype = C.c_int64
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("bdg_free",):
            fn.restype = C.c_int
    _LIB = L
    return L


class Context:
    """One bdg_ctx = one GPU.  Host-buffer calls take numpy arrays; *_dev calls take torch tensors
    that already live on the context's device."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.bdg_init(int(device), C.byref(h))
        if rc != 0:
            raise BadgerHipError(rc, self.lib.bdg_last_error(None).decode())
        self.h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "h", None):
            self.lib.bdg_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise BadgerHipError(rc, self.lib.bdg_last_error(self.h).decode())
        return rc

    # -- plumbing ----------------------------------------------------------
    def set_stream(self, hip_stream_handle):
        """hipStream_t handle as an int; 0 is the device's default stream (torch's default stream)."""
        self._check(self.lib.bdg_set_stream(self.h, C.c_void_p(hip_stream_handle or 0)))

    def synchronize(self):
        self._check(self.lib.bdg_synchronize(self.h))

    def set_overlap(self, on=True):
        """nearest16_recs_dev on an auxiliary stream: the match of batch i is queued behind the scan of batch i + 1 and runs
        beside its alignment kernels; results are complete after synchronize() (which also queues a match still waiting)"""
        self._check(self.lib.bdg_set_overlap(self.h, 1 if on else 0))

    def profile(self, on=True):
        self._check(self.lib.bdg_profile_enable(self.h, 1 if on else 0))

    def profile_only(self, kernel=None):
        """time only this kernel (None: every kernel again)"""
        self._check(self.lib.bdg_profile_only(self.h, kernel.encode() if kernel else None))

    def profile_reset(self):
        self._check(self.lib.bdg_profile_reset(self.h))

    def profile_read(self):
        buf = (KernelTime * 32)()
        n = self._check(self.lib.bdg_profile_read(self.h, buf, 32))
        return {buf[i].name.decode(): (int(buf[i].launches), float(buf[i].total_ms)) for i in range(min(n, 32))}

    # -- extraction ----------------------------------------------------------
    def extract_batch(self, bases, off, umi_len=12):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        out = np.zeros(max(n, 0), dtype=REC_DTYPE)
        if n <= 0:
            return out
        self._check(self.lib.bdg_extract_batch(self.h, bases.ctypes.data, off.ctypes.data, n, umi_len, out.ctypes.data))
        return out

    def extract_batch_dev(self, d_bases, d_off, n, total_bytes, umi_len, d_out):
        self._check(self.lib.bdg_extract_batch_dev(self.h, d_bases.data_ptr(), d_off.data_ptr(), n, total_bytes,
                                                   umi_len, d_out.data_ptr()))

    def extract_submit(self, slot, bases_ptr, off_ptr, n, umi_len=12):
        """enqueue one chunk (host pointers, best pinned) on staging set `slot`; returns at once"""
        self._check(self.lib.bdg_extract_submit(self.h, slot, bases_ptr, off_ptr, n, umi_len))

    def extract_collect(self, slot, n):
        """wait for the chunk submitted to `slot` and return its records"""
        out = np.zeros(n, dtype=REC_DTYPE)
        self._check(self.lib.bdg_extract_collect(self.h, slot, out.ctypes.data))
        return out

    def extract_keep_records(self, on=True):
        """keep (a copy of) every collected chunk's records on the device, in order, for the stage-2 hand-off"""
        self._check(self.lib.bdg_extract_keep_records(self.h, 1 if on else 0))

    def keep_observed(self, rank, usable):
        """the observed barcodes of a stage-1 TSV (rank uint32[n], usable bool[n]) as the kept records"""
        rank = np.ascontiguousarray(rank, dtype=np.uint32)
        usable = np.ascontiguousarray(usable, dtype=np.uint8)
        if len(rank) != len(usable):
            raise ValueError("rank and usable differ in length")
        self._check(self.lib.bdg_keep_observed(self.h, rank.ctypes.data, usable.ctypes.data, len(rank)))

    def kept_records(self):
        """-> (device pointer, count) of the records kept since extract_keep_records(True)"""
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self.lib.bdg_kept_records(self.h, C.byref(p), C.byref(n)))
        return p.value or 0, int(n.value)

    def kept_records_to_host(self):
        """the kept records as a numpy array (synchronises)"""
        _, n = self.kept_records()
        out = np.zeros(n, dtype=REC_DTYPE)
        if n:
            self._check(self.lib.bdg_kept_records_to_host(self.h, out.ctypes.data, n))
        return out

    def extract_status(self):
        bad, nwin = C.c_uint64(), C.c_uint64()
        rc = self.lib.bdg_extract_status(self.h, C.byref(bad), C.byref(nwin))
        return rc, bad.value, nwin.value

    def extract_set_queue_capacity(self, entries_per_segment):
        """entries per segment of the internal candidate queues (0 = automatic); an overflow grows it again"""
        self._check(self.lib.bdg_extract_set_queue_capacity(self.h, entries_per_segment))

    def extract_set_strand_rule(self, rule):
        """STRAND_RULE_DEFAULT (find_barcode_umi) or STRAND_RULE_NO_POLYA (find_barcode_umi_no_polya) for the launches that follow"""
        self._check(self.lib.bdg_extract_set_strand_rule(self.h, rule))

    def extract_counters(self):
        out = (C.c_uint64 * 8)()
        self._check(self.lib.bdg_extract_counters(self.h, out))
        names = ("hits", "clusters", "filter_in", "filter_skipped", "filter_kept", "requeued", "alignments", "filter_in_clusters")
        return dict(zip(names, [int(x) for x in out[:8]]))

    # -- nearest ---------------------------------------------------------------
    def nearest16(self, q, wl, max_ed=2):
        q = np.ascontiguousarray(q, dtype=np.uint32)
        wl = np.ascontiguousarray(wl, dtype=np.uint32)
        idx = np.zeros(len(q), np.uint32)
        ed = np.zeros(len(q), np.uint8)
        ties = np.zeros(len(q), np.uint16)
        self._check(self.lib.bdg_nearest16(self.h, q.ctypes.data, len(q), wl.ctypes.data, len(wl), max_ed,
                                           idx.ctypes.data, ed.ctypes.data, ties.ctypes.data))
        return idx, ed, ties

    def whitelist_load(self, wl):
        wl = np.ascontiguousarray(wl, dtype=np.uint32)
        self._check(self.lib.bdg_whitelist_load(self.h, wl.ctypes.data, len(wl)))

    def nearest16_dev(self, d_q, nq, max_ed, d_idx, d_ed, d_ties):
        self._check(self.lib.bdg_nearest16_dev(self.h, d_q.data_ptr(), nq, max_ed, d_idx.data_ptr(),
                                               d_ed.data_ptr(), d_ties.data_ptr()))

    def nearest16_recs_dev(self, d_recs, n, max_ed, d_idx, d_ed, d_ties):
        """nearest16 of every record's barcode (records without a 16-base ACGT barcode report no hit)"""
        self._check(self.lib.bdg_nearest16_recs_dev(self.h, d_recs.data_ptr(), n, max_ed, d_idx.data_ptr(),
                                                    d_ed.data_ptr(), d_ties.data_ptr()))

    def nearest16_set_algo(self, algo):
        self._check(self.lib.bdg_nearest16_set_algo(self.h, algo))

    def nearest16_index_bytes(self):
        """device bytes of the probe index of the loaded whitelist (0: never built)"""
        return int(self.lib.bdg_nearest16_index_bytes(self.h))

    # -- graph -----------------------------------------------------------------
    def graph_edges(self, ranks, thr, qgram_T):
        ranks = np.ascontiguousarray(ranks, dtype=np.uint32)
        cap = max(1024, 4 * len(ranks))
        while True:
            out = np.zeros(cap, dtype=EDGE_DTYPE)
            tot = C.c_uint64()
            rc = self.lib.bdg_graph_edges(self.h, ranks.ctypes.data, len(ranks), thr, qgram_T,
                                          out.ctypes.data, cap, C.byref(tot))
            if rc == E_CAPACITY and tot.value > cap:
                cap = int(tot.value)
                continue
            self._check(rc)
            return out[:tot.value]

    def graph_edges_dev(self, d_ranks, n, thr, qgram_T, d_out, cap, d_n_edges):
        self._check(self.lib.bdg_graph_edges_dev(self.h, d_ranks.data_ptr(), n, thr, qgram_T,
                                                 d_out.data_ptr(), cap, d_n_edges.data_ptr()))

    def graph_edges_rows_dev(self, d_ranks, n, row_begin, row_end, thr, qgram_T, d_out, cap, d_n_edges):
        """edges whose smaller rank is row row_begin <= i < row_end of the sorted array (one GPU's share)"""
        self._check(self.lib.bdg_graph_edges_rows_dev(self.h, d_ranks.data_ptr(), n, row_begin, row_end, thr, qgram_T,
                                                      d_out.data_ptr(), cap, d_n_edges.data_ptr()))

    def graph_edges_part_dev(self, d_ranks, n, part, nparts, thr, qgram_T, d_out, cap, d_n_edges):
        """one of nparts disjoint shares of the edge list, cut by the library so that the shares cost the same (one GPU's share)"""
        self._check(self.lib.bdg_graph_edges_part_dev(self.h, _ptr(d_ranks), n, part, nparts, thr, qgram_T,
                                                      _ptr(d_out), cap, _ptr(d_n_edges)))

    def graph_set_algo(self, algo):
        self._check(self.lib.bdg_graph_set_algo(self.h, algo))

    def graph_status(self):
        """waits for the stream; raises if a deletion-variant join of the last *_dev graph call could not group its input"""
        self._check(self.lib.bdg_graph_status(self.h))

    def rows_of_dev(self, d_sorted, n, d_values, m, stride_words, d_rows, value_offset_words=0):
        """d_rows[i] = position of d_values[value_offset_words + i * stride_words] in the ascending d_sorted[0..n), NONE if absent"""
        self._check(self.lib.bdg_rows_of_dev(self.h, _ptr(d_sorted), n, _ptr(d_values, 4 * value_offset_words), m, stride_words, _ptr(d_rows)))

    def touched_count_dev(self, d_ea, d_eb, m, nu, d_extra, n_extra):
        """how many of nu barcodes appear in the m edges (positions) or in d_extra (bdg_touched_count_dev)"""
        out = C.c_uint64(0)
        self._check(self.lib.bdg_touched_count_dev(self.h, _ptr(d_ea) if m else None, _ptr(d_eb) if m else None, m, nu,
                                                   _ptr(d_extra) if n_extra else None, n_extra, C.byref(out)))
        return int(out.value)

    def cluster_dev(self, d_ea, d_eb, m, nu, d_owner):
        """the two clustering levels over m edges given as positions in the distinct array (bdg_cluster_dev)"""
        self._check(self.lib.bdg_cluster_dev(self.h, _ptr(d_ea), _ptr(d_eb), m, nu, _ptr(d_owner)))

    def assign_reads_dev(self, d_recs, n, d_uniq, nu, d_assigned, d_has, d_out_rank, d_out_has):
        """per record: the barcode its observed barcode was corrected to (bdg_assign_reads_dev)"""
        self._check(self.lib.bdg_assign_reads_dev(self.h, _ptr(d_recs), n, _ptr(d_uniq), nu, _ptr(d_assigned), _ptr(d_has),
                                                  _ptr(d_out_rank), _ptr(d_out_has)))

    def distinct_dev(self, d_recs, n, d_uniq, d_count, d_first, d_n):
        """d_recs: a torch tensor of records or a raw device pointer (kept_records())"""
        p = d_recs if isinstance(d_recs, int) else d_recs.data_ptr()
        self._check(self.lib.bdg_distinct_dev(self.h, p, n, d_uniq.data_ptr(), d_count.data_ptr(),
                                              d_first.data_ptr(), d_n.data_ptr()))


def _ptr(x, byte_offset=0):
    return (x if isinstance(x, int) else x.data_ptr()) + byte_offset


class DeviceArray:
    """A zero-filled array in the context's device memory (bdg_mem_alloc): what the package's own command lines pass to
    the *_dev entry points in place of a torch tensor, so that they start without importing torch."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = (shape,) if isinstance(shape, int) else tuple(shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        ctx._check(ctx.lib.bdg_mem_alloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value or 0

    def data_ptr(self):
        return self.ptr

    def to_host(self, rows=None):
        """the first `rows` rows (all by default) as a numpy array; waits for the context's stream"""
        rows = self.shape[0] if rows is None else int(rows)
        out = np.zeros((rows,) + self.shape[1:], dtype=self.dtype)
        self.ctx._check(self.ctx.lib.bdg_mem_to_host(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    @classmethod
    def from_host(cls, ctx, arr):
        """a device copy of a numpy array (at least one element is allocated)"""
        arr = np.ascontiguousarray(arr)
        d = cls(ctx, arr.shape if arr.size else (1,) + tuple(arr.shape[1:]), arr.dtype)
        if arr.size:
            ctx._check(ctx.lib.bdg_mem_from_host(ctx.h, d.ptr, arr.ctypes.data, arr.nbytes))
        return d

    def free(self):
        if self.ptr and self.ctx.h:
            self.ctx.lib.bdg_mem_free(self.ctx.h, self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Ingest:
    """[gzipped / BGZF] FASTA / FASTQ / SAM / BAM -> chunks of reads in pinned host memory, parsed by native threads
    (bdg_ingest_*).  next() yields IngestChunk structures; release(chunk) hands the memory back to the readers."""

    def __init__(self, path, chunk_reads=100000, ring_chunks=4, pinned=True, inflate_threads=0, segment_bytes=0,
                 skip_secondary=False):
        """inflate_threads: reader threads (inflate + parse); 0 = min(12, cores), 1 = one sequential reader"""
        self.lib = load()
        h = C.c_void_p()
        o = IngestOpts(chunk_reads, ring_chunks, 1 if pinned else 0, inflate_threads, segment_bytes, 1 if skip_secondary else 0, 0)
        rc = self.lib.bdg_ingest_open_ex(os.fsencode(path), C.byref(o), C.byref(h))
        if rc != 0:
            raise BadgerHipError(rc, "cannot read %s (unknown extension or unreadable file)" % path)
        self.h = h

    def next(self):
        ch = IngestChunk()
        rc = self.lib.bdg_ingest_next(self.h, C.byref(ch))
        if rc == E_FORMAT:
            raise ValueError(self.lib.bdg_ingest_error(self.h).decode())
        if rc == E_NOSEQ:
            raise TypeError(self.lib.bdg_ingest_error(self.h).decode())      # the reference: len(None) in find_barcode_umi
        if rc != 0:
            raise BadgerHipError(rc, self.lib.bdg_ingest_error(self.h).decode())
        return ch

    def release(self, ch):
        self.lib.bdg_ingest_release(self.h, ch.id)

    def reads(self):
        """reads in the chunks made so far"""
        return int(self.lib.bdg_ingest_reads(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.bdg_ingest_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def chunk_reads(ch):
    """(read id, sequence) pairs of an ingest chunk, in order (tests and small tools; the pipeline never builds strings)"""
    n = ch.n
    if not n:
        return []
    off = np.ctypeslib.as_array(C.cast(ch.off, C.POINTER(C.c_uint64)), shape=(n + 1,)).tolist()
    idoff = np.ctypeslib.as_array(C.cast(ch.id_off, C.POINTER(C.c_uint64)), shape=(n + 1,)).tolist()
    bases = C.string_at(ch.bases + off[0], off[n] - off[0])
    ids = C.string_at(ch.ids + idoff[0], idoff[n] - idoff[0])
    return [(ids[idoff[i] - idoff[0]:idoff[i + 1] - idoff[0]].decode("ascii", "replace"),
             bases[off[i] - off[0]:off[i + 1] - off[0]].decode("ascii", "replace")) for i in range(n)]


def stage1_run(contexts, in_path, out_path, header, umi_len, threads=0, header_every=0, skip_secondary=False,
               chunk_reads=0, segment_bytes=0, format_threads=0):
    """bdg_stage1_run: input file -> TSV in native threads over the given contexts.  Returns the Stage1Result; raises what
    the reference raises: KeyError for a base outside ACGTN, ValueError for a malformed file, TypeError for a record
    without a sequence."""
    L = load()
    arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    o = Stage1Opts(umi_len, threads, format_threads, header_every, chunk_reads, 1 if skip_secondary else 0, segment_bytes)
    res = Stage1Result()
    rc = L.bdg_stage1_run(arr, len(contexts), os.fsencode(in_path), os.fsencode(out_path), header.encode("ascii"), C.byref(o), C.byref(res))
    if rc != 0:
        msg = L.bdg_last_error(contexts[0].h).decode()
        if rc == E_BADBASE:
            raise KeyError(msg)
        if rc == E_FORMAT:
            raise ValueError(msg)
        if rc == E_NOSEQ:
            raise TypeError(msg)
        raise BadgerHipError(rc, msg)
    return res


class IdStore:
    """read ids of a run, kept natively (bdg_idstore_*)"""

    def __init__(self, ids=None):
        self.lib = load()
        self.h = C.c_void_p(self.lib.bdg_idstore_new())
        if ids:
            self.extend(ids)

    def __len__(self):
        return int(self.lib.bdg_idstore_count(self.h))

    def extend(self, ids):
        """append a list of str"""
        if not len(ids):
            return
        text = "".join(ids).encode("ascii", "replace")
        off = np.zeros(len(ids) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(np.fromiter((len(x) for x in ids), dtype=np.uint64, count=len(ids)))
        if int(off[-1]) != len(text):
            raise ValueError("read ids must be ASCII")
        if self.lib.bdg_idstore_append(self.h, text, off.ctypes.data, len(ids)) != 0:
            raise BadgerHipError(E_ARG, "bdg_idstore_append")

    def __getitem__(self, i):
        p, n = C.c_void_p(), C.c_uint32()
        if self.lib.bdg_idstore_get(self.h, i, C.byref(p), C.byref(n)) != 0:
            raise IndexError(i)
        return C.string_at(p.value, n.value).decode("ascii", "replace")

    def to_list(self):
        return [self[i] for i in range(len(self))]

    def close(self):
        if getattr(self, "h", None):
            self.lib.bdg_idstore_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def stage1_collect(ctx, in_path, umi_len, ids, threads=0, skip_secondary=False, chunk_reads=0, segment_bytes=0):
    """bdg_stage1_collect: every read of the file through the context (records stay on the device if it keeps them), the
    read ids into the IdStore.  Raises like stage1_run."""
    L = load()
    o = Stage1Opts(umi_len, threads, 0, 0, chunk_reads, 1 if skip_secondary else 0, segment_bytes)
    res = Stage1Result()
    rc = L.bdg_stage1_collect(ctx.h, os.fsencode(in_path), C.byref(o), ids.h, C.byref(res))
    if rc != 0:
        msg = L.bdg_last_error(ctx.h).decode()
        if rc == E_BADBASE:
            raise KeyError(msg)
        if rc == E_FORMAT:
            raise ValueError(msg)
        if rc == E_NOSEQ:
            raise TypeError(msg)
        raise BadgerHipError(rc, msg)
    return res


def import_stage1_tsv(path, bc_len=16):
    """a stage-1 TSV the way badger.py:91-111 reads it -> (IdStore of the read ids, rank uint32[n], usable bool[n])"""
    L = load()
    ids = IdStore()
    pr, pu, n, bad = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
    rc = L.bdg_import_stage1_tsv(os.fsencode(path), bc_len, ids.h, C.byref(pr), C.byref(pu), C.byref(n), C.byref(bad))
    if rc == E_BADBASE:
        raise KeyError("the barcode in line %d of %s holds a letter outside ACGT" % (bad.value, path))
    if rc == E_FORMAT:
        raise ValueError("%s is empty or has no '#read_id' / 'barcode' column" % path)
    if rc != 0:
        raise BadgerHipError(rc, "cannot read %s" % path)
    k = int(n.value)
    if k == 0 or not pr.value or not pu.value:                # (a header and no rows)
        for q in (pr, pu):
            if q.value:
                L.bdg_host_free(q)
        return ids, np.zeros(0, np.uint32), np.zeros(0, bool)
    rank = np.ctypeslib.as_array(C.cast(pr, C.POINTER(C.c_uint32)), shape=(max(k, 1),))[:k].copy()
    usable = np.ctypeslib.as_array(C.cast(pu, C.POINTER(C.c_uint8)), shape=(max(k, 1),))[:k].astype(bool)
    L.bdg_host_free(pr)
    L.bdg_host_free(pu)
    return ids, rank, usable


def write_assignments(ids, rank, has, path):
    """<path>: "readID\tbarcode" and one line per read (bdg_write_assignments)"""
    L = load()
    rank = np.ascontiguousarray(rank, dtype=np.uint32)
    has = np.ascontiguousarray(has, dtype=np.uint8)
    rc = L.bdg_write_assignments(ids.h, rank.ctypes.data, has.ctypes.data, len(rank), os.fsencode(path))
    if rc != 0:
        raise BadgerHipError(rc, "bdg_write_assignments(%s): %d reads, %d ids" % (path, len(rank), len(ids)))


def format_rows(ch, recs):
    """TSV rows of a chunk as bytes (one "\n"-terminated line per read) + (reads, barcodes, polyT, R1) counts"""
    L = load()
    recs = np.ascontiguousarray(recs)
    counts = (C.c_uint64 * 4)()
    need = L.bdg_format_rows(C.byref(ch), recs.ctypes.data, None, 0, counts)
    if need < 0:
        raise BadgerHipError(int(need), "bdg_format_rows")
    buf = C.create_string_buffer(int(need) + 1)
    got = L.bdg_format_rows(C.byref(ch), recs.ctypes.data, buf, int(need), counts)
    if got < 0 or got > need:
        raise BadgerHipError(int(got), "bdg_format_rows")
    return buf.raw[:got], tuple(int(x) for x in counts)


_DEFAULT = {}


def device_count():
    """devices this process may open (no context needed)"""
    return int(load().bdg_device_count())


def default_context(device=0, instance=0):
    """Process-wide context per device (created on first use; raises without a GPU).  `instance` > 0 gives further,
    independent contexts on the same device (each with its own stream and workspaces)."""
    key = (device, instance)
    if key not in _DEFAULT:
        _DEFAULT[key] = Context(device)
    return _DEFAULT[key]
