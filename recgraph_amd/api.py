"""Host-side mirror of the reference's call surface for the DP hot path.

Function names, argument meaning and defaults follow the reference (file:line relative to the
RecGraph tree) so that the parity tests read like the reference's own tests:

* ``align_global_no_gap`` / ``align_global_gap``  — ``src/api.rs:11``, ``src/api.rs:43``
* ``create_score_matrix_i32`` / ``_f32``           — ``src/api.rs:131``, ``src/api.rs:153``
* ``pathwise_alignment_exec``                      — ``src/pathwise_alignment.rs:5`` (+ ``main.rs:255-261``)
* ``pathwise_alignment_recombination_exec``        — ``src/pathwise_alignment_recombination.rs:23``
* ``GAFStruct``                                    — ``src/gaf_output.rs:6-94``
* ``align_batch``                                  — the per-read loops of ``src/main.rs:56-105,174-213,257-261,297-312``

Everything is computed by ``librecgraph_hip.so`` on the GPU; nothing here falls back to a CPU path.
"""
import ctypes as C
from dataclasses import dataclass, field

from . import _lib
from ._lib import Params, check



class _LazyNumpy:
    """numpy is imported at first use: the CLI's streaming path starts without it (200 ms of a 3 s job)."""

    def __getattr__(self, name):
        import numpy
        return getattr(numpy, name)


np = _LazyNumpy()

ALPHABET = "ACGTN-"
SCORE_MISSING = -536870912

MODE_GLOBAL_POA = 0
MODE_GLOBAL_POA_SCALAR = 10
MODE_GAP_POA = 2
MODE_PATHWISE = 4
MODE_RECOMBINATION = 8
MODE_PATHWISE_SEMI = 5
MODE_RECOMBINATION_SEMI = 9
MODE_LOCAL_POA = 1
MODE_LOCAL_POA_SCALAR = 11
MODE_GAP_LOCAL_POA = 3

READ_BAND_WARNING, READ_BAND_NOT_ENOUGH, READ_WOULD_PANIC, READ_BAD_BASE = 1, 2, 4, 8


# ----------------------------------------------------------------------------------------------
# score matrices (HashMap<(char,char), i32|f32> of the reference as a dict)
# ----------------------------------------------------------------------------------------------
def create_score_matrix_i32(match_score=None, mismatch_score=None, matrix_file_path=None):
    """api.rs:131-152: a .mtx file (score_matrix.rs:67-105, every gap entry -200) or match/mismatch scores, where any
    pairing with '-' scores 2*mismatch (score_matrix.rs:35-51).  Without a file both scores are required: the reference
    unwraps them (api.rs:143-146)."""
    if matrix_file_path is not None:
        return _matrix_from_mtx(matrix_file_path)
    if match_score is None or mismatch_score is None:
        raise _lib.RecGraphError(-1, "called `Option::unwrap()` on a `None` value (api.rs:143-146)")
    out = (C.c_int32 * 36)()
    _lib.load().rg_scores_match_mis(match_score, mismatch_score, 0, out)
    return _dict_from_table(out)


def create_score_matrix_f32(match_score=None, mismatch_score=None, matrix_type=None):
    """api.rs:153-164: the i32 matrix above with every value as f32 — so its gap entries are 2*mismatch, unlike the
    DEFAULT matrix of align_global_no_gap / align_local_no_gap (score_matrix.rs:52-66, gap = mismatch)."""
    return {k: float(v) for k, v in create_score_matrix_i32(match_score, mismatch_score, matrix_type).items()}


def _score_matrix_match_mis_f32(m, x):
    """score_matrix::create_score_matrix_match_mis_f32 (score_matrix.rs:52-66): gap entries equal the mismatch score."""
    out = (C.c_int32 * 36)()
    _lib.load().rg_scores_match_mis(int(m), int(x), 1, out)
    return {k: float(v) for k, v in _dict_from_table(out).items()}


def _matrix_from_mtx(path):
    rows = [ln.split() for ln in open(path).read().splitlines() if ln.strip()]
    cols = rows[0]
    d = {}
    for r in rows[1:]:
        for c, v in zip(cols, r[1:]):
            d[(r[0][0], c[0])] = int(v)
    for ch in "ACGTN":
        d[(ch, "-")] = -200
        d[("-", ch)] = -200
    d.pop(("-", "-"), None)
    return d


def _dict_from_table(t):
    return {(a, b): int(t[i * 6 + j]) for i, a in enumerate(ALPHABET) for j, b in enumerate(ALPHABET)
            if t[i * 6 + j] != SCORE_MISSING}


def _table_from_dict(d):
    t = [SCORE_MISSING] * 36
    for (a, b), v in d.items():
        if float(v) != int(v):
            raise ValueError("non-integer scores are outside the exact range of the f32 path (SURVEY A.1 item 9)")
        t[ALPHABET.index(a) * 6 + ALPHABET.index(b)] = int(v)
    return t


# ----------------------------------------------------------------------------------------------
@dataclass
class GAFStruct:
    """gaf_output.rs:6-20; ``to_string`` is gaf_output.rs:70-94."""
    query_name: str = ""
    query_length: int = 0
    query_start: int = 0
    query_end: int = 0
    strand: str = " "
    path: list = field(default_factory=lambda: [0])
    path_length: int = 0
    path_start: int = 0
    path_end: int = 0
    residue_matches_number: int = 0
    alignment_block_length: str = ""
    mapping_quality: str = ""
    comments: str = ""

    def to_string(self):
        return "\t".join([self.query_name, str(self.query_length), str(self.query_start), str(self.query_end),
                          self.strand, ">" + ">".join(str(p) for p in self.path), str(self.path_length),
                          str(self.path_start), str(self.path_end), str(self.residue_matches_number),
                          self.alignment_block_length, self.mapping_quality, self.comments])

    @classmethod
    def from_line(cls, line):
        f = line.split("\t", 12)
        path = [int(x) for x in f[5].split(">")[1:]] if f[5] != ">" else []
        return cls(f[0], int(f[1]), int(f[2]), int(f[3]), f[4], path, int(f[6]), int(f[7]), int(f[8]), int(f[9]),
                   f[10], f[11], f[12])


class Graph:
    """A flattened graph resident in HBM (LnzGraph, graph.rs:23-27, and — when the GFA has P lines —
    PathGraph + reverse PathGraph + distance tables, pathwise_graph.rs:10-18,250-354)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_gfa_text(cls, text):
        lib = _lib.load()
        h = C.c_void_p()
        b = text.encode()
        check(lib.rg_graph_from_gfa(b, len(b), C.byref(h)))
        return cls(h)

    @classmethod
    def from_gfa(cls, path):
        """graph::read_graph (graph.rs:11) / pathwise_graph::read_graph_w_path (pathwise_graph.rs:127)."""
        with open(path) as f:
            return cls.from_gfa_text(f.read())

    @classmethod
    def from_lnz(cls, lnz, pred_hash, node_id=None):
        """LnzGraph literal as the reference's unit tests build it (global_abpoa.rs:576-755):
        ``pred_hash`` = {row: [predecessor rows]}, nwp = rows that have an entry."""
        lib = _lib.load()
        L = len(lnz)
        off, rows = [0], []
        for i in range(L):
            rows += list(pred_hash.get(i, []))
            off.append(len(rows))
        offa = (C.c_int64 * len(off))(*off)
        rowa = (C.c_int64 * max(1, len(rows)))(*rows)
        ida = (C.c_uint64 * L)(*node_id) if node_id is not None else None
        h = C.c_void_p()
        check(lib.rg_graph_create_lnz(lnz.encode(), L, offa, rowa, ida, C.byref(h)))
        return cls(h)

    @classmethod
    def from_path_arrays(cls, lnz, paths_number, row_paths, pred_hash, node_id):
        """PathGraph literal (pathwise_graph.rs:10-18) through rg_graph_create_path: ``row_paths[i]`` = set of path ids
        through row i, ``pred_hash`` = {row: {pred_row: set of path ids}} (PredHash, pathwise_graph.rs:75-125)."""
        lib = _lib.load()
        L, W = len(lnz), (paths_number + 63) // 64

        def words(paths):
            w = [0] * W
            for k in paths:
                w[k >> 6] |= 1 << (k & 63)
            return w
        rm = [x for i in range(L) for x in words(row_paths[i])]
        off, preds, masks = [0], [], []
        for i in range(L):
            for p_, ks in sorted(pred_hash.get(i, {}).items()):
                preds.append(p_)
                masks += words(ks)
            off.append(len(preds))
        h = C.c_void_p()
        check(lib.rg_graph_create_path(lnz.encode(), L, paths_number, (C.c_uint64 * len(rm))(*rm), (C.c_int64 * len(off))(*off),
                                       (C.c_int64 * max(1, len(preds)))(*preds), (C.c_uint64 * max(1, len(masks)))(*masks),
                                       (C.c_uint64 * L)(*node_id), C.byref(h)))
        return cls(h)

    def __del__(self):
        try:
            _lib.load().rg_graph_destroy(self._h)
        except Exception:
            pass

    @property
    def rows(self):
        return _lib.load().rg_graph_rows(self._h)

    @property
    def paths_number(self):
        return _lib.load().rg_graph_paths(self._h)

    @property
    def path_error(self):
        """Why a GFA with P lines has no PathGraph view ('' when it has one): modes 0-3 still work on such a graph."""
        return _lib.load().rg_graph_path_error(self._h).decode()

    def dump(self, which):
        lib = _lib.load()
        n = lib.rg_graph_dump(self._h, which, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib.rg_graph_dump(self._h, which, buf, n + 1)
        return buf.value.decode()


def make_params(mode, score_matrix=None, o=None, e=None, b=None, f=None, bta=None, R=None, r=None, B=None, amb=None):
    p = Params()
    _lib.load().rg_params_default(C.byref(p), mode)
    if score_matrix is not None:
        t = score_matrix if isinstance(score_matrix, (list, tuple)) else _table_from_dict(score_matrix)
        for i in range(36):
            p.scores[i] = t[i]
    if o is not None:
        p.gap_open = o
    if e is not None:
        p.gap_ext = e
    if b is not None:
        p.band_b = b
    if f is not None:
        p.band_f = f
    if bta is not None:
        p.bta_override = int(bta)
    if R is not None:
        p.base_rec_cost = R
    if r is not None:
        p.multi_rec_cost = r
    if B is not None:
        p.rec_band_width = B
    if amb is not None:
        p.amb_mode = amb
    return p


class Batch:
    """One rg_batch: reads resident in HBM, re-runnable (what bench.py times)."""

    def __init__(self, graph, reads, params):
        lib = _lib.load()
        self.graph = graph
        self.n = len(reads)
        blob = "".join(reads).encode()
        offs = np.zeros(self.n + 1, dtype=np.int64)
        np.cumsum([len(x) for x in reads], out=offs[1:])
        self._h = C.c_void_p()
        check(lib.rg_batch_create(graph._h, C.byref(params), blob, offs.ctypes.data_as(C.POINTER(C.c_int64)), self.n,
                                  C.byref(self._h)))

    @staticmethod
    def pack_reads(reads):
        """The C ABI's read-set form: (concatenated bases as bytes, int64 offsets[n + 1])."""
        offs = np.zeros(len(reads) + 1, dtype=np.int64)
        np.cumsum([len(x) for x in reads], out=offs[1:])
        return "".join(reads).encode(), offs

    def set_reads(self, reads):
        """Replace the reads of this handle (work buffers in HBM are kept): rg_batch_set_reads.  ``reads`` is a list of
        strings or an already packed ``(bytes, offsets)`` pair (``pack_reads``)."""
        blob, offs = reads if isinstance(reads, tuple) else self.pack_reads(reads)
        self.n = len(offs) - 1
        check(_lib.load().rg_batch_set_reads(self._h, blob, offs.ctypes.data_as(C.POINTER(C.c_int64)), self.n))

    def __del__(self):
        try:
            _lib.load().rg_batch_destroy(self._h)
        except Exception:
            pass

    def run(self):
        check(_lib.load().rg_batch_run(self._h))

    def fetch(self):
        check(_lib.load().rg_batch_fetch(self._h))

    def status(self, i):
        return _lib.load().rg_result_status(self._h, i)

    def score(self, i):
        return _lib.load().rg_result_score(self._h, i)

    def gaf_text(self, i, name, seq_index=1):
        lib = _lib.load()
        nb = name.encode()
        n = lib.rg_result_gaf(self._h, i, nb, seq_index, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib.rg_result_gaf(self._h, i, nb, seq_index, buf, n + 1)
        return buf.value.decode()

    def fields(self, i, name=""):
        """The record of read i as a GAFStruct built from rg_result_fields (no text parsing); None when the
        reference produces no GAFStruct for it (it panics on this read)."""
        lib = _lib.load()
        f = _lib.GafFields()
        check(lib.rg_result_fields(self._h, i, C.byref(f), None, 0, None, 0))
        if not f.has_record:
            return None
        ids = (C.c_uint64 * max(1, f.n_path_ids))()
        com = C.create_string_buffer(f.comments_len + 1)
        check(lib.rg_result_fields(self._h, i, C.byref(f), ids, f.n_path_ids, com, f.comments_len + 1))
        star = "" if f.empty else "*"
        return GAFStruct("" if f.empty else name, f.query_length, f.query_start, f.query_end, f.strand.decode(),
                         [int(ids[k]) for k in range(f.n_path_ids)], f.path_length, f.path_start, f.path_end,
                         f.residue_matches_number, star, star, com.value.decode())

    def format_all(self, names=None, seq_index_base=1, nthreads=8):
        """All GAF text of the batch in one buffer (C++ host threads)."""
        lib = _lib.load()
        arr = None
        if names is not None:
            arr = (C.c_char_p * self.n)(*[x.encode() for x in names])
        # one formatting pass when the buffer of the previous call (or the estimate) is large enough
        cap = getattr(self, "_fmt_cap", 0) or (4096 + 2048 * self.n)
        while True:
            buf = C.create_string_buffer(cap)
            need = lib.rg_batch_format_all(self._h, arr, seq_index_base, buf, cap, nthreads)
            if need < 0:
                raise _lib.RecGraphError(need, lib.rg_last_error().decode())
            if need + 1 <= cap:
                break
            cap = need + 4096
        self._fmt_cap = need + 4096
        return buf.raw[:need]

    @property
    def cell_updates(self):
        return _lib.load().rg_batch_cell_updates(self._h)

    @property
    def cell_updates_performed(self):
        return _lib.load().rg_batch_cell_updates_performed(self._h)

    def kernel_stats(self):
        lib = _lib.load()
        return {lib.rg_batch_kernel_name(self._h, k).decode(): (lib.rg_batch_kernel_ms(self._h, k),
                                                                 lib.rg_batch_kernel_launches(self._h, k))
                for k in range(lib.rg_batch_kernel_count(self._h))}


class _ShardView(Batch):
    """A shard of an rg_multi: borrowed rg_batch handle (owned by the rg_multi)."""

    def __init__(self, handle, n, owner):
        self._h, self.n, self._owner = handle, n, owner

    def __del__(self):
        pass


class MultiBatch:
    """rg_align_batch_multi: the read loop over several GPUs behind one call (the streaming engine inside: tiles of
    reads pulled by the batch handles of every device, one results-only shard per tile, results in input order)."""

    def __init__(self, graph, reads, params, device_ids=None):
        lib = _lib.load()
        self.graph = graph
        self.n = len(reads)
        blob = "".join(reads).encode()
        offs = np.zeros(self.n + 1, dtype=np.int64)
        np.cumsum([len(x) for x in reads], out=offs[1:])
        self._h = C.c_void_p()
        devs = (C.c_int32 * len(device_ids))(*device_ids) if device_ids is not None else None
        check(lib.rg_align_batch_multi(graph._h, C.byref(params), blob, offs.ctypes.data_as(C.POINTER(C.c_int64)), self.n,
                                       devs, len(device_ids) if device_ids is not None else 0, C.byref(self._h)))
        k = lib.rg_multi_shards(self._h)
        self.begin = [lib.rg_multi_shard_begin(self._h, i) for i in range(k + 1)]
        self.shards = [_ShardView(C.c_void_p(lib.rg_multi_batch(self._h, i)), self.begin[i + 1] - self.begin[i], self) for i in range(k)]

    def __del__(self):
        try:
            _lib.load().rg_multi_destroy(self._h)
        except Exception:
            pass

    def locate(self, i):
        """(shard, index inside the shard) of read i."""
        import bisect
        k = bisect.bisect_right(self.begin, i) - 1
        return self.shards[k], i - self.begin[k]

    def format_all(self, names, seq_index_base=1, nthreads=8):
        lib = _lib.load()
        arr = (C.c_char_p * self.n)(*[x.encode() for x in names])
        cap = 4096 + 2048 * self.n
        while True:
            buf = C.create_string_buffer(cap)
            need = lib.rg_multi_format_all(self._h, arr, seq_index_base, buf, cap, nthreads)
            if need < 0:
                raise _lib.RecGraphError(need, lib.rg_last_error().decode())
            if need + 1 <= cap:
                return buf.raw[:need]
            cap = need + 4096


class Reads:
    """sequences::get_sequences (sequences.rs:5-45) behind the C ABI (rg_reads_from_fasta): the bases of all reads in one
    blob + offsets + names, in the input form of ``Batch`` / ``Stream.push`` (no per-character Python)."""

    def __init__(self, handle):
        lib = _lib.load()
        self._h = handle
        self.n = lib.rg_reads_count(handle)
        self._bases = lib.rg_reads_bases(handle)
        self._off = lib.rg_reads_offsets(handle)
        self._names = lib.rg_reads_names(handle)

    @classmethod
    def from_fasta_text(cls, text):
        b = text if isinstance(text, bytes) else text.encode()
        h = C.c_void_p()
        check(_lib.load().rg_reads_from_fasta(b, len(b), C.byref(h)))
        return cls(h)

    @classmethod
    def from_fasta(cls, path):
        with open(path, "rb") as f:
            return cls.from_fasta_text(f.read())

    def __del__(self):
        try:
            _lib.load().rg_reads_destroy(self._h)
        except Exception:
            pass

    def __len__(self):
        return self.n

    @property
    def offsets(self):
        return np.ctypeslib.as_array(self._off, shape=(self.n + 1,))

    @property
    def names(self):
        return [self._names[i].decode() for i in range(self.n)]

    def sequences(self):
        """The reads as strings (tests / the ``-s true`` path)."""
        off = self.offsets
        blob = C.string_at(self._bases, int(off[-1]))
        return [blob[off[i]:off[i + 1]].decode() for i in range(self.n)]


def fasta_check(path, block=8 << 20):
    """rg_fasta_check over a file, block by block: the number of reads, or RecGraphError("wrong fasta file format") where
    the reference panics (sequences.rs:41-43).  Holds one block."""
    lib = _lib.load()
    st = (C.c_int64 * 4)()
    n = C.c_int64(0)
    with open(path, "rb") as f:
        while True:
            b = f.read(block)
            check(lib.rg_fasta_check(b, len(b), int(not b), st, C.byref(n)))
            if not b:
                return n.value


class StreamTile:
    """One rg_stream_result, copied out of the library's buffers."""
    __slots__ = ("first", "n", "text", "text_off", "status", "score", "device", "cell_updates", "cell_updates_performed", "records")

    def text_of(self, i):
        return self.text[self.text_off[i]:self.text_off[i + 1]]


class Stream:
    """rg_stream: the reference's read loop (main.rs:56,174,257,297-312) as the pipeline hidden behind the C ABI: tiles of
    reads pulled by ``handles_per_device`` batch handles per device from one queue, results in input order."""

    def __init__(self, graph, params, device_ids=None, handles_per_device=0, tile_reads=0, format_threads=0,
                 seq_index_base=1, keep_records=False, no_text=False, spin_wait=False, max_queued_tiles=0,
                 max_undelivered_bytes=0, amb_strand=False):
        """``amb_strand``: ``-s true`` inside the workers (POA modes).  ``max_queued_tiles`` / ``max_undelivered_bytes``:
        bounds on what the stream holds (pushes / workers wait): the pushing and the consuming side must then be
        different threads, or one thread that drains whenever ``pending`` says so."""
        lib = _lib.load()
        self.graph = graph
        o = _lib.StreamOpts()
        lib.rg_stream_opts_default(C.byref(o))
        o.handles_per_device, o.tile_reads, o.format_threads = handles_per_device, tile_reads, format_threads
        o.seq_index_base, o.keep_records, o.no_text, o.spin_wait = seq_index_base, int(keep_records), int(no_text), int(spin_wait)
        o.max_queued_tiles, o.max_undelivered_bytes, o.amb_strand = max_queued_tiles, max_undelivered_bytes, int(amb_strand)
        devs = (C.c_int32 * len(device_ids))(*device_ids) if device_ids is not None else None
        self._h = C.c_void_p()
        check(lib.rg_stream_create(graph._h, C.byref(params), devs, len(device_ids) if device_ids is not None else 0,
                                   C.byref(o), C.byref(self._h)))

    def close(self):
        if self._h:
            _lib.load().rg_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, reads, names=None):
        """``reads``: list of strings, a packed ``(bytes, int64 offsets)`` pair (``Batch.pack_reads``) or a ``Reads``
        (its names are used).  The library copies everything before the call returns."""
        lib = _lib.load()
        if isinstance(reads, Reads):
            check(lib.rg_stream_push(self._h, reads._bases, reads._off, reads.n, reads._names))
            return reads.n
        blob, offs = reads if isinstance(reads, tuple) else Batch.pack_reads(reads)
        n = len(offs) - 1
        arr = (C.c_char_p * n)(*[x.encode() for x in names]) if names is not None else None
        check(lib.rg_stream_push(self._h, blob, offs.ctypes.data_as(C.POINTER(C.c_int64)), n, arr))
        return n

    def push_fasta(self, text):
        """rg_stream_push_fasta: FASTA text (bytes) parsed inside the library (sequences.rs:5-45), its reads pushed tile by
        tile while the rest is parsed.  Returns the number of reads."""
        n = C.c_int64(0)
        check(_lib.load().rg_stream_push_fasta(self._h, text, len(text), C.byref(n)))
        return n.value

    def feed_fasta(self, piece, final=False):
        """rg_stream_feed_fasta: the FASTA text in pieces (any split); ``final`` closes it.  Returns the reads completed
        and pushed by this call."""
        n = C.c_int64(0)
        check(_lib.load().rg_stream_feed_fasta(self._h, piece, len(piece), int(final), C.byref(n)))
        return n.value

    @property
    def pending(self):
        """Tiles pushed and not yet delivered."""
        return _lib.load().rg_stream_pending(self._h)

    def release(self, tile):
        """keep_records: gives the tile's record handle back (rg_stream_release)."""
        if tile.records:
            _lib.load().rg_stream_release(self._h, tile.records)
            tile.records = None

    def finish(self):
        check(_lib.load().rg_stream_finish(self._h))

    def abort(self):
        """rg_stream_abort: the error exit — queued tiles are dropped and every thread blocked in ``push`` / ``feed_fasta`` /
        ``next`` returns with an error, so that a feeder thread can be joined before the stream is closed."""
        if self._h:
            check(_lib.load().rg_stream_abort(self._h))

    def next(self, copy_text=True):
        """The next tile in input order (blocks), or None after ``finish`` when everything was delivered."""
        lib = _lib.load()
        r = _lib.StreamResult()
        rc = lib.rg_stream_next(self._h, C.byref(r))
        if rc == 1:
            return None
        check(rc)
        t = StreamTile()
        t.first, t.n, t.device, t.cell_updates, t.records = r.first_read, r.nreads, r.device, r.cell_updates, r.records
        t.cell_updates_performed = r.cell_updates_performed
        t.text = C.string_at(r.text, r.text_len) if copy_text else (r.text, r.text_len)
        t.text_off = np.ctypeslib.as_array(r.text_off, shape=(r.nreads + 1,)).copy()
        t.status = np.ctypeslib.as_array(r.status, shape=(r.nreads,)).copy()
        t.score = np.ctypeslib.as_array(r.score, shape=(r.nreads,)).copy()
        return t

    def __iter__(self):
        while True:
            t = self.next()
            if t is None:
                return
            yield t

    def kernel_stats(self):
        """{name: (ms summed over the tiles so far, launches)}; the host phases are listed as ``host:*``."""
        lib = _lib.load()
        return {lib.rg_stream_kernel_name(self._h, k).decode(): (lib.rg_stream_kernel_ms(self._h, k), lib.rg_stream_kernel_launches(self._h, k))
                for k in range(lib.rg_stream_kernel_count(self._h))}

    @property
    def handles(self):
        return _lib.load().rg_stream_handles(self._h)


def set_option(name, value):
    """rg_set_option: process-wide diagnostic switches ("sweep_i32", "three_sweeps", "no_frec", "debug")."""
    check(_lib.load().rg_set_option(name.encode(), int(value)))


def align_stream(graph, reads, names=None, mode=MODE_GLOBAL_POA, seq_index_base=1, device_ids=None, handles_per_device=0,
                 tile_reads=0, amb_strand=False, **kw):
    """``align_batch`` through the streaming engine (every visible GPU unless ``device_ids`` says otherwise): returns the
    per-read texts and status bits in input order.  ``amb_strand``: the ``-s true`` retry inside the library."""
    st = Stream(graph, make_params(mode, **kw), device_ids=device_ids, handles_per_device=handles_per_device,
                tile_reads=tile_reads, seq_index_base=seq_index_base, amb_strand=amb_strand)
    st.push(reads, names)
    st.finish()
    texts, status = [], []
    for t in st:
        texts += [t.text_of(i).decode() for i in range(t.n)]
        status += [int(x) for x in t.status]
    st.close()
    return texts, status


_COMPLEMENT = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def rev_and_compl(read):
    """sequences::rev_and_compl (sequences.rs:65-82) on a read without its '$'."""
    try:
        return "".join(_COMPLEMENT[c] for c in reversed(read))
    except KeyError as ex:
        raise _lib.RecGraphError(-1, "wrong char: %s, unable to rev&compl" % ex.args[0])


def align_batch(graph, reads, names=None, mode=MODE_GLOBAL_POA, seq_index_base=1, amb_strand=False, **kw):
    """The reference's per-read loop as one device batch.  Returns, per read, exactly the text the
    reference prints on stdout (warning lines + GAF line), and the per-read status bits.

    ``amb_strand`` is ``-s true`` (main.rs:82-106, 132-165, 188-212, 229-253; POA modes only): a second batch aligns
    the reverse complement of the reads that qualify against the same graph, labelled with the reversed handle order,
    and the reference's per-mode comparison picks which record is written."""
    p = make_params(mode, **kw)
    b = Batch(graph, reads, p)
    b.run()
    b.fetch()
    n = len(reads)
    names = names or ["read%d" % i for i in range(n)]
    texts = [b.gaf_text(i, names[i], seq_index_base + i) for i in range(n)]
    status = [b.status(i) for i in range(n)]
    poa_global = (MODE_GLOBAL_POA, MODE_GLOBAL_POA_SCALAR, MODE_GAP_POA)
    poa_local = (MODE_LOCAL_POA, MODE_LOCAL_POA_SCALAR, MODE_GAP_LOCAL_POA)
    if not amb_strand or mode not in poa_global + poa_local:
        return texts, status
    ok = [i for i in range(n) if not status[i] & (READ_WOULD_PANIC | READ_BAD_BASE)]
    sel = [i for i in ok if mode in poa_local or b.score(i) < 0]          # main.rs:82,188 `alignment.0 < 0`
    if not sel:
        return texts, status
    rmode = MODE_GLOBAL_POA_SCALAR if mode == MODE_GLOBAL_POA else mode    # main.rs:88: the retry uses the scalar exec
    amb = 1 if mode == MODE_GAP_LOCAL_POA else 3                           # main.rs:240 passes amb_mode = false
    canon = ["".join("N" if c == "-" else c.upper() for c in reads[i]) for i in sel]   # sequences.rs:13-22
    rb = Batch(graph, [rev_and_compl(r) for r in canon], make_params(rmode, amb=amb, **kw))
    rb.run()
    rb.fetch()
    for k, i in enumerate(sel):
        if rb.status(k) & (READ_WOULD_PANIC | READ_BAD_BASE):
            status[i] |= rb.status(k) & (READ_WOULD_PANIC | READ_BAD_BASE)
            continue
        fl = texts[i].splitlines(True)
        rl = rb.gaf_text(k, names[i], seq_index_base + i).splitlines(True)
        if mode in (MODE_LOCAL_POA, MODE_LOCAL_POA_SCALAR):
            take_rev = not (b.score(i) < rb.score(k))                      # main.rs:160-164 (sic)
        else:
            take_rev = rb.score(k) > b.score(i)
        # warning lines are printed while the two exec calls run; write_gaf then prints the chosen record
        texts[i] = "".join(fl[:-1]) + "".join(rl[:-1]) + (rl[-1] if take_rev else fl[-1])
    return texts, status


def align_batch_multi(graph, reads, names=None, mode=MODE_GLOBAL_POA, seq_index_base=1, device_ids=None, **kw):
    """``align_batch`` over several GPUs behind one C call (``rg_align_batch_multi``: the streaming engine, one
    results-only shard per tile; ``device_ids`` None = every visible device).  Same return value; no ``-s`` retry."""
    n = len(reads)
    names = names or ["read%d" % i for i in range(n)]
    m = MultiBatch(graph, reads, make_params(mode, **kw), device_ids=device_ids)
    texts, status = [], []
    for k, sh in enumerate(m.shards):
        for j in range(sh.n):
            i = m.begin[k] + j
            texts.append(sh.gaf_text(j, names[i], seq_index_base + i))
            status.append(sh.status(j))
    return texts, status


def _single(graph, read, name, mode, seq_index=1, **kw):
    p = make_params(mode, **kw)
    b = Batch(graph, [read], p)
    b.run()
    b.fetch()
    st = b.status(0)
    if st & (READ_WOULD_PANIC | READ_BAD_BASE):
        raise _lib.RecGraphError(-1, "the reference panics on this input (status %d)" % st)
    return b, b.gaf_text(0, name, seq_index)


def _f32_usize(x):
    return int(np.float32(x))


def align_global_no_gap(read, graph, sequence_name=None, score_matrix=None, bases_to_add=None):
    """api.rs:11-40.  Defaults: f32 matrix (2, -4, gaps -4), bases_to_add = len * 0.1, name ("no_name", 1)."""
    sm = score_matrix if score_matrix is not None else _score_matrix_match_mis_f32(2, -4)
    bta = _f32_usize(np.float32(len(read)) * np.float32(0.1 if bases_to_add is None else bases_to_add))
    name, idx = sequence_name if sequence_name is not None else ("no_name", 1)
    if idx == 0:
        raise _lib.RecGraphError(-1, "alignment.1.unwrap() on None (api.rs:38)")
    _, text = _single(graph, read, name, MODE_GLOBAL_POA, idx, score_matrix=sm, bta=bta)
    lines = text.rstrip("\n").split("\n")
    return GAFStruct.from_line(lines[-1]) if "band not enough" not in text else GAFStruct()


def align_global_gap(read, graph, sequence_name=None, score_matrix=None, bases_to_add=None, o=None, e=None):
    """api.rs:43-72.  Defaults: i32 matrix (2, -4), o = -10, e = -6, bases_to_add = len * 0.1."""
    sm = score_matrix if score_matrix is not None else create_score_matrix_i32(2, -4)
    bta = _f32_usize(np.float32(len(read)) * np.float32(0.1 if bases_to_add is None else bases_to_add))
    name, idx = sequence_name if sequence_name is not None else ("no_name", 1)
    if idx == 0:
        raise _lib.RecGraphError(-1, "alignment.1.unwrap() on None (api.rs:70)")
    _, text = _single(graph, read, name, MODE_GAP_POA, idx, score_matrix=sm, bta=bta, o=-10 if o is None else o,
                      e=-6 if e is None else e)
    return GAFStruct.from_line(text.rstrip("\n").split("\n")[-1])


def align_local_no_gap(read, graph, sequence_name=None, score_matrix=None):
    """api.rs:76-99.  Defaults: f32 matrix (2, -4, gaps -4), name ("no_name", 1)."""
    sm = score_matrix if score_matrix is not None else _score_matrix_match_mis_f32(2, -4)
    name, idx = sequence_name if sequence_name is not None else ("no_name", 1)
    if idx == 0:
        raise _lib.RecGraphError(-1, "alignment.1.unwrap() on None (api.rs:97)")
    _, text = _single(graph, read, name, MODE_LOCAL_POA, idx, score_matrix=sm)
    return GAFStruct.from_line(text.rstrip("\n").split("\n")[-1])


def align_local_gap(read, graph, sequence_name=None, score_matrix=None, o=None, e=None):
    """api.rs:102-128.  Defaults: i32 matrix (2, -4), o = -10, e = -6."""
    sm = score_matrix if score_matrix is not None else create_score_matrix_i32(2, -4)
    name, idx = sequence_name if sequence_name is not None else ("no_name", 1)
    if idx == 0:
        raise _lib.RecGraphError(-1, "alignment.1.unwrap() on None (api.rs:127)")
    _, text = _single(graph, read, name, MODE_GAP_LOCAL_POA, idx, score_matrix=sm, o=-10 if o is None else o,
                      e=-6 if e is None else e)
    return GAFStruct.from_line(text.rstrip("\n").split("\n")[-1])


def local_poa_exec(sequence, seq_name, graph, score_matrix, scalar=True):
    """local_poa::exec (local_poa.rs:176) / exec_simd (:9): returns (score, GAFStruct | None)."""
    mode = MODE_LOCAL_POA_SCALAR if scalar else MODE_LOCAL_POA
    b, text = _single(graph, "".join(sequence[1:]), seq_name[0], mode, seq_name[1], score_matrix=score_matrix)
    gaf = GAFStruct.from_line(text.rstrip("\n").split("\n")[-1]) if seq_name[1] != 0 and text else None
    return b.score(0), gaf


def gap_local_poa_exec(sequence, seq_name, graph, score_matrix, o, e):
    """gap_local_poa::exec (gap_local_poa.rs:6): returns (score, GAFStruct | None)."""
    b, text = _single(graph, "".join(sequence[1:]), seq_name[0], MODE_GAP_LOCAL_POA, seq_name[1],
                      score_matrix=score_matrix, o=o, e=e)
    gaf = GAFStruct.from_line(text.rstrip("\n").split("\n")[-1]) if seq_name[1] != 0 and text else None
    return b.score(0), gaf


def global_abpoa_exec(sequence, seq_name, graph, score_matrix, bta, scalar=True):
    """global_abpoa::exec (global_abpoa.rs:260) / exec_simd (:10): returns (score, GAFStruct | None).
    ``sequence`` carries the leading '$' like the reference's read arrays."""
    mode = MODE_GLOBAL_POA_SCALAR if scalar else MODE_GLOBAL_POA
    b, text = _single(graph, "".join(sequence[1:]), seq_name[0], mode, seq_name[1], score_matrix=score_matrix, bta=bta)
    gaf = GAFStruct.from_line(text.rstrip("\n").split("\n")[-1]) if seq_name[1] != 0 and text and "band not" not in text else None
    return b.score(0), gaf


def gap_global_abpoa_exec(sequence, seq_name, graph, score_matrix, o, e, bta):
    """gap_global_abpoa::exec (gap_global_abpoa.rs:11): returns (score, GAFStruct | None)."""
    b, text = _single(graph, "".join(sequence[1:]), seq_name[0], MODE_GAP_POA, seq_name[1], score_matrix=score_matrix,
                      bta=bta, o=o, e=e)
    gaf = GAFStruct.from_line(text.rstrip("\n").split("\n")[-1]) if seq_name[1] != 0 and text else None
    return b.score(0), gaf


def pathwise_alignment_exec(sequence, graph, score_matrix=None):
    """pathwise_alignment::exec (pathwise_alignment.rs:5): query_name is "Temp" until the caller
    overwrites it (main.rs:259)."""
    _, text = _single(graph, "".join(sequence[1:]), "Temp", MODE_PATHWISE, 1, score_matrix=score_matrix)
    return GAFStruct.from_line(text.rstrip("\n"))


def pathwise_alignment_semiglobal_exec(sequence, graph, score_matrix=None):
    """pathwise_alignment_semiglobal::exec (pathwise_alignment_semiglobal.rs:6)."""
    _, text = _single(graph, "".join(sequence[1:]), "Temp", MODE_PATHWISE_SEMI, 1, score_matrix=score_matrix)
    return GAFStruct.from_line(text.rstrip("\n"))


def pathwise_alignment_recombination_exec(aln_mode, sequence, graph, score_matrix=None, base_rec_cost=4,
                                          multi_rec_cost=0.1, rbw=1.0):
    """pathwise_alignment_recombination::exec (pathwise_alignment_recombination.rs:23); the reverse graph
    and the displacement matrix arguments of the reference are derived inside the graph handle."""
    if aln_mode not in (8, 9):
        raise _lib.RecGraphError(-1, "aln_mode must be 8 (global) or 9 (semiglobal)")
    _, text = _single(graph, "".join(sequence[1:]), "Temp", MODE_RECOMBINATION if aln_mode == 8 else MODE_RECOMBINATION_SEMI, 1,
                      score_matrix=score_matrix,
                      R=base_rec_cost, r=multi_rec_cost, B=rbw)
    return GAFStruct.from_line(text.rstrip("\n"))
