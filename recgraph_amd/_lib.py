"""ctypes binding of librecgraph_hip.so (the C ABI declared in include/recgraph_hip.h)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# RG_LIB_PATH: another build of the same library (timing-only kernel variants of tools/sweep_variants.sh); never set in tests
_SO = os.environ.get("RG_LIB_PATH") or os.path.join(_HERE, "librecgraph_hip.so")


class RecGraphError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"recgraph_hip error {code}: {msg}")
        self.code = code


class GafFields(C.Structure):
    _fields_ = [("has_record", C.c_int32), ("empty", C.c_int32), ("warning", C.c_uint32), ("strand", C.c_char),
                ("query_length", C.c_uint64), ("query_start", C.c_uint64), ("query_end", C.c_uint64),
                ("path_length", C.c_uint64), ("path_start", C.c_uint64), ("path_end", C.c_uint64),
                ("residue_matches_number", C.c_uint64), ("n_path_ids", C.c_int64), ("comments_len", C.c_int64)]


class StreamOpts(C.Structure):
    _fields_ = [("handles_per_device", C.c_int32), ("tile_reads", C.c_int32), ("format_threads", C.c_int32),
                ("keep_records", C.c_int32), ("seq_index_base", C.c_int64), ("no_text", C.c_int32), ("spin_wait", C.c_int32),
                ("max_queued_tiles", C.c_int32), ("amb_strand", C.c_int32), ("max_undelivered_bytes", C.c_int64)]


class StreamResult(C.Structure):
    _fields_ = [("first_read", C.c_int64), ("nreads", C.c_int64), ("text", C.c_void_p), ("text_len", C.c_int64),
                ("text_off", C.POINTER(C.c_int64)), ("status", C.POINTER(C.c_uint32)), ("score", C.POINTER(C.c_int32)),
                ("device", C.c_int32), ("reserved", C.c_int32), ("cell_updates", C.c_uint64), ("records", C.c_void_p),
                ("cell_updates_performed", C.c_uint64)]


class Params(C.Structure):
    _fields_ = [("mode", C.c_int32), ("scores", C.c_int32 * 36), ("gap_open", C.c_int32), ("gap_ext", C.c_int32),
                ("band_b", C.c_float), ("band_f", C.c_float), ("bta_override", C.c_int64),
                ("base_rec_cost", C.c_int32), ("multi_rec_cost", C.c_float), ("rec_band_width", C.c_float),
                ("amb_mode", C.c_int32)]


def library_path():
    return _SO


def build_library(force=False):
    """Compile every HIP source for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return _SO


# every symbol include/recgraph_hip.h declares
SYMBOLS = ["rg_params_default", "rg_scores_match_mis", "rg_graph_from_gfa", "rg_graph_create_lnz",
           "rg_graph_create_path", "rg_graph_destroy", "rg_graph_path_error", "rg_graph_rows", "rg_graph_paths", "rg_graph_dump",
           "rg_batch_create", "rg_batch_set_reads", "rg_batch_run", "rg_batch_fetch", "rg_batch_destroy", "rg_batch_size",
           "rg_result_status", "rg_result_score", "rg_result_gaf", "rg_result_fields", "rg_batch_format_all", "rg_batch_cell_updates", "rg_batch_cell_updates_performed", "rg_batch_kernel_count",
           "rg_batch_kernel_name", "rg_batch_kernel_ms", "rg_batch_kernel_launches", "rg_align_batch", "rg_align_batch_multi", "rg_multi_shards", "rg_multi_batch", "rg_multi_shard_begin",
           "rg_multi_format_all", "rg_multi_destroy", "rg_last_error",
           "rg_device_count", "rg_set_device",
           "rg_reads_from_fasta", "rg_reads_count", "rg_reads_bases", "rg_reads_offsets", "rg_reads_names", "rg_reads_destroy", "rg_fasta_check",
           "rg_stream_opts_default", "rg_stream_create", "rg_stream_push", "rg_stream_push_fasta", "rg_stream_feed_fasta", "rg_stream_pending",
           "rg_stream_release", "rg_stream_finish", "rg_stream_abort", "rg_stream_next",
           "rg_stream_destroy", "rg_stream_kernel_count", "rg_stream_kernel_name", "rg_stream_kernel_ms",
           "rg_stream_kernel_launches", "rg_stream_tiles_done", "rg_stream_handles", "rg_set_option", "rg_get_option"]

_lib = None


def load():
    """Load the HIP library; fails loudly when it has not been built (there is no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise RecGraphError(-3, f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(the product has no CPU fallback)")
    l = C.CDLL(_SO)
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    P = C.POINTER
    l.rg_params_default.argtypes = [P(Params), i32]
    l.rg_scores_match_mis.argtypes = [i32, i32, i32, P(i32)]
    l.rg_graph_from_gfa.argtypes = [C.c_char_p, i64, P(vp)]
    l.rg_graph_create_lnz.argtypes = [C.c_char_p, i64, P(i64), P(i64), P(u64), P(vp)]
    l.rg_graph_create_path.argtypes = [C.c_char_p, i64, i32, P(u64), P(i64), P(i64), P(u64), P(u64), P(vp)]
    l.rg_graph_destroy.argtypes = [vp]
    l.rg_graph_path_error.argtypes = [vp]
    l.rg_graph_path_error.restype = C.c_char_p
    l.rg_graph_rows.argtypes = [vp]
    l.rg_graph_rows.restype = i64
    l.rg_graph_paths.argtypes = [vp]
    l.rg_graph_dump.argtypes = [vp, i32, C.c_char_p, i64]
    l.rg_graph_dump.restype = i64
    l.rg_batch_create.argtypes = [vp, P(Params), C.c_char_p, P(i64), i64, P(vp)]
    l.rg_batch_set_reads.argtypes = [vp, C.c_char_p, P(i64), i64]
    l.rg_align_batch.argtypes = [vp, P(Params), C.c_char_p, P(i64), i64, P(vp)]
    for f in ("rg_batch_run", "rg_batch_fetch"):
        getattr(l, f).argtypes = [vp]
    l.rg_batch_destroy.argtypes = [vp]
    l.rg_batch_size.argtypes = [vp]
    l.rg_batch_size.restype = i64
    l.rg_result_status.argtypes = [vp, i64]
    l.rg_result_status.restype = C.c_uint32
    l.rg_result_score.argtypes = [vp, i64]
    l.rg_result_gaf.argtypes = [vp, i64, C.c_char_p, i64, C.c_char_p, i64]
    l.rg_result_gaf.restype = i64
    l.rg_result_fields.argtypes = [vp, i64, P(GafFields), P(u64), i64, C.c_char_p, i64]
    l.rg_align_batch_multi.argtypes = [vp, P(Params), C.c_char_p, P(i64), i64, P(i32), i32, P(vp)]
    l.rg_multi_shards.argtypes = [vp]
    l.rg_multi_batch.argtypes = [vp, i32]
    l.rg_multi_batch.restype = vp
    l.rg_multi_shard_begin.argtypes = [vp, i32]
    l.rg_multi_shard_begin.restype = i64
    l.rg_multi_format_all.argtypes = [vp, P(C.c_char_p), i64, C.c_char_p, i64, i32]
    l.rg_multi_format_all.restype = i64
    l.rg_multi_destroy.argtypes = [vp]
    l.rg_batch_format_all.argtypes = [vp, P(C.c_char_p), i64, C.c_char_p, i64, i32]
    l.rg_batch_format_all.restype = i64
    l.rg_batch_cell_updates.argtypes = [vp]
    l.rg_batch_cell_updates.restype = u64
    l.rg_batch_cell_updates_performed.argtypes = [vp]
    l.rg_batch_cell_updates_performed.restype = u64
    l.rg_batch_kernel_count.argtypes = [vp]
    l.rg_batch_kernel_name.argtypes = [vp, i32]
    l.rg_batch_kernel_name.restype = C.c_char_p
    l.rg_batch_kernel_ms.argtypes = [vp, i32]
    l.rg_batch_kernel_ms.restype = C.c_double
    l.rg_batch_kernel_launches.argtypes = [vp, i32]
    l.rg_batch_kernel_launches.restype = i64
    l.rg_last_error.restype = C.c_char_p
    l.rg_set_device.argtypes = [i32]
    l.rg_reads_from_fasta.argtypes = [C.c_char_p, i64, P(vp)]
    l.rg_reads_count.argtypes = [vp]
    l.rg_reads_count.restype = i64
    l.rg_reads_bases.argtypes = [vp]
    l.rg_reads_bases.restype = vp
    l.rg_reads_offsets.argtypes = [vp]
    l.rg_reads_offsets.restype = P(i64)
    l.rg_reads_names.argtypes = [vp]
    l.rg_reads_names.restype = P(C.c_char_p)
    l.rg_reads_destroy.argtypes = [vp]
    l.rg_fasta_check.argtypes = [C.c_char_p, i64, i32, P(i64), P(i64)]
    l.rg_stream_opts_default.argtypes = [P(StreamOpts)]
    l.rg_stream_create.argtypes = [vp, P(Params), P(i32), i32, P(StreamOpts), P(vp)]
    l.rg_stream_push.argtypes = [vp, vp, P(i64), i64, P(C.c_char_p)]
    l.rg_stream_push_fasta.argtypes = [vp, C.c_char_p, i64, P(i64)]
    l.rg_stream_feed_fasta.argtypes = [vp, C.c_char_p, i64, i32, P(i64)]
    l.rg_stream_pending.argtypes = [vp]
    l.rg_stream_pending.restype = i64
    l.rg_stream_release.argtypes = [vp, vp]
    l.rg_stream_release.restype = None
    l.rg_stream_finish.argtypes = [vp]
    l.rg_stream_abort.argtypes = [vp]
    l.rg_stream_next.argtypes = [vp, P(StreamResult)]
    l.rg_stream_destroy.argtypes = [vp]
    l.rg_stream_kernel_count.argtypes = [vp]
    l.rg_stream_kernel_name.argtypes = [vp, i32]
    l.rg_stream_kernel_name.restype = C.c_char_p
    l.rg_stream_kernel_ms.argtypes = [vp, i32]
    l.rg_stream_kernel_ms.restype = C.c_double
    l.rg_stream_kernel_launches.argtypes = [vp, i32]
    l.rg_stream_kernel_launches.restype = i64
    l.rg_stream_tiles_done.argtypes = [vp]
    l.rg_stream_tiles_done.restype = i64
    l.rg_stream_handles.argtypes = [vp]
    l.rg_set_option.argtypes = [C.c_char_p, i64]
    l.rg_get_option.argtypes = [C.c_char_p]
    l.rg_get_option.restype = i64
    _lib = l
    return l


def check(rc):
    if rc != 0:
        raise RecGraphError(rc, load().rg_last_error().decode())
