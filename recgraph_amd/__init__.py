"""recgraph_amd — MI355X-native implementation of RecGraph's sequence-to-graph DP hot path.

Python here is only the host-side mirror of the reference's call surface (``api.rs`` and the
per-mode ``exec`` functions) over the C ABI in ``include/recgraph_hip.h``; all computation happens in
``librecgraph_hip.so`` (hand-written HIP for gfx950).  Importing this package never imports the
test oracle.
"""
from .api import (GAFStruct, Graph, Params, align_batch, align_global_gap, align_global_no_gap,  # noqa: F401
                  create_score_matrix_f32, create_score_matrix_i32, pathwise_alignment_exec,
                  pathwise_alignment_recombination_exec)
from ._lib import RecGraphError, build_library, library_path, load  # noqa: F401

__all__ = ["GAFStruct", "Graph", "Params", "align_batch", "align_global_gap", "align_global_no_gap",
           "create_score_matrix_f32", "create_score_matrix_i32", "pathwise_alignment_exec",
           "pathwise_alignment_recombination_exec", "RecGraphError", "build_library", "library_path", "load"]
