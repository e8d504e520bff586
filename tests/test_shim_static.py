"""CPU: the Rust side (shim/) cannot be compiled here (no toolchain), so it is checked statically: the raw binding
shim/src/hip_ffi.rs is REGENERATED from include/recgraph_hip.h and must equal the committed file; it declares every
symbol the library exports; every `rg_*` function / constant the hand-written Rust files use exists in it; the
#[repr(C)] structs have the header's fields in the header's order with the widths ctypes uses."""
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "shim", "src")


def _ffi():
    return open(os.path.join(SHIM, "hip_ffi.rs")).read()


def test_generated_binding_is_current():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py")], capture_output=True, text=True, check=True)
    assert out.stdout == _ffi(), "run: python3 tools/gen_rust_ffi.py --write"


def test_binding_covers_the_library():
    from recgraph_amd import _lib
    ffi = _ffi()
    fns = set(re.findall(r"pub fn (rg_[a-z_0-9]+)\(", ffi))
    assert fns == set(_lib.SYMBOLS)
    lib = _lib.load()
    for f in fns:
        getattr(lib, f)


def test_hand_written_rust_only_uses_declared_items():
    ffi = _ffi()
    declared = set(re.findall(r"pub fn (rg_[a-z_0-9]+)", ffi)) | set(re.findall(r"pub const (RG_[A-Z0-9_]+)", ffi)) | \
        set(re.findall(r"pub struct (rg_[a-z_]+)", ffi))
    for name in ("hip.rs", "api.rs", "main_loop.rs"):
        src = open(os.path.join(SHIM, name)).read()
        used = set(re.findall(r"\b(rg_[a-z_0-9]+|RG_[A-Z0-9_]+)\b", src))
        missing = {u for u in used if u not in declared}
        assert not missing, (name, sorted(missing))
    hip = open(os.path.join(SHIM, "hip.rs")).read()
    assert "pub use crate::hip_ffi::*;" in hip
    # the read loop of main.rs for every accelerated mode, over the stream
    loop = open(os.path.join(SHIM, "main_loop.rs")).read()
    for mode in ("RG_MODE_GLOBAL_POA", "RG_MODE_GAP_POA", "RG_MODE_PATHWISE", "RG_MODE_RECOMBINATION", "RG_MODE_PATHWISE_SEMI",
                 "RG_MODE_RECOMBINATION_SEMI", "RG_MODE_LOCAL_POA", "RG_MODE_GAP_LOCAL_POA"):
        assert mode in loop
    assert "feed_fasta" in loop and "write_gaf" in loop
    # bounded like the reference's loop, and `-s true` goes into the library
    assert "max_queued_tiles" in loop and "max_undelivered_bytes" in loop and "amb_strand" in loop


def test_repr_c_structs_match_the_ctypes_mirror():
    """Field names, order and widths of the Rust structs == the ctypes structures the GPU tests run through."""
    from recgraph_amd import _lib
    ffi = _ffi()
    width = {"i32": 4, "u32": 4, "f32": 4, "i64": 8, "u64": 8, "f64": 8, "c_char": 1}
    for rust, ct in (("rg_params", _lib.Params), ("rg_gaf_fields", _lib.GafFields), ("rg_stream_opts", _lib.StreamOpts),
                     ("rg_stream_result", _lib.StreamResult)):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % rust, ffi, flags=re.S).group(1)
        fields = re.findall(r"pub (\w+): ([^,]+),", body)
        assert [f for f, _ in fields] == [f[0] for f in ct._fields_], rust
        for (fname, rtype), cf in zip(fields, ct._fields_):
            csize = C.sizeof(cf[1])
            arr = re.match(r"\[(\w+); (\d+)\]", rtype)
            rsize = width[arr.group(1)] * int(arr.group(2)) if arr else (8 if rtype.startswith("*") else width[rtype])
            assert rsize == csize, (rust, fname, rtype)
