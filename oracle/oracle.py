"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front-end of oracle/liboracle.so, the CPU restatement of RecGraph's DP hot path
(see oracle/orc_common.hpp for what it follows and how it is pinned).  Only tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module;
the product (``recgraph_amd``) never does.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

M0_SIMD, M0_SCALAR, M2, M4, M4_ABS, M8, M8_PRUNED, M8_ABS = 0, 10, 2, 4, 14, 8, 18, 28
M5, M5_ABS, M9, M9_PRUNED, M9_ABS = 5, 15, 9, 19, 29
M1_SIMD, M1_SCALAR, M3 = 1, 11, 3


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp"))]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None
_native = None        # (path, description) once use_native() succeeded


def use_native():
    """Timing build for bench.py's cpu_baseline legs: the same sources compiled ``-O3 -march=native`` ON THE MACHINE THAT
    RUNS THEM (oracle/_native/liboracle-<cpu tag>.so; the portable liboracle.so travels between machines, native code
    must not).  Call before the first ``lib()``.  Returns a description, or None when the build failed (the portable
    library is used then)."""
    global _native
    import hashlib
    try:
        info = open("/proc/cpuinfo").read()
        model = [ln for ln in info.splitlines() if ln.startswith("model name")][:1]
        flags = [ln for ln in info.splitlines() if ln.startswith("flags")][:1]
        tag = hashlib.sha256(("".join(model) + "".join(flags)).encode()).hexdigest()[:12]
    except OSError:
        return None
    d = os.path.join(_HERE, "_native")
    so = os.path.join(d, "liboracle-%s.so" % tag)
    srcs = sorted(os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".cpp"))
    hdrs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".hpp")]
    flagsv = ["-O3", "-march=native", "-std=c++17", "-fPIC", "-Wno-sign-compare", "-ffp-contract=off", "-pthread"]
    try:
        if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in srcs + hdrs):
            os.makedirs(d, exist_ok=True)
            tmp = so + ".%d.tmp" % os.getpid()
            subprocess.check_call(["g++"] + flagsv + ["-shared", "-o", tmp] + srcs, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            os.replace(tmp, so)
    except (OSError, subprocess.CalledProcessError):
        return None
    _native = (so, "g++ " + " ".join(flagsv[:2]) + " on " + ("".join(model).split(":")[-1].strip() or "this host"))
    return _native[1]


def lib():
    global _lib
    if _lib is None:
        if _native is None:
            build()
        l = C.CDLL(_native[0] if _native else _LIB)
        l.orc_graph_new.restype = C.c_void_p
        l.orc_graph_new.argtypes = [C.c_char_p, C.c_int]
        l.orc_lnz_literal.restype = C.c_void_p
        l.orc_lnz_literal.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        l.orc_graph_free.argtypes = [C.c_void_p]
        l.orc_graph_error.restype = C.c_char_p
        l.orc_graph_error.argtypes = [C.c_void_p]
        l.orc_graph_dump.restype = C.c_longlong
        l.orc_graph_dump.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_longlong]
        l.orc_align.restype = C.c_longlong
        l.orc_align.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_longlong, C.POINTER(C.c_int),
                                C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_float, C.c_float, C.c_char_p,
                                C.c_longlong, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_ulonglong)]
        l.orc_align_amb.restype = C.c_longlong
        l.orc_align_amb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_longlong, C.POINTER(C.c_int),
                                    C.c_int, C.c_int, C.c_longlong, C.c_char_p, C.c_longlong, C.POINTER(C.c_int),
                                    C.POINTER(C.c_int)]
        l.orc_bench.restype = C.c_double
        l.orc_bench.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_longlong), C.c_longlong,
                                C.POINTER(C.c_int), C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float,
                                C.c_float, C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
        l.orc_bench_text.restype = C.c_double
        l.orc_bench_text.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_longlong), C.c_longlong,
                                     C.POINTER(C.c_int), C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float,
                                     C.c_float, C.c_int, C.c_char_p, C.c_longlong, C.c_longlong, C.c_char_p, C.c_longlong,
                                     C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_ulonglong)]
        l.orc_bench_faithful.restype = C.c_double
        l.orc_bench_faithful.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong), C.c_longlong, C.POINTER(C.c_int),
                                         C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_double),
                                         C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        l.orc_scores_match_mis.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        l.orc_scores_from_mtx.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        l.orc_missing_value.restype = C.c_int
        l.orc_f32_display.restype = C.c_int
        l.orc_f32_display.argtypes = [C.c_uint, C.c_char_p, C.c_int]
        l.orc_f32_cell_roundtrip_limit.restype = C.c_longlong
        l.orc_f32_cell_roundtrip_limit.argtypes = [C.c_longlong]
        _lib = l
    return _lib


ALPHABET = "ACGTN-"


def scores_match_mis(m=2, x=-4, f32_variant=False):
    out = (C.c_int * 36)()
    lib().orc_scores_match_mis(m, x, 1 if f32_variant else 0, out)
    return list(out)


def scores_from_mtx(text):
    out = (C.c_int * 36)()
    lib().orc_scores_from_mtx(text.encode(), out)
    return list(out)


def scores_from_dict(d):
    """{(a,b): v} -> 36 ints, absent keys = MISSING (the reference would panic on lookup)."""
    miss = lib().orc_missing_value()
    t = [miss] * 36
    for (a, b), v in d.items():
        t[ALPHABET.index(a) * 6 + ALPHABET.index(b)] = int(v)
    return t


class Graph:
    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_gfa_text(cls, text, want_path=True):
        h = lib().orc_graph_new(text.encode(), 1 if want_path else 0)
        err = lib().orc_graph_error(h).decode()
        if err:
            lib().orc_graph_free(h)
            raise ValueError(err)
        return cls(h)

    @classmethod
    def from_gfa(cls, path, want_path=True):
        with open(path) as f:
            return cls.from_gfa_text(f.read(), want_path)

    @classmethod
    def lnz_literal(cls, lnz, preds):
        """LnzGraph literal as in the reference's unit tests: preds = {row: [pred rows]}; nwp = rows with preds."""
        L = len(lnz)
        nwp = bytes(1 if i in preds else 0 for i in range(L))
        off = [0]
        rows = []
        for i in range(L):
            rows += preds.get(i, [])
            off.append(len(rows))
        offa = (C.c_longlong * len(off))(*off)
        rowa = (C.c_longlong * max(1, len(rows)))(*rows)
        return cls(lib().orc_lnz_literal(lnz.encode(), nwp, offa, rowa))

    def __del__(self):
        try:
            lib().orc_graph_free(self.h)
        except Exception:
            pass

    def dump(self, which):
        n = lib().orc_graph_dump(self.h, which, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib().orc_graph_dump(self.h, which, buf, n + 1)
        return buf.value.decode()

    def align(self, mode, read, name="read", idx=1, scores=None, o=-4, e=-2, bta=None, b=1.0, f=0.01, R=4, r=0.1,
              B=1.0):
        """Returns (stdout text, score, would_panic, cells)."""
        import numpy as np
        if scores is None:
            scores = scores_match_mis(2, -4)
        if bta is None:
            # main.rs:57: (b + f * seq.len() as f32) as usize, seq.len() = n + 1, all in f32
            bta = int(np.float32(b) + np.float32(f) * np.float32(len(read) + 1))
        sc = (C.c_int * 36)(*scores)
        score = C.c_int(0)
        flags = C.c_int(0)
        cells = C.c_ulonglong(0)
        cap = 1 << 16
        while True:
            buf = C.create_string_buffer(cap)
            n = lib().orc_align(self.h, mode, read.encode(), name.encode(), idx, sc, o, e, bta, R, r, B, buf, cap,
                                C.byref(score), C.byref(flags), C.byref(cells))
            if n + 1 <= cap:
                break
            cap = n + 16
        return buf.value.decode(), score.value, bool(flags.value), cells.value

    def align_amb(self, mode, amb, read, name="read", idx=1, scores=None, o=-4, e=-2, bta=0):
        """POA modes with the handle table / strand of `-s true` (amb bit 0: reversed handles, bit 1: strand '-')."""
        if scores is None:
            scores = scores_match_mis(2, -4)
        sc = (C.c_int * 36)(*scores)
        score = C.c_int(0)
        flags = C.c_int(0)
        cap = 1 << 16
        while True:
            buf = C.create_string_buffer(cap)
            n = lib().orc_align_amb(self.h, mode, amb, read.encode(), name.encode(), idx, sc, o, e, bta, buf, cap,
                                    C.byref(score), C.byref(flags))
            if n + 1 <= cap:
                break
            cap = n + 16
        return buf.value.decode(), score.value, bool(flags.value)

    def main_rs_amb_strand(self, m, read, name, idx, scores=None, o=-4, e=-2, b=1.0, f=0.01, avx2=True):
        """What main.rs prints for one read under `-s true`, modes 0-3 (main.rs:47-253).  Returns stdout text."""
        import numpy as np
        bta = int(np.float32(b) + np.float32(f) * np.float32(len(read) + 1))
        comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
        rc = "".join(comp[c] for c in reversed(read.upper().replace("-", "N")))   # sequences.rs:65-82
        def pick(fwd, rev, take_rev):
            # warnings are printed inside exec (both runs, in order); write_gaf then prints the chosen record
            fl, rl = fwd[0].splitlines(True), rev[0].splitlines(True)
            return "".join(fl[:-1]) + "".join(rl[:-1]) + (rl[-1] if take_rev else fl[-1])

        if m in (0, 2):
            fmode = (M0_SIMD if avx2 else M0_SCALAR) if m == 0 else M2
            rmode = M0_SCALAR if m == 0 else M2          # main.rs:88: the retry always uses the scalar exec
            fwd = self.align_amb(fmode, 0, read, name, idx, scores, o, e, bta)
            if fwd[1] < 0:
                rev = self.align_amb(rmode, 3, rc, name, idx, scores, o, e, bta)
                assert not rev[2]
                return pick(fwd, rev, rev[1] > fwd[1])
            return fwd[0]
        if m == 1:
            mode = M1_SIMD if avx2 else M1_SCALAR
            fwd = self.align_amb(mode, 0, read, name, idx, scores, o, e, bta)
            rev = self.align_amb(mode, 3, rc, name, idx, scores, o, e, bta)
            return pick(fwd, rev, not (fwd[1] < rev[1]))    # main.rs:160-164 (sic)
        if m == 3:
            fwd = self.align_amb(M3, 0, read, name, idx, scores, o, e, bta)
            rev = self.align_amb(M3, 1, rc, name, idx, scores, o, e, bta)   # amb_mode = false, reversed handles (:240)
            return pick(fwd, rev, rev[1] > fwd[1])
        raise ValueError(m)

    def bench(self, mode, reads, scores=None, o=-4, e=-2, b=1.0, f=0.01, R=4, r=0.1, B=1.0, nthreads=1):
        """Time `reads` (list of str) on `nthreads` host threads.  Returns (seconds, cells, checksum)."""
        if scores is None:
            scores = scores_match_mis(2, -4)
        blob = "".join(reads).encode()
        offs = [0]
        for rd in reads:
            offs.append(offs[-1] + len(rd))
        offa = (C.c_longlong * len(offs))(*offs)
        sc = (C.c_int * 36)(*scores)
        cells = C.c_ulonglong(0)
        chk = C.c_ulonglong(0)
        secs = lib().orc_bench(self.h, mode, blob, offa, len(reads), sc, o, e, b, f, R, r, B, nthreads,
                               C.byref(cells), C.byref(chk))
        return secs, cells.value, chk.value

    def bench_text(self, mode, reads, scores=None, o=-4, e=-2, b=1.0, f=0.01, R=4, r=0.1, B=1.0, nthreads=1,
                   name_prefix="read", idx_base=1, name_base=0):
        """Like ``bench`` but returns (seconds, cells, [stdout text per read]); read i is named
        ``name_prefix + str(name_base + i)`` with seq index ``idx_base + i`` (what ``rg_batch_format_all`` uses by default;
        a stream names its reads by their position in the whole stream)."""
        if scores is None:
            scores = scores_match_mis(2, -4)
        blob = "".join(reads).encode()
        offs = [0]
        for rd in reads:
            offs.append(offs[-1] + len(rd))
        offa = (C.c_longlong * len(offs))(*offs)
        sc = (C.c_int * 36)(*scores)
        cells = C.c_ulonglong(0)
        need = C.c_longlong(0)
        cap = 4096 * len(reads) + 65536
        buf = C.create_string_buffer(cap)
        toff = (C.c_longlong * (len(reads) + 1))()
        secs = lib().orc_bench_text(self.h, mode, blob, offa, len(reads), sc, o, e, b, f, R, r, B, nthreads,
                                    name_prefix.encode(), name_base, idx_base, buf, cap, toff, C.byref(need), C.byref(cells))
        if need.value > cap:
            raise RuntimeError("orc_bench_text: output larger than the buffer (%d > %d)" % (need.value, cap))
        raw = buf.raw
        return secs, cells.value, [raw[toff[i]:toff[i + 1]] for i in range(len(reads))]

    def bench_faithful(self, reads, scores=None, R=4, r=0.1, B=1.0, nthreads=1, col_stride=1):
        """FAITHFUL -m 8 (literal DP + unpruned O(L^2 n) scan) with the scan sampled on every ``col_stride``-th
        column.  Returns dict(wall, dp_secs, scan_secs, cols_visited, cols_total); sums over the reads."""
        if scores is None:
            scores = scores_match_mis(2, -4)
        blob = "".join(reads).encode()
        offs = [0]
        for rd in reads:
            offs.append(offs[-1] + len(rd))
        offa = (C.c_longlong * len(offs))(*offs)
        sc = (C.c_int * 36)(*scores)
        dp, scan = C.c_double(0), C.c_double(0)
        cv, ct = C.c_longlong(0), C.c_longlong(0)
        wall = lib().orc_bench_faithful(self.h, blob, offa, len(reads), sc, R, r, B, nthreads, col_stride,
                                        C.byref(dp), C.byref(scan), C.byref(cv), C.byref(ct))
        return dict(wall=wall, dp_secs=dp.value, scan_secs=scan.value, cols_visited=cv.value, cols_total=ct.value)
