// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.hpp header).
// extern "C" surface used by tests/ (through oracle/oracle.py) and by bench.py's cpu_baseline leg.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>

#include <malloc.h>

#include "orc_common.hpp"

using namespace orc;

struct OrcGraph {
    Gfa gfa;
    bool has_lnz = false, has_path = false;
    LnzGraph lnz;
    std::vector<size_t> r_values;
    PathGraph pg, rpg;
    std::vector<int64_t> dfs, dfe;
    std::string err;
};

static Scores scores_from(const int* t36) {
    Scores s;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) s.t[i][j] = t36[i * 6 + j];
    return s;
}

// The timed runners keep large blocks inside the malloc arenas: with glibc's defaults every matrix / row vector above
// 128 KB is an mmap + munmap per read, and on a many-core host the threads then serialise on the address-space lock
// (the reference itself links jemalloc on linux for the same reason, main.rs:21-23).
static void tune_malloc_once() {
    static const bool done = [] {
        mallopt(M_MMAP_THRESHOLD, 32 << 20);   // (the largest value glibc accepts)
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
        mallopt(M_TOP_PAD, 64 << 20);
        return true;
    }();
    (void)done;
}

extern "C" {

// mode ids
enum { ORC_M0_SIMD = 0, ORC_M0_SCALAR = 10, ORC_M2 = 2, ORC_M4 = 4, ORC_M4_ABS = 14, ORC_M8 = 8, ORC_M8_PRUNED = 18, ORC_M8_ABS = 28,
       ORC_M5 = 5, ORC_M5_ABS = 15, ORC_M9 = 9, ORC_M9_PRUNED = 19, ORC_M9_ABS = 29, ORC_M1_SIMD = 1, ORC_M1_SCALAR = 11,
       ORC_M3 = 3 };

void orc_scores_match_mis(int m, int x, int f32_variant, int* out36) {
    Scores s = f32_variant ? make_scores_match_mis_f32(m, x) : make_scores_match_mis(m, x);
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) out36[i * 6 + j] = s.t[i][j];
}
void orc_scores_from_mtx(const char* text, int* out36) {
    Scores s = make_scores_from_mtx(text);
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) out36[i * 6 + j] = s.t[i][j];
}
int orc_missing_value() { return MISSING; }

// build every flattened form from GFA text; want_path=0 skips the PathGraph (graphs without P lines)
void* orc_graph_new(const char* gfa_text, int want_path) {
    auto* g = new OrcGraph();
    if (!parse_gfa_text(gfa_text, g->gfa, g->err)) return g;
    g->lnz = create_graph_struct(g->gfa);
    g->r_values = set_r_values(g->lnz.nwp, g->lnz.pred_hash, g->lnz.lnz.size());
    g->has_lnz = true;
    if (want_path && !g->gfa.paths.empty()) {
        g->pg = create_path_graph(g->gfa);
        g->rpg = create_reverse_path_graph(g->pg);
        g->dfe = get_distance_from_end(g->pg);
        g->dfs = get_distance_from_start(g->rpg);
        g->has_path = true;
    }
    return g;
}

// hand-written LnzGraph literal, as the reference's unit tests build them
// (global_abpoa.rs:576-755, gap_global_abpoa.rs:464-757): preds in CSR form over rows with nwp set
void* orc_lnz_literal(const char* lnz, const unsigned char* nwp, const long long* pred_off,
                      const long long* pred_rows) {
    auto* g = new OrcGraph();
    g->lnz.lnz = lnz;
    size_t L = g->lnz.lnz.size();
    g->lnz.nwp.assign(nwp, nwp + L);
    for (size_t i = 0; i < L; ++i)
        for (long long k = pred_off[i]; k < pred_off[i + 1]; ++k) g->lnz.pred_hash[i].push_back((size_t)pred_rows[k]);
    g->lnz.hofp.assign(L - 1, "0");
    g->r_values = set_r_values(g->lnz.nwp, g->lnz.pred_hash, L);
    g->has_lnz = true;
    return g;
}

void orc_graph_free(void* h) { delete (OrcGraph*)h; }
const char* orc_graph_error(void* h) { return ((OrcGraph*)h)->err.c_str(); }

static long long put(const std::string& s, char* buf, long long cap) {
    if ((long long)s.size() + 1 <= cap) { memcpy(buf, s.c_str(), s.size() + 1); }
    return (long long)s.size();
}

// text dumps of the flattened graphs for the construction tests
// which: 0 lnz, 1 nwp, 2 preds "row:p,p;...", 3 hofp "id,id,...", 4 r_values
//        10 path lnz, 11 path nwp, 12 path preds "row:pred=bits,pred=bits;", 13 paths_nodes rows "bits;bits", 14 alphas,
//        15 nodes_id_pos, 16 rev nwp, 17 rev preds, 18 dfs, 19 dfe
long long orc_graph_dump(void* h, int which, char* buf, long long cap) {
    auto* g = (OrcGraph*)h;
    std::string s;
    auto bits = [](const std::vector<uint8_t>& v) { std::string o; for (auto b : v) o += b ? '1' : '0'; return o; };
    auto ph = [&](const PathGraph& pg) {
        std::string o;
        for (auto& nk : pg.pred_hash) {
            o += std::to_string(nk.first) + ":";
            bool first = true;
            for (auto& pk : nk.second) { if (!first) o += ","; first = false; o += std::to_string(pk.first) + "=" + bits(pk.second); }
            o += ";";
        }
        return o;
    };
    switch (which) {
        case 0: s = g->lnz.lnz; break;
        case 1: s = bits(g->lnz.nwp); break;
        case 2:
            for (auto& kv : g->lnz.pred_hash) {
                s += std::to_string(kv.first) + ":";
                for (size_t k = 0; k < kv.second.size(); ++k) { if (k) s += ","; s += std::to_string(kv.second[k]); }
                s += ";";
            }
            break;
        case 3: for (size_t i = 0; i < g->lnz.hofp.size(); ++i) { if (i) s += ","; s += g->lnz.hofp[i]; } break;
        case 4: for (size_t i = 0; i < g->r_values.size(); ++i) { if (i) s += ","; s += std::to_string((long long)g->r_values[i]); } break;
        case 10: s = g->pg.lnz; break;
        case 11: s = bits(g->pg.nwp); break;
        case 12: s = ph(g->pg); break;
        case 13: for (auto& r : g->pg.paths_nodes) { s += bits(r); s += ";"; } break;
        case 14: for (size_t i = 0; i < g->pg.alphas.size(); ++i) { if (i) s += ","; s += std::to_string(g->pg.alphas[i]); } break;
        case 15: for (size_t i = 0; i < g->pg.nodes_id_pos.size(); ++i) { if (i) s += ","; s += std::to_string(g->pg.nodes_id_pos[i]); } break;
        case 16: s = bits(g->rpg.nwp); break;
        case 17: s = ph(g->rpg); break;
        case 18: for (size_t i = 0; i < g->dfs.size(); ++i) { if (i) s += ","; s += std::to_string(g->dfs[i]); } break;
        case 19: for (size_t i = 0; i < g->dfe.size(); ++i) { if (i) s += ","; s += std::to_string(g->dfe[i]); } break;
        default: break;
    }
    return put(s, buf, cap);
}

static Result run_one(OrcGraph* g, int mode, const std::string& read_dollar, const std::string& name, size_t idx,
                      const Scores& sc, int o, int e, size_t bta, int brc, float mrc, float rbw, uint64_t* cells) {
    switch (mode) {
        case ORC_M0_SIMD: return m0_simd(read_dollar, name, idx, g->lnz, sc, bta, g->r_values, cells);
        case ORC_M0_SCALAR: return m0_scalar(read_dollar, name, idx, g->lnz, sc, bta, cells);
        case ORC_M2: return m2_gap(read_dollar, name, idx, g->lnz, sc, o, e, bta, cells);
        case ORC_M1_SIMD: return m1_simd(read_dollar, name, idx, g->lnz, sc, cells);
        case ORC_M1_SCALAR: return m1_scalar(read_dollar, name, idx, g->lnz, sc, cells);
        case ORC_M3: return m3_gap_local(read_dollar, name, idx, g->lnz, sc, o, e, cells);
        case ORC_M4: return m4_literal(read_dollar, name, g->pg, sc);
        case ORC_M4_ABS: return m4_abs(read_dollar, name, g->pg, sc);
        case ORC_M8: return m8_literal(read_dollar, name, g->pg, g->rpg, g->dfs, g->dfe, sc, brc, mrc, rbw, false);
        case ORC_M8_PRUNED: return m8_literal(read_dollar, name, g->pg, g->rpg, g->dfs, g->dfe, sc, brc, mrc, rbw, true);
        case ORC_M8_ABS: return m8_abs(read_dollar, name, g->pg, g->rpg, g->dfs, g->dfe, sc, brc, mrc, rbw);
        case ORC_M5: return m5_literal(read_dollar, name, g->pg, sc);
        case ORC_M5_ABS: return m5_abs(read_dollar, name, g->pg, sc);
        case ORC_M9: return m9_literal(read_dollar, name, g->pg, g->rpg, g->dfs, g->dfe, sc, brc, mrc, rbw, false);
        case ORC_M9_PRUNED: return m9_literal(read_dollar, name, g->pg, g->rpg, g->dfs, g->dfe, sc, brc, mrc, rbw, true);
        case ORC_M9_ABS: return m9_abs(read_dollar, name, g->pg, g->rpg, g->dfs, g->dfe, sc, brc, mrc, rbw);
        default: { Result r; r.would_panic = true; return r; }
    }
}

// `-s true` support (main.rs:82-106, 132-165, 188-212, 229-253): the reference aligns the reverse complement of the
// read against the SAME LnzGraph, with the handle table of the reversed handle order and (modes 0-2) strand '-'.
// amb bit 0: use hofp_rev; bit 1: strand '-'.  sequences.rs:65-82 rev_and_compl is applied by the caller.
long long orc_align_amb(void* h, int mode, int amb, const char* read, const char* name, long long idx, const int* scores36,
                        int o, int e, long long bta, char* out, long long cap, int* score, int* flags) {
    auto* g = (OrcGraph*)h;
    std::string rd = "$";
    for (const char* p = read; *p; ++p) rd += (*p == '-') ? 'N' : (char)std::toupper(*p);
    Scores sc = scores_from(scores36);
    OrcGraph view;
    view.lnz = g->lnz;
    view.r_values = g->r_values;
    if ((amb & 1) && !g->lnz.hofp_rev.empty()) view.lnz.hofp = g->lnz.hofp_rev;
    view.lnz.strand = (amb & 2) ? '-' : '+';
    uint64_t c = 0;
    Result r = run_one(&view, mode, rd, name, (size_t)idx, sc, o, e, (size_t)bta, 0, 0.f, 0.f, &c);
    if (score) *score = r.score;
    if (flags) *flags = r.would_panic ? 1 : 0;
    return put(r.out, out, cap);
}

// One read (bases without the '$'; sequences.rs:48-61 build_align_string is applied here).
// idx is seq_name.1 of the reference (0 = score only).  Returns the length of the stdout text.
long long orc_align(void* h, int mode, const char* read, const char* name, long long idx, const int* scores36,
                    int o, int e, long long bta, int brc, float mrc, float rbw, char* out, long long cap,
                    int* score, int* flags, unsigned long long* cells) {
    auto* g = (OrcGraph*)h;
    std::string rd = "$";
    for (const char* p = read; *p; ++p) rd += (*p == '-') ? 'N' : (char)std::toupper(*p);
    Scores sc = scores_from(scores36);
    uint64_t c = 0;
    Result r = run_one(g, mode, rd, name, (size_t)idx, sc, o, e, (size_t)bta, brc, mrc, rbw, &c);
    if (score) *score = r.score;
    if (flags) *flags = r.would_panic ? 1 : 0;
    if (cells) *cells = c;
    return put(r.out, out, cap);
}

// cpu_baseline leg of bench.py: run `nreads` reads sharded over `nthreads` host threads, return
// wall seconds (reads are independent; the reference itself is single-threaded, main.rs:56).
double orc_bench(void* h, int mode, const char* reads_concat, const long long* offsets, long long nreads,
                 const int* scores36, int o, int e, float b, float f, int brc, float mrc, float rbw, int nthreads,
                 unsigned long long* cells_out, unsigned long long* checksum_out) {
    tune_malloc_once();
    auto* g = (OrcGraph*)h;
    Scores sc0 = scores_from(scores36);
    std::atomic<long long> next{0};
    std::atomic<unsigned long long> cells{0}, checksum{0};
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&]() {
        Scores sc = sc0;
        while (true) {
            long long r = next.fetch_add(1);
            if (r >= nreads) break;
            std::string rd = "$" + std::string(reads_concat + offsets[r], reads_concat + offsets[r + 1]);
            size_t bta = (size_t)(b + f * (float)rd.size());  // main.rs:57
            uint64_t c = 0;
            Result res = run_one(g, mode, rd, "r", (size_t)r + 1, sc, o, e, bta, brc, mrc, rbw, &c);
            unsigned long long hsum = 1469598103934665603ull;
            for (unsigned char ch : res.out) { hsum ^= ch; hsum *= 1099511628211ull; }
            cells += c;
            checksum += hsum;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work);
    for (auto& t : th) t.join();
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (cells_out) *cells_out = cells.load();
    if (checksum_out) *checksum_out = checksum.load();
    return secs;
}

// Same sharded run, returning every read's stdout text (read r is named "<name_prefix><r>" with seq index
// idx_base + r, like rg_batch_format_all with names == NULL): text_out holds the texts back to back in read order,
// text_off[r] .. text_off[r + 1] is read r.  Used by bench.py's in-run parity gate and by the full-size GPU tests.
// Returns wall seconds; *need = bytes required (nothing is written past `cap`).
double orc_bench_text(void* h, int mode, const char* reads_concat, const long long* offsets, long long nreads,
                      const int* scores36, int o, int e, float b, float f, int brc, float mrc, float rbw, int nthreads,
                      const char* name_prefix, long long name_base, long long idx_base, char* text_out, long long cap,
                      long long* text_off, long long* need, unsigned long long* cells_out) {
    tune_malloc_once();
    auto* g = (OrcGraph*)h;
    Scores sc0 = scores_from(scores36);
    std::atomic<long long> next{0};
    std::atomic<unsigned long long> cells{0};
    std::vector<std::string> outs((size_t)nreads);
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&]() {
        Scores sc = sc0;
        while (true) {
            long long r = next.fetch_add(1);
            if (r >= nreads) break;
            std::string rd = "$";
            for (const char* p = reads_concat + offsets[r]; p < reads_concat + offsets[r + 1]; ++p)
                rd += (*p == '-') ? 'N' : (char)std::toupper(*p);
            size_t bta = (size_t)(b + f * (float)rd.size());  // main.rs:57
            uint64_t c = 0;
            Result res = run_one(g, mode, rd, std::string(name_prefix) + std::to_string(name_base + r), (size_t)(idx_base + r), sc, o, e, bta,
                                 brc, mrc, rbw, &c);
            cells += c;
            outs[(size_t)r] = res.would_panic ? std::string("<would panic>\n") : res.out;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work);
    for (auto& t : th) t.join();
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long long total = 0;
    for (long long r = 0; r < nreads; ++r) {
        if (text_off) text_off[r] = total;
        if (text_out && total + (long long)outs[(size_t)r].size() <= cap) memcpy(text_out + total, outs[(size_t)r].data(), outs[(size_t)r].size());
        total += (long long)outs[(size_t)r].size();
    }
    if (text_off) text_off[nreads] = total;
    if (need) *need = total;
    if (cells_out) *cells_out = cells.load();
    return secs;
}

// FAITHFUL -m 8 timing (the literal transliteration with the UNPRUNED best_alignment scan,
// pathwise_alignment_recombination.rs:808-864), one read per worker at a time, the scan sampled on every
// `col_stride`-th column (FaithfulProbe).  Outputs are sums over the reads; the caller extrapolates
// scan_secs * cols_total / cols_visited.  Returns wall seconds.
double orc_bench_faithful(void* h, const char* reads_concat, const long long* offsets, long long nreads, const int* scores36,
                          int brc, float mrc, float rbw, int nthreads, int col_stride, double* dp_secs, double* scan_secs,
                          long long* cols_visited, long long* cols_total) {
    tune_malloc_once();
    auto* g = (OrcGraph*)h;
    Scores sc0 = scores_from(scores36);
    std::atomic<long long> next{0};
    std::vector<FaithfulProbe> probes((size_t)std::max(1, nthreads));
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&](int t) {
        Scores sc = sc0;
        probes[(size_t)t].col_stride = std::max(1, col_stride);
        g_faithful_probe = &probes[(size_t)t];
        while (true) {
            long long r = next.fetch_add(1);
            if (r >= nreads) break;
            std::string rd = "$" + std::string(reads_concat + offsets[r], reads_concat + offsets[r + 1]);
            (void)m8_literal(rd, "r", g->pg, g->rpg, g->dfs, g->dfe, sc, brc, mrc, rbw, false);
        }
        g_faithful_probe = nullptr;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < std::max(1, nthreads); ++t) th.emplace_back(work, t);
    for (auto& t : th) t.join();
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double dp = 0, sc = 0; long long cv = 0, ct = 0;
    for (auto& p : probes) { dp += p.dp_secs; sc += p.scan_secs; cv += p.cols_visited; ct += p.cols_total; }
    if (dp_secs) *dp_secs = dp;
    if (scan_secs) *scan_secs = sc;
    if (cols_visited) *cols_visited = cv;
    if (cols_total) *cols_total = ct;
    return secs;
}

// the oracle's Rust-`{}` formatter on one f32 given by its bit pattern (tests pin it against numpy's shortest-unique form)
int orc_f32_display(unsigned bits, char* out, int cap) {
    float v;
    memcpy(&v, &bits, 4);
    const std::string s = f32_display(v);
    if ((int)s.size() + 1 > cap) return -1;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}

// decode check used by tests: the f32 path-cell encoding "pred + 0.1/0.2/0.3" parsed back through
// Display + split('.') (gaf_output.rs:783-786); returns first pred for which it fails, or -1
long long orc_f32_cell_roundtrip_limit(long long upto) {
    const float mv[3] = {0.1f, 0.2f, 0.3f};
    for (long long p = 0; p < upto; ++p)
        for (int d = 0; d < 3; ++d) {
            float v = (float)p + mv[d];
            std::string s = f32_display(v);
            size_t dot = s.find('.');
            if (dot == std::string::npos) return p;
            if (std::stoll(s.substr(0, dot)) != p) return p;
            std::string fr = s.substr(dot + 1);
            if (fr.size() > 9 || std::stol(fr) != d + 1) return p;
        }
    return -1;
}

}  // extern "C"
