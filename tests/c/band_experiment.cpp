// TEST / ANALYSIS TOOL (uses the oracle's graph code: test infrastructure, never linked into the product).
//
// VERDICT r3 asked for "exact banding" of the pathwise sweeps: drop the DP cells whose value plus the best possible rest
// cannot reach the verified lower bound of the optimum.  This program measures what that would do to the cells that are
// KEPT.  The pathwise DP is not a max-plus recurrence per path: per (row, edge group) the group's alpha path takes
// max(d, u, l) and every other member FOLLOWS the alpha's direction with its own values (pathwise_alignment.rs:185-299,
// SURVEY A.4).  A member's value at a cell is therefore the score of the one alignment the alphas chose along its chain,
// and it depends on every alpha decision along that chain — each of which depends on three neighbour values of the
// alpha, and so on: the dependency set of a near-diagonal cell is its whole lower-left quadrant, whatever the scores.
//
// The experiment: forward sweep of `-m 4 / -m 8` (absolute form) in full, then again with every path's rolling row cut to
// the window |column - (rows of the path so far)| <= w (cells outside read as minus infinity, as a banded kernel would
// have it), and per read: how many kept cells changed value, how deep inside the band (distance from the band edge) the
// deepest change lies, and whether a cell that matters downstream changed (a sink value, or a cell within `core` columns of
// its path's diagonal).
//
//   band_experiment graph.gfa reads.txt core w1 [w2 ...]         (reads: one per line, without '$')
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../oracle/orc_common.hpp"

using namespace orc;

namespace {

constexpr int NEGV = -100000000;

struct Group { size_t pred; std::vector<size_t> members; size_t ga; };

std::vector<std::vector<Group>> program(const PathGraph& g) {
    const size_t L = g.lnz.size(), P = g.paths_number;
    std::vector<std::vector<Group>> rows(L);
    auto mk = [&](size_t i, size_t p, const std::vector<uint8_t>& paths) {
        Group gr;
        gr.pred = p;
        for (size_t k = 0; k < P; ++k) if (paths[k] && g.paths_nodes[i][k]) gr.members.push_back(k);
        if (gr.members.empty()) return;
        auto has = [&](size_t k) { return std::find(gr.members.begin(), gr.members.end(), k) != gr.members.end(); };
        gr.ga = has(g.alphas[p]) ? g.alphas[p] : has(g.alphas[i]) ? g.alphas[i] : gr.members[0];   // pathwise_alignment.rs:190,235-239
        rows[i].push_back(gr);
    };
    for (size_t i = 1; i + 1 < L; ++i) {
        if (g.nwp[i]) {
            auto it = g.pred_hash.find(i);
            if (it != g.pred_hash.end()) for (auto& pk : it->second) mk(i, pk.first, pk.second);
        } else mk(i, i - 1, g.paths_nodes[i - 1]);
    }
    return rows;
}

// forward sweep; w < 0: full width.  layer[i][k]: the row of path k after row i (members only)
void sweep(const PathGraph& g, const std::vector<std::vector<Group>>& pr, const std::vector<std::vector<int>>& tk, const std::string& seq,
           const Scores& sc, int w, std::vector<std::vector<std::vector<int>>>& layer) {
    const size_t L = g.lnz.size(), W = seq.size(), P = g.paths_number;
    layer.assign(L, std::vector<std::vector<int>>(P));
    std::vector<std::vector<int>> roll(P, std::vector<int>(W, 0));
    auto inwin = [&](size_t i, size_t k, size_t j) { return w < 0 || std::abs((long)j - (long)tk[i][k]) <= w; };
    {
        std::vector<int> r0(W, 0);
        for (size_t j = 1; j < W; ++j) r0[j] = r0[j - 1] + sc.get(seq[j], '-');
        for (size_t k = 0; k < P; ++k) for (size_t j = 0; j < W; ++j) roll[k][j] = inwin(0, k, j) ? r0[j] : NEGV;
    }
    std::vector<int> na(W), tmp(W);
    std::vector<uint8_t> dir(W);
    for (size_t i = 1; i + 1 < L; ++i) {
        const int g_i = sc.get(g.lnz[i], '-');
        for (const Group& gr : pr[i]) {
            std::vector<int>& ra = roll[gr.ga];
            const size_t a = gr.ga;
            for (size_t j = 0; j < W; ++j) {
                if (!inwin(i, a, j)) { na[j] = NEGV; dir[j] = 0; continue; }
                if (j == 0) { na[0] = ra[0] + g_i; dir[0] = 2; continue; }
                const int d = ra[j - 1] <= NEGV / 2 ? NEGV : ra[j - 1] + sc.get(g.lnz[i], seq[j]);
                const int u = ra[j] <= NEGV / 2 ? NEGV : ra[j] + g_i;
                const int l = na[j - 1] <= NEGV / 2 ? NEGV : na[j - 1] + sc.get(seq[j], '-');
                const int b = std::max(std::max(d, u), l);
                na[j] = b;
                dir[j] = b == d ? 1 : b == u ? 2 : 3;
            }
            for (size_t k : gr.members) {
                if (k == a) continue;
                std::vector<int>& rk = roll[k];
                for (size_t j = 0; j < W; ++j) {
                    if (!inwin(i, k, j) || dir[j] == 0) { tmp[j] = NEGV; continue; }    // no alpha decision there: not computable
                    if (j == 0) { tmp[0] = rk[0] <= NEGV / 2 ? NEGV : rk[0] + g_i; continue; }
                    const int src = dir[j] == 1 ? rk[j - 1] : dir[j] == 2 ? rk[j] : tmp[j - 1];
                    const int add = dir[j] == 1 ? sc.get(g.lnz[i], seq[j]) : dir[j] == 2 ? g_i : sc.get(seq[j], '-');
                    tmp[j] = src <= NEGV / 2 ? NEGV : src + add;
                }
                rk.swap(tmp);
            }
            ra.swap(na);
        }
        for (size_t k = 0; k < P; ++k) if (g.paths_nodes[i][k]) layer[i][k] = roll[k];
    }
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s graph.gfa reads.txt core w1 [w2 ...]\n", argv[0]); return 2; }
    std::ifstream gf(argv[1]);
    std::stringstream ss;
    ss << gf.rdbuf();
    Gfa gfa;
    std::string err;
    if (!parse_gfa_text(ss.str(), gfa, err)) { fprintf(stderr, "gfa: %s\n", err.c_str()); return 1; }
    const PathGraph g = create_path_graph(gfa);
    const Scores sc = make_scores_match_mis(2, -4);
    const auto pr = program(g);
    const size_t L = g.lnz.size(), P = g.paths_number;
    std::vector<std::vector<int>> tk(L, std::vector<int>(P, 0));       // rows of path k up to and including row i
    for (size_t k = 0; k < P; ++k) { int c = 0; for (size_t i = 1; i + 1 < L; ++i) { if (g.paths_nodes[i][k]) ++c; tk[i][k] = c; } }
    std::vector<std::string> reads;
    { std::ifstream rf(argv[2]); std::string ln; while (std::getline(rf, ln)) if (!ln.empty()) reads.push_back("$" + ln); }
    const int core = atoi(argv[3]);
    printf("{\"rows\": %zu, \"paths\": %zu, \"reads\": %zu, \"core\": %d, \"bands\": [", L, P, reads.size(), core);
    for (int a = 4; a < argc; ++a) {
        const int w = atoi(argv[a]);
        long reads_changed = 0, reads_core_changed = 0, reads_sink_changed = 0, cells_kept = 0, cells_changed = 0, deepest = 0;
        std::vector<long> depth_hist(8, 0);     // deepest change per read: edge .. diagonal in eighths of the band
        for (const std::string& seq : reads) {
            std::vector<std::vector<std::vector<int>>> full, band;
            sweep(g, pr, tk, seq, sc, -1, full);
            sweep(g, pr, tk, seq, sc, w, band);
            const size_t W = seq.size();
            long ch = 0, dp = -1;
            bool corech = false, sinkch = false;
            for (size_t i = 1; i + 1 < L; ++i)
                for (size_t k = 0; k < P; ++k) {
                    if (band[i][k].empty()) continue;
                    for (size_t j = 0; j < W; ++j) {
                        const long off = std::abs((long)j - (long)tk[i][k]);
                        if (off > w) continue;
                        ++cells_kept;
                        if (band[i][k][j] != full[i][k][j]) {
                            ++ch;
                            dp = std::max(dp, (long)w - off);
                            if (off <= core) corech = true;
                        }
                    }
                }
            // sink values: every path's last row, column n
            for (size_t k = 0; k < P; ++k) {
                size_t last = 0;
                for (size_t i = 1; i + 1 < L; ++i) if (g.paths_nodes[i][k]) last = i;
                if (last && std::abs((long)(W - 1) - (long)tk[last][k]) <= w && band[last][k][W - 1] != full[last][k][W - 1]) sinkch = true;
            }
            cells_changed += ch;
            if (ch) { ++reads_changed; deepest = std::max(deepest, dp); ++depth_hist[(size_t)std::min<long>(7, dp * 8 / std::max(1, w + 1))]; }
            if (corech) ++reads_core_changed;
            if (sinkch) ++reads_sink_changed;
        }
        printf("%s{\"w\": %d, \"reads_with_changed_kept_cells\": %ld, \"reads_with_changed_core_cells\": %ld, \"reads_with_changed_sink_value\": %ld, "
               "\"kept_cells\": %ld, \"changed_cells\": %ld, \"deepest_change_columns_from_edge\": %ld, \"deepest_change_histogram_eighths\": [",
               a > 4 ? ", " : "", w, reads_changed, reads_core_changed, reads_sink_changed, cells_kept, cells_changed, deepest);
        for (size_t b = 0; b < 8; ++b) printf("%s%ld", b ? ", " : "", depth_hist[b]);
        printf("]}");
        fflush(stdout);
    }
    printf("]}\n");
    return 0;
}
