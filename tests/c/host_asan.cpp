// Sanitizer harness for the host-only, text-eating code of librecgraph_hip (`make -C recgraph_amd/csrc asan`: g++
// -fsanitize=address,undefined over rg_graph.cpp, rg_gaf.cpp, rg_reads.cpp — no HIP, no GPU).  Feeds it hand-made
// malformed inputs and seeded byte / line mutations of valid GFA and FASTA text: every call must come back with a
// status code (never crash, never trip ASan / UBSan).  Usage: host_asan graph.gfa reads.fa [iterations] [seed]
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "rg_host.hpp"

using namespace rg;

static std::string slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

static long ok_gfa = 0, bad_gfa = 0, ok_fa = 0, bad_fa = 0;

static void try_gfa(const std::string& t) {
    HostGraph g;
    const int rc = build_from_gfa(t.data(), (int64_t)t.size(), g);
    if (rc == RG_OK) {
        ++ok_gfa;
        for (int which = 0; which < 8; ++which) (void)dump_graph(g, which);     // walks every flattened array
        // the step tables of the pathwise sweeps, their split form and the path-retirement tables (rg_steps.cpp)
        for (int which = 30; which <= 34; ++which) (void)dump_graph(g, which);
    } else {
        ++bad_gfa;
    }
}

// the same text in random pieces through FastaFeeder (what rg_stream_feed_fasta runs) and through fasta_count (what
// rg_fasta_check runs): same reads, same verdict as the one-piece parser
static void try_feeder(const std::string& t, bool whole_ok, const FastaReads& whole, uint64_t seed) {
    std::mt19937_64 rng(seed);
    FastaFeeder fd;
    FastaReads out;
    int64_t st[4] = {0, 0, 0, 0};
    size_t pos = 0;
    while (pos < t.size()) {
        const size_t cnt = std::min<size_t>(t.size() - pos, (size_t)(rng() % 7 == 0 ? 0 : 1 + rng() % 40));
        fd.feed(t.data() + pos, (int64_t)cnt, false, out);
        fasta_count(t.data() + pos, (int64_t)cnt, false, st);
        pos += cnt;
    }
    fd.feed(nullptr, 0, true, out);
    fasta_count(nullptr, 0, true, st);
    if (fd.balanced() != whole_ok || (st[0] == st[1]) != whole_ok) { fprintf(stderr, "feeder verdict differs\n"); abort(); }
    if (st[0] != fd.names_total || st[1] != fd.seqs_total) { fprintf(stderr, "fasta_count differs from the feeder\n"); abort(); }
    if (whole_ok && (out.names != whole.names || out.bases != whole.bases || out.off != whole.off)) { fprintf(stderr, "feeder reads differ\n"); abort(); }
}

static void try_fasta(const std::string& t, int64_t batch) {
    FastaReads r;
    int64_t emitted = 0;
    const bool ok = parse_fasta(t.data(), (int64_t)t.size(), r, batch, [&](int64_t first, int64_t count) {
        if (first != emitted || count < 1) { fprintf(stderr, "emit order broken\n"); abort(); }
        // what rg_stream_push_fasta reads: names [first, first + count), offsets [first, first + count]
        for (int64_t i = first; i < first + count; ++i) {
            if ((size_t)i >= r.names.size() || (size_t)(i + 1) >= r.off.size() || r.off[(size_t)i + 1] <= r.off[(size_t)i]) { fprintf(stderr, "incomplete read emitted\n"); abort(); }
        }
        emitted += count;
    });
    try_feeder(t, ok, r, (uint64_t)t.size() * 31 + (uint64_t)batch);
    if (ok) {
        ++ok_fa;
        if (emitted != (int64_t)r.names.size()) { fprintf(stderr, "reads lost\n"); abort(); }
        if (!r.names.empty()) {
            std::vector<uint8_t> codes(r.bases.size() + 1), bad(r.names.size());
            (void)canonicalise_reads(r.bases.data(), r.off.data(), (int64_t)r.names.size(), codes.data(), bad.data());
        }
    } else {
        ++bad_fa;
    }
}

static std::string mutate(const std::string& base, std::mt19937_64& rng) {
    std::string t = base;
    const int nmut = 1 + (int)(rng() % 4);
    for (int m = 0; m < nmut && !t.empty(); ++m) {
        const size_t pos = rng() % t.size();
        switch (rng() % 7) {
            case 0: t[pos] = (char)(rng() % 256); break;
            case 1: t.erase(pos, 1 + rng() % 8); break;
            case 2: t.insert(pos, 1, "\t\n+-,0123456789SLPACGT>"[rng() % 23]); break;
            case 3: {   // duplicate a line
                size_t a = t.rfind('\n', pos), b = t.find('\n', pos);
                a = a == std::string::npos ? 0 : a + 1;
                b = b == std::string::npos ? t.size() : b + 1;
                t.insert(a, t.substr(a, b - a));
                break;
            }
            case 4: {   // drop a line
                size_t a = t.rfind('\n', pos), b = t.find('\n', pos);
                a = a == std::string::npos ? 0 : a + 1;
                b = b == std::string::npos ? t.size() : b + 1;
                t.erase(a, b - a);
                break;
            }
            case 5: t.resize(pos); break;                                   // truncate
            default: {  // overwrite a number with a huge / odd one
                static const char* nums[] = {"18446744073709551615", "99999999999999999999999", "0", "-1", "4294967296", ""};
                t.insert(pos, nums[rng() % 6]);
                break;
            }
        }
    }
    return t;
}

int main(int argc, char** argv) {
    // `host_asan --f32`: one f32 bit pattern (hex) per stdin line -> rg::f32_display of it per stdout line: the product's
    // formatter of the recombination score (`{}` of an f32 in the reference, recombination_output.rs:363-631), for the
    // independent shortest-round-trip check of tests/test_host_cpu.py
    if (argc == 2 && std::string(argv[1]) == "--f32") {
        char line[64];
        while (fgets(line, sizeof line, stdin)) {
            const uint32_t bits = (uint32_t)strtoul(line, nullptr, 16);
            float v;
            memcpy(&v, &bits, 4);
            puts(rg::f32_display(v).c_str());
        }
        return 0;
    }
    if (argc < 3) { fprintf(stderr, "usage: %s graph.gfa reads.fa [iterations] [seed]\n", argv[0]); return 2; }
    const std::string gfa = slurp(argv[1]), fa = slurp(argv[2]);
    const int iters = argc > 3 ? atoi(argv[3]) : 2000;
    std::mt19937_64 rng(argc > 4 ? strtoull(argv[4], nullptr, 10) : 1);
    // hand-made malformed GFAs (duplicate ids, links to nothing, back-links, empty segments, odd paths ...)
    const char* gfas[] = {"", "\n\n", "S\t1\tA\n", "S\t1\tA\nS\t1\tC\n", "S\tx\tA\n", "S\t1\t\n", "S\t1\n", "L\t1\t+\t2\t+\t0M\n",
                          "S\t1\tA\nS\t2\tC\nL\t2\t+\t1\t+\t0M\n", "S\t1\tA\nS\t2\tC\nL\t1\t-\t2\t+\t0M\n", "S\t1\tA\nL\t1\t+\t1\t+\t0M\n",
                          "S\t1\tA\nS\t2\tC\nL\t1\t+\t2\t+\t0M\nP\tp\t2+,1+\t*\n", "S\t1\tA\nS\t2\tC\nL\t1\t+\t2\t+\t0M\nP\tp\t1+,3+\t*\n",
                          "S\t1\tA\nS\t2\tC\nL\t1\t+\t2\t+\t0M\nP\tp\t1-,2+\t*\n", "S\t1\tA\nS\t2\tC\nL\t1\t+\t2\t+\t0M\nP\tp\t\t*\n",
                          "S\t1\tA\nS\t2\tC\nL\t1\t+\t2\t+\t0M\nP\tp\t,,,\t*\n", "S\t1\tA\nS\t3\tC\nL\t1\t+\t3\t+\t0M\nP\tp\t1+\t*\nP\tq\t1+,3+\t*\n",
                          "S\t18446744073709551615\tA\n", "S\t1\tA\r\nS\t2\tC\r\nL\t1\t+\t2\t+\t0M\r\n", "S\t1\tacgtn-\n", "H\tVN:Z:1.0\n",
                          "S\t1\tA\nS\t2\tC\nS\t3\tG\nL\t1\t+\t2\t+\t0M\nL\t1\t+\t3\t+\t0M\nP\ta\t1+,2+\t*\nP\tb\t1+,3+\t*\n"};
    for (const char* g : gfas) try_gfa(g);
    { std::string big = "S\t1\t" + std::string(70000, 'A') + "\n"; try_gfa(big); }
    const char* fas[] = {"", ">", ">\n", "A", ">a\n", ">a\nA\n>b\n", "A\n>a\n", ">a\r\nAC\r", "\n\n>a\n\nAC\n\n", ">a\nA\n>b\nC\n>c\nG", ">>>\n>\nA\n"};
    for (const char* f : fas) for (int64_t batch : {0, 1, 2, 5}) try_fasta(f, batch);
    for (int it = 0; it < iters; ++it) {
        try_gfa(mutate(gfa, rng));
        try_fasta(mutate(fa, rng), (int64_t)(rng() % 9));
    }
    try_gfa(gfa);
    try_fasta(fa, 7);
    printf("gfa ok %ld rejected %ld, fasta ok %ld rejected %ld\n", ok_gfa, bad_gfa, ok_fa, bad_fa);
    return 0;
}
