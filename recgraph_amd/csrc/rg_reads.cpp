// Host-only read ingestion for librecgraph_hip: FASTA text -> reads (sequences::get_sequences, sequences.rs:5-45) and
// the canonicalisation of a read set into base codes (sequences.rs:13-22 + the score-matrix alphabet).  No HIP here: this
// file, rg_graph.cpp and rg_gaf.cpp are what `make asan` builds with AddressSanitizer / UBSan (tests/c/host_asan.cpp).
#include <algorithm>
#include <cctype>
#include <cstring>

#include "rg_host.hpp"

namespace rg {

// Incremental form of the parser: `emit(first, count)` is called whenever `batch` more reads are complete (read i is
// complete once sequence i is closed AND name i exists: the reference pairs the two lists by index) and once more at the
// end for the rest.  Returns false ("wrong fasta file format") when the counts differ at the end of the text.
bool parse_fasta(const char* text, int64_t len, FastaReads& r, int64_t batch, const std::function<void(int64_t, int64_t)>& emit) {
    r.bases.reserve((size_t)len);
    r.off.push_back(0);
    // the reference pushes a name at every header and a sequence whenever the one being collected is non-empty at the
    // next header / at the end of the file; the two lists are paired by index afterwards (:41-43 panics when the counts differ)
    size_t cur_begin = 0;               // start of the sequence being collected inside r.bases
    int64_t emitted = 0;
    auto ready = [&] { return std::min<int64_t>((int64_t)r.off.size() - 1, (int64_t)r.names.size()); };
    const char* p = text;
    const char* end = text + len;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        const char* q = le;
        if (nl && q > p && q[-1] == '\r') --q;              // BufRead::lines drops "\n" or "\r\n" (a '\r' at the very end stays)
        if (q > p) {                                        // empty lines are skipped (:14)
            if (*p == '>') {
                r.names.emplace_back(p + 1, q);
                if (r.bases.size() > cur_begin) { r.off.push_back((int64_t)r.bases.size()); cur_begin = r.bases.size(); }
                if (batch > 0 && ready() - emitted >= batch) { emit(emitted, batch); emitted += batch; }
            } else {
                for (const char* c = p; c < q; ++c) {
                    const unsigned char ch = (unsigned char)*c;
                    // '-' -> 'N', ASCII upper-casing (char::to_ascii_uppercase leaves everything else alone)
                    r.bases.push_back(ch == '-' ? 'N' : (ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : (char)ch);
                }
            }
        }
        p = nl ? nl + 1 : end;
    }
    if (r.bases.size() > cur_begin) r.off.push_back((int64_t)r.bases.size());
    if (r.off.size() - 1 != r.names.size()) return false;
    if (ready() > emitted) emit(emitted, ready() - emitted);
    return true;
}


// Base codes of a read set: one table pass over the blob (canonical character: '-' -> 'N', upper case; code 0..4, a
// character outside ACGTN marks its read `bad` and is stored as N).  Returns the longest read.
int64_t canonicalise_reads(const char* reads, const int64_t* read_off, int64_t nreads, uint8_t* codes, uint8_t* bad) {
    static const struct Canon {
        uint8_t code[256];
        Canon() {
            for (int c = 0; c < 256; ++c) {
                const char u = c == '-' ? 'N' : (char)toupper(c);
                code[c] = u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : u == 'N' ? 4 : 0xff;
            }
        }
    } canon;
    const int64_t base = read_off[0];
    const size_t total = (size_t)(read_off[nreads] - base);
    const unsigned char* src = reinterpret_cast<const unsigned char*>(reads) + base;
    for (size_t k = 0; k < total; ++k) codes[k] = canon.code[src[k]];
    int64_t max_n = 0;
    for (int64_t r = 0; r < nreads; ++r) {
        const int64_t lo = read_off[r] - base, n = read_off[r + 1] - read_off[r];
        max_n = std::max(max_n, n);
        bad[r] = 0;
        // a character outside ACGTN (after canonicalisation): the reference panics on the score lookup
        if (n > 0 && memchr(codes + lo, 0xff, (size_t)n)) {
            bad[r] = 1;
            for (int64_t k = lo; k < lo + n; ++k) if (codes[k] == 0xff) codes[k] = 4;
        }
    }
    return max_n;
}

}  // namespace rg
