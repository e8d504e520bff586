// Host-only read ingestion for librecgraph_hip: FASTA text -> reads (sequences::get_sequences, sequences.rs:5-45) and
// the canonicalisation of a read set into base codes (sequences.rs:13-22 + the score-matrix alphabet).  No HIP here: this
// file, rg_graph.cpp and rg_gaf.cpp are what `make asan` builds with AddressSanitizer / UBSan (tests/c/host_asan.cpp).
#include <algorithm>
#include <cctype>
#include <cstring>
#include <deque>

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


// The same rules with the text arriving in pieces (any split, even inside a line or between '\r' and '\n'): what a piece
// leaves unfinished — the unterminated last line, the sequence being collected, names / sequences that have no partner
// yet — stays in the feeder, complete reads (sequence i closed AND name i seen: the reference pairs the two lists by
// index) are appended to `out`.  Memory held is what the current piece completes, not the file.
void FastaFeeder::line(const char* p, const char* q, FastaReads& out) {
    if (q <= p) return;                                     // empty lines are skipped (sequences.rs:14)
    if (*p == '>') {
        names.emplace_back(p + 1, q);
        ++names_total;
        if (!cur.empty()) { seqs.push_back(std::move(cur)); cur.clear(); ++seqs_total; }
    } else {
        for (const char* c = p; c < q; ++c) {
            const unsigned char ch = (unsigned char)*c;
            cur.push_back(ch == '-' ? 'N' : (ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : (char)ch);
        }
    }
    pair_up(out);
}

void FastaFeeder::pair_up(FastaReads& out) {
    if (out.off.empty()) out.off.push_back(0);
    while (!names.empty() && !seqs.empty()) {
        out.bases += seqs.front();
        out.off.push_back((int64_t)out.bases.size());
        out.names.push_back(std::move(names.front()));
        names.pop_front();
        seqs.pop_front();
    }
}

void FastaFeeder::feed(const char* text, int64_t len, bool final, FastaReads& out) {
    const char* p = text;
    const char* end = text + len;
    if (out.off.empty()) out.off.push_back(0);
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        if (!nl) { carry.append(p, end); break; }          // unterminated: wait for the rest of the line
        if (!carry.empty()) {
            carry.append(p, nl);
            const char* b = carry.data();
            const char* q = b + carry.size();
            if (q > b && q[-1] == '\r') --q;                // BufRead::lines drops "\n" or "\r\n"
            line(b, q, out);
            carry.clear();
        } else {
            const char* q = nl;
            if (q > p && q[-1] == '\r') --q;
            line(p, q, out);
        }
        p = nl + 1;
    }
    if (final) {
        if (!carry.empty()) { line(carry.data(), carry.data() + carry.size(), out); carry.clear(); }   // (a '\r' at the very end stays)
        if (!cur.empty()) { seqs.push_back(std::move(cur)); cur.clear(); ++seqs_total; }
        pair_up(out);
    }
}

// What sequences::get_sequences would collect, counted without keeping it (the reference parses the whole file — and
// panics on a name / sequence count mismatch, sequences.rs:41-43 — before it aligns anything; a caller that streams the
// file runs this ahead of its output).  st: {names, sequences, current sequence non-empty, line state}; zero before the
// first piece.  Line states: 0 at the start of a line, 1 inside a line, 2 after a '\r' that opened the line.
void fasta_count(const char* text, int64_t len, bool final, int64_t st[4]) {
    const char* p = text;
    const char* end = text + len;
    while (p < end) {
        if (st[3] == 1) {                                   // the rest of the line decides nothing
            const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
            if (!nl) return;
            p = nl + 1;
            st[3] = 0;
            continue;
        }
        const char c = *p++;
        if (st[3] == 2) {                                   // "\r\n" is an empty line; '\r' + anything else is content
            if (c == '\n') { st[3] = 0; continue; }
            st[2] = 1;
            st[3] = 1;
            continue;
        }
        if (c == '\n') continue;                            // empty line
        if (c == '\r') { st[3] = 2; continue; }
        if (c == '>') { ++st[0]; if (st[2]) { ++st[1]; st[2] = 0; } }
        else st[2] = 1;
        st[3] = 1;
    }
    if (final) {
        if (st[3] == 2) st[2] = 1;                          // a '\r' at the very end stays: a one-character line
        if (st[2]) { ++st[1]; st[2] = 0; }
        st[3] = 0;
    }
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
