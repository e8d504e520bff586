// Streaming engine of librecgraph_hip (rg_stream_*, rg_reads_*, rg_align_batch_multi in include/recgraph_hip.h): the
// reference's read loop (main.rs:56,174,257,297-312) as a pipeline hidden behind the C ABI.
//
//   caller thread(s)   rg_stream_push: copy reads + names, cut into tiles, append to ONE queue
//   worker threads     `handles_per_device` per device, each owns one rg_batch handle (HIP stream + HBM work buffers):
//                      pop a tile -> canonicalise + upload (rg_batch_set_reads) -> kernels (rg_batch_run) -> record
//                      fetch -> GAF text on `format_threads` host threads -> publish
//   caller thread      rg_stream_next: tiles in input order
//
// The queue is shared by every handle of every device (a tile goes to whichever handle is free first), so devices
// balance without a static split; the small latency-bound kernels of one tile run beside the sweeps of the other
// handles' tiles, and all host work of a tile overlaps the device work of the others.  There is no CPU fallback:
// rg_stream_create fails with RG_ERR_NO_DEVICE when no HIP device is usable.
#include <sched.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <thread>

#include "rg_batch_impl.hpp"

// ---------------------------------------------------------------------------------------------------------------------
// FASTA (sequences::get_sequences, sequences.rs:5-45)
struct rg_reads : rg::FastaReads {
    std::vector<const char*> name_ptrs;
};

extern "C" {

int32_t rg_reads_from_fasta(const char* text, int64_t len, rg_reads** out) {
    if (!text || !out || len < 0) return fail(RG_ERR_ARG, "null argument");
    auto r = std::make_unique<rg_reads>();
    if (!parse_fasta(text, len, *r, 0, [](int64_t, int64_t) {})) return fail(RG_ERR_ARG, "wrong fasta file format");
    r->name_ptrs.reserve(r->names.size());
    for (auto& n : r->names) r->name_ptrs.push_back(n.c_str());
    *out = r.release();
    return RG_OK;
}
int64_t rg_reads_count(const rg_reads* r) { return r ? (int64_t)r->names.size() : 0; }
const char* rg_reads_bases(const rg_reads* r) { return r ? r->bases.c_str() : nullptr; }
const int64_t* rg_reads_offsets(const rg_reads* r) { return r ? r->off.data() : nullptr; }
const char* const* rg_reads_names(const rg_reads* r) { return r ? r->name_ptrs.data() : nullptr; }
void rg_reads_destroy(rg_reads* r) { delete r; }

int32_t rg_fasta_check(const char* piece, int64_t len, int32_t final, int64_t* state4, int64_t* nreads_out) {
    if ((!piece && len > 0) || len < 0 || !state4) return fail(RG_ERR_ARG, "null argument");
    fasta_count(piece, len, final != 0, state4);
    if (nreads_out) *nreads_out = std::min(state4[0], state4[1]);
    if (final && state4[0] != state4[1]) return fail(RG_ERR_ARG, "wrong fasta file format");
    return RG_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct Tile {
    int64_t id = 0;                     // position in the output order
    int64_t first = 0, n = 0;           // reads [first, first + n) of the stream
    std::string bases;
    std::vector<int64_t> off;           // n + 1, relative to `bases`
    std::vector<std::string> names;     // empty: "read<first + i>"
    // results
    int rc = RG_OK;
    std::string err;
    std::string text;
    std::vector<int64_t> text_off;
    std::vector<uint32_t> status;
    std::vector<int32_t> score;
    int device = -1;
    uint64_t cells = 0, cells_performed = 0;
    int64_t out_bytes = 0;              // what the finished tile holds until rg_stream_next has delivered it
    rg_batch* records = nullptr;        // keep_records: owned by the stream once delivered
    ~Tile() { if (records) rg_batch_destroy_impl(records); }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// CPUs this process may actually use: hardware threads capped by the scheduler affinity and by a cgroup CPU quota (a
// container that shows 256 CPUs and carries a quota of 16 is throttled beyond 16 busy threads).
int usable_cpus() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        long long per = 0;
        if (fscanf(f, "%31s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) {
            const long long lim = (atoll(q) + per - 1) / per;
            if (lim >= 1) n = (int)std::min<long long>(n, lim);
        }
        fclose(f);
    }
    return n;
}

bool mode_is_pathwise(int mode) {
    return mode == RG_MODE_PATHWISE || mode == RG_MODE_RECOMBINATION || mode == RG_MODE_PATHWISE_SEMI || mode == RG_MODE_RECOMBINATION_SEMI;
}
bool mode_is_local(int mode) { return mode == RG_MODE_LOCAL_POA || mode == RG_MODE_LOCAL_POA_SCALAR || mode == RG_MODE_GAP_LOCAL_POA; }

}  // namespace

struct rg_stream {
    const rg_graph* g = nullptr;
    rg_params p;
    rg_stream_opts o;
    std::vector<int> devs;                  // one entry per device slot (a device may be listed more than once)
    std::vector<size_t> budget;             // HBM share of one handle, per slot
    int tile_reads = 4096;
    int format_threads = 1;
    bool amb = false;                       // `-s true` retry inside the workers (POA modes)

    std::mutex mu;
    std::condition_variable cv_work, cv_done, cv_space;
    std::deque<Tile*> queue;
    std::map<int64_t, Tile*> done;
    int64_t tiles_pushed = 0, next_out = 0, reads_pushed = 0;
    int64_t undelivered_bytes = 0;          // out_bytes of the tiles in `done`
    bool finished = false, stopping = false;
    bool aborted = false;                   // rg_stream_abort: queued tiles are dropped, every blocked push / feed / next returns an error
    std::vector<std::thread> workers;
    std::unique_ptr<Tile> cur;              // the tile rg_stream_next last returned
    std::vector<rg_batch*> kept;            // keep_records: results-only handles of the delivered tiles (until released)

    std::mutex pmu;                         // one push at a time: the tiles of a push get consecutive ids even when it has to wait for
                                            // room in the queue (pushes from several threads queue up behind each other)
    std::mutex fmu;                         // rg_stream_feed_fasta: one text at a time
    rg::FastaFeeder feeder;
    rg::FastaReads fbuf;                    // complete reads not yet pushed (less than one tile between calls)

    std::mutex smu;                         // statistics
    std::vector<KernelStat> kstats;
    double host_s[4] = {0, 0, 0, 0};        // set_reads, run, fetch, format
    double first_s[3] = {0, 0, 0};          // the same phases of every handle's FIRST tile (handle creation, buffer allocation) + context warm-up
    int64_t tiles_done = 0;
    int handles_used = 0;
    std::vector<std::string> stat_names;    // storage for rg_stream_kernel_name

    ~rg_stream() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stopping = true;
        }
        cv_work.notify_all();
        cv_space.notify_all();
        cv_done.notify_all();
        // a thread still inside rg_stream_push / rg_stream_feed_fasta (blocked on a full queue until `stopping` woke it) leaves
        // through members of this object: wait until it is out (a daemon feeder at interpreter teardown)
        { std::lock_guard<std::mutex> f(fmu); }
        { std::lock_guard<std::mutex> q(pmu); }
        for (auto& t : workers) t.join();
        for (Tile* t : queue) delete t;
        for (auto& kv : done) delete kv.second;
        for (rg_batch* b : kept) rg_batch_destroy_impl(b);
    }

    // `-s true` (main.rs:82-106, 132-165, 188-212, 229-253): the reads of the fetched batch `h` that qualify are aligned
    // again, reverse-complemented (sequences::rev_and_compl, sequences.rs:65-82, on the canonical read), on the worker's
    // second handle: scalar exec for -m 0 (main.rs:88), reversed handle labels + strand '-' (amb_mode = true) except for
    // -m 3, whose retry passes amb_mode = false (main.rs:240: labels reversed by hofp_reverse only).
    int amb_retry(rg_batch* h, rg_batch*& h2, size_t mem_budget, std::vector<int64_t>& rev_index) {
        const int mode = p.mode;
        const int64_t n = h->nreads;
        rev_index.assign((size_t)n, -1);
        std::string blob;
        std::vector<int64_t> off(1, 0);
        int64_t k = 0;
        for (int64_t i = 0; i < n; ++i) {
            const DevRecord& d = h->rec[(size_t)i];
            if (d.status & (ST_BAD_BASE | ST_WOULD_PANIC)) continue;
            if (!mode_is_local(mode) && !(d.score < 0)) continue;                 // main.rs:82,188 `alignment.0 < 0`
            const long long lo = h->off[(size_t)i], hi = h->off[(size_t)i + 1];
            for (long long c = hi; c-- > lo;) blob.push_back("TGCAN"[h->codes[(size_t)c]]);
            off.push_back((int64_t)blob.size());
            rev_index[(size_t)i] = k++;
        }
        if (!k) return RG_OK;
        int rc;
        if (!h2) {
            rg_params p2 = p;
            p2.mode = mode == RG_MODE_GLOBAL_POA ? RG_MODE_GLOBAL_POA_SCALAR : mode;
            p2.amb_mode = mode == RG_MODE_GAP_LOCAL_POA ? 1 : 3;
            rc = rg_batch_create(g, &p2, blob.data(), off.data(), k, &h2);
            if (rc == RG_OK) { h2->mem_budget = mem_budget; h2->spin_wait = o.spin_wait != 0; }
        } else {
            rc = rg_batch_set_reads(h2, blob.data(), off.data(), k);
        }
        if (rc == RG_OK) rc = rg_batch_run(h2);
        if (rc == RG_OK) rc = rg_batch_fetch(h2);
        return rc;
    }

    // One tile on this worker's handle(s); every failure is reported in the tile, never thrown past the worker
    void run_tile(Tile* t, int slot, int dev, rg_batch*& h, rg_batch*& h2, bool& counted) {
        double ts[5];
        ts[0] = now_s();
        const bool dbg = options().debug != 0;
        if (dbg) fprintf(stderr, "[rg] stream worker %d: tile %lld (%lld reads) starts\n", slot, (long long)t->id, (long long)t->n);
        int rc;
        if (!h) {
            // (the handle is created on the first tile: its work buffers are sized by what it actually aligns)
            rc = rg_batch_create(g, &p, t->bases.data(), t->off.data(), t->n, &h);
            if (rc == RG_OK) { h->mem_budget = budget[(size_t)slot] / (amb ? 2 : 1); h->spin_wait = o.spin_wait != 0; }
        } else {
            rc = rg_batch_set_reads(h, t->bases.data(), t->off.data(), t->n);
        }
        ts[1] = now_s();
        if (rc == RG_OK) { std::string().swap(t->bases); std::vector<int64_t>().swap(t->off); }    // the handle holds its own copy now
        if (rc == RG_OK) rc = rg_batch_run(h);
        ts[2] = now_s();
        if (rc == RG_OK) rc = rg_batch_fetch(h);
        std::vector<int64_t> rev_index;
        if (rc == RG_OK && amb) rc = amb_retry(h, h2, budget[(size_t)slot] / 2, rev_index);
        ts[3] = now_s();
        t->device = dev;
        bool have_rev = false;
        if (rc == RG_OK) {
            t->cells = h->cells;
            t->cells_performed = h->cells_performed;
            t->status.resize((size_t)t->n);
            t->score.resize((size_t)t->n);
            for (int64_t i = 0; i < t->n; ++i) { t->status[(size_t)i] = h->rec[(size_t)i].status & 0xffu; t->score[(size_t)i] = h->rec[(size_t)i].score; }
            AmbRetry ar;
            have_rev = amb && h2 && !rev_index.empty() && std::any_of(rev_index.begin(), rev_index.end(), [](int64_t v) { return v >= 0; });
            if (have_rev) {
                ar.rb = h2;
                ar.rev_index = rev_index.data();
                t->cells += h2->cells;
                t->cells_performed += h2->cells_performed;
                for (int64_t i = 0; i < t->n; ++i) {
                    const int64_t k = rev_index[(size_t)i];
                    if (k < 0) continue;
                    const DevRecord& rd = h2->rec[(size_t)k];
                    if (rd.status & (ST_BAD_BASE | ST_WOULD_PANIC)) { t->status[(size_t)i] |= rd.status & (ST_BAD_BASE | ST_WOULD_PANIC) & 0xffu; continue; }
                    if (amb_take_rev(p.mode, h->rec[(size_t)i].score, rd.score)) t->score[(size_t)i] = rd.score;    // score of the record written
                }
            }
            if (!o.no_text) {
                std::vector<const char*> np;
                if (!t->names.empty()) { np.reserve(t->names.size()); for (auto& s : t->names) np.push_back(s.c_str()); }
                format_batch(h, np.empty() ? nullptr : np.data(), t->first, o.seq_index_base + t->first, format_threads, t->text, &t->text_off,
                             have_rev ? &ar : nullptr);
            } else {
                t->text_off.assign((size_t)t->n + 1, 0);
            }
            if (o.keep_records) t->records = rg_batch_detach_results(h);
        } else {
            t->rc = rc;
            t->err = g_last_error;
        }
        std::vector<std::string>().swap(t->names);
        ts[4] = now_s();
        if (dbg) fprintf(stderr, "[rg] stream worker %d: tile %lld done rc %d: set_reads %.1f run %.1f fetch %.1f format %.1f ms\n", slot, (long long)t->id, rc,
                         (ts[1] - ts[0]) * 1e3, (ts[2] - ts[1]) * 1e3, (ts[3] - ts[2]) * 1e3, (ts[4] - ts[3]) * 1e3);
        std::lock_guard<std::mutex> lk(smu);
        for (int k = 0; k < 4; ++k) host_s[k] += ts[k + 1] - ts[k];
        ++tiles_done;
        if (h && !counted) { ++handles_used; counted = true; first_s[0] += ts[1] - ts[0]; first_s[1] += ts[2] - ts[1]; }
        if (rc == RG_OK)
            for (rg_batch* hb : {h, have_rev ? h2 : (rg_batch*)nullptr})       // (the retry handle's kernels count when it ran for this tile)
                if (hb)
                    for (auto& s : hb->stats) {
                        bool found = false;
                        for (auto& a : kstats) if (a.name == s.name) { a.ms += s.ms; a.launches += s.launches; found = true; break; }
                        if (!found) kstats.push_back(s);
                    }
    }

    void worker(int slot) {
        const int dev = devs[(size_t)slot];
        // A worker whose device cannot be selected must not run tiles at all: rg_batch_create would bind its handle to
        // whatever device is current (device 0) while the tile reports `dev`.
        bool dev_ok = hipSetDevice(dev) == hipSuccess;
        if (!dev_ok) (void)hipGetLastError();
        rg_batch* h = nullptr;
        rg_batch* h2 = nullptr;             // amb_strand: the reverse-complement pass
        bool counted = false;
        if (dev_ok) {
            // device context of this thread's device now, while the caller still parses / pushes its reads
            const double w0 = now_s();
            (void)hipFree(nullptr);
            std::lock_guard<std::mutex> lk(smu);
            first_s[2] += now_s() - w0;
        }
        for (;;) {
            Tile* t = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                // back-pressure on the output side: no new tile while the finished, undelivered ones hold more than the
                // cap — except the tile rg_stream_next is waiting for (the queue is in id order: that is its front)
                cv_work.wait(lk, [&] {
                    if (stopping || aborted) return true;
                    if (queue.empty()) return false;
                    return o.max_undelivered_bytes <= 0 || undelivered_bytes < o.max_undelivered_bytes || queue.front()->id == next_out;
                });
                if (stopping || aborted) break;
                t = queue.front();
                queue.pop_front();
            }
            cv_space.notify_all();
            if (!dev_ok) {
                t->rc = RG_ERR_NO_DEVICE;
                t->err = "HIP device " + std::to_string(dev) + " could not be selected by its worker thread";
                t->device = dev;
            } else {
                try {
                    run_tile(t, slot, dev, h, h2, counted);
                } catch (const std::exception& ex) {        // std::bad_alloc while copying / formatting a tile: the tile fails, the process lives
                    t->rc = RG_ERR_CAPACITY;
                    t->err = std::string("host memory: ") + ex.what();
                    t->device = dev;
                    std::string().swap(t->text);
                }
            }
            t->out_bytes = (int64_t)(t->text.capacity() + t->text_off.capacity() * 8 + t->status.capacity() * 4 + t->score.capacity() * 4);
            if (t->records) t->out_bytes += (int64_t)(t->records->ops.capacity() + t->records->oprows.capacity() * 4 + t->records->rec.capacity() * sizeof(DevRecord) + t->records->codes_own.capacity());
            {
                std::lock_guard<std::mutex> lk(mu);
                done[t->id] = t;
                undelivered_bytes += t->out_bytes;
            }
            cv_done.notify_all();
        }
        if (h) rg_batch_destroy_impl(h);
        if (h2) rg_batch_destroy_impl(h2);
    }

    // Cut [0, nreads) into tiles and queue them (blocking while max_queued_tiles are waiting).  Tiles hold at most tile_reads
    // reads and are even-sized.  Only the FIRST push of a stream and pushes larger than one tile are spread over the
    // device slots (at least one tile per slot, a multiple of the slot count): a push of one tile — what
    // rg_stream_feed_fasta and a caller that cuts its own tiles hand over — stays one tile.
    int push(const char* reads, const int64_t* read_off, int64_t nreads, const char* const* names, const std::vector<std::string>* own_names) {
        std::lock_guard<std::mutex> one_push(pmu);
        const int64_t T = tile_reads, D = (int64_t)devs.size();
        int64_t nt = (nreads + T - 1) / T;
        bool first_push;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (aborted) return fail(RG_ERR_ARG, "stream aborted");
            if (finished) return fail(RG_ERR_ARG, "rg_stream_push after rg_stream_finish");
            first_push = tiles_pushed == 0;
        }
        if (first_push || nreads > T) {
            nt = std::max(nt, std::min<int64_t>(D, nreads));
            if (D > 1 && nreads >= nt + D) nt = (nt + D - 1) / D * D;
        }
        int64_t base = -1;
        for (int64_t k = 0; k < nt; ++k) {
            const int64_t lo = nreads * k / nt, hi = nreads * (k + 1) / nt;
            if (hi == lo) continue;
            auto t = std::make_unique<Tile>();
            t->n = hi - lo;
            t->bases.assign(reads + read_off[lo], reads + read_off[hi]);
            t->off.resize((size_t)t->n + 1);
            for (int64_t i = 0; i <= t->n; ++i) t->off[(size_t)i] = read_off[lo + i] - read_off[lo];
            if (names) { t->names.reserve((size_t)t->n); for (int64_t i = lo; i < hi; ++i) t->names.emplace_back(names[i] ? names[i] : ""); }
            else if (own_names) { t->names.assign(own_names->begin() + lo, own_names->begin() + hi); }
            {
                std::unique_lock<std::mutex> lk(mu);
                if (o.max_queued_tiles > 0)
                    cv_space.wait(lk, [&] { return stopping || finished || (int64_t)queue.size() < o.max_queued_tiles; });
                if (aborted || stopping) return fail(RG_ERR_ARG, "stream aborted");
                if (finished) return fail(RG_ERR_ARG, "rg_stream_push after rg_stream_finish");
                if (base < 0) { base = reads_pushed; reads_pushed += nreads; }
                t->id = tiles_pushed++;
                t->first = base + lo;
                queue.push_back(t.release());
            }
            cv_work.notify_all();
        }
        return RG_OK;
    }
};

extern "C" {

void rg_stream_opts_default(rg_stream_opts* o) {
    if (o) memset(o, 0, sizeof *o);
    if (o) o->seq_index_base = 1;
}

int32_t rg_stream_create(const rg_graph* g, const rg_params* p, const int32_t* device_ids, int32_t ndev, const rg_stream_opts* opts,
                         rg_stream** out) {
    if (!g || !p || !out) return fail(RG_ERR_ARG, "null argument");
    auto s = std::make_unique<rg_stream>();
    s->g = g;
    s->p = *p;
    if (opts) s->o = *opts; else rg_stream_opts_default(&s->o);
    if (s->o.max_queued_tiles < 0 || s->o.max_undelivered_bytes < 0) return fail(RG_ERR_ARG, "negative stream bound");
    s->amb = s->o.amb_strand != 0 && !mode_is_pathwise(p->mode);      // modes 4+ ignore -s (main.rs:254-313)
    if (s->amb && s->o.keep_records) return fail(RG_ERR_ARG, "amb_strand and keep_records exclude each other (the kept record would be the forward one)");
    if (s->amb && p->amb_mode) return fail(RG_ERR_ARG, "amb_strand runs the retry itself: rg_params.amb_mode must be 0");
    const int visible = rg_device_count();
    if (visible < 1) return fail(RG_ERR_NO_DEVICE, "no HIP device (the product has no CPU path)");
    if (device_ids) {
        if (ndev < 1) return fail(RG_ERR_ARG, "empty device list");
        s->devs.assign(device_ids, device_ids + ndev);
        for (int d : s->devs) if (d < 0 || d >= visible) return fail(RG_ERR_NO_DEVICE, "no HIP device " + std::to_string(d));
    } else {
        for (int d = 0; d < visible; ++d) s->devs.push_back(d);
    }
    const int K = s->o.handles_per_device > 0 ? std::min(s->o.handles_per_device, 8) : 3;
    s->tile_reads = s->o.tile_reads > 0 ? s->o.tile_reads : (mode_is_pathwise(p->mode) ? 4096 : 8192);
    const int nworkers = K * (int)s->devs.size();
    const int hw = usable_cpus();
    s->format_threads = s->o.format_threads > 0 ? std::min(s->o.format_threads, 64) : std::max(1, std::min(16, hw / std::max(1, nworkers)));
    // HBM share of one handle: what is free on its device once the graph tables are there, less head-room for what is
    // not part of a run's work buffers (input block, records, ops, a retry pass), split over the handles of that device
    s->budget.assign(s->devs.size(), 0);
    for (size_t k = 0; k < s->devs.size(); ++k) {
        DevGuard dg(s->devs[k]);
        HIPCHK(dg.err);
        GraphTables* gt = nullptr;
        int rc = upload_graph(const_cast<rg_graph*>(g), &gt);
        if (rc) return rc;
        size_t fr = 0, tot = 0;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        const size_t same = (size_t)std::count(s->devs.begin(), s->devs.end(), s->devs[k]);
        const size_t head = std::min<size_t>(fr / 20, (size_t)4 << 30);      // 5 %, at most 4 GB
        s->budget[k] = (fr - head) / 100 * 92 / ((size_t)K * same);
    }
    rg_stream* sp = s.get();
    for (size_t k = 0; k < s->devs.size(); ++k)
        for (int j = 0; j < K; ++j) s->workers.emplace_back([sp, k] { sp->worker((int)k); });
    *out = s.release();
    return RG_OK;
}

int32_t rg_stream_push(rg_stream* s, const char* reads, const int64_t* read_off, int64_t nreads, const char* const* names) {
    if (!s || !reads || !read_off || nreads < 1) return fail(RG_ERR_ARG, "null/empty argument");
    for (int64_t r = 0; r < nreads; ++r)
        if (read_off[r + 1] - read_off[r] < 1) return fail(RG_ERR_ARG, "empty read");
    try {
        return s->push(reads, read_off, nreads, names, nullptr);
    } catch (const std::exception& ex) {
        return fail(RG_ERR_CAPACITY, std::string("host memory: ") + ex.what());
    }
}

int32_t rg_stream_feed_fasta(rg_stream* s, const char* piece, int64_t len, int32_t final, int64_t* nreads_out) {
    if (nreads_out) *nreads_out = 0;
    if (!s || (!piece && len > 0) || len < 0) return fail(RG_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> fl(s->fmu);
    int64_t total = 0;
    try {
        // the text is fed in blocks so that the reads completed by one block are pushed (and their text dropped) before the
        // next one is parsed: the stream never holds more parsed text than a block completes
        const int64_t block = 4 << 20;
        auto flush = [&](bool all) -> int {
            rg::FastaReads& r = s->fbuf;
            int64_t have = (int64_t)r.names.size(), at = 0;
            while (have - at >= s->tile_reads || (all && have > at)) {
                const int64_t cnt = std::min<int64_t>(s->tile_reads, have - at);
                for (int64_t i = 0; i < cnt; ++i)
                    if (r.off[(size_t)(at + i + 1)] - r.off[(size_t)(at + i)] < 1) return fail(RG_ERR_ARG, "empty read");
                // ONE tile per push (see rg_stream::push): names handed over by value
                std::vector<std::string> nm(std::make_move_iterator(r.names.begin() + at), std::make_move_iterator(r.names.begin() + at + cnt));
                const int rc = s->push(r.bases.data(), r.off.data() + at, cnt, nullptr, &nm);
                if (rc) return rc;
                at += cnt;
                total += cnt;
            }
            if (at) {       // drop what was pushed
                const int64_t cut = r.off[(size_t)at];
                r.bases.erase(0, (size_t)cut);
                r.names.erase(r.names.begin(), r.names.begin() + at);
                r.off.erase(r.off.begin(), r.off.begin() + at);
                for (auto& v : r.off) v -= cut;
            }
            return RG_OK;
        };
        int rc = RG_OK;
        for (int64_t pos = 0; pos < len && rc == RG_OK; pos += block) {
            const int64_t cnt = std::min(block, len - pos);
            s->feeder.feed(piece + pos, cnt, false, s->fbuf);
            rc = flush(false);
        }
        if (rc == RG_OK && final) {
            s->feeder.feed(nullptr, 0, true, s->fbuf);
            rc = flush(true);
        }
        if (nreads_out) *nreads_out = total;
        if (rc) {
            // a failed feed leaves nothing behind: reads already pushed are not pushed again, no stale carry
            s->feeder = rg::FastaFeeder();
            s->fbuf = rg::FastaReads();
            return rc;
        }
        if (final) {
            const bool ok = s->feeder.balanced();
            s->feeder = rg::FastaFeeder();
            s->fbuf = rg::FastaReads();
            if (!ok) return fail(RG_ERR_ARG, "wrong fasta file format");
        }
        return RG_OK;
    } catch (const std::exception& ex) {
        if (nreads_out) *nreads_out = total;
        s->feeder = rg::FastaFeeder();
        s->fbuf = rg::FastaReads();
        return fail(RG_ERR_CAPACITY, std::string("host memory: ") + ex.what());
    }
}

int32_t rg_stream_push_fasta(rg_stream* s, const char* fasta_text, int64_t len, int64_t* nreads_out) {
    if (!s || !fasta_text || len < 0) return fail(RG_ERR_ARG, "null argument");
    return rg_stream_feed_fasta(s, fasta_text, len, 1, nreads_out);
}

int32_t rg_stream_finish(rg_stream* s) {
    if (!s) return fail(RG_ERR_ARG, "null stream");
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->finished = true;
    }
    s->cv_done.notify_all();
    s->cv_space.notify_all();
    return RG_OK;
}

int32_t rg_stream_abort(rg_stream* s) {
    if (!s) return fail(RG_ERR_ARG, "null stream");
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->aborted = true;
        s->finished = true;
    }
    s->cv_done.notify_all();
    s->cv_space.notify_all();
    s->cv_work.notify_all();
    return RG_OK;
}

int64_t rg_stream_pending(rg_stream* s) {
    if (!s) return 0;
    std::lock_guard<std::mutex> lk(s->mu);
    return s->tiles_pushed - s->next_out;
}

int32_t rg_stream_next(rg_stream* s, rg_stream_result* out) {
    if (!s || !out) return fail(RG_ERR_ARG, "null argument");
    Tile* t = nullptr;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv_done.wait(lk, [&] { return s->aborted || s->done.count(s->next_out) || (s->finished && s->next_out == s->tiles_pushed); });
        if (s->aborted) return fail(RG_ERR_ARG, "stream aborted");
        auto it = s->done.find(s->next_out);
        if (it == s->done.end()) return RG_STREAM_END;
        t = it->second;
        s->done.erase(it);
        ++s->next_out;
        s->undelivered_bytes -= t->out_bytes;
    }
    s->cv_work.notify_all();        // workers held back by max_undelivered_bytes (or waiting for the tile that is now the next one out)
    s->cur.reset(t);
    memset(out, 0, sizeof *out);
    out->first_read = t->first;
    out->nreads = t->n;
    out->device = t->device;
    if (t->rc != RG_OK) return fail(t->rc, "tile of reads [" + std::to_string(t->first) + ", " + std::to_string(t->first + t->n) + ") on device " +
                                               std::to_string(t->device) + ": " + t->err);
    out->text = t->text.c_str();
    out->text_len = (int64_t)t->text.size();
    out->text_off = t->text_off.data();
    out->status = t->status.data();
    out->score = t->score.data();
    out->cell_updates = t->cells;
    out->cell_updates_performed = t->cells_performed;
    if (t->records) {
        std::lock_guard<std::mutex> lk(s->mu);
        s->kept.push_back(t->records);
        out->records = t->records;
        t->records = nullptr;
    }
    return RG_OK;
}

void rg_stream_release(rg_stream* s, rg_batch* records) {
    if (!s || !records) return;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        auto it = std::find(s->kept.begin(), s->kept.end(), records);
        if (it == s->kept.end()) return;            // not a handle of this stream (or released already)
        s->kept.erase(it);
    }
    rg_batch_destroy_impl(records);
}

void rg_stream_destroy(rg_stream* s) { delete s; }

int32_t rg_stream_kernel_count(rg_stream* s) {
    if (!s) return 0;
    std::lock_guard<std::mutex> lk(s->smu);
    return (int32_t)s->kstats.size() + 7;
}
static bool stream_stat(rg_stream* s, int32_t k, std::string* name, double* ms, int64_t* launches) {
    static const char* host_names[7] = {"host:set_reads", "host:run", "host:fetch", "host:format", "host:first_tile_create",
                                        "host:first_tile_run", "host:context_warmup"};
    std::lock_guard<std::mutex> lk(s->smu);
    const int32_t nk = (int32_t)s->kstats.size();
    if (k < 0 || k >= nk + 7) return false;
    if (k < nk) {
        if (name) *name = s->kstats[(size_t)k].name;
        if (ms) *ms = s->kstats[(size_t)k].ms;
        if (launches) *launches = s->kstats[(size_t)k].launches;
    } else {
        if (name) *name = host_names[k - nk];
        if (ms) *ms = (k - nk < 4 ? s->host_s[k - nk] : s->first_s[k - nk - 4]) * 1e3;
        if (launches) *launches = k - nk < 4 ? s->tiles_done : s->handles_used;
    }
    return true;
}
const char* rg_stream_kernel_name(rg_stream* s, int32_t k) {
    if (!s) return "";
    std::string nm;
    if (!stream_stat(s, k, &nm, nullptr, nullptr)) return "";
    std::lock_guard<std::mutex> lk(s->smu);
    for (auto& x : s->stat_names) if (x == nm) return x.c_str();
    s->stat_names.reserve(64);          // (pointers handed out stay valid: never more than a few dozen names)
    s->stat_names.push_back(nm);
    return s->stat_names.back().c_str();
}
double rg_stream_kernel_ms(rg_stream* s, int32_t k) {
    double ms = 0;
    if (s) stream_stat(s, k, nullptr, &ms, nullptr);
    return ms;
}
int64_t rg_stream_kernel_launches(rg_stream* s, int32_t k) {
    int64_t n = 0;
    if (s) stream_stat(s, k, nullptr, nullptr, &n);
    return n;
}
int64_t rg_stream_tiles_done(rg_stream* s) {
    if (!s) return 0;
    std::lock_guard<std::mutex> lk(s->smu);
    return s->tiles_done;
}
int32_t rg_stream_handles(rg_stream* s) {
    if (!s) return 0;
    std::lock_guard<std::mutex> lk(s->smu);
    return s->handles_used;
}

// ---- all visible GPUs behind one call ("one call may use all visible GPUs") ----
struct rg_multi {
    rg_stream* stream = nullptr;            // owns the shards (results-only handles of its tiles)
    std::vector<rg_batch*> shards;
    std::vector<int64_t> begin;             // shards.size() + 1
    ~rg_multi() { delete stream; }
};

int32_t rg_align_batch_multi(const rg_graph* g, const rg_params* p, const char* reads, const int64_t* read_off, int64_t nreads,
                             const int32_t* device_ids, int32_t ndev, rg_multi** out) {
    if (!g || !p || !reads || !read_off || !out || nreads < 1) return fail(RG_ERR_ARG, "null/empty argument");
    rg_stream_opts o;
    rg_stream_opts_default(&o);
    o.keep_records = 1;
    o.no_text = 1;
    auto m = std::make_unique<rg_multi>();
    int rc = rg_stream_create(g, p, device_ids, ndev, &o, &m->stream);
    if (rc) return rc;
    if ((rc = rg_stream_push(m->stream, reads, read_off, nreads, nullptr)) || (rc = rg_stream_finish(m->stream))) return rc;
    int first_err = RG_OK;
    std::string err;
    for (;;) {
        rg_stream_result r;
        rc = rg_stream_next(m->stream, &r);
        if (rc == RG_STREAM_END) break;
        if (rc != RG_OK) { if (first_err == RG_OK) { first_err = rc; err = g_last_error; } continue; }   // drain: the workers finish their tiles
        m->begin.push_back(r.first_read);
        m->shards.push_back(r.records);
    }
    if (first_err != RG_OK) return fail(first_err, err);
    m->begin.push_back(nreads);
    *out = m.release();
    return RG_OK;
}
int32_t rg_multi_shards(const rg_multi* m) { return m ? (int32_t)m->shards.size() : 0; }
rg_batch* rg_multi_batch(const rg_multi* m, int32_t k) { return m && k >= 0 && k < (int32_t)m->shards.size() ? m->shards[(size_t)k] : nullptr; }
int64_t rg_multi_shard_begin(const rg_multi* m, int32_t k) { return m && k >= 0 && k <= (int32_t)m->shards.size() ? m->begin[(size_t)k] : -1; }
int64_t rg_multi_format_all(const rg_multi* m, const char* const* names, int64_t seq_index_base, char* buf, int64_t cap,
                            int32_t nthreads) {
    if (!m) return fail(RG_ERR_ARG, "null handle");
    int64_t total = 0;
    for (size_t k = 0; k < m->shards.size(); ++k) {
        const int64_t left = buf && cap > total ? cap - total : 0;
        const int64_t need = rg_batch_format_all(m->shards[k], names ? names + m->begin[k] : nullptr, seq_index_base + m->begin[k],
                                                 left ? buf + total : nullptr, left, nthreads);
        if (need < 0) return need;
        total += need;
    }
    return total;
}
void rg_multi_destroy(rg_multi* m) { delete m; }

}  // extern "C"
