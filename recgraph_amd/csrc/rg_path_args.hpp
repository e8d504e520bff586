// Argument blocks and work buffers of the pathwise (-m 4 / -m 8) kernels.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "rg_device.hpp"
#include "rg_host.hpp"

namespace rg {

// flattened PathGraph (+ reverse PredHash, DP programs) in HBM
struct PathGraphDev {
    int L, P;
    const uint8_t* lnz;
    const uint64_t* row_mask;   // RG_PW words per row
    const int* knm;
    const int* dfs;
    const int* dfe;
    const int* fgoff;
    const int* rgoff;
    const GroupDesc* fgroups;
    const GroupDesc* rgroups;
    int fslots, rslots;
    const unsigned long long* node_id;
    const int* segfirst;
    const int* seglast;
    const int* eoff;
    const int* epred;
    const uint64_t* emask;      // RG_PW words per edge: path k is bit (k & 63) of word (k >> 6)
    const int* roff;
    const int* rsucc;
    const uint64_t* rmask;      // RG_PW words per edge
    const uint8_t* pnwp;
    const uint8_t* rnwp;
};

struct PathWorkImpl;
struct PathWork {
    PathWorkImpl* impl = nullptr;
    bool spin_wait = false;     // the owning handle waits for the device with hipStreamSynchronize (rg_stream_opts.spin_wait)
    ~PathWork();
};

}  // namespace rg
