// Codes shared by the kernels and the host formatter (no HIP here: rg_gaf.cpp is part of the host-only sanitizer build).
#pragma once
#include <stdint.h>

namespace rg {

// traceback op codes (one byte per op, walk order)
enum : uint8_t { OP_D = 1, OP_U = 2, OP_L = 3, OP_CONT = 0x80 };

// status bits mirror include/recgraph_hip.h
enum : uint32_t { ST_BAND_WARNING = 1u, ST_BAND_NOT_ENOUGH = 2u, ST_WOULD_PANIC = 4u, ST_BAD_BASE = 8u, ST_OVERFLOW = 0x100u };

}  // namespace rg

// PATH RETIREMENT of k_sweep16: the sweeps look for hopeless paths every 2^RG_SWEEP16_RETIRE_SHIFT step records;
// the kernel and the builder of the tables that go with it (rg_steps.cpp) share the constant
#ifndef RG_SWEEP16_RETIRE_SHIFT
#define RG_SWEEP16_RETIRE_SHIFT 8
#endif
