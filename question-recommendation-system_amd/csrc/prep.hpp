// prep.hpp -- device-side pre-processing (prep.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include "kernels.hpp"
#include "plan.hpp"

namespace mfx {

// ids must fit the 24-bit fields of the sort key (they do whenever they arrived as floats)
bool device_prep_supported(int m, int n);

// Build the stripe/task layout from ratings resident in HBM (dR: nnz mf_node records).  Fills the
// host-side Plan (maps, counts, statistics, tasks, slot_task_ptr, n_entries) and returns the entry
// array in a fresh hipMalloc'd buffer (*d_entries, caller frees).  Throws std::runtime_error /
// std::invalid_argument / std::bad_alloc.
void build_plan_device(const void *dR, long long nnz, int m, int n, const PlanConfig &cfg, int cu_count,
                       hipStream_t s, Plan &p, EntryD **d_entries);

// init_model (reference mf/mf.cpp:952-1007) on the device, bit-identical to the host's init_factors:
// row counts (internal order) already in HBM, factors written with stride ka.  p_at/q_at (host, may be
// null = identity): internal row at each position of the reference's row order (Plan::p_at).
void init_factors_device(const int *d_omega_p, int m, const int *d_omega_q, int n, const int *p_at_host,
                         const int *q_at_host, int k, int ka, int cu_count, hipStream_t s, float *dP, float *dQ);

} // namespace mfx
