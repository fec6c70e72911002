// kernels.hpp -- launch interface between the host pipeline and kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace mfx {

// device views of plan.hpp's Entry / TaskDesc (same layout)
struct EntryD { uint32_t own; int32_t gat; float r; };
struct TaskDescD { uint64_t off; uint32_t nsteps; uint32_t pad; };
struct WgTaskD { uint64_t off; uint32_t nsteps; uint32_t visit0; uint32_t nvisits; uint32_t swapped; };  // plan.hpp WgTask
struct WgVisitD { uint32_t row; uint32_t nsteps; uint32_t len; uint32_t info; uint32_t slot; uint32_t pad; }; // WgVisitRec

constexpr int LOSS_SLOTS = 256; // the epoch loss is kept as this many partial sums
constexpr int HOT_SUB = 8;      // a hot row's combine slot is kept as this many partial sums of (ka + HOT_EXTRA) floats:
constexpr int HOT_EXTRA = 5;    // the row, then both accumulator slots, squared errors, ratings, chains

// Arguments of one SGD round (= one launch = NS stripe-disjoint blocks).
struct RoundArgs {
    float *own_rows;  // factors of the owner side   (n_own x ka)
    float *gat_rows;  // factors of the gathered side (n_gat x ka)
    float *own_acc;   // Adagrad slots, 2 per row (reference PG/QG, mf.cpp:2835)
    float *gat_acc;
    float *hot_acc;   // combine slots of the hot rows (HOT_SUB x (ka + HOT_EXTRA) floats each), zero between rounds
    const EntryD *entries;
    const TaskDescD *tasks;
    const long long *slot_task_ptr; // ns+1 task offsets of this round
    int *slot_cursor;               // ns ints, zero before the launch
    const WgTaskD *wg_tasks;        // workgroup tasks (heavy rows), claimed by whole workgroups before the wave tasks
    const WgVisitD *wg_visits;
    const long long *slot_wg_ptr;   // ns+1 workgroup-task offsets of this round
    int *wg_cursor;                 // ns ints, zero before the launch
    double *loss;                   // LOSS_SLOTS partial sums of e^2 (scaled units), accumulated
    float lambda_own, lambda_gat, eta, rk1;
    int ka, slow_only, ns;
    int n_xcc;                      // XCDs that take work
    int active_waves;               // waves of a workgroup that take work (1..4)
#ifdef MFX_STAMPS
    unsigned long long *stamps;     // diagnostic build only: 8 cycle sums per wave (kernels.hip, STAMP)
#endif
    // stripes: internal-id boundaries (ns+1 ints each, device memory); slot s of round r works on owner
    // stripe s and gathered stripe (s + r) mod ns.  The kernel addresses the gathered side through buffer
    // descriptors over that stripe (a stripe must stay below 4 GB) and can stream both stripes into L2.
    const int *own_begin, *gat_begin;
    int round;
    int merge_back;                 // 1: the plan swaps roles for heavy gathered rows: owner rows are written back by merging
    int warm;                       // 1: stream the stripes of each slot into L2 before the first step
    int waves_per_xcd;              // waves that take work in one XCD (warm-up chunks are dealt over them)
    signed char xcc_rank[16];       // HW_REG_XCC_ID -> rank in [0, n_xcc), -1 = takes no work
};

hipError_t launch_sgd_round(int lanes, const RoundArgs &a, int grid, hipStream_t s);
hipError_t launch_fold_hot(float *own_rows, float *own_acc, float *gat_rows, float *gat_acc, float *hot_acc, const int *hot_row,
                           int n_slots, int ka, float eta, float rk1, int slow_only, hipStream_t s);
hipError_t launch_visibility_probe(int *ticket, float *row, int *flag, int *ack, int rounds, int *out, int grid, hipStream_t s);
hipError_t launch_probe_xcc(unsigned *mask, int grid, hipStream_t s);
hipError_t launch_sq_err_entries(int lanes, const float *own_rows, const float *gat_rows,
                                 const EntryD *entries, long long n_entries, int ka, double *out,
                                 int grid, hipStream_t s);
hipError_t launch_reg2(const float *rows, const int *omega, int nrows, int ka, double *out,
                       int grid, hipStream_t s);
hipError_t launch_predict(const float *model, int m, int n, int k, float b, const float *pairs,
                          long long npairs, float *out, int grid, hipStream_t s);
hipError_t launch_sq_err_nodes(const float *model, int m, int n, int k, float b, const EntryD *R,
                               long long nnz, double *out, int grid, hipStream_t s);
hipError_t launch_export(const float *rows, const int *map, int nrows, int k, int ka, float f,
                         int do_scale, float *out, int grid, hipStream_t s);
hipError_t launch_epoch_reset(double *loss, int *cursor, const long long *slot_task_ptr, const long long *slot_wg_ptr, int nb,
                              int check, int *sticky, hipStream_t s);
hipError_t launch_fill(float *p, long long n, float v, int grid, hipStream_t s);
hipError_t launch_triplets(const float *tri, long long count, void *out, int *mn_bad, int grid, hipStream_t s);
hipError_t launch_synth(unsigned long long seed, unsigned long long shard, long long first,
                        long long count, int m, int n, void *out, int grid, hipStream_t s);

} // namespace mfx
