// kernels.hip -- gfx950 kernels of the SGD matrix-factorisation path.
//
// sgd_round<LANES> is the hot loop: it replaces the reference's per-rating worker loop
// SolverBase::run + calc_z + L2_MFR::prepare_for_sg_update + MFSolver::sg_update
// (reference mf/mf.cpp:1201-1238, 1264-1273, 1720-1728, 1462-1548) and the block
// Scheduler (mf.cpp:49-312).  Design (DESIGN.md has the long form):
//
//  * A rating is processed by a group of LANES lanes, 4 consecutive factors per lane
//    (one 16-byte access per lane, LANES*16 B = one coalesced row segment); a 64-wide
//    wavefront runs 64/LANES ratings per step.  The k-dim dot and the gradient-norm sums
//    are DPP butterflies inside the group -- no LDS, no MFMA (nothing here is a dense
//    contraction).
//  * "Owner computes": ratings arrive grouped by the row of one side (the owner side).
//    The owner row and its two Adagrad slots stay in registers across the whole run and
//    are written back once; only the other ("gathered") side's row is read and written
//    per rating.
//  * The conflict-free idea of the reference scheduler (no two live blocks share a row or
//    column stripe, mf.cpp:133-141) is kept at XCD granularity: one launch = one round of
//    NS stripe-disjoint blocks; a block is worked by the waves of ONE XCD only, so all of
//    its rows live in that XCD's L2 and no cross-XCD coherence is needed inside a launch
//    (the per-XCD L2s are not coherent with each other).  A wave reads HW_REG_XCC_ID and
//    serves the blocks of its own XCD (rank, rank+X, ...): which workgroup lands where
//    changes speed, never results.  Inside a block the XCD's waves run lock-free (Hogwild)
//    on the gathered side; factor loads bypass the per-CU L1 so they see the XCD's L2.
//  * The number of ratings in flight is capped (lost updates cost RMSE), so the time of a
//    launch is (steps per wave) x (latency of one step), and a step is a chain: L2 round
//    trip + the wave's own instructions.  vmcnt retires in order on gfx9 and hipcc waits
//    conservatively across branches, so the loop is laid out around its waits: one burst of
//    memory operations per step with one settle point, entries streamed through a per-wave
//    LDS ring in blocks, the next task claimed/fetched while the current one runs, the
//    block's stripes streamed into L2 before the first step.  Comments at each piece.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace mfx {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// ---- cross-lane helpers -------------------------------------------------------------

template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// x + the value of the lane 16 (32) positions across: the butterfly step that crosses 16-lane rows.  gfx950 has VALU
// instructions for it (v_permlane16_swap / v_permlane32_swap: with the same value in both operands the two results are "mine" and
// "the other row's"); ds_bpermute (__shfl_xor) goes through the LDS and costs a round trip per value -- five values per step at
// k >= 128, fourteen per step of a workgroup task at k = 64.  Bit-identical to the __shfl_xor form
// (scripts/probes/permlane_probe.hip).  Inline asm: given the builtin with one value for both operands, hipcc 7.2 folds the two
// results into one.
__device__ __forceinline__ float xor16_sum(float x)
{
#ifdef MFX_SHFL_XOR // experiment/bisect: the LDS form
    return x + __shfl_xor(x, 16);
#else
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); // (the wait states a freshly written source needs)
    return a + b;
#endif
}
__device__ __forceinline__ float xor32_sum(float x)
{
#ifdef MFX_SHFL_XOR
    return x + __shfl_xor(x, 32);
#else
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
#endif
}

// Sum over the LANES-lane group this lane belongs to; every lane gets the total.
// quad_perm swaps for 1 and 2, row_half_mirror / row_mirror for 4 and 8 (valid because
// the halves are already uniform at that point), permlane swaps beyond a 16-lane row.
template <int LANES>
__device__ __forceinline__ float group_sum(float x)
{
    if (LANES >= 2) x += dpp<0xB1>(x);  // quad_perm [1,0,3,2]
    if (LANES >= 4) x += dpp<0x4E>(x);  // quad_perm [2,3,0,1]
    if (LANES >= 8) x += dpp<0x141>(x); // row_half_mirror
    if (LANES >= 16) x += dpp<0x140>(x); // row_mirror
    if (LANES >= 32) x = xor16_sum(x);
    if (LANES >= 64) x = xor32_sum(x);
    return x;
}

// Sum over the lanes of a wavefront that hold the same factors of DIFFERENT ratings (same position inside their LANES-lane
// group): rotations inside a 16-lane row (DPP row_ror), ds_bpermute across rows.  Every lane gets the total.
template <int LANES>
__device__ __forceinline__ float cross_group_sum(float x)
{
    if (LANES <= 2) x += dpp<0x122>(x);  // row_ror:2
    if (LANES <= 4) x += dpp<0x124>(x);  // row_ror:4
    if (LANES <= 8) x += dpp<0x128>(x);  // row_ror:8
    if (LANES <= 16) x = xor16_sum(x);
    if (LANES <= 32) x = xor32_sum(x);
    return x;
}

// LDS float add without a return value (ds_add_f32)
__device__ __forceinline__ void lds_add(float *p, float v)
{
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ int xcc_id()
{
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x;
}

// factor loads: served by the XCD's L2, never by a stale per-CU L1 line
#ifndef MFX_LD_PLAIN
__device__ __forceinline__ f4 ld_row(const float *p) { return __builtin_nontemporal_load((const f4 *)p); }
__device__ __forceinline__ f2 ld_acc(const float *p) { return __builtin_nontemporal_load((const f2 *)p); }
#else // experiment only: L1-cached loads (can read a stale line)
__device__ __forceinline__ f4 ld_row(const float *p) { return *(const f4 *)p; }
__device__ __forceinline__ f2 ld_acc(const float *p) { return *(const f2 *)p; }
#endif

// Gathered-side accesses through a raw buffer descriptor over the block's gathered stripe: 32-bit offsets
// (one or two VALU instructions instead of a 64-bit address chain), and the hardware range check does
// the pad entries and the lanes past k_a -- an offset beyond the stripe loads zeros and drops the store.
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
constexpr unsigned BUF_OOB = 0xFFFFFF00u; // beyond any stripe
constexpr int BUF_NT = 2;                  // aux: non-temporal, like ld_row (past the per-CU L1)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f4 bld_row(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, BUF_NT));
}
__device__ __forceinline__ f2 bld_acc(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    return __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, BUF_NT));
}
__device__ __forceinline__ void bst_row(__amdgpu_buffer_rsrc_t r, unsigned off, f4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), r, (int)off, 0, 0);
}
__device__ __forceinline__ void bst_acc(__amdgpu_buffer_rsrc_t r, unsigned off, f2 v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, v), r, (int)off, 0, 0);
}

#ifdef MFX_STAMPS
// Diagnostic build only (`make diag` -> lib_diag/): s_memtime stamps around the segments of a step.
// The sums go to a buffer of their own; no result depends on them; the shipped library has none of this.
#define STAMP(t)                                                                                 \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
// 100 MHz wall clock shared by the whole device (s_memtime counters of different CUs are offset)
#define RSTAMP(t)                                                                                \
    do {                                                                                         \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");           \
    } while (0)
#else
#define STAMP(t) do { } while (0)
#endif

// ---- the SGD round ----------------------------------------------------------------------

// LANES  lanes per rating (power of two, LANES*4 >= k_a)
// FULL   k_a == LANES*4: every lane carries factors (no per-lane bounds test)
// SLOW   epoch 0 of the reference: only factors [0,8) move (slow_only, mf.cpp:2834, 1230-1231)
#ifdef MFX_WAVES_PER_EU // experiment (make variant VFLAGS=-DMFX_WAVES_PER_EU=5): ask the register allocator for that occupancy
#define MFX_OCC __attribute__((amdgpu_waves_per_eu(MFX_WAVES_PER_EU, 8)))
#else
#define MFX_OCC
#endif
template <int LANES, bool FULL, bool SLOW>
__global__ __launch_bounds__(256) MFX_OCC void sgd_round(RoundArgs a)
{
    constexpr int G = 64 / LANES;
    constexpr int EBLK = 128; // entries per block of the entry stream (two per lane); EBLK/G steps
    static_assert(EBLK / G >= 4, "a block of the entry stream must span at least four steps");
    constexpr unsigned NONE = 0xFFFFFFFFu, IDMASK = 0x3FFFFFFFu; // (bit 31 of `own`: a visit starts; bit 30: roles swapped)
    constexpr int GAT_ID = 0x3FFFFFFF, GAT_RO = 0x40000000; // bit 30 of `gat`: read the row, do not write it (plan.hpp ENTRY_READ_ONLY)
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    __shared__ u4 ering_all[4][2 * EBLK]; // per wave: a ring of two blocks, 16-byte slots
    u4 *const ering = ering_all[threadIdx.x >> 6];
    __shared__ int wg_first;
    // workgroup tasks: the heavy row of the visit in progress lives here (LDS float atomics from every list of the workgroup)
    __shared__ f4 lrow4_all[2][64]; // two buffers: the next visit's row is staged while the current one is worked on
    __shared__ float lacc_all[2][2];
    __shared__ float lE_all[2];
    __shared__ float lg_init[2]; // the smaller accumulator slot as the visit found it (read by every wave: must not change under it)
#ifdef MFX_OWNER_LDS
    // experiment (make variant VFLAGS=-DMFX_OWNER_LDS): the "LDS-staged latent tile" for the one side that has re-use -- the
    // owner row of a visit lives in LDS (read before and written after every update) instead of in registers
    __shared__ volatile f4 own_tile[256];
#endif
    const int lane = threadIdx.x & 63;
    const int lig = lane % LANES;
    const int grp = lane / LANES;
    const int d0 = lig * 4;
    const int ka = FULL ? LANES * 4 : a.ka; // a compile-time constant when every lane carries factors
    const bool lane_ok = FULL || d0 < ka;
    const bool slot1 = d0 >= 8;               // dims [8,k_a) use accumulator slot 1
    const bool upd = lane_ok && !(SLOW && slot1);
    const float lam_o = a.lambda_own, lam_g = a.lambda_gat, eta = a.eta;
    const float rk0 = 0.125f, rk1 = a.rk1;
    const f4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};

    // Blocks ("slots") of this round that belong to this wave's XCD: rank, rank+X, ...
    // (xcc_rank comes from a probe launch; an XCD the probe did not see takes no work and
    // the host's cursor check reports the unfinished blocks -- never a silent wrong answer).
    const int rank = a.xcc_rank[xcc_id() & 15];                   // the same for every wave of a workgroup
    const bool wave_on = (int)(threadIdx.x >> 6) < a.active_waves; // tiny problems run fewer waves
    double lsum = 0.0;
#ifdef MFX_STAMPS
    unsigned long long tk0 = 0, tk1 = 0, ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, tsb = 0, c_burst = 0;
    unsigned long long c_task = 0, c_wait = 0, c_win = 0, c_rest = 0, n_steps = 0, n_tasks = 0, c_total = 0, tstart = 0;
    unsigned long long rstart = 0, rend = 0;
    STAMP(tstart);
    RSTAMP(rstart);
#endif
    if (rank >= 0)
        for (int slot = rank; slot < a.ns; slot += a.n_xcc) {
            const long long tbeg = a.slot_task_ptr[slot];
            const int ntask = (int)(a.slot_task_ptr[slot + 1] - tbeg);
            // the block's stripes: owner stripe = slot, gathered stripe = (slot + round) mod ns
            const int gs = (slot + a.round) % a.ns;
            const int ofirst = a.own_begin[slot], on_rows = a.own_begin[slot + 1] - ofirst;
            const int gfirst = a.gat_begin[gs], gn_rows = a.gat_begin[gs + 1] - gfirst;
            const __amdgpu_buffer_rsrc_t rs_rows = make_rsrc(a.gat_rows + (size_t)gfirst * ka, (unsigned)gn_rows * (unsigned)(ka * 4));
            const __amdgpu_buffer_rsrc_t rs_acc = make_rsrc(a.gat_acc + (size_t)gfirst * 2, (unsigned)gn_rows * 8u);
            // A task costs a chain of dependent misses before its first step: claim (atomic), descriptor,
            // entries, rows.  Only the first task of a wave pays it in full: while a task runs, its
            // first three steps claim the NEXT one, read its descriptor and fetch its first block of
            // entries into registers (nb0/nb1), each one step apart and right behind a wait.
            auto claim = [&]() { // -> VGPR, lane 0's value is the claimed index
                int c = 0;
                if (lane == 0) c = atomicAdd(&a.slot_cursor[slot], 1);
                return c;
            };
            auto fetch_first = [&](unsigned long long off, int nsteps_, EntryD &b0, EntryD &b1) {
                const EntryD *const base = a.entries + off;
                const int n = nsteps_ * G;
                b0 = base[lane < n ? lane : n - 1];
                b1 = base[lane + 64 < n ? lane + 64 : n - 1];
            };
            auto uniform_off = [&](const TaskDescD &d) {
                return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(d.off >> 32)) << 32) |
                       (unsigned)__builtin_amdgcn_readfirstlane((int)(d.off & 0xFFFFFFFFu));
            };
            // ---- workgroup tasks: the heavy rows of the block (plan.hpp "workgroup tasks") ----
            // Claimed by whole workgroups before any wave task.  The W waves x G lane groups of the workgroup advance through
            // the ratings of ONE heavy row side by side; the row and its two accumulator slots live in LDS for the length
            // of the visit and every list updates them with LDS float atomics -- one copy, no lost update: the sequential
            // meaning of the order up to the W x G ratings in flight.  The other side of these ratings is read-modified-
            // written through the XCD's L2 exactly as in a wave task.  With the roles swapped (heavy row of the GATHERED side)
            // the same code runs on the other side's pointers, descriptors and lambda.
            STAMP(tk1);
            {
                const long long wbeg = a.slot_wg_ptr[slot];
                const int nwg = (int)(a.slot_wg_ptr[slot + 1] - wbeg);
                const int wv = (int)(threadIdx.x >> 6);
                while (nwg > 0) {
                    __syncthreads(); // (wg_first is reused)
                    if (threadIdx.x == 0) wg_first = atomicAdd(&a.wg_cursor[slot], 1);
                    __syncthreads();
                    const int wt = __builtin_amdgcn_readfirstlane(wg_first);
                    if (wt >= nwg) break;
                    const WgTaskD *const tp = a.wg_tasks + wbeg + wt;
                    const unsigned long long toff = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(tp->off >> 32)) << 32) |
                                                    (unsigned)__builtin_amdgcn_readfirstlane((int)(tp->off & 0xFFFFFFFFu));
                    const int tsteps = __builtin_amdgcn_readfirstlane((int)tp->nsteps);
                    const int visit0 = __builtin_amdgcn_readfirstlane((int)tp->visit0);
                    const int nvisits = __builtin_amdgcn_readfirstlane((int)tp->nvisits);
                    const bool sw = __builtin_amdgcn_readfirstlane((int)tp->swapped) != 0;
                    float *const o_rows = sw ? a.gat_rows : a.own_rows, *const o_acc = sw ? a.gat_acc : a.own_acc;
                    // the side that is read-modified-written: the gathered stripe, or (roles swapped) the owner stripe
                    const __amdgpu_buffer_rsrc_t w_rows = sw ? make_rsrc(a.own_rows + (size_t)ofirst * ka, (unsigned)on_rows * (unsigned)(ka * 4)) : rs_rows;
                    const __amdgpu_buffer_rsrc_t w_acc = sw ? make_rsrc(a.own_acc + (size_t)ofirst * 2, (unsigned)on_rows * 8u) : rs_acc;
                    const int wfirst = sw ? ofirst : gfirst;
                    const float wl_o = sw ? lam_g : lam_o, wl_g = sw ? lam_o : lam_g;
                    const EntryD *const ebase = a.entries + toff + (size_t)wv * (size_t)tsteps * G;
                    const int nent = tsteps * G;
                    auto fetch_block = [&](int blk, EntryD &r0, EntryD &r1) {
                        const int i0 = blk * EBLK + lane, i1 = i0 + 64;
                        r0 = ebase[i0 < nent ? i0 : nent - 1];
                        r1 = ebase[i1 < nent ? i1 : nent - 1];
                    };
                    auto park_block = [&](int blk, const EntryD &r0, const EntryD &r1) {
                        const int i0 = (blk * EBLK + lane) & (2 * EBLK - 1);
                        ering[i0] = u4{r0.own, (unsigned)r0.gat, __builtin_bit_cast(unsigned, r0.r), 0u};
                        ering[i0 + 64] = u4{r1.own, (unsigned)r1.gat, __builtin_bit_cast(unsigned, r1.r), 0u};
                    };
                    auto entry_of = [&](int step) {
                        const u4 v = ering[(step * G + grp) & (2 * EBLK - 1)];
                        const unsigned rbits = v.z;
                        return EntryD{v.x, (int)v.y, __builtin_bit_cast(float, rbits)};
                    };
                    EntryD r0 = {0u, -1, 0.0f}, r1 = {0u, -1, 0.0f}, e = {0u, -1, 0.0f};
                    unsigned grow = BUF_OOB, gacc = BUF_OOB;
                    f4 gn = zero4;
                    f2 ggn = {1.0f, 1.0f};
                    if (wave_on) {
                        fetch_block(0, r0, r1);
                        park_block(0, r0, r1);
                        e = entry_of(0);
                        grow = e.gat >= 0 && lane_ok ? (unsigned)((e.gat & GAT_ID) - wfirst) * (unsigned)(ka * 4) + d0 * 4 : BUF_OOB;
                        gacc = e.gat >= 0 ? (unsigned)((e.gat & GAT_ID) - wfirst) * 8u : BUF_OOB;
                        gn = bld_row(w_rows, grow);
                        ggn = bld_acc(w_acc, gacc);
                    }
                    float tsum = 0.0f;
                    int step = 0;
                    // Visit records and rows are fetched one visit ahead: a record read at the top of its visit would cost a
                    // memory round trip before the first step, the row another one.
                    auto visit_rec = [&](int v, unsigned &row, int &vsteps, unsigned &vlen, unsigned &vinfo, unsigned &vslot) {
                        const WgVisitD *const vp = a.wg_visits + visit0 + (v < nvisits ? v : nvisits - 1);
                        row = (unsigned)__builtin_amdgcn_readfirstlane((int)vp->row);
                        vsteps = __builtin_amdgcn_readfirstlane((int)vp->nsteps);
                        vlen = (unsigned)__builtin_amdgcn_readfirstlane((int)vp->len);
                        vinfo = (unsigned)__builtin_amdgcn_readfirstlane((int)vp->info);
                        vslot = (unsigned)__builtin_amdgcn_readfirstlane((int)vp->slot);
                    };
                    const bool stager = wv == 0 && grp == 0; // the lanes that move the heavy row between memory and LDS
                    unsigned row, vlen, vinfo, vslot, row_n, vlen_n, vinfo_n, vslot_n;
                    int vsteps, vsteps_n;
                    visit_rec(0, row, vsteps, vlen, vinfo, vslot);
                    f4 x0 = zero4, xn = zero4;   // the state the visit started from (what its change is measured against) / the next one's
                    f2 g0 = {1.0f, 1.0f}, g0n = {1.0f, 1.0f};
                    __syncthreads(); // (the task before is through with both buffers)
                    if (stager) {
                        if (lane_ok) {
                            x0 = ld_row(o_rows + (size_t)row * ka + d0);
                            lrow4_all[0][lig] = x0;
                        }
                        if (lig == 0) {
                            g0 = ld_acc(o_acc + (size_t)row * 2);
                            lacc_all[0][0] = g0.x;
                            lacc_all[0][1] = g0.y;
                            lE_all[0] = 0.0f;
                            lg_init[0] = SLOW ? g0.x : fminf(g0.x, g0.y);
                        }
                    }
                    __syncthreads();
                    for (int v = 0; v < nvisits; ++v) {
                        const int cur = v & 1;
                        f4 *const lrow4 = lrow4_all[cur];
                        float *const lacc = lacc_all[cur];
                        visit_rec(v + 1, row_n, vsteps_n, vlen_n, vinfo_n, vslot_n);
                        if (stager && v + 1 < nvisits) { // the next visit's row, in flight while this visit runs
                            if (lane_ok) xn = ld_row(o_rows + (size_t)row_n * ka + d0);
                            if (lig == 0) g0n = ld_acc(o_acc + (size_t)row_n * 2);
                        }
                        const float tsum0 = tsum;
                        auto wg_step = [&]() {
                                const EntryD enext = entry_of(step + 1);
                                const bool act = e.gat >= 0;
                                const float rating = e.r;
                                f4 g = gn;
                                f2 gg = ggn;
                                const unsigned grow_c = grow, gacc_c = gacc;
                                asm volatile("" : "+v"(g.x), "+v"(g.y), "+v"(g.z), "+v"(g.w), "+v"(gg.x), "+v"(gg.y));
                                const int sib = step & (EBLK / G - 1);
                                const int blk1 = step / (EBLK / G) + 1;
                                if (sib == 0 && blk1 * EBLK < nent) fetch_block(blk1, r0, r1);
                                if (sib == EBLK / G - 2 && blk1 * EBLK < nent) park_block(blk1, r0, r1);
                                // the heavy row as it is NOW (other lists keep adding to it)
                                const f4 o = lane_ok ? lrow4[lig] : zero4;
                                const float og0 = lacc[0], og1 = lacc[1];
                                f2 o01 = {o.x, o.y}, o23 = {o.z, o.w}, g01 = {g.x, g.y}, g23 = {g.z, g.w};
                                const f2 zz = o01 * g01 + o23 * g23;
                                const float z = group_sum<LANES>(zz.x + zz.y);
                                const float err = act ? rating - z : 0.0f;
                                tsum += err * err;
                                const float eta_o = eta * __builtin_amdgcn_rsqf(slot1 ? og1 : og0);
                                const float eta_g = eta * __builtin_amdgcn_rsqf(slot1 ? gg.y : gg.x);
                                float so = 0.0f, sg = 0.0f;
                                f2 d01 = {0.0f, 0.0f}, d23 = {0.0f, 0.0f};
                                const bool move = upd && act;
                                // The G lists of the wave step the heavy row from the same snapshot and their steps are summed.  Where
                                // the lists pull the same way (one pair repeated over and over, or simply the direction all rows have
                                // in common) the sum overshoots what G ratings one after the other would do: with S = sum over the
                                // lists of step size x curvature, a sequential pass contracts by exp(-S) where the sum contracts by S.
                                // The summed step is scaled by (1 - exp(-S)) / S -- the fold's 1-D model, one wave-step at a time; 1
                                // to first order while S is small (it is, except in the first steps of a run and for large eta:
                                // eta = 0.2 overflowed without it).
                                float wdamp;
                                {
                                    const f2 q2 = g01 * g01 + g23 * g23;
                                    const float S = cross_group_sum<LANES>(group_sum<LANES>(move ? eta_o * (q2.x + q2.y) : 0.0f));
                                    wdamp = S > 1e-3f ? (1.0f - __expf(-S)) / S : 1.0f - 0.5f * S;
                                }
                                if (move) {
                                    const f2 go01 = wl_o * o01 - err * g01, go23 = wl_o * o23 - err * g23;
                                    const f2 gq01 = wl_g * g01 - err * o01, gq23 = wl_g * g23 - err * o23;
                                    const f2 so2 = go01 * go01 + go23 * go23, sg2 = gq01 * gq01 + gq23 * gq23;
                                    so = so2.x + so2.y;
                                    sg = sg2.x + sg2.y;
                                    d01 = -(eta_o * wdamp) * go01;
                                    d23 = -(eta_o * wdamp) * go23;
                                    g01 -= eta_g * gq01;
                                    g23 -= eta_g * gq23;
                                    g = f4{g01.x, g01.y, g23.x, g23.y};
                                }
                                const bool nact = enext.gat >= 0 && step + 1 < tsteps;
                                grow = nact && lane_ok ? (unsigned)((enext.gat & GAT_ID) - wfirst) * (unsigned)(ka * 4) + d0 * 4 : BUF_OOB;
                                gacc = nact ? (unsigned)((enext.gat & GAT_ID) - wfirst) * 8u : BUF_OOB;
                                float sg0 = group_sum<LANES>(slot1 ? 0.0f : sg), sg1 = 0.0f;
                                if (!SLOW) sg1 = group_sum<LANES>(slot1 ? sg : 0.0f);
                                // All lists of the wave on the SAME row of the other side (the synthetic streams repeat a pair --
                                // heavy user, heavy item -- thousands of times; sorted, those ratings fill list after list): every
                                // list read the row before any of them wrote it, so the row would keep ONE of their steps and
                                // one rating's accumulator growth -- steps that stay too large on a row that is hit that often
                                // (eta = 0.2: the row oscillates until it overflows).  The lists' changes are summed across the
                                // wave instead, row and accumulators, and every list stores the same result.
                                {
                                    const unsigned long long m_act = __ballot(act);
                                    if (m_act != 0ull && G > 1) {
                                        const int gid = __builtin_amdgcn_readlane(e.gat, (int)__builtin_ctzll(m_act));
                                        const bool same = __ballot(act && e.gat != gid) == 0ull && __builtin_popcountll(m_act) > LANES;
                                        if (same) {
                                            const f4 gold = gn; // (what every list loaded)
                                            g.x = gold.x + cross_group_sum<LANES>(act ? g.x - gold.x : 0.0f);
                                            g.y = gold.y + cross_group_sum<LANES>(act ? g.y - gold.y : 0.0f);
                                            g.z = gold.z + cross_group_sum<LANES>(act ? g.z - gold.z : 0.0f);
                                            g.w = gold.w + cross_group_sum<LANES>(act ? g.w - gold.w : 0.0f);
                                            sg0 = cross_group_sum<LANES>(sg0);
                                            if (!SLOW) sg1 = cross_group_sum<LANES>(sg1);
                                        }
                                    }
                                }
                                const bool ro = (e.gat & GAT_RO) != 0; // the other row of a pair of two heavy rows: read, never written here
                                bst_row(w_rows, ro ? BUF_OOB : grow_c, g);
                                gg.x = gg.x + sg0 * rk0;
                                if (!SLOW) gg.y = gg.y + sg1 * rk1;
                                bst_acc(w_acc, ro ? BUF_OOB : gacc_c, gg);
                                gn = bld_row(w_rows, grow);
                                ggn = bld_acc(w_acc, gacc);
                                // The heavy row's own update.  The G lists of this wave all hold the SAME row: what they change is
                                // summed across the lane groups in registers first (DPP / ds_bpermute), and one group adds the
                                // total with LDS float atomics (ds_add_f32) -- the other waves of the workgroup add theirs the
                                // same way, so no list's change is lost.  (Every group adding for itself is G lanes on every
                                // address: the LDS runs those one after the other, and a step took four times as long.)
                                {
                                    const float t0 = cross_group_sum<LANES>(d01.x), t1 = cross_group_sum<LANES>(d01.y);
                                    const float t2 = cross_group_sum<LANES>(d23.x), t3 = cross_group_sum<LANES>(d23.y);
                                    if (grp == 0 && upd) {
                                        float *const lr = (float *)lrow4 + d0;
                                        lds_add(lr + 0, t0);
                                        lds_add(lr + 1, t1);
                                        lds_add(lr + 2, t2);
                                        lds_add(lr + 3, t3);
                                    }
                                }
                                const float so0 = cross_group_sum<LANES>(group_sum<LANES>(slot1 ? 0.0f : so));
                                if (lane == 0) lds_add(&lacc[0], so0 * rk0);
                                if (!SLOW) {
                                    const float so1 = cross_group_sum<LANES>(group_sum<LANES>(slot1 ? so : 0.0f));
                                    if (lane == 0) lds_add(&lacc[1], so1 * rk1);
                                }
                                e = enext;
                        };
                        // While the row's accumulators are still small (epoch 0: G = 1, step size eta itself) the FIRST step of a
                        // visit is taken by the waves in turn.  Side by side, W x G lists would all read the row as the visit found
                        // it and add their steps to it at once; that sum overshoots -- 128 simultaneous ratings at k = 8 move the
                        // row 4 x beyond its optimum and the visit never recovers.  After one step in turn the accumulators hold
                        // W x G ratings' growth and the steps are small enough to add.  The bound is where eta/sqrt(G) x (a row
                        // norm of 4) x (W x G ratings) reaches 1; every heavy row is past it after an epoch or two.
                        const float gmin = lg_init[cur];
                        const float gneed = eta * 4.0f * (float)(a.active_waves * G);
                        const bool in_turn = __builtin_amdgcn_readfirstlane((int)(gmin < gneed * gneed)) != 0;
                        if (in_turn) {
                            for (int tw = 0; tw < a.active_waves; ++tw) {
                                if (wv == tw) wg_step();
                                __syncthreads();
                            }
                            ++step;
                        }
                        if (wave_on)
                            for (int s = in_turn ? 1 : 0; s < vsteps; ++s, ++step) wg_step();
                        else
                            step += vsteps - (in_turn ? 1 : 0);
                        if (wave_on && lig == 0) lds_add(&lE_all[cur], tsum - tsum0);
                        if (stager && v + 1 < nvisits) { // stage the next visit's row in the other buffer (its last user is long done)
                            if (lane_ok) lrow4_all[cur ^ 1][lig] = xn;
                            if (lig == 0) {
                                lacc_all[cur ^ 1][0] = g0n.x;
                                lacc_all[cur ^ 1][1] = g0n.y;
                                lE_all[cur ^ 1] = 0.0f;
                                lg_init[cur ^ 1] = SLOW ? g0n.x : fminf(g0n.x, g0n.y);
                            }
                        }
                        __syncthreads(); // every list's adds have landed; the next visit's row is staged
                        if (stager) {
                            // One workgroup held ALL of the row's ratings of this block: the copy is the row, written back like the
                            // owner row of a wave task.  A row split over several workgroups: every copy adds what it CHANGED
                            // (end state minus the state it started from -- sums of end states cancel catastrophically for a
                            // row of many copies), its squared errors, its ratings and 1 to the row's combine slot with
                            // fire-and-forget float atomics; fold_hot_rows, launched behind the round, folds the copies.
                            // (The other waves are already in the next visit, on the other buffer.)
                            const unsigned ncop = vinfo >> 1;
                            if (ncop <= 1 && !a.merge_back) {
                                if (lane_ok) *(f4 *)(o_rows + (size_t)row * ka + d0) = lrow4[lig];
                                if (lig == 0) *(f2 *)(o_acc + (size_t)row * 2) = f2{lacc[0], lacc[1]};
                            } else if (ncop <= 1) { // (roles swapped somewhere in the plan: other visits may have touched the row meanwhile)
                                float *const dst = o_rows + (size_t)row * ka;
                                if (lane_ok) {
                                    const f4 x1 = lrow4[lig];
                                    unsafeAtomicAdd(dst + d0 + 0, x1.x - x0.x);
                                    unsafeAtomicAdd(dst + d0 + 1, x1.y - x0.y);
                                    unsafeAtomicAdd(dst + d0 + 2, x1.z - x0.z);
                                    unsafeAtomicAdd(dst + d0 + 3, x1.w - x0.w);
                                }
                                if (lig == 0) {
                                    unsafeAtomicAdd(o_acc + (size_t)row * 2, lacc[0] - g0.x);
                                    unsafeAtomicAdd(o_acc + (size_t)row * 2 + 1, lacc[1] - g0.y);
                                }
                            } else {
                                float *const dst = a.hot_acc + ((size_t)vslot * HOT_SUB + (size_t)(wt & (HOT_SUB - 1))) * (size_t)(ka + HOT_EXTRA);
                                if (lane_ok) {
                                    const f4 x1 = lrow4[lig];
                                    unsafeAtomicAdd(dst + d0 + 0, x1.x - x0.x);
                                    unsafeAtomicAdd(dst + d0 + 1, x1.y - x0.y);
                                    unsafeAtomicAdd(dst + d0 + 2, x1.z - x0.z);
                                    unsafeAtomicAdd(dst + d0 + 3, x1.w - x0.w);
                                }
                                if (lig == 0) {
                                    unsafeAtomicAdd(dst + ka, lacc[0] - g0.x);
                                    unsafeAtomicAdd(dst + ka + 1, lacc[1] - g0.y);
                                    unsafeAtomicAdd(dst + ka + 2, lE_all[cur]);  // squared errors of this copy
                                    unsafeAtomicAdd(dst + ka + 3, (float)vlen);  // its ratings
                                    unsafeAtomicAdd(dst + ka + 4, 1.0f);         // one more copy
                                }
                            }
                        }
                        x0 = xn;
                        g0 = g0n;
                        row = row_n;
                        vsteps = vsteps_n;
                        vlen = vlen_n;
                        vinfo = vinfo_n;
                        vslot = vslot_n;
                    }
                    if (wave_on && lig == 0) lsum += (double)tsum;
                }
            }
            STAMP(tk0);
#ifdef MFX_STAMPS
            c_burst += tk0 - tk1; // (diagnostic) cycles this wave spent in the workgroup-task phase
#endif
            // The first claim of a launch is made once per workgroup, not once per wave: every wave
            // of the XCD asks at the same moment, and atomics on one address take ~70 cycles each.
            __syncthreads(); // (wg_first is reused per slot)
            if (threadIdx.x == 0) wg_first = atomicAdd(&a.slot_cursor[slot], a.active_waves);
            __syncthreads();
            if (!wave_on) continue;
            int c = __builtin_amdgcn_readfirstlane(wg_first) + (int)(threadIdx.x >> 6);
            // ---- L2 warm-up ----
            // The L2 is invalidated between launches, so the first touch of every row is a miss to
            // MALL/HBM, and the first steps of a wave are chains of such misses (measured: the
            // first 16 steps of a task cost 3.3x the later ones, 20 % of a launch).  When the two
            // stripes of the block fit the L2, every wave first streams its share of them with
            // independent, wide loads: one memory latency instead of a dozen in a row.
            if (a.warm && c < a.waves_per_xcd) {
                f4 sink = zero4;
                auto stream_in = [&](const float *base, size_t floats) {
                    for (size_t off = (size_t)c * 256 + (size_t)lane * 4; off < floats; off += (size_t)a.waves_per_xcd * 256) {
                        const f4 v = *(const f4 *)(base + off);
                        sink.x += v.x; sink.y += v.y; sink.z += v.z; sink.w += v.w;
                    }
                };
                const size_t of = (size_t)ofirst, on_ = (size_t)on_rows, gf = (size_t)gfirst, gn_ = (size_t)gn_rows;
                stream_in(a.gat_rows + gf * ka, gn_ * ka & ~(size_t)3);
                stream_in(a.own_rows + of * ka, on_ * ka & ~(size_t)3);
                stream_in(a.gat_acc + gf * 2, gn_ * 2 & ~(size_t)3);
                stream_in(a.own_acc + of * 2, on_ * 2 & ~(size_t)3);
                asm volatile("" ::"v"(sink.x), "v"(sink.y), "v"(sink.z), "v"(sink.w)); // keep the loads
            }
            if (c >= ntask) continue;
            // the descriptor is the same for every lane: keep it in SGPRs, so that the step loop
            // below branches on scalars (real branches, no exec-masked loop exits)
            unsigned long long toff;
            int nsteps, trole; // (role 1: the lists of this task are visits of rows of the GATHERED side, roles swapped)
            EntryD nb0, nb1;
            {
                const TaskDescD td = a.tasks[tbeg + c];
                toff = uniform_off(td);
                nsteps = __builtin_amdgcn_readfirstlane((int)td.nsteps);
                trole = __builtin_amdgcn_readfirstlane((int)td.pad);
                fetch_first(toff, nsteps, nb0, nb1);
            }
            // the other role's descriptors: the owner stripe is the one that is read-modified-written
            const __amdgpu_buffer_rsrc_t rs_rows_sw = make_rsrc(a.own_rows + (size_t)ofirst * ka, (unsigned)on_rows * (unsigned)(ka * 4));
            const __amdgpu_buffer_rsrc_t rs_acc_sw = make_rsrc(a.own_acc + (size_t)ofirst * 2, (unsigned)on_rows * 8u);
            for (;;) {
                const bool tsw = trole != 0;
                float *const t_own_rows = tsw ? a.gat_rows : a.own_rows, *const t_own_acc = tsw ? a.gat_acc : a.own_acc;
                const __amdgpu_buffer_rsrc_t t_rows = tsw ? rs_rows_sw : rs_rows, t_acc = tsw ? rs_acc_sw : rs_acc;
                const int t_first = tsw ? ofirst : gfirst;
                const float t_lam_o = tsw ? lam_g : lam_o, t_lam_g = tsw ? lam_o : lam_g;
                int cn_v = 0, cn = ntask; // the next task: claim in flight / claimed index
                // The claim goes out eight steps before the task ends: late enough that the waves of an
                // XCD, which start their tasks together, do not all ask at once (atomics on one address
                // take ~70 cycles each, 14k cycles for 196 waves, and the wait behind the claim would
                // cover them), early enough for descriptor and entries to arrive under the last steps;
                // and the waves that run ahead are the ones that get what is left.
                const int claim_at = nsteps > 12 ? nsteps - 8 : 0;
                TaskDescD tdn_v = {0, 0, 0};
                unsigned long long toff_n = 0;
                int nsteps_n = 0, trole_n = 0;

                unsigned cur = NONE;  // owner row held in registers
                f4 o = zero4;
                float og0 = 1.0f, og1 = 1.0f;
                unsigned pf = NONE;   // owner row in flight for the next visit
                f4 on = zero4;
                f2 ogn = {1.0f, 1.0f};
                float tsum = 0.0f;

                // ---- the entry stream ----
                // The entries of a task are streamed once from HBM, so their loads are the slow ones
                // (a microsecond and more under load) and, vmcnt being in-order, the first wait
                // behind such a load waits for it.  One entry load per step therefore costs one HBM
                // latency per step.  Instead the wave fetches EBLK entries (EBLK/G steps) with two
                // loads per lane and parks them in its own LDS ring of two such blocks; the steps
                // take their entries from LDS (lgkmcnt, a queue of its own).  That leaves one slow
                // load per block, issued right behind a wait so that it has a whole step to arrive.
                const EntryD *const ebase = a.entries + toff;
                const int nent = nsteps * G;
                auto fetch_block = [&](int blk, EntryD &r0, EntryD &r1) { // lane -> its two entries of a block
                    const int i0 = blk * EBLK + lane, i1 = i0 + 64;
                    r0 = ebase[i0 < nent ? i0 : nent - 1];
                    r1 = ebase[i1 < nent ? i1 : nent - 1];
                };
                auto park_block = [&](int blk, const EntryD &r0, const EntryD &r1) {
                    const int i0 = (blk * EBLK + lane) & (2 * EBLK - 1);
                    ering[i0] = u4{r0.own, (unsigned)r0.gat, __builtin_bit_cast(unsigned, r0.r), 0u};
                    ering[i0 + 64] = u4{r1.own, (unsigned)r1.gat, __builtin_bit_cast(unsigned, r1.r), 0u};
                };
                auto entry_of = [&](int step) { // this lane group's entry of a step
#ifdef MFX_ENTRY_GLOBAL // experiment/bisect: straight from memory
                    return ebase[(step < nsteps ? step : nsteps - 1) * G + grp];
#else
                    const u4 v = ering[(step * G + grp) & (2 * EBLK - 1)];
                    const unsigned rbits = v.z; // (bit_cast straight from the swizzle v.z takes element 0 with this hipcc)
                    return EntryD{v.x, (int)v.y, __builtin_bit_cast(float, rbits)};
#endif
                };
                EntryD r0 = nb0, r1 = nb1; // block 0 was fetched while the task before ran
                park_block(0, r0, r1);
                EntryD e = entry_of(0);
                if (e.gat != -1) { // every list starts with a visit: fetch its owner row now
                    pf = e.own & IDMASK;
                    if (lane_ok) on = ld_row(t_own_rows + (size_t)pf * ka + d0);
                    ogn = ld_acc(t_own_acc + (size_t)pf * 2);
                }
                // Every memory operation of a step goes out in ONE burst at the end of its update
                // window, in this order: row store, accumulator store, the NEXT step's gathered row
                // and accumulators, the next visit's owner row.  The wait at the top of the next step
                // then covers exactly that burst, and the owner update of a step runs under the L2
                // round trip of the next one.  The time between a row's load and its store (the
                // window in which a concurrent update of the same row is lost) keeps its length; it
                // only starts earlier.
                // The gathered side is addressed through buffer descriptors over the block's gathered
                // stripe (32-bit offsets); pad entries and lanes past k_a use an offset beyond it --
                // the range check loads zeros and drops the store -- so every access is unconditional.
                unsigned grow = e.gat >= 0 && lane_ok ? (unsigned)((e.gat & GAT_ID) - t_first) * (unsigned)(ka * 4) + d0 * 4 : BUF_OOB;
                unsigned gacc = e.gat >= 0 ? (unsigned)((e.gat & GAT_ID) - t_first) * 8u : BUF_OOB;
                f4 gn = bld_row(t_rows, grow);
                f2 ggn = bld_acc(t_acc, gacc);
                STAMP(tk1);
#ifdef MFX_STAMPS
                c_task += tk1 - tk0;
                n_tasks++;
#endif
                // Plans that run heavy rows of the gathered side with the roles swapped (mfx_options.swap_heavy): those visits
                // read-modify-write owner rows while a wave task holds them in registers.  The visit then writes back what
                // memory holds NOW (om, fetched one step before the visit ends) plus what it changed, instead of its copy.
                f4 o_start = zero4, om = zero4;
                f2 og_start = {1.0f, 1.0f}, ogm = {1.0f, 1.0f};
                auto close_visit = [&]() {
#ifdef MFX_OWNER_LDS
                    o = own_tile[threadIdx.x];
#endif
                    if (a.merge_back) {
                        if (lane_ok) *(f4 *)(t_own_rows + (size_t)cur * ka + d0) = om + (o - o_start);
                        if (lig == 0) *(f2 *)(t_own_acc + (size_t)cur * 2) = f2{ogm.x + (og0 - og_start.x), ogm.y + (og1 - og_start.y)};
                        return;
                    }
                    if (lane_ok) *(f4 *)(t_own_rows + (size_t)cur * ka + d0) = o;
                    if (lig == 0) *(f2 *)(t_own_acc + (size_t)cur * 2) = f2{og0, og1};
                };
                for (int step = 0; step < nsteps; ++step) {
                    STAMP(ts0);
                    // (read under the wait below; stale past the end of the list, see nact; the
                    //  block it may belong to was parked one step ago)
                    const EntryD enext = entry_of(step + 1);
                    const bool act = e.gat >= 0;
                    const unsigned id = e.own & IDMASK;
                    const float rating = e.r;
                    const bool ro = (e.gat & GAT_RO) != 0; // (plan.hpp ENTRY_READ_ONLY)
                    const bool newvisit = act && (e.own >> 31) && id != cur;
                    if (newvisit) { // switch the owner row: write the old one back, take the prefetched one
                        if (cur != NONE) close_visit();
                        if (pf != id) { // not prefetched (cannot happen for lists built by plan.cpp)
                            if (lane_ok) on = ld_row(t_own_rows + (size_t)id * ka + d0);
                            ogn = ld_acc(t_own_acc + (size_t)id * 2);
                        }
                        o = on;
#ifdef MFX_OWNER_LDS
                        own_tile[threadIdx.x] = o; // experiment: the owner row of the visit staged in LDS, not held in registers
#endif
                        og0 = ogn.x;
                        og1 = ogn.y;
                        o_start = on;
                        og_start = ogn;
                        cur = id;
                    }
                    f4 g = gn;
                    f2 gg = ggn;
                    const unsigned grow_c = grow, gacc_c = gacc;
                    // One explicit settle point for the loads this step consumes.  Without it hipcc
                    // re-waits with vmcnt(0) at later uses of these registers, i.e. behind the
                    // stores below, which costs a full store round trip per step.
                    asm volatile("" : "+v"(g.x), "+v"(g.y), "+v"(g.z), "+v"(g.w), "+v"(gg.x), "+v"(gg.y));
                    if ((unsigned)(step - claim_at) < 3u) { // the next task, one dependent access per step (scalar branches)
                        if (step == claim_at) {
                            cn_v = claim();
                        } else if (step == claim_at + 1) {
                            cn = __builtin_amdgcn_readfirstlane(cn_v);
                            if (cn < ntask) tdn_v = a.tasks[tbeg + cn];
                        } else if (cn < ntask) {
                            toff_n = uniform_off(tdn_v);
                            nsteps_n = __builtin_amdgcn_readfirstlane((int)tdn_v.nsteps);
                            trole_n = __builtin_amdgcn_readfirstlane((int)tdn_v.pad);
                            fetch_first(toff_n, nsteps_n, nb0, nb1);
                        }
                    }
                    // block bookkeeping, on scalars: fetch the next block at the start of a block
                    // (right behind the wait above), park it two steps before it is needed
                    const int sib = step & (EBLK / G - 1); // step within its block
                    const int blk1 = step / (EBLK / G) + 1;
                    if (sib == 0 && blk1 * EBLK < nent) fetch_block(blk1, r0, r1);
                    if (sib == EBLK / G - 2 && blk1 * EBLK < nent) park_block(blk1, r0, r1);

                    STAMP(ts1);
                    // ---- compute: z = p.q (calc_z), err = r - z (prepare_for_sg_update) ----
                    // (written on float pairs: one v_pk_* instruction per two factors)
#ifdef MFX_OWNER_LDS
                    o = own_tile[threadIdx.x];
#endif
                    f2 o01 = {o.x, o.y}, o23 = {o.z, o.w}, g01 = {g.x, g.y}, g23 = {g.z, g.w};
                    const f2 zz = o01 * g01 + o23 * g23;
                    const float z = group_sum<LANES>(zz.x + zz.y);
                    const float err = act ? rating - z : 0.0f;
                    tsum += err * err;

                    // sg_update: eta scaled by rsqrt of the slot this lane's dims belong to
                    const float eta_o = eta * __builtin_amdgcn_rsqf(slot1 ? og1 : og0);
                    const float eta_g = eta * __builtin_amdgcn_rsqf(slot1 ? gg.y : gg.x);
                    float so = 0.0f, sg = 0.0f;
                    if (upd && act) {
                        // both gradients use the OLD values of the other side
                        const f2 go01 = t_lam_o * o01 - err * g01, go23 = t_lam_o * o23 - err * g23;
                        const f2 gq01 = t_lam_g * g01 - err * o01, gq23 = t_lam_g * g23 - err * o23;
                        const f2 so2 = go01 * go01 + go23 * go23, sg2 = gq01 * gq01 + gq23 * gq23;
                        so = so2.x + so2.y;
                        sg = sg2.x + sg2.y;
                        o01 -= eta_o * go01;
                        o23 -= eta_o * go23;
                        g01 -= eta_g * gq01;
                        g23 -= eta_g * gq23;
                        o = f4{o01.x, o01.y, o23.x, o23.y};
                        g = f4{g01.x, g01.y, g23.x, g23.y};
#ifdef MFX_OWNER_LDS
                        own_tile[threadIdx.x] = o;
#endif
                    }
                    // ---- one burst of memory operations ----
                    const bool nact = enext.gat >= 0 && step + 1 < nsteps;
                    grow = nact && lane_ok ? (unsigned)((enext.gat & GAT_ID) - t_first) * (unsigned)(ka * 4) + d0 * 4 : BUF_OOB;
                    gacc = nact ? (unsigned)((enext.gat & GAT_ID) - t_first) * 8u : BUF_OOB;
                    const unsigned id1 = enext.own & IDMASK;
                    auto owner_prefetch = [&]() {
                        const bool switches = nact && (enext.own >> 31) && id1 != cur; // a visit starts at the next step
                        if (switches) {
                            pf = id1;
                            if (lane_ok) on = ld_row(t_own_rows + (size_t)id1 * ka + d0);
                            ogn = ld_acc(t_own_acc + (size_t)id1 * 2);
                        }
                        if (a.merge_back && cur != NONE && (switches || step + 1 >= nsteps || !nact)) { // the visit in progress ends
                            if (lane_ok) om = ld_row(t_own_rows + (size_t)cur * ka + d0);
                            ogm = ld_acc(t_own_acc + (size_t)cur * 2);
                        }
                    };
                    auto next_loads = [&]() {
                        gn = bld_row(t_rows, grow);
                        ggn = bld_acc(t_acc, gacc);
                    };
                    auto acc_store = [&]() {
                        const float sg0 = group_sum<LANES>(slot1 ? 0.0f : sg);
                        gg.x = gg.x + sg0 * rk0;
                        if (!SLOW) {
                            const float sg1 = group_sum<LANES>(slot1 ? sg : 0.0f);
                            gg.y = gg.y + sg1 * rk1;
                        }
                        bst_acc(t_acc, ro ? BUF_OOB : gacc_c, gg); // every lane of the group writes the same pair
                    };
                    STAMP(tsb);
                    // the gathered row goes back first: the time between its load and this store is the
                    // window in which another wave's update of the same row is lost
                    bst_row(t_rows, ro ? BUF_OOB : grow_c, g);
                    // accumulator store, then the next step's loads: the wait at the top of the next step
                    // covers the whole burst.  (Sending the loads ahead of the accumulator store, with the
                    // wait leaving that store in flight, paid for wide rows before the buffer addressing and
                    // stopped paying with it: profiles/experiments/r01_burst_order.log.)  Stores before loads
                    // also means a duplicate rating next in the list sees this step's result.
                    acc_store();
                    next_loads();
                    owner_prefetch();
                    e = enext;
                    STAMP(ts2);
                    const float so0 = group_sum<LANES>(slot1 ? 0.0f : so);
                    og0 = og0 + so0 * rk0;
                    if (!SLOW) {
                        const float so1 = group_sum<LANES>(slot1 ? so : 0.0f);
                        og1 = og1 + so1 * rk1;
                    }
                    STAMP(ts3);
#ifdef MFX_STAMPS
                    c_wait += ts1 - ts0;
                    c_win += ts2 - ts1;
                    c_rest += ts3 - ts2;
                    n_steps++;
#endif
                }
                if (cur != NONE) close_visit();
                if (lig == 0) lsum += (double)tsum;
                // a task that ended before the hand-over was through does the rest now
                STAMP(tk0);
                if (nsteps < claim_at + 1) cn_v = claim();
                if (nsteps < claim_at + 2) {
                    cn = __builtin_amdgcn_readfirstlane(cn_v);
                    if (cn < ntask) tdn_v = a.tasks[tbeg + cn];
                }
                if (nsteps < claim_at + 3 && cn < ntask) {
                    toff_n = uniform_off(tdn_v);
                    nsteps_n = __builtin_amdgcn_readfirstlane((int)tdn_v.nsteps);
                    trole_n = __builtin_amdgcn_readfirstlane((int)tdn_v.pad);
                    fetch_first(toff_n, nsteps_n, nb0, nb1);
                }
                if (cn >= ntask) break;
                c = cn;
                toff = toff_n;
                nsteps = nsteps_n;
                trole = trole_n;
            }
        }

#ifdef MFX_STAMPS
    {
        unsigned long long tend;
        STAMP(tend);
        RSTAMP(rend);
        c_total = tend - tstart;
        if (lane == 0 && a.stamps) {
            unsigned long long *o = a.stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
            o[0] += c_task; o[1] += c_wait; o[2] += c_win; o[3] += c_rest; o[4] += n_steps; o[5] += n_tasks; o[6] += c_total; o[7] += 1;
            // timeline of the latest launch: start, end, XCC id, steps of every wave
            unsigned long long *tl = a.stamps + (size_t)65536 * 8 + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
            tl[0] = rstart; tl[1] = rend; tl[2] = (unsigned long long)(xcc_id() & 15) + 1; tl[3] = n_steps | (c_burst << 20);
        }
    }
#endif
    // online loss (Scheduler::get_loss, mf.cpp:237-241): wave -> workgroup through LDS, then ONE
    // double atomic per workgroup, spread over LOSS_SLOTS addresses (every wave adding to one
    // word serialises ~1500 atomics at the end of a short launch)
    __shared__ double wg_loss[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off);
    if (lane == 0) wg_loss[threadIdx.x >> 6] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = wg_loss[0] + wg_loss[1] + wg_loss[2] + wg_loss[3];
        if (s != 0.0) atomicAdd(a.loss + (blockIdx.x % LOSS_SLOTS), s);
    }
}

// ---- rows wider than one float4 per lane (256 < k_a <= 1024) --------------------------------------------------------
// The same update on the same plan, one rating per wavefront step (G = 1) and V float4 per lane (factors (v*64 + lane)*4 ..
// of chunk v: a chunk is one contiguous kilobyte across the wave).  The plan of such a problem has wave tasks only -- a heavy
// row is one long list, the roles are not swapped (plan.cpp: PlanConfig::wide_rows) -- and this kernel is the step loop
// without its latency hiding (no entry ring, no hand-over under the last steps): correct first, the reference's k in use
// is 8 .. 128.  Same arithmetic as sgd_round: gradients from the OLD values, accumulator slots [0,8) / [8,k_a), rk = 1/8 for
// both (quirk Q1), v_rsq_f32, epoch 0 moves the first eight factors only.
template <int V, bool SLOW>
__global__ __launch_bounds__(256) void sgd_round_wide(RoundArgs a)
{
    constexpr unsigned NONE = 0xFFFFFFFFu, IDMASK = 0x3FFFFFFFu;
    const int lane = threadIdx.x & 63;
    const int ka = a.ka;
    const float lam_o = a.lambda_own, lam_g = a.lambda_gat, eta = a.eta;
    const float rk0 = 0.125f, rk1 = a.rk1;
    const f4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int rank = a.xcc_rank[xcc_id() & 15];
    const bool wave_on = (int)(threadIdx.x >> 6) < a.active_waves;
    double lsum = 0.0;
    bool ok[V], s1[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int d0 = (v * 64 + lane) * 4;
        ok[v] = d0 < ka;
        s1[v] = d0 >= 8;
    }
    if (rank >= 0 && wave_on)
        for (int slot = rank; slot < a.ns; slot += a.n_xcc) {
            const long long tbeg = a.slot_task_ptr[slot];
            const int ntask = (int)(a.slot_task_ptr[slot + 1] - tbeg);
            for (;;) {
                int c = 0;
                if (lane == 0) c = atomicAdd(&a.slot_cursor[slot], 1);
                c = __builtin_amdgcn_readfirstlane(c);
                if (c >= ntask) break;
                const TaskDescD td = a.tasks[tbeg + c];
                const EntryD *const ebase = a.entries + td.off;
                const int nsteps = (int)td.nsteps;
                unsigned cur = NONE;
                f4 o[V];
#pragma unroll
                for (int v = 0; v < V; ++v) o[v] = zero4;
                float og0 = 1.0f, og1 = 1.0f, tsum = 0.0f;
                auto write_back = [&]() {
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        if (ok[v]) *(f4 *)(a.own_rows + (size_t)cur * ka + (v * 64 + lane) * 4) = o[v];
                    if (lane == 0) *(f2 *)(a.own_acc + (size_t)cur * 2) = f2{og0, og1};
                };
                for (int step = 0; step < nsteps; ++step) {
                    const EntryD e = ebase[step];
                    if (e.gat < 0) continue; // padding
                    const unsigned id = e.own & IDMASK;
                    if (id != cur) { // a visit starts: the row of the one before goes back
                        if (cur != NONE) write_back();
#pragma unroll
                        for (int v = 0; v < V; ++v) o[v] = ok[v] ? ld_row(a.own_rows + (size_t)id * ka + (v * 64 + lane) * 4) : zero4;
                        const f2 og = ld_acc(a.own_acc + (size_t)id * 2);
                        og0 = og.x;
                        og1 = og.y;
                        cur = id;
                    }
                    float *const gp = a.gat_rows + (size_t)(e.gat & 0x3FFFFFFF) * ka;
                    float *const gap = a.gat_acc + (size_t)(e.gat & 0x3FFFFFFF) * 2;
                    f4 g[V];
                    float z = 0.0f;
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        g[v] = ok[v] ? ld_row(gp + (v * 64 + lane) * 4) : zero4;
                        z += o[v].x * g[v].x + o[v].y * g[v].y + o[v].z * g[v].z + o[v].w * g[v].w;
                    }
                    f2 gg = ld_acc(gap);
                    z = group_sum<64>(z);
                    const float err = e.r - z;
                    tsum += err * err;
                    const float eo0 = eta * __builtin_amdgcn_rsqf(og0), eo1 = eta * __builtin_amdgcn_rsqf(og1);
                    const float eg0 = eta * __builtin_amdgcn_rsqf(gg.x), eg1 = eta * __builtin_amdgcn_rsqf(gg.y);
                    float so0 = 0.0f, so1 = 0.0f, sg0 = 0.0f, sg1 = 0.0f;
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        if (!ok[v] || (SLOW && s1[v])) continue;
                        const f4 go = lam_o * o[v] - err * g[v], gq = lam_g * g[v] - err * o[v];
                        const float so = go.x * go.x + go.y * go.y + go.z * go.z + go.w * go.w;
                        const float sg = gq.x * gq.x + gq.y * gq.y + gq.z * gq.z + gq.w * gq.w;
                        if (s1[v]) {
                            so1 += so;
                            sg1 += sg;
                            o[v] -= eo1 * go;
                            g[v] -= eg1 * gq;
                        } else {
                            so0 += so;
                            sg0 += sg;
                            o[v] -= eo0 * go;
                            g[v] -= eg0 * gq;
                        }
                    }
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        if (ok[v]) *(f4 *)(gp + (v * 64 + lane) * 4) = g[v];
                    gg.x += group_sum<64>(sg0) * rk0;
                    og0 += group_sum<64>(so0) * rk0;
                    if (!SLOW) {
                        gg.y += group_sum<64>(sg1) * rk1;
                        og1 += group_sum<64>(so1) * rk1;
                    }
                    if (lane == 0) *(f2 *)gap = gg;
                }
                if (cur != NONE) write_back();
                if (lane == 0) lsum += (double)tsum;
            }
        }
    __shared__ double wg_loss[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off);
    if (lane == 0) wg_loss[threadIdx.x >> 6] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = wg_loss[0] + wg_loss[1] + wg_loss[2] + wg_loss[3];
        if (s != 0.0) atomicAdd(a.loss + (blockIdx.x % LOSS_SLOTS), s);
    }
}

// sum over the plan's ratings of (r - p.q)^2 in scaled units, rows of any width (one rating per wavefront)
__global__ __launch_bounds__(256) void sq_err_entries_wide(const float *own_rows, const float *gat_rows, const EntryD *entries,
                                                           long long n_entries, int ka, double *out)
{
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    double lsum = 0.0;
    for (long long i = wave; i < n_entries; i += nwaves) {
        const EntryD e = entries[i];
        if (e.gat < 0) continue;
        const float *const o = own_rows + (size_t)(e.own & 0x3FFFFFFFu) * ka, *const g = gat_rows + (size_t)(e.gat & 0x3FFFFFFF) * ka;
        float z = 0.0f;
        for (int d = lane * 4; d < ka; d += 256) {
            const f4 x = *(const f4 *)(o + d), y = *(const f4 *)(g + d);
            z += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
        z = group_sum<64>(z);
        const float err = e.r - z;
        if (lane == 0) lsum += (double)(err * err);
    }
    if (lane == 0 && lsum != 0.0) atomicAdd(out, lsum);
}

// Fold the copies of the rows that are split over several workgroups into their rows (launched behind every round that
// holds such a row; one wave per combine slot).  A row with more ratings in a block than one workgroup does in a launch
// is worked on by n workgroups, each on its own LDS copy that starts from the row p0 of before the launch; every copy adds
// what it changed -- row, both accumulator slots, squared errors, ratings -- to the row's combine slot.  Here the
// accumulators grow by ALL copies' growth and the row moves by the copies' summed change times a damping factor, a 1-D
// model of what one sequential pass over all n parts does (DESIGN.md "Heavy rows"): with S = sum over the ratings of
// (step size x curvature), a sequential pass moves the row (1 - exp(-S_seq)) of the way to the block's optimum, one copy
// (1 - exp(-S_copy)); the sum of the n copies is therefore scaled by (1 - exp(-S_seq)) / (n (1 - exp(-S_copy))): exactly 1
// for one copy, the plain sum (first-order equivalence) while S is small, the mean of the copies when every copy
// converges by itself.  Step sizes come from the Adagrad accumulators (sequentially G runs from G0 to G0 + A, in a copy
// to G0 + A/n: mean step 2 eta / (sqrt(G_end) + sqrt(G0))), the curvature |q|^2 from the accumulator growth over the
// squared errors (A = rk * sum e^2 |q|^2): nothing extra is computed per rating, and NO constant is fitted (rounds 1-2
// cut such rows into hundreds of 128-rating chains and needed a calibrated gain on S; with a handful of long copies the
// result no longer depends on the rule: the mean of the copies lands within 0.2 % of it, oracle/plan_order.c).
// hot_row[slot] = internal row | side << 31 (1: a row of the plan's gathered side, whose visits ran with the roles swapped).
__global__ __launch_bounds__(256) void fold_hot_rows(float *own_rows, float *own_acc, float *gat_rows, float *gat_acc, float *hot_acc,
                                                     const int *hot_row, int n_slots, int ka, float eta, float rk1, int slow_only)
{
    const int lane = threadIdx.x & 63;
    const int slot_i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot_i >= n_slots) return;
    const int stride = ka + HOT_EXTRA;
    float *const s0 = hot_acc + (size_t)slot_i * HOT_SUB * stride;
    float ex[HOT_EXTRA] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f}; // growth of both accumulator slots, squared errors, ratings, copies
    if (lane < HOT_EXTRA)
        for (int sub = 0; sub < HOT_SUB; ++sub) ex[0] += s0[(size_t)sub * stride + ka + lane];
    const float v = ex[0];
#pragma unroll
    for (int j = 0; j < HOT_EXTRA; ++j) ex[j] = __shfl(v, j);
    const float n = ex[4];
    if (!(n > 0.0f)) return; // the row was not split in this round
    const unsigned hr = (unsigned)hot_row[slot_i];
    const int row = (int)(hr & 0x7FFFFFFFu);
    float *const rows = (hr >> 31) ? gat_rows : own_rows, *const acc = (hr >> 31) ? gat_acc : own_acc;
    const float g00 = acc[(size_t)row * 2], g01 = acc[(size_t)row * 2 + 1];
    const float A0 = fmaxf(ex[0], 0.0f), A1 = fmaxf(ex[1], 0.0f), E = ex[2], N = ex[3];
    const float rn = 1.0f / n;
    const float r00 = __builtin_sqrtf(g00), r01 = __builtin_sqrtf(g01);
    // mean step sizes (/ 2 eta) of a sequential pass and of a copy, per accumulator slot
    const float ts0 = 1.0f / (__builtin_sqrtf(g00 + A0) + r00), ts1 = 1.0f / (__builtin_sqrtf(g01 + A1) + r01);
    const float tc0 = 1.0f / (__builtin_sqrtf(g00 + A0 * rn) + r00), tc1 = 1.0f / (__builtin_sqrtf(g01 + A1 * rn) + r01);
    const float cq = E > 0.0f ? 2.0f * eta * N / E : 0.0f; // sum of |q|^2 over the ratings, per slot: N * A / (rk * E)
    const float c0 = cq * A0 * 8.0f, c1 = cq * A1 / rk1;
    const float Sseq = ts0 * c0 + ts1 * c1, Sch = (tc0 * c0 + tc1 * c1) * rn;
    auto damp = [](float S) { return S > 1e-3f ? (1.0f - __expf(-S)) / S : 1.0f - 0.5f * S; };
    const float phi = damp(Sseq) / damp(Sch); // (1-exp(-S)) / (n (1-exp(-S/n))) in the symmetric case
    const float sc0 = phi * ts0 / tc0, sc1 = phi * ts1 / tc1;
    for (int d = lane; d < (slow_only ? 8 : ka); d += 64) { // (epoch 0 moves the first eight factors only)
        float sum = 0.0f;
        for (int sub = 0; sub < HOT_SUB; ++sub) sum += s0[(size_t)sub * stride + d];
        rows[(size_t)row * ka + d] += (d >= 8 ? sc1 : sc0) * sum;
    }
    if (lane == 0) {
        acc[(size_t)row * 2] = g00 + A0;
        acc[(size_t)row * 2 + 1] = g01 + A1;
    }
    // leave zeros for the next round
    for (int i = lane; i < HOT_SUB * stride; i += 64) s0[i] = 0.0f;
}

// Self-test of what the lock-free gathered side relies on: inside one XCD, a row stored by one CU (plain store,
// write-through to the XCD's L2) is seen by the non-temporal loads (global nt and raw buffer aux = 2, the two forms
// sgd_round uses) of ANOTHER CU -- they must not be served from that CU's own, stale L1 line.  The first two
// workgroups that arrive on XCC 0 play ping-pong: A writes a row and raises a flag, B polls the flag with nt loads,
// reads the row with both load forms, answers; every wait is bounded, nothing can hang.
// out: [0] rounds completed, [1] stale rows seen, [2] polls that ran out, [3] CU of A, [4] CU of B
__global__ __launch_bounds__(64) void visibility_probe(int *ticket, float *row, int *flag, int *ack, int rounds, int *out)
{
    if ((xcc_id() & 15) != 0) return;
    __shared__ int role_s;
    if (threadIdx.x == 0) role_s = atomicAdd(ticket, 1);
    __syncthreads();
    const int role = role_s;
    if (role > 1) return;
    const int lane = threadIdx.x;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (lane == 0) out[3 + role] = (int)((hw >> 8) & 0xF) | (int)(((hw >> 13) & 0x7) << 4); // CU id | SE id << 4
    const int SPIN = 400000;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(row, 64u * 4u);
    int stale = 0, lost = 0, done = 0;
    for (int i = 1; i <= rounds; ++i) {
        if (role == 0) {
            row[lane] = (float)i;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) *flag = i;
            int spins = 0;
            while (__builtin_nontemporal_load(ack) != i && ++spins < SPIN) asm volatile("" ::: "memory"); // (reload every time)
            if (spins >= SPIN) { ++lost; break; }
        } else {
            int spins = 0;
            while (__builtin_nontemporal_load(flag) != i && ++spins < SPIN) asm volatile("" ::: "memory");
            if (spins >= SPIN) { ++lost; break; }
            const float a = __builtin_nontemporal_load(row + lane);
            const float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, 0, BUF_NT));
            if (a != (float)i || b != (float)i) ++stale;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) *ack = i;
        }
        ++done;
    }
    for (int off = 32; off > 0; off >>= 1) stale += __shfl_down(stale, off);
    if (lane == 0) {
        if (role == 1) { out[0] = done; atomicAdd(&out[1], stale); }
        atomicAdd(&out[2], lost);
    }
}

// Which XCC ids does a grid land on?  One bit per id seen (run once per trainer).
__global__ void probe_xcc(unsigned *mask)
{
    if (threadIdx.x == 0) atomicOr(mask, 1u << (xcc_id() & 15));
}

// ---- metrics ------------------------------------------------------------------------

// sum over the plan's ratings of (r - p.q)^2 in scaled units
template <int LANES>
__global__ __launch_bounds__(256) void sq_err_entries(const float *own_rows, const float *gat_rows,
                                                      const EntryD *entries, long long n_entries,
                                                      int ka, double *out)
{
    constexpr int G = 64 / LANES;
    const int lane = threadIdx.x & 63;
    const int lig = lane % LANES, grp = lane / LANES;
    const int d0 = lig * 4;
    const bool lane_ok = d0 < ka;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    double lsum = 0.0;
    for (long long base = wave * G; base < n_entries; base += nwaves * G) {
        long long i = base + grp;
        float z = 0.0f, r = 0.0f;
        bool act = false;
        if (i < n_entries) {
            EntryD e = entries[i];
            act = e.gat >= 0;
            if (act && lane_ok) {
                const bool sw = (e.own & 0x40000000u) != 0; // a heavy row of the gathered side: `own` indexes that side
                f4 o = *(const f4 *)((sw ? gat_rows : own_rows) + (size_t)(e.own & 0x3FFFFFFFu) * ka + d0);
                f4 g = *(const f4 *)((sw ? own_rows : gat_rows) + (size_t)(e.gat & 0x3FFFFFFF) * ka + d0);
                z = o.x * g.x + o.y * g.y + o.z * g.z + o.w * g.w;
            }
            r = e.r;
        }
        z = group_sum<LANES>(z);
        if (act && lig == 0) {
            float err = r - z;
            lsum += (double)(err * err);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off);
    if (lane == 0 && lsum != 0.0) atomicAdd(out, lsum);
}

// calc_reg2 (mf.cpp:608-633): sum_i omega[i] * |row_i|^2 for one side
__global__ __launch_bounds__(256) void reg2_side(const float *rows, const int *omega, int nrows,
                                                 int ka, double *out)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nth = (long long)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (long long i = tid; i < nrows; i += nth) {
        int w = omega[i];
        if (w <= 0) continue;
        const float *r = rows + (size_t)i * ka;
        float s = 0.0f;
        for (int d = 0; d < ka; ++d) s += r[d] * r[d];
        acc += (double)((float)w * s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0 && acc != 0.0) atomicAdd(out, acc);
}

// mf_predict (mf.cpp:4295-4314) over (u,v) float pairs against a facade array on the
// device: out of range -> b, NaN -> b.  One 16-lane group per pair.
__global__ __launch_bounds__(256) void predict_pairs(const float *model, int m, int n, int k,
                                                     float b, const float *pairs,
                                                     long long npairs, float *out)
{
    const int lane = threadIdx.x & 63, lig = lane & 15, grp = lane >> 4;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const float *P = model + 5, *Q = model + 5 + (size_t)m * k;
    for (long long base = wave * 4; base < npairs; base += nwaves * 4) {
        long long i = base + grp;
        float z = 0.0f;
        bool in = false, valid = i < npairs;
        if (valid) {
            int u = (int)pairs[2 * i], v = (int)pairs[2 * i + 1];
            in = u >= 0 && u < m && v >= 0 && v < n;
            if (in) {
                const float *p = P + (size_t)u * k, *q = Q + (size_t)v * k;
                for (int d = lig; d < k; d += 16) z += p[d] * q[d];
            }
        }
        z = group_sum<16>(z);
        if (valid && lig == 0) out[i] = (in && z == z) ? z : b;
    }
}

// calc_rmse (mf.cpp:4316-4331) of a facade array over mf_node ratings
__global__ __launch_bounds__(256) void sq_err_nodes(const float *model, int m, int n, int k, float b,
                                                    const EntryD *R, long long nnz, double *out)
{
    const int lane = threadIdx.x & 63, lig = lane & 15, grp = lane >> 4;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const float *P = model + 5, *Q = model + 5 + (size_t)m * k;
    double lsum = 0.0;
    for (long long base = wave * 4; base < nnz; base += nwaves * 4) {
        long long i = base + grp;
        float z = 0.0f, r = 0.0f;
        bool in = false, valid = i < nnz;
        if (valid) {
            int u = (int)R[i].own, v = R[i].gat; // mf_node {u, v, r} viewed through EntryD
            r = R[i].r;
            in = u >= 0 && u < m && v >= 0 && v < n;
            if (in) {
                const float *p = P + (size_t)u * k, *q = Q + (size_t)v * k;
                for (int d = lig; d < k; d += 16) z += p[d] * q[d];
            }
        }
        z = group_sum<16>(z);
        if (valid && lig == 0) {
            float pred = (in && z == z) ? z : b;
            float err = r - pred;
            lsum += (double)(err * err);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off);
    if (lane == 0 && lsum != 0.0) atomicAdd(out, lsum);
}

// scale_model + shrink_model + shuffle_model (mf.cpp:529-553, 1057-1074, 1027-1055):
// out[orig][d] = rows[map[orig]][d] * f for d < k
__global__ __launch_bounds__(256) void export_side(const float *rows, const int *map, int nrows,
                                                   int k, int ka, float f, int do_scale, float *out)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)nrows * k;
    const long long nth = (long long)gridDim.x * blockDim.x;
    for (long long i = tid; i < total; i += nth) {
        long long row = i / k;
        int d = (int)(i - row * k);
        float x = rows[(size_t)map[row] * ka + d];
        out[i] = do_scale ? x * f : x;
    }
}

// Start of an epoch: every task of the epoch before must have been handed out (cursor >= task count per
// block) -- if not, a sticky flag is raised that no later reset clears -- then the loss sums and the
// cursors are zeroed.  One launch in place of the memset that used to do only the second half.
__global__ __launch_bounds__(256) void epoch_reset(double *loss, int *cursor, const long long *slot_task_ptr,
                                                  const long long *slot_wg_ptr, int nb, int check, int *sticky)
{
    for (int i = threadIdx.x; i < nb; i += 256) { // wave-task cursors [0, nb), workgroup-task cursors [nb, 2 nb)
        if (check && (cursor[i] < (int)(slot_task_ptr[i + 1] - slot_task_ptr[i]) ||
                      cursor[nb + i] < (int)(slot_wg_ptr[i + 1] - slot_wg_ptr[i])))
            atomicOr(sticky, 1);
        cursor[i] = 0;
        cursor[nb + i] = 0;
    }
    for (int i = threadIdx.x; i < LOSS_SLOTS; i += 256) loss[i] = 0.0;
}

__global__ void fill_f32(float *p, long long n, float v)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nth = (long long)gridDim.x * blockDim.x;
    for (long long i = tid; i < n; i += nth) p[i] = v;
}

} // namespace mfx

#include "synth.hpp"

namespace mfx {

__global__ __launch_bounds__(256) void synth_kernel(unsigned long long seed,
                                                    unsigned long long shard, long long first,
                                                    long long count, int m, int n, SynthNode *out)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nth = (long long)gridDim.x * blockDim.x;
    for (long long i = tid; i < count; i += nth) out[i] = synth_rating(seed, shard, first + i, m, n);
}

// ---- launchers (the only symbols the host code sees) ---------------------------------

template <int LANES>
static void launch_round_t(const RoundArgs &a, int grid, hipStream_t s)
{
    const bool full = a.ka == LANES * 4;
    if (full && a.slow_only)
        hipLaunchKernelGGL((sgd_round<LANES, true, true>), dim3(grid), dim3(256), 0, s, a);
    else if (full)
        hipLaunchKernelGGL((sgd_round<LANES, true, false>), dim3(grid), dim3(256), 0, s, a);
    else if (a.slow_only)
        hipLaunchKernelGGL((sgd_round<LANES, false, true>), dim3(grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((sgd_round<LANES, false, false>), dim3(grid), dim3(256), 0, s, a);
}

template <int V>
static void launch_round_wide_t(const RoundArgs &a, int grid, hipStream_t s)
{
    if (a.slow_only)
        hipLaunchKernelGGL((sgd_round_wide<V, true>), dim3(grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((sgd_round_wide<V, false>), dim3(grid), dim3(256), 0, s, a);
}

hipError_t launch_sgd_round(int lanes, const RoundArgs &a, int grid, hipStream_t s)
{
    if (a.ka > 256) { // V float4 per lane (plans of such problems hold wave tasks only)
        if (lanes != 64 || a.ka > 1024) return hipErrorInvalidValue;
        switch ((a.ka + 255) / 256) {
        case 2: launch_round_wide_t<2>(a, grid, s); break;
        case 3: launch_round_wide_t<3>(a, grid, s); break;
        default: launch_round_wide_t<4>(a, grid, s); break;
        }
        return hipGetLastError();
    }
    switch (lanes) {
    case 2: launch_round_t<2>(a, grid, s); break;
    case 4: launch_round_t<4>(a, grid, s); break;
    case 8: launch_round_t<8>(a, grid, s); break;
    case 16: launch_round_t<16>(a, grid, s); break;
    case 32: launch_round_t<32>(a, grid, s); break;
    case 64: launch_round_t<64>(a, grid, s); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_fold_hot(float *own_rows, float *own_acc, float *gat_rows, float *gat_acc, float *hot_acc, const int *hot_row,
                           int n_slots, int ka, float eta, float rk1, int slow_only, hipStream_t s)
{
    if (n_slots <= 0) return hipSuccess;
    hipLaunchKernelGGL(fold_hot_rows, dim3((n_slots + 3) / 4), dim3(256), 0, s, own_rows, own_acc, gat_rows, gat_acc, hot_acc,
                       hot_row, n_slots, ka, eta, rk1, slow_only);
    return hipGetLastError();
}

hipError_t launch_visibility_probe(int *ticket, float *row, int *flag, int *ack, int rounds, int *out, int grid, hipStream_t s)
{
    hipLaunchKernelGGL(visibility_probe, dim3(grid), dim3(64), 0, s, ticket, row, flag, ack, rounds, out);
    return hipGetLastError();
}

hipError_t launch_probe_xcc(unsigned *mask, int grid, hipStream_t s)
{
    hipLaunchKernelGGL(probe_xcc, dim3(grid), dim3(64), 0, s, mask);
    return hipGetLastError();
}

hipError_t launch_sq_err_entries(int lanes, const float *own_rows, const float *gat_rows,
                                 const EntryD *entries, long long n_entries, int ka, double *out,
                                 int grid, hipStream_t s)
{
#define MFX_CASE(L)                                                                             \
    case L:                                                                                     \
        hipLaunchKernelGGL(sq_err_entries<L>, dim3(grid), dim3(256), 0, s, own_rows, gat_rows,  \
                           entries, n_entries, ka, out);                                        \
        break;
    if (ka > 256) {
        hipLaunchKernelGGL(sq_err_entries_wide, dim3(grid), dim3(256), 0, s, own_rows, gat_rows, entries, n_entries, ka, out);
        return hipGetLastError();
    }
    switch (lanes) {
        MFX_CASE(2) MFX_CASE(4) MFX_CASE(8) MFX_CASE(16) MFX_CASE(32) MFX_CASE(64)
    default: return hipErrorInvalidValue;
    }
#undef MFX_CASE
    return hipGetLastError();
}

hipError_t launch_reg2(const float *rows, const int *omega, int nrows, int ka, double *out,
                       int grid, hipStream_t s)
{
    hipLaunchKernelGGL(reg2_side, dim3(grid), dim3(256), 0, s, rows, omega, nrows, ka, out);
    return hipGetLastError();
}

hipError_t launch_predict(const float *model, int m, int n, int k, float b, const float *pairs,
                          long long npairs, float *out, int grid, hipStream_t s)
{
    hipLaunchKernelGGL(predict_pairs, dim3(grid), dim3(256), 0, s, model, m, n, k, b, pairs, npairs,
                       out);
    return hipGetLastError();
}

hipError_t launch_sq_err_nodes(const float *model, int m, int n, int k, float b, const EntryD *R,
                               long long nnz, double *out, int grid, hipStream_t s)
{
    hipLaunchKernelGGL(sq_err_nodes, dim3(grid), dim3(256), 0, s, model, m, n, k, b, R, nnz, out);
    return hipGetLastError();
}

hipError_t launch_export(const float *rows, const int *map, int nrows, int k, int ka, float f,
                         int do_scale, float *out, int grid, hipStream_t s)
{
    hipLaunchKernelGGL(export_side, dim3(grid), dim3(256), 0, s, rows, map, nrows, k, ka, f,
                       do_scale, out);
    return hipGetLastError();
}

hipError_t launch_epoch_reset(double *loss, int *cursor, const long long *slot_task_ptr, const long long *slot_wg_ptr, int nb,
                              int check, int *sticky, hipStream_t s)
{
    hipLaunchKernelGGL(epoch_reset, dim3(1), dim3(256), 0, s, loss, cursor, slot_task_ptr, slot_wg_ptr, nb, check, sticky);
    return hipGetLastError();
}

hipError_t launch_fill(float *p, long long n, float v, int grid, hipStream_t s)
{
    hipLaunchKernelGGL(fill_f32, dim3(grid), dim3(256), 0, s, p, n, v);
    return hipGetLastError();
}

// read_triplet (reference mf/mf.cpp:3367-3394) on the device: (u, v, r) float triples -> nodes, with
// m = max u + 1, n = max v + 1 and a flag for negative ids
__global__ __launch_bounds__(256) void triplets_kernel(const float *tri, long long count, SynthNode *out, int *mn_bad)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nth = (long long)gridDim.x * blockDim.x;
    int mx = 0, nx = 0, bad = 0;
    for (long long j = tid; j < count; j += nth) {
        SynthNode nd;
        nd.u = (int)tri[3 * j];
        nd.v = (int)tri[3 * j + 1];
        nd.r = tri[3 * j + 2];
        bad |= (nd.u < 0) | (nd.v < 0);
        mx = nd.u + 1 > mx ? nd.u + 1 : mx;
        nx = nd.v + 1 > nx ? nd.v + 1 : nx;
        out[j] = nd;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int a = __shfl_down(mx, off), b = __shfl_down(nx, off);
        mx = a > mx ? a : mx;
        nx = b > nx ? b : nx;
        bad |= __shfl_down(bad, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&mn_bad[0], mx);
        atomicMax(&mn_bad[1], nx);
        if (bad) atomicOr(&mn_bad[2], 1);
    }
}

hipError_t launch_triplets(const float *tri, long long count, void *out, int *mn_bad, int grid, hipStream_t s)
{
    hipLaunchKernelGGL(triplets_kernel, dim3(grid), dim3(256), 0, s, tri, count, (SynthNode *)out, mn_bad);
    return hipGetLastError();
}

hipError_t launch_synth(unsigned long long seed, unsigned long long shard, long long first,
                        long long count, int m, int n, void *out, int grid, hipStream_t s)
{
    hipLaunchKernelGGL(synth_kernel, dim3(grid), dim3(256), 0, s, seed, shard, first, count, m, n,
                       (SynthNode *)out);
    return hipGetLastError();
}

} // namespace mfx
