// prep.hip -- pre-processing of a training problem on the device.
//
// GPU version of the reference's fpsg() prologue (reference mf/mf.cpp:2972-3016): collect_info
// (462-484), shuffle_problem (775-791), scale_problem (517-527) and the bucketing + in-block
// sort of grid_problem (793-858), for ratings that are already resident in HBM.  The result is
// the same stripe/task layout plan.cpp builds on the host (tests compare the two entry for
// entry); only the visit table (one record per owner row and block, ~nnz/25) crosses PCIe, the
// packing of visits into tasks (plan.cpp: finish_plan) runs on the host, the entries are written
// by a kernel.
//
//   ratings --key_build--> (block|role|primary|secondary) keys, scaled r, row counts
//           (primary = owner id, or -- role "swapped" -- the id of a heavy gathered row: plan_hot_gathered)
//           --radix sort (hipCUB)--> block-major, primary-major order
//           --run-length encode on (block|role|primary)--> visit table --D2H--> finish_plan (host)
//           --H2D placements--> emit_entries --> entries[] in HBM
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "kernels.hpp"
#include "plan.hpp"
#include "prep.hpp"

namespace mfx {

namespace {

constexpr int ID_BITS = 24; // ids travel as floats through the facade: exact below 2^24

struct HipErr : std::runtime_error {
    explicit HipErr(const std::string &m) : std::runtime_error(m) {}
};

#define PREP_TRY(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) throw HipErr(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T> struct Buf {
    T *p = nullptr;
    void alloc(size_t n)
    {
        if (n == 0) n = 1;
        PREP_TRY(hipMalloc((void **)&p, n * sizeof(T)));
    }
    ~Buf()
    {
        if (p) (void)hipFree(p);
    }
};

// sum and sum of squares in double + id range check (collect_info, mf.cpp:462-484), and the
// ratings of every row by ORIGINAL id (omega, mf.cpp:815-816; plan_maps balances the stripes with them).
// A row that holds a few per cent of all ratings would receive hundreds of thousands of atomics on one
// address (measured: 6.5 ms for 10M ratings); each workgroup therefore counts in a small LDS hash table
// first -- whoever claims a slot is counted there, everything else goes straight to memory -- and
// flushes one atomic per slot.
constexpr int CNT_SLOTS = 2048; // per side and workgroup

__device__ __forceinline__ void count_id(int id, int *keys, int *cnts, int *global)
{
    const unsigned h = ((unsigned)id * 2654435761u) >> (32 - 11); // CNT_SLOTS = 2^11
    const int old = atomicCAS(&keys[h], -1, id);
    if (old == -1 || old == id) atomicAdd(&cnts[h], 1);
    else atomicAdd(&global[id], 1);
}

__global__ __launch_bounds__(256) void stats_kernel(const Node *R, long long nnz, int m, int n,
                                                    double *sums, int *bad, int *cnt_p, int *cnt_q)
{
    __shared__ int key_p[CNT_SLOTS], key_q[CNT_SLOTS], num_p[CNT_SLOTS], num_q[CNT_SLOTS];
    for (int i = threadIdx.x; i < CNT_SLOTS; i += blockDim.x) {
        key_p[i] = key_q[i] = -1;
        num_p[i] = num_q[i] = 0;
    }
    __syncthreads();
    // contiguous chunk per workgroup, so that the table sees as many repeats as possible
    const long long per = (nnz + gridDim.x - 1) / gridDim.x;
    const long long beg = (long long)blockIdx.x * per, end = beg + per < nnz ? beg + per : nnz;
    double a = 0.0, q = 0.0;
    int b = 0;
    for (long long i = beg + threadIdx.x; i < end; i += blockDim.x) {
        const Node x = R[i];
        a += (double)x.r;
        q += (double)x.r * x.r;
        const int out = (x.u < 0) | (x.u >= m) | (x.v < 0) | (x.v >= n);
        b |= out;
        if (!out) {
            count_id(x.u, key_p, num_p, cnt_p);
            count_id(x.v, key_q, num_q, cnt_q);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off);
        q += __shfl_down(q, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sums[0], a);
        atomicAdd(&sums[1], q);
    }
    if (b) atomicOr(bad, 1);
    __syncthreads();
    for (int i = threadIdx.x; i < CNT_SLOTS; i += blockDim.x) {
        if (key_p[i] >= 0) atomicAdd(&cnt_p[key_p[i]], num_p[i]);
        if (key_q[i] >= 0) atomicAdd(&cnt_q[key_q[i]], num_q[i]);
    }
}

// relabel, scale, count rows, build the sort key (block | role | primary id | secondary id)
// bounds: own_begin[ns+1] then gat_begin[ns+1] (internal-id boundaries of the stripes)
// omega_own / omega_gat: ratings per internal row; hot_gat: heavy rows of the gathered side (plan_hot_gathered)
__global__ __launch_bounds__(256) void key_build(const Node *R, long long nnz, const int *p_map,
                                                 const int *q_map, int owner_is_q, float inv_scale,
                                                 int do_scale, const int *bounds, int ns, const int *omega_own,
                                                 const int *omega_gat, int swap_heavy,
                                                 unsigned long long *keys, float *vals)
{
    __shared__ int sb[2 * 257];
    for (int i = threadIdx.x; i < 2 * (ns + 1); i += blockDim.x) sb[i] = bounds[i];
    __syncthreads();
    const int *own_begin = sb, *gat_begin = sb + ns + 1;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nth = (long long)gridDim.x * blockDim.x;
    for (long long i = tid; i < nnz; i += nth) {
        const Node x = R[i];
        const unsigned u = (unsigned)p_map[x.u], v = (unsigned)q_map[x.v];
        const unsigned own = owner_is_q ? v : u, gat = owner_is_q ? u : v;
        const unsigned long long blk =
            (unsigned long long)stripe_of(own_begin, ns, own) * ns + stripe_of(gat_begin, ns, gat);
        const bool sw = swap_heavy && omega_gat[gat] > omega_own[own];
        keys[i] = (blk << (2 * ID_BITS + 1)) | ((unsigned long long)(sw ? 1 : 0) << (2 * ID_BITS)) |
                  ((unsigned long long)(sw ? gat : own) << ID_BITS) | (sw ? own : gat);
        vals[i] = do_scale ? x.r * inv_scale : x.r;
    }
}

struct VisitKey { // (block | role | primary) part of a sort key
    __host__ __device__ unsigned long long operator()(unsigned long long k) const { return k >> ID_BITS; }
};

__global__ __launch_bounds__(256) void fill_entries(EntryD *e, long long n)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nth = (long long)gridDim.x * blockDim.x;
    for (long long i = tid; i < n; i += nth) e[i] = EntryD{0u, -1, 0.0f};
}

// one lane per placement: entries[dst + x*G] <- sorted rating src + x*stride (plan.hpp: Placement)
// omega_own / hot_gat / heavy_thr: the read-only rule of plan.cpp (a pair of two heavy rows moves the heavier one only)
__global__ __launch_bounds__(256) void emit_entries(const Placement *pl, long long npl,
                                                    const unsigned long long *keys, const float *vals, long long nnz,
                                                    int G, const int *omega_own, const char *hot_gat, long long heavy_thr,
                                                    int swap_heavy, EntryD *entries)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nth = (long long)gridDim.x * blockDim.x;
    for (long long i = tid; i < npl; i += nth) {
        const Placement p = pl[i];
        const unsigned stride = p.stride_flags & 0xFFFFu;
        const unsigned flags = (p.stride_flags >> 31) ? ENTRY_SWAPPED : 0u;
        const bool visit_start = ((p.stride_flags >> 30) & 1u) != 0;
        for (unsigned x = 0; x < p.len; ++x) {
            const unsigned long long si = p.src + (unsigned long long)x * stride;
            const unsigned long long k = keys[si];
            EntryD e;
            e.own = (unsigned)((k >> ID_BITS) & ((1u << ID_BITS) - 1)) | flags | ((visit_start && x == 0) ? 0x80000000u : 0u);
            e.gat = (int)(k & ((1u << ID_BITS) - 1));
            // inside a long run of one pair of two heavy rows (plan.cpp, build_plan): the key holds block, role and both ids
            if (swap_heavy && (flags ? (long long)omega_own[e.gat] > heavy_thr : hot_gat[e.gat] != 0) &&
                si >= (unsigned long long)(RUN_READ_ONLY / 2) && si + RUN_READ_ONLY / 2 < (unsigned long long)nnz &&
                keys[si - RUN_READ_ONLY / 2] == k && keys[si + RUN_READ_ONLY / 2] == k)
                e.gat |= ENTRY_READ_ONLY;
            e.r = vals[p.src + (unsigned long long)x * stride];
            entries[p.dst + (unsigned long long)x * G] = e;
        }
    }
}

// ---- init_model on the device ----------------------------------------------------------------
// The reference draws the initial factors from ONE std::minstd_rand0 stream, P rows then Q rows in
// internal order, k draws per row that has ratings (mf.cpp:952-1007).  x_{i+1} = 16807 x_i mod (2^31-1)
// can be entered anywhere: x_{i+j} = 16807^j x_i, so every row starts from its own stream position
// (k times the number of seen rows before it, an exclusive scan) and the result is bit-identical.

// i = position in the reference's row order (P rows, then Q rows); at = position -> internal row (null: identity)
__device__ __forceinline__ long long row_at(long long i, int m, const int *p_at, const int *q_at)
{
    if (i < m) return p_at ? p_at[i] : i;
    return q_at ? q_at[i - m] : i - m;
}

__global__ __launch_bounds__(256) void seen_flags(const int *omega_p, int m, const int *omega_q, int n,
                                                  const int *p_at, const int *q_at, unsigned long long *flag)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nth = (long long)gridDim.x * blockDim.x, rows = (long long)m + n;
    for (long long i = tid; i < rows; i += nth) {
        const long long row = row_at(i, m, p_at, q_at);
        flag[i] = (i < m ? omega_p[row] : omega_q[row]) > 0 ? 1ull : 0ull;
    }
}

__device__ __forceinline__ unsigned minstd_mul(unsigned a, unsigned b)
{
    return (unsigned)(((unsigned long long)a * b) % 2147483647ull);
}

__global__ __launch_bounds__(256) void init_rows(const unsigned long long *seen_before, const int *omega_p, int m,
                                                 const int *omega_q, int n, const int *p_at, const int *q_at,
                                                 int k, int ka, float scale, float *P, float *Q)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nth = (long long)gridDim.x * blockDim.x, rows = (long long)m + n;
    for (long long i = tid; i < rows; i += nth) {
        const bool isP = i < m;
        const long long row = row_at(i, m, p_at, q_at);
        float *dst = (isP ? P : Q) + row * ka;
        const bool seen = (isP ? omega_p[row] : omega_q[row]) > 0;
        if (seen) {
            unsigned long long steps = seen_before[i] * (unsigned long long)k; // draws before this row
            unsigned mult = 1u, base = 16807u;                                  // 16807^steps mod (2^31-1)
            while (steps) {
                if (steps & 1ull) mult = minstd_mul(mult, base);
                base = minstd_mul(base, base);
                steps >>= 1;
            }
            unsigned x = mult; // default seed 1
            for (int d = 0; d < k; ++d) {
                x = minstd_mul(x, 16807u);
                float f = (float)(x - 1u) * 4.6566128730773926e-10f; // / 2^31, exact
                if (!(f < 1.0f)) f = 0.99999994f;                    // generate_canonical's clamp
                dst[d] = f * scale;
            }
        } else {
            for (int d = 0; d < k; ++d) dst[d] = __builtin_nanf("");
        }
        for (int d = k; d < ka; ++d) dst[d] = 0.0f;
    }
}

int grid_of(long long n, int cu) { return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, (long long)cu * 8)); }

} // namespace

void init_factors_device(const int *d_omega_p, int m, const int *d_omega_q, int n, const int *p_at_host,
                         const int *q_at_host, int k, int ka, int cu_count, hipStream_t s, float *dP, float *dQ)
{
    const long long rows = (long long)m + n;
    Buf<unsigned long long> dFlag, dScan;
    Buf<int> dPat, dQat;
    if (p_at_host) {
        dPat.alloc(m);
        PREP_TRY(hipMemcpyAsync(dPat.p, p_at_host, (size_t)m * 4, hipMemcpyHostToDevice, s));
    }
    if (q_at_host) {
        dQat.alloc(n);
        PREP_TRY(hipMemcpyAsync(dQat.p, q_at_host, (size_t)n * 4, hipMemcpyHostToDevice, s));
    }
    dFlag.alloc((size_t)rows);
    dScan.alloc((size_t)rows);
    hipLaunchKernelGGL(seen_flags, dim3(grid_of(rows, cu_count)), dim3(256), 0, s, d_omega_p, m, d_omega_q, n, dPat.p,
                       dQat.p, dFlag.p);
    size_t tmp_bytes = 0;
    PREP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, dFlag.p, dScan.p, (int)rows, s));
    Buf<char> dTmp;
    dTmp.alloc(tmp_bytes);
    PREP_TRY(hipcub::DeviceScan::ExclusiveSum(dTmp.p, tmp_bytes, dFlag.p, dScan.p, (int)rows, s));
    const float scale = (float)std::sqrt(1.0 / k); // mf.cpp:971
    hipLaunchKernelGGL(init_rows, dim3(grid_of(rows, cu_count)), dim3(256), 0, s, dScan.p, d_omega_p, m, d_omega_q,
                       n, dPat.p, dQat.p, k, ka, scale, dP, dQ);
    PREP_TRY(hipGetLastError());
    PREP_TRY(hipStreamSynchronize(s));
}

bool device_prep_supported(int m, int n) { return m <= (1 << ID_BITS) && n <= (1 << ID_BITS); }

void build_plan_device(const void *dR_v, long long nnz, int m, int n, const PlanConfig &cfg, int cu_count,
                       hipStream_t s, Plan &p, EntryD **d_entries_out)
{
    const Node *dR = (const Node *)dR_v;
    *d_entries_out = nullptr;
    plan_header(nnz, m, n, cfg, p);
    if (!device_prep_supported(m, n)) throw std::invalid_argument("ids beyond 2^24 need the host plan builder");
    int threads = cfg.threads > 0 ? cfg.threads : (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    const int NS = p.ns, NB = NS * NS, G = p.groups;
    // MFX_PLAN_TIMING=1: wall time of every phase to stderr (each phase ends in a stream synchronise when it is on)
    const bool timing = getenv("MFX_PLAN_TIMING") && atoi(getenv("MFX_PLAN_TIMING")) != 0;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "mfx plan: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };

    // 1. statistics + id validation
    Buf<double> dSums;
    Buf<int> dBad;
    dSums.alloc(2);
    dBad.alloc(1);
    PREP_TRY(hipMemsetAsync(dSums.p, 0, 2 * sizeof(double), s));
    PREP_TRY(hipMemsetAsync(dBad.p, 0, sizeof(int), s));
    Buf<int> dCntP, dCntQ;
    dCntP.alloc(m);
    dCntQ.alloc(n);
    PREP_TRY(hipMemsetAsync(dCntP.p, 0, (size_t)m * 4, s));
    PREP_TRY(hipMemsetAsync(dCntQ.p, 0, (size_t)n * 4, s));
    hipLaunchKernelGGL(stats_kernel, dim3(grid_of(nnz, cu_count)), dim3(256), 0, s, dR, nnz, m, n, dSums.p, dBad.p,
                       dCntP.p, dCntQ.p);
    double sums[2];
    int bad = 0;
    std::vector<int> cnt_p(m), cnt_q(n);
    PREP_TRY(hipMemcpyAsync(sums, dSums.p, sizeof(sums), hipMemcpyDeviceToHost, s));
    PREP_TRY(hipMemcpyAsync(&bad, dBad.p, sizeof(int), hipMemcpyDeviceToHost, s));
    PREP_TRY(hipMemcpyAsync(cnt_p.data(), dCntP.p, (size_t)m * 4, hipMemcpyDeviceToHost, s));
    PREP_TRY(hipMemcpyAsync(cnt_q.data(), dCntQ.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    PREP_TRY(hipStreamSynchronize(s));
    if (bad) throw std::invalid_argument("rating with id outside [0,m) x [0,n)");
    if (cfg.use_stats) {
        p.avg = cfg.stats_avg;
        p.std_dev = cfg.stats_std;
    } else {
        const double ex = sums[0] / (double)nnz, ex2 = sums[1] / (double)nnz;
        p.avg = (float)ex;
        p.std_dev = (float)std::sqrt(ex2 - ex * ex);
    }
    plan_scale(p);
    lap("stats + counts");
    // id maps, stripe boundaries, omega (host: the glibc-compatible shuffle is serial)
    plan_maps(cfg, p, cnt_p.data(), cnt_q.data());
    plan_hot_gathered(cfg, p);
    lap("id maps (host)");

    // 2. keys
    Buf<int> dPmap, dQmap, dBounds, dOmOwn, dOmGat;
    Buf<char> dHotGat;
    dPmap.alloc(m);
    dQmap.alloc(n);
    dBounds.alloc(2 * (NS + 1));
    PREP_TRY(hipMemcpyAsync(dPmap.p, p.p_map.data(), (size_t)m * 4, hipMemcpyHostToDevice, s));
    PREP_TRY(hipMemcpyAsync(dQmap.p, p.q_map.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    {
        const std::vector<int> &oo = p.owner_is_q ? p.omega_q : p.omega_p, &og = p.owner_is_q ? p.omega_p : p.omega_q;
        dOmOwn.alloc(oo.size());
        dOmGat.alloc(og.size());
        dHotGat.alloc(p.hot_gat.size());
        PREP_TRY(hipMemcpyAsync(dOmOwn.p, oo.data(), oo.size() * 4, hipMemcpyHostToDevice, s));
        PREP_TRY(hipMemcpyAsync(dOmGat.p, og.data(), og.size() * 4, hipMemcpyHostToDevice, s));
        PREP_TRY(hipMemcpyAsync(dHotGat.p, p.hot_gat.data(), p.hot_gat.size(), hipMemcpyHostToDevice, s));
    }
    {
        const std::vector<int> &ob = p.owner_is_q ? p.q_begin : p.p_begin, &gb = p.owner_is_q ? p.p_begin : p.q_begin;
        PREP_TRY(hipMemcpyAsync(dBounds.p, ob.data(), (size_t)(NS + 1) * 4, hipMemcpyHostToDevice, s));
        PREP_TRY(hipMemcpyAsync(dBounds.p + NS + 1, gb.data(), (size_t)(NS + 1) * 4, hipMemcpyHostToDevice, s));
    }
    Buf<unsigned long long> dKeyA, dKeyB;
    Buf<float> dValA, dValB;
    dKeyA.alloc(nnz);
    dKeyB.alloc(nnz);
    dValA.alloc(nnz);
    dValB.alloc(nnz);
    lap("maps H2D + key buffers");
    hipLaunchKernelGGL(key_build, dim3(grid_of(nnz, cu_count)), dim3(256), 0, s, dR, nnz, dPmap.p, dQmap.p,
                       p.owner_is_q ? 1 : 0, p.inv_scale, p.inv_scale != 1.0f ? 1 : 0, dBounds.p, NS, dOmOwn.p, dOmGat.p,
                       p.swap_heavy ? 1 : 0, dKeyA.p, dValA.p);

    lap("keys");
    // 3. sort by (block, owner, gathered); stable, so equal pairs keep their input order
    int blk_bits = 1;
    while ((1 << blk_bits) < NB) ++blk_bits;
    const int end_bit = 2 * ID_BITS + 1 + blk_bits;
    if (end_bit > 64) throw std::invalid_argument("too many stripes for the device builder's sort key");
    size_t tmp_bytes = 0;
    PREP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, dKeyA.p, dKeyB.p, dValA.p, dValB.p, nnz, 0, end_bit, s));
    Buf<char> dTmp;
    dTmp.alloc(tmp_bytes);
    lap("sort scratch");
    PREP_TRY(hipcub::DeviceRadixSort::SortPairs(dTmp.p, tmp_bytes, dKeyA.p, dKeyB.p, dValA.p, dValB.p, nnz, 0, end_bit, s));

    lap("radix sort");
    // 4. visit table: run-length encode the (block | owner) part of the sorted keys
    if (nnz > 2147483647LL) throw std::invalid_argument("more than 2^31-1 ratings per trainer: shard the problem");
    Buf<int> dRunLen, dRuns; // run keys go to dKeyA (free after the sort)
    dRunLen.alloc(nnz);
    dRuns.alloc(1);
    hipcub::TransformInputIterator<unsigned long long, VisitKey, const unsigned long long *> vk(dKeyB.p, VisitKey());
    size_t rle_bytes = 0;
    PREP_TRY(hipcub::DeviceRunLengthEncode::Encode(nullptr, rle_bytes, vk, dKeyA.p, dRunLen.p, dRuns.p, (int)nnz, s));
    Buf<char> dTmp2;
    dTmp2.alloc(rle_bytes);
    PREP_TRY(hipcub::DeviceRunLengthEncode::Encode(dTmp2.p, rle_bytes, vk, dKeyA.p, dRunLen.p, dRuns.p, (int)nnz, s));
    int runs = 0;
    PREP_TRY(hipMemcpyAsync(&runs, dRuns.p, sizeof(int), hipMemcpyDeviceToHost, s));
    PREP_TRY(hipStreamSynchronize(s));
    std::vector<unsigned long long> run_key((size_t)runs);
    std::vector<int> run_len((size_t)runs);
    PREP_TRY(hipMemcpyAsync(run_key.data(), dKeyA.p, (size_t)runs * 8, hipMemcpyDeviceToHost, s));
    PREP_TRY(hipMemcpyAsync(run_len.data(), dRunLen.p, (size_t)runs * 4, hipMemcpyDeviceToHost, s));
    PREP_TRY(hipStreamSynchronize(s));

    lap("run-length encode + D2H");
    // 5. host: visits per block, packing into tasks (same code as the host builder)
    std::vector<std::vector<Visit>> block_visits(NB);
    {
        // the runs are sorted by block: every block is one range of them, filled by its own thread
        // (run key = block << (ID_BITS + 1) | role << ID_BITS | primary id)
        std::vector<size_t> first((size_t)NB + 1, (size_t)runs);
        for (int b = 0; b <= NB; ++b) {
            const unsigned long long key = (unsigned long long)b << (ID_BITS + 1);
            first[b] = (size_t)(std::lower_bound(run_key.begin(), run_key.end(), key) - run_key.begin());
        }
        std::vector<uint64_t> start_of((size_t)NB + 1, 0); // sorted position of a block's first rating
        std::atomic<int> nb(0);
        auto sum = [&]() {
            for (;;) {
                const int b = nb.fetch_add(1);
                if (b >= NB) break;
                uint64_t t = 0;
                for (size_t i = first[b]; i < first[b + 1]; ++i) t += (uint64_t)run_len[i];
                start_of[b + 1] = t;
            }
        };
        auto fill = [&]() {
            for (;;) {
                const int b = nb.fetch_add(1);
                if (b >= NB) break;
                std::vector<Visit> &bv = block_visits[b];
                bv.reserve(first[b + 1] - first[b]);
                uint64_t start = start_of[b];
                for (size_t i = first[b]; i < first[b + 1]; ++i) {
                    bv.push_back({(uint32_t)(run_key[i] & ((1u << ID_BITS) - 1)), (uint32_t)run_len[i], start,
                                  (uint32_t)((run_key[i] >> ID_BITS) & 1u)});
                    start += (uint64_t)run_len[i];
                }
            }
        };
        const int nt = std::max(1, std::min(threads, NB));
        {
            std::vector<std::thread> pool;
            for (int t = 0; t < nt; ++t) pool.emplace_back(sum);
            for (auto &th : pool) th.join();
        }
        for (int b = 0; b < NB; ++b) start_of[b + 1] += start_of[b];
        nb = 0;
        {
            std::vector<std::thread> pool;
            for (int t = 0; t < nt; ++t) pool.emplace_back(fill);
            for (auto &th : pool) th.join();
        }
    }
    lap("visit table -> blocks (host)");
    PlaceVec places;
    finish_plan(block_visits, cfg, p, places, threads);
    lap("visits -> tasks (host)");

    // 6. entries on the device
    EntryD *dEntries = nullptr;
    PREP_TRY(hipMalloc((void **)&dEntries, (size_t)std::max<long long>(1, p.n_entries) * sizeof(EntryD)));
    try {
        Buf<Placement> dPlaces;
        dPlaces.alloc(places.size());
        PREP_TRY(hipMemcpyAsync(dPlaces.p, places.data(), places.size() * sizeof(Placement), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(fill_entries, dim3(grid_of(p.n_entries, cu_count)), dim3(256), 0, s, dEntries, p.n_entries);
        hipLaunchKernelGGL(emit_entries, dim3(grid_of((long long)places.size(), cu_count)), dim3(256), 0, s,
                           dPlaces.p, (long long)places.size(), dKeyB.p, dValB.p, nnz, G, dOmOwn.p, dHotGat.p, (long long)p.heavy_thr,
                           p.swap_heavy ? 1 : 0, dEntries);
        PREP_TRY(hipGetLastError());
        PREP_TRY(hipStreamSynchronize(s));
        lap("placements H2D + entries");
    } catch (...) {
        (void)hipFree(dEntries);
        throw;
    }
    *d_entries_out = dEntries;
}

} // namespace mfx
