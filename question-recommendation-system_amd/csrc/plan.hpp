// plan.hpp -- host-side pre-processing of one training problem into the layout the
// SGD kernel streams from HBM.
//
// Mirrors the reference's fpsg() prologue (reference mf/mf.cpp:2972-3016):
// collect_info -> gen_random_map -> shuffle_problem -> scale_problem -> grid_problem,
// with grid_problem's nr_bins^2 CPU blocks replaced by a stripes^2 grid of
// (owner-stripe, gather-stripe) blocks cut into wavefront tasks.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

namespace mfx {

struct Node { int u; int v; float r; }; // = mf_node (reference mf/mf.h:36-41)

// One rating as the kernel reads it.  `own` = id on the register-resident ("owner")
// side, bit 31 set when the owner row must be (re)loaded before this rating;
// `gat` = id on the gathered side, -1 for a padding slot.
struct Entry { uint32_t own; int32_t gat; float r; };
static_assert(sizeof(Entry) == 12, "Entry must stay 12 bytes (mf_node sized)");

// One wavefront task: `nsteps` steps of G = 64/lanes ratings, stored step-major
// (entry index = off + step*G + group).
struct TaskDesc { uint64_t off; uint32_t nsteps; uint32_t pad; };

// All ratings of one owner row inside one block: `len` consecutive ratings of the block-sorted
// rating array, starting at global index `start`.  A visit longer than the hot-chain length is cut into
// `nch` chains (nch = 0: an ordinary visit); `hot` is then the row's combine slot (Plan::n_hot_slots).
struct Visit { uint32_t own; uint32_t len; uint64_t start; uint32_t nch = 0; uint32_t hot = 0; uint32_t idx = 0; };

// Header entry of a hot chain (written into the entry stream just before the chain's first rating):
// entries[dst] = {own | bit 31, -(1 + (nch | idx << 15)), bits of `hot`}: nch chains of this row in this block
// (2 <= nch < 2^15), this one is number idx in the order of the sorted visit; hot = combine slot (bits 0..19) |
// chain length << 20.  The kernel gives every chain of a hot row its
// own register copy and folds the chains' changes together when the last one of the launch ends.
struct HeaderRec { uint64_t dst; uint32_t own; uint32_t nch; uint32_t hot; uint32_t idx; };

// "Put sorted ratings [src, src+len) into entries[dst + x*G], x = 0..len-1, the first one flagged."
struct Placement { uint64_t src; uint64_t dst; uint32_t len; uint32_t pad; };
// vector whose resize() leaves trivially constructible elements uninitialised: the merged placement list of a plan is
// ~100 MB at 100 M ratings and is filled by several threads, which should also be the ones to touch its pages first
template <class T> struct DefaultInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef DefaultInitAlloc<U> other; };
    template <class U> void construct(U *ptr) { ::new ((void *)ptr) U; }
    template <class U, class... A> void construct(U *ptr, A &&...a) { ::new ((void *)ptr) U(std::forward<A>(a)...); }
};
typedef std::vector<Placement, DefaultInitAlloc<Placement>> PlaceVec;

// Result of packing one block; entry offsets are relative to the block.
struct BlockPack {
    std::vector<TaskDesc> tasks;
    std::vector<Placement> places;
    std::vector<HeaderRec> headers;
    uint64_t n_entries = 0;
    long long hot = 0, padding = 0;
};

struct PlanConfig {
    int k = 8;
    int stripes = 8;          // NS
    int lanes = 2;            // lanes per rating (power of two)
    int task_steps = 0;       // 0 = auto
    int owner_side = 0;       // 0 auto, 1 users, 2 items
    int map_mode = 0;         // 0 mass-balanced stripes, 1 identity, 2 the reference's shuffle (equal-count stripes)
    // map_mode 0: balance with these counts (per ORIGINAL id) instead of the data's own, so that several
    // trainers -- the stripe trainers of one rank, the ranks of a job -- agree on the row of every id
    const int *layout_cnt_p = nullptr, *layout_cnt_q = nullptr;
    bool use_stats = false;   // take avg/std from below instead of collect_info
    float stats_avg = 0, stats_std = 0;
    int waves_per_stripe = 256; // for auto task sizing
    int threads = 0;          // host worker threads, 0 = hardware_concurrency
};

struct Plan {
    int m = 0, n = 0, k = 0, ka = 0;
    long long nnz = 0;
    float avg = 0, std_dev = 0, scale = 1, inv_scale = 1;
    bool owner_is_q = true;
    int ns = 8, lanes = 2, groups = 32;
    std::vector<int> p_map, q_map;       // original id -> internal id
    std::vector<int> p_begin, q_begin;   // ns+1 internal-id boundaries of the user / item stripes
    // Internal row that sits at position i of the REFERENCE's row order (its shuffled ids,
    // mf.cpp:1009-1017).  init_model draws the factors in that order, so every original id starts
    // from the reference's values whatever the layout.  Empty = the layout is that order.
    std::vector<int> p_at, q_at;
    std::vector<int> omega_p, omega_q;   // ratings per internal row
    std::vector<Entry> entries;
    std::vector<TaskDesc> tasks;
    std::vector<long long> slot_task_ptr; // ns*ns+1, ordered (round, slot)
    std::vector<HeaderRec> headers;       // hot-chain header entries (already part of `entries` on the host path)
    long long n_hot_slots = 0;            // distinct rows that are cut into chains somewhere
    std::vector<int> hot_rows;            // combine slot -> internal owner row
    std::vector<char> round_hot;          // ns flags: does round r hold any chain (is there anything to fold behind it)?
    long long n_entries = 0;   // entries.size() on the host path; on the device path the array lives in HBM only
    long long n_hot_rows = 0;
    long long n_padding = 0;
};

// the entry a header record stands for
inline Entry header_entry(const HeaderRec &h)
{
    Entry e;
    e.own = h.own | 0x80000000u;
    e.gat = -(int32_t)(1u + (h.nch | (h.idx << 15))); // <= -3: nch >= 2 (a pad slot is -1)
    uint32_t bits = h.hot;
    static_assert(sizeof(float) == 4, "");
    __builtin_memcpy(&e.r, &bits, 4);
    return e;
}

// pieces shared by the host builder (build_plan) and the device builder (prep.hip)
void plan_header(long long nnz, int m, int n, const PlanConfig &cfg, Plan &p);
void plan_scale(Plan &p);                                  // scale from std_dev
// id permutations and stripe boundaries; cnt_* = ratings per ORIGINAL id (needed for map_mode 0);
// also fills omega_p / omega_q (ratings per internal row)
void plan_maps(const PlanConfig &cfg, Plan &p, const int *cnt_p, const int *cnt_q);
// stripe of an internal id: largest s with begin[s] <= id
#ifdef __HIPCC__
__host__ __device__
#endif
inline int stripe_of(const int *begin, int ns, unsigned id)
{
    int lo = 0, hi = ns; // begin[lo] <= id < begin[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((unsigned)begin[mid] <= id) lo = mid; else hi = mid;
    }
    return lo;
}
void pack_visits(std::vector<Visit> &visits, int G, int target, int hot_len, BlockPack &out,
                 int one_task_waves = 0, const std::vector<int> *hot_slot_of_row = nullptr);
void finish_plan(std::vector<std::vector<Visit>> &block_visits, const PlanConfig &cfg, Plan &p,
                 PlaceVec &places, int threads);

// number of floats per padded row: 8*ceil(k/8) (reference mf/mf.cpp:959)
inline int k_aligned(int k) { return (k + 7) / 8 * 8; }

// lanes per rating for a padded width: next power of two >= ka/4
int lanes_for(int ka);

// Throws std::runtime_error / std::bad_alloc; callers at the C boundary catch.
void build_plan(const Node *R, long long nnz, int m, int n, const PlanConfig &cfg, Plan &out);

// init_model (reference mf/mf.cpp:952-1007) in internal ids, padded stride ka.
// omega_*_override: counts in INTERNAL row order, or null for the plan's own.
void init_factors(const Plan &plan, const int *omega_p_override, const int *omega_q_override,
                  std::vector<float> &P, std::vector<float> &Q, int threads);

void gen_random_map(int size, std::vector<int> &map); // reference mf/mf.cpp:1009-1017

// run fn(begin,end) over [0,n) on `threads` std::threads
void parallel_ranges(long long n, int threads, const std::function<void(long long, long long, int)> &fn);

} // namespace mfx
