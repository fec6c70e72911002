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
// In a workgroup task (below) `own` is the heavy row the visit belongs to; bit 30 (ENTRY_SWAPPED) says that this
// row lives on the plan's GATHERED side -- `own` then indexes the gathered side's factors and `gat` the owner side's.
struct Entry { uint32_t own; int32_t gat; float r; };
constexpr uint32_t ENTRY_SWAPPED = 0x40000000u, ENTRY_ID_MASK = 0x3FFFFFFFu;
// Bit 30 of `gat` (ENTRY_READ_ONLY): this rating is inside a long run of ONE pair of two heavy rows (synthetic streams repeat
// a pair -- heavy user, heavy item -- thousands of times).  The lighter row of the pair lives in the LDS of its own workgroup
// visits during the launch; the run moves the heavier row only (plan.cpp, build_plan).  RUN_READ_ONLY: a rating is "inside" a
// run when the ratings RUN_READ_ONLY / 2 places before and after it in the sorted order are the same pair.
constexpr int32_t ENTRY_READ_ONLY = 0x40000000;
constexpr int RUN_READ_ONLY = 64;
static_assert(sizeof(Entry) == 12, "Entry must stay 12 bytes (mf_node sized)");

// One wavefront task: `nsteps` steps of G = 64/lanes ratings, stored step-major
// (entry index = off + step*G + group).
struct TaskDesc { uint64_t off; uint32_t nsteps; uint32_t pad; };

// All ratings of one row inside one block: `len` consecutive ratings of the block-sorted rating array, starting at
// global index `start`.  swapped = 0: a row of the plan's owner side (the ratings are sorted by it); swapped = 1: a heavy
// row of the GATHERED side whose ratings were taken out of the owner-major order and grouped by that row instead.
struct Visit { uint32_t own; uint32_t len; uint64_t start; uint32_t swapped = 0; };

// ---- workgroup tasks: the heavy rows ----
// A row with more ratings in a block than one list should hold (hot_len) is not cut into independent chains any more
// (rounds 1-2: private register copies folded by a calibrated model).  Its ratings are dealt over ALL lists of one
// workgroup -- W waves x G lane groups advance through them side by side -- and the row lives in LDS for the length of that
// visit, updated by every list with LDS float atomics: one copy, no lost update, the sequential meaning of the order up
// to the 4 x G ratings in flight.  Only a row with more ratings than one workgroup does in a launch is split over several
// workgroups (n_copies > 1); those copies add their change to the row's combine slot and fold_hot_rows folds them.
// Heavy rows of the gathered side get the same treatment with the roles swapped, instead of being read-modified-written
// by hundreds of lists at once (where most of their updates were lost).
//
// A workgroup task = `nvisits` visits run one after the other; its entries are stored wave-major (wave w of the
// workgroup streams entries [off + w*nsteps*G, off + (w+1)*nsteps*G), step-major like a wave task), the visits tile the
// steps [0, nsteps).  All visits of a task have the same role (info bit 0).
struct WgVisit { uint32_t row; uint32_t nsteps; uint32_t len; uint32_t info; }; // info: n_copies << 1 | swapped; slot in `slot`
struct WgVisitRec { WgVisit v; uint32_t slot; uint32_t pad; };                  // 24 bytes, what the kernel reads
struct WgTask { uint64_t off; uint32_t nsteps; uint32_t visit0; uint32_t nvisits; uint32_t swapped; };
static_assert(sizeof(WgVisitRec) == 24 && sizeof(WgTask) == 24, "kernel-visible layouts");

// "Put sorted ratings src, src+stride, ... (len of them) into entries[dst + x*G], x = 0..len-1."  Bit 30: the visit of a
// wave task, whose first entry is flagged "owner row changes here"; bit 31: a list of a workgroup visit with the roles swapped.
struct Placement { uint64_t src; uint64_t dst; uint32_t len; uint32_t stride_flags; }; // stride (bits 0..15) | flags
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
    std::vector<WgTask> wg_tasks;       // visit0 relative to wg_visits
    std::vector<WgVisitRec> wg_visits;
    std::vector<Placement> places;
    uint64_t n_entries = 0;
    long long hot = 0, padding = 0;
    bool folds = false;                 // some row of this block is split over several workgroups
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
    int waves_per_stripe = 256; // for auto task sizing: the waves the concurrency cap allows per XCD
    int wgs_hw = 0;             // workgroups per XCD the launch has (0: waves_per_stripe / waves_per_wg); more than the cap allows = a wide launch (plan.cpp block_shape)
    int waves_per_wg = 4;     // waves of a workgroup that take work (W of a workgroup task)
    bool swap_heavy = false;  // also run the heavy rows of the GATHERED side in workgroup tasks, roles swapped (experimental)
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
    std::vector<WgTask> wg_tasks;         // workgroup tasks, ordered like the wave tasks
    std::vector<WgVisitRec> wg_visits;
    std::vector<long long> slot_wg_ptr;   // ns*ns+1
    int waves_per_wg = 4;
    int hot_len = 128;                    // a row with more ratings in a block goes to a workgroup task
    bool swap_heavy = false;              // every rating is worked from its heavier row (roles per task; PlanConfig::swap_heavy)
    bool merge_back = false;              // visits write back "memory now + what I changed" instead of their copy (finish_plan)
    std::vector<char> hot_gat;            // per internal row of the gathered side: heavy by its global count (roles swapped)
    long long heavy_thr = 0;              // ... = more than this many ratings in all (hot_len per block on average)
    long long n_hot_slots = 0;            // rows that are split over several workgroups somewhere (combine slots)
    std::vector<int> hot_rows;            // combine slot -> internal row | side << 31 (1 = gathered side of the plan)
    std::vector<char> round_hot;          // ns flags: does round r hold a split row (is there anything to fold behind it)?
    long long n_entries = 0;   // entries.size() on the host path; on the device path the array lives in HBM only
    long long n_hot_rows = 0;
    long long n_padding = 0;
};

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
// task sizes of a plan: wave-task target and the length beyond which a row goes to a workgroup task
void plan_sizes(long long nnz, int NB, int G, const PlanConfig &cfg, int &target, int &hot_len);
// which rows of the gathered side are heavy enough to be taken out of the lock-free side (cnt = ratings per INTERNAL row)
void plan_hot_gathered(const PlanConfig &cfg, Plan &p);
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
