// trainer.cpp -- the mfx_* C-ABI (include/mfx.h) over the HIP kernels.
//
// Host orchestration that replaces the reference's fpsg()/fpsg_core() driver
// (reference mf/mf.cpp:2774-3042): no worker threads, no mutex scheduler -- an epoch is
// `stripes` kernel launches on one HIP stream.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mfx.h"
#include "kernels.hpp"
#include "knobs.hpp"
#include "plan.hpp"
#include "prep.hpp"
#include "synth.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(MFX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

// documented product switches (knobs.hpp), each read once per process
int product_switch(const char *name, int dflt)
{
    return mfx::env_int_raw(name, dflt);
}
const int g_host_plan = product_switch("MFX_HOST_PLAN", 0);
const int g_host_init = product_switch("MFX_HOST_INIT", 0);
const int g_host_threads = product_switch("MFX_HOST_THREADS", 0);
const int g_plan_timing = product_switch("MFX_PLAN_TIMING", 0);
const int g_predict_cache = product_switch("MFX_PREDICT_CACHE", 0);
using mfx::knob_int; // experiment knobs: constants unless the library is built with -DMFX_EXPERIMENTS

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc((void **)&p, count * sizeof(T));
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void adopt(T *ptr, size_t count) // take ownership of a hipMalloc'd buffer
    {
        release();
        p = ptr;
        n = count;
    }
    ~DevBuf() { release(); }
};

int grid_for(long long work_items_per_thread_total, int cu)
{
    long long blocks = (work_items_per_thread_total + 255) / 256;
    long long cap = (long long)cu * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

} // namespace

struct mfx_trainer {
    mfx_options opt;
    mfx::Plan plan; // host copy (entries/tasks dropped after upload)
    int device = 0, cu_count = 256, xcd_count = 8, wg_per_cu = 2, wgs_per_xcd = 64, waves_per_wg = 4;
    int wgs_grid = 64; // workgroups per XCD a launch starts: wgs_per_xcd (what the concurrency cap allows), or more -- a wide launch
    signed char xcc_rank[16];
    hipStream_t stream = nullptr;
    long long n_entries = 0, n_tasks = 0;
    float lambda_p = 0, lambda_q = 0; // scaled (reference mf/mf.cpp:2805-2806)
    float rk1 = 0.125f;

    DevBuf<mfx::EntryD> dEntries;
    DevBuf<mfx::TaskDescD> dTasks;
    DevBuf<long long> dSlotPtr;
    DevBuf<mfx::WgTaskD> dWgTasks;   // workgroup tasks (heavy rows) and their visits
    DevBuf<mfx::WgVisitD> dWgVisits;
    DevBuf<long long> dSlotWgPtr;
    long long n_wg_tasks = 0, n_wg_visits = 0;
    DevBuf<double> dEpochState; // zeroed once per epoch: LOSS_SLOTS loss sums, then 2 * ns*ns cursors (wave tasks, workgroup tasks)
    int *dSlotStateP = nullptr;  // -> the cursors inside dEpochState
    DevBuf<float> dHotAcc;   // combine slots of the rows split over several workgroups (kernels.hip: fold_hot_rows)
    DevBuf<int> dHotRow;     // combine slot -> internal row | side << 31
    float max_wg_per_cu = 0;     // launch width that was chosen (for mfx_info)
    int warm = 0;
    DevBuf<int> dSticky;     // raised by epoch_reset when an epoch left a block unfinished; never cleared
    DevBuf<double> dScalars; // [0..3] scratch for metrics
    double *dLossP = nullptr;    // -> the loss sums inside dEpochState
    DevBuf<int> dOwnBegin, dGatBegin; // RoundArgs::own_begin / gat_begin (ns+1 each)
    DevBuf<int> dOmegaP, dOmegaQ, dPmap, dQmap;
    DevBuf<float> oP, oQ, oPG, oQG; // owned factor storage
    float *dP = nullptr, *dQ = nullptr, *dPG = nullptr, *dQG = nullptr;
    bool model_ready = false;
    long long epochs_done = 0;
    double last_loss = 0;
    bool loss_pending = false;
    bool cursors_live = false; // an epoch has run in THIS trainer since the cursors were last zeroed

    bool timing = false;
    long long timed_launches = 0;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;

    ~mfx_trainer()
    {
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

#pragma GCC visibility push(default)
extern "C" {

int mfx_abi_version(void) { return MFX_ABI_VERSION; }

const char *mfx_last_error(void) { return g_err.c_str(); }

int mfx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void mfx_default_options(mfx_options *o)
{
    memset(o, 0, sizeof(*o));
    o->k = 8; // mf_get_default_param, reference mf/mf.cpp:4538-4557
    o->lambda_p2 = 0.1f;
    o->lambda_q2 = 0.1f;
    o->eta = 0.1f;
    o->device = -1;
}

// options -> plan configuration (shared by the device trainer and the host-only plan)
// Launch width.  The gathered side is updated lock-free, so the number of ratings in flight
// per XCD is capped at (rows of one gathered stripe) / conflict_div: beyond that, lost
// updates on shared rows start to cost final RMSE (measured, DESIGN.md "Concurrency cap").
static int wgs_per_xcd_for(const mfx_options &opt, int m, int n, int ns, int cu_per_xcd,
                           int *waves_per_wg)
{
    *waves_per_wg = 4;
    if (opt.wg_per_cu > 0) return opt.wg_per_cu * cu_per_xcd;
    int env = knob_int("MFX_WG_PER_CU", 0);
    if (env > 0) return env * cu_per_xcd;
    env = knob_int("MFX_WGS_PER_XCD", 0); // experiment knob: exact launch width (1 = four waves per XCD)
    if (env > 0) return env;
    const int ka = mfx::k_aligned(opt.k);
    const int G = 64 / mfx::lanes_for(ka);
    const bool owner_is_q = opt.owner_side == 0 ? (m >= n) : opt.owner_side == 2;
    const long long n_gat = owner_is_q ? m : n;
    const long long stripe_rows = (n_gat + ns - 1) / ns;
    const int div = std::max(1, knob_int("MFX_CONFLICT_DIV", opt.conflict_div > 0 ? opt.conflict_div : 32));
    long long waves = stripe_rows / ((long long)div * G);
    // (Round 1 cut the cap further, with the square of the stripe size, below 7500 rows per stripe -- at a divisor of 8 .. 12.
    //  At 32 that rule only pushed 2 M-rating problems down to ONE workgroup per XCD, which then has nobody to run the heavy
    //  rows' workgroup tasks beside the ordinary rows: +3.2 % on 20 k x 10 k.  It is gone.)
    if (waves < 4) { // tiny problem: one workgroup per XCD with 1..3 live waves
        *waves_per_wg = (int)std::max<long long>(1, waves);
        return 1;
    }
    long long wgs = (waves + 3) / 4;
    // Occupancy cap: about 64 ratings in flight per CU (2 workgroups of 4 waves at 8 ratings per wave).
    // The step is bound by instruction issue once a SIMD holds two waves; more waves only shorten the
    // tasks and crowd the L2 (sweep: profiles/experiments/r01_occupancy_sweep.log).
    const long long cap = (long long)cu_per_xcd * std::max(1, knob_int("MFX_MAX_WG_PER_CU", std::max(1, 16 / G)));
    if (wgs > cap) wgs = cap;
    if (wgs < 1) wgs = 1;
    return (int)wgs;
}

// Workgroups per XCD a launch starts: what the concurrency cap allows -- or, on request (mfx_options.wide), what the chip runs
// anyway (the occupancy cap of wgs_per_xcd_for), the workgroups beyond the cap taking the heavy rows only (plan.cpp
// block_shape).  An explicit width (mfx_options.wg_per_cu, tests) and the one-workgroup launches of tiny problems stay.
static int wgs_grid_for(const mfx_options &opt, int wgs_cap, int waves_per_wg, int cu_per_xcd)
{
    if (opt.wide == 0 || opt.wg_per_cu > 0 || waves_per_wg < 4) return wgs_cap;
    const int G = 64 / mfx::lanes_for(mfx::k_aligned(opt.k));
    return std::max(wgs_cap, cu_per_xcd * std::max(1, 16 / G));
}

// Stripes per side (= launches per epoch) and the launch width that goes with them: one stripe per XCD, whatever the size
// of the problem.  (Round 1 took half the stripes for small problems; it cost parity -- four times the block means four
// times the split of a heavy row and half the folds per epoch -- and is gone.)
static int choose_stripes(const mfx_options &opt, long long nnz, int m, int n, int xcd_count, int cu_per_xcd,
                          int *wgs_per_xcd, int *waves_per_wg)
{
    (void)nnz;
    int stripes = opt.stripes > 0 ? opt.stripes : std::max(1, knob_int("MFX_STRIPES", xcd_count));
    *wgs_per_xcd = wgs_per_xcd_for(opt, m, n, stripes, cu_per_xcd, waves_per_wg);
    return stripes;
}

static mfx::PlanConfig plan_config(const mfx_options &opt, int stripes, int wgs_per_xcd, int waves_per_wg, int wgs_grid = 0)
{
    mfx::PlanConfig cfg;
    cfg.k = opt.k;
    cfg.stripes = stripes;
    cfg.lanes = mfx::lanes_for(mfx::k_aligned(opt.k));
    cfg.task_steps = opt.task_steps > 0 ? opt.task_steps : knob_int("MFX_TASK_STEPS", 0);
    cfg.owner_side = opt.owner_side;
    // 0 mass-balanced stripes, 1 identity, 2 the reference's shuffle; an explicit option wins over the experiment knob
    cfg.map_mode = opt.identity_maps != 0 ? opt.identity_maps : knob_int("MFX_MAP_MODE", 0);
    cfg.use_stats = opt.use_stats != 0;
    cfg.stats_avg = opt.stats_avg;
    cfg.stats_std = opt.stats_std;
    cfg.waves_per_stripe = wgs_per_xcd * waves_per_wg;
    cfg.waves_per_wg = waves_per_wg;
    cfg.wgs_hw = wgs_grid;
    cfg.swap_heavy = opt.no_swap == 0 && knob_int("MFX_NO_SWAP", 0) == 0 && mfx::k_aligned(opt.k) <= 256; // (wide rows: one role, kernels.hip sgd_round_wide)
    cfg.threads = g_host_threads;
    return cfg;
}

static int check_options(const mfx_options &opt)
{
    // check_parameter, reference mf/mf.cpp:3115-3184
    if (opt.k < 1) return fail(MFX_E_ARG, "number of factors must be greater than zero");
    if (opt.lambda_p2 < 0 || opt.lambda_q2 < 0)
        return fail(MFX_E_ARG, "regularization coefficient must be non-negative");
    if (!(opt.eta > 0)) return fail(MFX_E_ARG, "learning rate must be greater than zero");
    if (mfx::k_aligned(opt.k) > 1024)
        return fail(MFX_E_UNSUPPORTED, "k > 1024 is not supported by the gfx950 kernel family (four float4 per lane at most)");
    return MFX_OK;
}

static int create_impl(const mfx::Node *R, const void *R_dev, long long nnz, int m, int n,
                       const mfx_options *opt_in, mfx_trainer **out, const int *layout_cnt_p = nullptr,
                       const int *layout_cnt_q = nullptr)
{
    if (!out) return fail(MFX_E_ARG, "null output handle");
    *out = nullptr;
    if (!opt_in) return fail(MFX_E_ARG, "null options");
    mfx_options opt = *opt_in;
    if (int rc0 = check_options(opt)) return rc0;
    if ((!R && !R_dev) || nnz <= 0 || m <= 0 || n <= 0) return fail(MFX_E_EMPTY, "train on an empty training set");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MFX_E_HIP, "no HIP device: the MI355X path cannot run (there is no CPU fallback)");
    int dev = opt.device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= ndev) return fail(MFX_E_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));

    mfx_trainer *t = new (std::nothrow) mfx_trainer();
    if (!t) return fail(MFX_E_NOMEM, "out of host memory");
    t->opt = opt;
    t->device = dev;
    t->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        // which XCC ids does a grid land on?  (probe launch, once per trainer)
        DevBuf<unsigned> dMask;
        unsigned mask = 0;
        hipError_t e = dMask.alloc(1);
        if (e == hipSuccess) e = hipMemset(dMask.p, 0, sizeof(unsigned));
        if (e == hipSuccess) e = mfx::launch_probe_xcc(dMask.p, t->cu_count * 4, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(&mask, dMask.p, sizeof(unsigned), hipMemcpyDeviceToHost);
        if (e != hipSuccess || mask == 0) {
            delete t;
            return fail(MFX_E_HIP, std::string("XCC probe failed: ") + hipGetErrorString(e));
        }
        int rank = 0;
        for (int i = 0; i < 16; ++i) t->xcc_rank[i] = (mask >> i) & 1 ? (signed char)rank++ : (signed char)-1;
        t->xcd_count = rank;
    }
    const int cu_per_xcd = std::max(1, t->cu_count / t->xcd_count);
    const int stripes = choose_stripes(opt, nnz, m, n, t->xcd_count, cu_per_xcd, &t->wgs_per_xcd, &t->waves_per_wg);
    t->wg_per_cu = (t->wgs_per_xcd + cu_per_xcd - 1) / cu_per_xcd;

    t->wgs_grid = wgs_grid_for(opt, t->wgs_per_xcd, t->waves_per_wg, cu_per_xcd);
    mfx::PlanConfig cfg = plan_config(opt, stripes, t->wgs_per_xcd, t->waves_per_wg, t->wgs_grid);
    cfg.layout_cnt_p = layout_cnt_p;
    cfg.layout_cnt_q = layout_cnt_q;

    // Pre-processing: on the device (prep.hip) unless forced to the host builder (MFX_HOST_PLAN=1)
    // or the ids do not fit the sort key.  R may be host memory (uploaded once) or already in HBM.
    const bool device_plan = g_host_plan == 0 && mfx::device_prep_supported(m, n);
    mfx::EntryD *dev_entries = nullptr;
    const bool plan_timing = g_plan_timing != 0;
    const auto t_create = std::chrono::steady_clock::now();
    auto since = [&](const char *what) {
        if (plan_timing)
            fprintf(stderr, "mfx create: %-26s %8.2f ms since the start\n", what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count());
    };
    try {
        if (device_plan) {
            hipStream_t ps = nullptr;
            hipError_t e = hipStreamCreateWithFlags(&ps, hipStreamNonBlocking);
            if (e != hipSuccess) throw std::runtime_error(std::string("hipStreamCreate: ") + hipGetErrorString(e));
            DevBuf<mfx::Node> up;
            const void *dR = R_dev;
            try {
                if (!dR) {
                    e = up.alloc((size_t)nnz);
                    if (e == hipSuccess) e = hipMemcpyAsync(up.p, R, (size_t)nnz * sizeof(mfx::Node), hipMemcpyHostToDevice, ps);
                    if (e != hipSuccess) throw std::runtime_error(std::string("upload of ratings: ") + hipGetErrorString(e));
                    dR = up.p;
                }
                mfx::build_plan_device(dR, nnz, m, n, cfg, t->cu_count, ps, t->plan, &dev_entries);
            } catch (...) {
                (void)hipStreamDestroy(ps);
                throw;
            }
            (void)hipStreamDestroy(ps);
        } else {
            std::vector<mfx::Node> host;
            if (!R) { // ratings live in HBM but the host builder was asked for
                host.resize((size_t)nnz);
                hipError_t e = hipMemcpy(host.data(), R_dev, (size_t)nnz * sizeof(mfx::Node), hipMemcpyDeviceToHost);
                if (e != hipSuccess) throw std::runtime_error(std::string("download of ratings: ") + hipGetErrorString(e));
                R = host.data();
            }
            mfx::build_plan(R, nnz, m, n, cfg, t->plan);
        }
    } catch (const std::bad_alloc &) {
        delete t;
        return fail(MFX_E_NOMEM, "out of host memory while building the plan");
    } catch (const std::invalid_argument &e) {
        delete t;
        return fail(MFX_E_ARG, e.what());
    } catch (const std::exception &e) {
        delete t;
        return fail(MFX_E_HIP, e.what());
    }
    since("plan built, scratch freed");
    mfx::Plan &p = t->plan;
    t->lambda_p = opt.lambda_p2 / p.scale;
    t->lambda_q = opt.lambda_q2 / p.scale;
    t->rk1 = (opt.rk_mode == 1 && p.ka > 8) ? (float)1.0 / (p.ka - 8) : 0.125f;
    t->n_entries = dev_entries ? p.n_entries : (long long)p.entries.size();
    t->n_tasks = (long long)p.tasks.size();
    t->n_wg_tasks = (long long)p.wg_tasks.size();
    t->n_wg_visits = (long long)p.wg_visits.size();

    int rc = MFX_OK;
    auto up = [&]() -> int {
        HIP_TRY(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
        if (dev_entries) t->dEntries.adopt(dev_entries, (size_t)p.n_entries);
        else HIP_TRY(t->dEntries.alloc(p.entries.size()));
        HIP_TRY(t->dTasks.alloc(p.tasks.size()));
        HIP_TRY(t->dSlotPtr.alloc(p.slot_task_ptr.size()));
        static_assert(sizeof(mfx::WgTask) == sizeof(mfx::WgTaskD) && sizeof(mfx::WgVisitRec) == sizeof(mfx::WgVisitD), "plan.hpp <-> kernels.hpp");
        HIP_TRY(t->dWgTasks.alloc(std::max<size_t>(1, p.wg_tasks.size())));
        HIP_TRY(t->dWgVisits.alloc(std::max<size_t>(1, p.wg_visits.size())));
        HIP_TRY(t->dSlotWgPtr.alloc(p.slot_wg_ptr.size()));
        if (!p.wg_tasks.empty())
            HIP_TRY(hipMemcpy(t->dWgTasks.p, p.wg_tasks.data(), p.wg_tasks.size() * sizeof(mfx::WgTask), hipMemcpyHostToDevice));
        if (!p.wg_visits.empty())
            HIP_TRY(hipMemcpy(t->dWgVisits.p, p.wg_visits.data(), p.wg_visits.size() * sizeof(mfx::WgVisitRec), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t->dSlotWgPtr.p, p.slot_wg_ptr.data(), p.slot_wg_ptr.size() * sizeof(long long), hipMemcpyHostToDevice));
        const size_t epoch_state_doubles = mfx::LOSS_SLOTS + (size_t)p.ns * p.ns + 1;
        HIP_TRY(t->dEpochState.alloc(epoch_state_doubles));
        HIP_TRY(hipMemset(t->dEpochState.p, 0, epoch_state_doubles * sizeof(double)));
        t->dLossP = t->dEpochState.p;
        t->dSlotStateP = (int *)(t->dEpochState.p + mfx::LOSS_SLOTS);
        HIP_TRY(t->dScalars.alloc(4));
        {
            const size_t slots = (size_t)std::max<long long>(1, p.n_hot_slots);
            const size_t words = slots * (size_t)mfx::HOT_SUB * (size_t)(p.ka + mfx::HOT_EXTRA);
            HIP_TRY(t->dHotAcc.alloc(words));
            HIP_TRY(t->dHotRow.alloc(slots));
            HIP_TRY(hipMemset(t->dHotAcc.p, 0, words * sizeof(float)));
            HIP_TRY(hipMemset(t->dHotRow.p, 0, slots * sizeof(int)));
            if (p.n_hot_slots > 0)
                HIP_TRY(hipMemcpy(t->dHotRow.p, p.hot_rows.data(), (size_t)p.n_hot_slots * sizeof(int), hipMemcpyHostToDevice));
        }
        HIP_TRY(t->dSticky.alloc(1));
        HIP_TRY(hipMemset(t->dSticky.p, 0, sizeof(int)));
        {
            // stripe boundaries for the kernel (buffer descriptors over the stripe that is read-modified-written, L2 warm-up)
            const std::vector<int> &ob = p.owner_is_q ? p.q_begin : p.p_begin, &gb = p.owner_is_q ? p.p_begin : p.q_begin;
            for (int x = 0; x < p.ns; ++x)
                if ((unsigned long long)(gb[x + 1] - gb[x]) * p.ka * 4ull >= 0xFFFFFF00ull ||
                    (unsigned long long)(ob[x + 1] - ob[x]) * p.ka * 4ull >= 0xFFFFFF00ull) {
                    rc = fail(MFX_E_UNSUPPORTED, "a stripe of 4 GB or more: use more stripes (mfx_options.stripes)");
                    return rc;
                }
            HIP_TRY(t->dOwnBegin.alloc(ob.size()));
            HIP_TRY(t->dGatBegin.alloc(gb.size()));
            HIP_TRY(hipMemcpy(t->dOwnBegin.p, ob.data(), ob.size() * sizeof(int), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(t->dGatBegin.p, gb.data(), gb.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        HIP_TRY(t->dOmegaP.alloc(m));
        HIP_TRY(t->dOmegaQ.alloc(n));
        HIP_TRY(t->dPmap.alloc(m));
        HIP_TRY(t->dQmap.alloc(n));
        if (!dev_entries)
            HIP_TRY(hipMemcpy(t->dEntries.p, p.entries.data(), p.entries.size() * sizeof(mfx::Entry),
                              hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t->dTasks.p, p.tasks.data(), p.tasks.size() * sizeof(mfx::TaskDesc),
                          hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t->dSlotPtr.p, p.slot_task_ptr.data(),
                          p.slot_task_ptr.size() * sizeof(long long), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t->dOmegaP.p, p.omega_p.data(), (size_t)m * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t->dOmegaQ.p, p.omega_q.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t->dPmap.p, p.p_map.data(), (size_t)m * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t->dQmap.p, p.q_map.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(t->dScalars.p, 0, 4 * sizeof(double)));
        return MFX_OK;
    };
    rc = up();
    if (rc != MFX_OK) {
        if (dev_entries && t->dEntries.p != dev_entries) (void)hipFree(dev_entries); // not adopted yet
        delete t;
        return rc;
    }
    std::vector<mfx::Entry>().swap(p.entries);
    std::vector<mfx::TaskDesc>().swap(p.tasks);
    std::vector<mfx::WgTask>().swap(p.wg_tasks);
    std::vector<mfx::WgVisitRec>().swap(p.wg_visits);
    {
        // L2 warm-up: only when every block's two stripes (rows + accumulators) fit the XCD's L2 with room to spare
        const std::vector<int> &ob = p.owner_is_q ? p.q_begin : p.p_begin, &gb = p.owner_is_q ? p.p_begin : p.q_begin;
        int max_own = 0, max_gat = 0;
        for (int x = 0; x < p.ns; ++x) {
            max_own = std::max(max_own, ob[x + 1] - ob[x]);
            max_gat = std::max(max_gat, gb[x + 1] - gb[x]);
        }
        const size_t warm_bytes = ((size_t)max_own + max_gat) * ((size_t)p.ka * 4 + 8);
        t->warm = warm_bytes <= (size_t)knob_int("MFX_WARM_KB", 3072) * 1024 ? 1 : 0;
    }
    since("plan tables on the device");
    *out = t;
    return MFX_OK;
}

int mfx_trainer_create(const mfx_node *R_host, long long nnz, int m, int n, const mfx_options *opt,
                       mfx_trainer **out)
{
    try {
        return create_impl((const mfx::Node *)R_host, nullptr, nnz, m, n, opt, out);
    } catch (const std::exception &e) {
        return fail(MFX_E_STATE, e.what());
    } catch (...) {
        return fail(MFX_E_STATE, "unknown failure");
    }
}

int mfx_trainer_create_layout(const mfx_node *R_host, const void *R_dev, long long nnz, int m, int n,
                              const mfx_options *opt, const int *layout_cnt_p, const int *layout_cnt_q,
                              mfx_trainer **out)
{
    try {
        return create_impl((const mfx::Node *)R_host, R_host ? nullptr : R_dev, nnz, m, n, opt, out, layout_cnt_p,
                           layout_cnt_q);
    } catch (const std::exception &e) {
        return fail(MFX_E_STATE, e.what());
    } catch (...) {
        return fail(MFX_E_STATE, "unknown failure");
    }
}

int mfx_trainer_create_device(const void *R_dev, long long nnz, int m, int n, const mfx_options *opt,
                              mfx_trainer **out)
{
    // ratings already resident in HBM: nothing but the visit table crosses PCIe (prep.hip)
    try {
        return create_impl(nullptr, R_dev, nnz, m, n, opt, out);
    } catch (const std::exception &e) {
        return fail(MFX_E_STATE, e.what());
    } catch (...) {
        return fail(MFX_E_STATE, "unknown failure");
    }
}

void mfx_trainer_destroy(mfx_trainer *t)
{
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    delete t;
}

int mfx_trainer_bind_model(mfx_trainer *t, void *dP, void *dQ, void *dPG, void *dQG)
{
    if (!t || !dP || !dQ || !dPG || !dQG) return fail(MFX_E_ARG, "null pointer");
    t->dP = (float *)dP;
    t->dQ = (float *)dQ;
    t->dPG = (float *)dPG;
    t->dQG = (float *)dQG;
    return MFX_OK;
}

static int ensure_model_storage(mfx_trainer *t)
{
    const mfx::Plan &p = t->plan;
    if (!t->dP) {
        HIP_TRY(t->oP.alloc((size_t)p.m * p.ka));
        HIP_TRY(t->oQ.alloc((size_t)p.n * p.ka));
        HIP_TRY(t->oPG.alloc((size_t)p.m * 2));
        HIP_TRY(t->oQG.alloc((size_t)p.n * 2));
        t->dP = t->oP.p;
        t->dQ = t->oQ.p;
        t->dPG = t->oPG.p;
        t->dQG = t->oQG.p;
    }
    return MFX_OK;
}

int mfx_trainer_init_model_counts(mfx_trainer *t, const int *omega_p, const int *omega_q)
{
    if (!t) return fail(MFX_E_ARG, "null trainer");
    try {
        HIP_TRY(hipSetDevice(t->device));
        int rc = ensure_model_storage(t);
        if (rc) return rc;
        const mfx::Plan &p = t->plan;
        // overrides arrive in original ids; rows are stored in internal (permuted) order
        std::vector<int> op, oq;
        if (omega_p) {
            op.resize(p.m);
            for (int i = 0; i < p.m; ++i) op[p.p_map[i]] = omega_p[i];
        }
        if (omega_q) {
            oq.resize(p.n);
            for (int i = 0; i < p.n; ++i) oq[p.q_map[i]] = omega_q[i];
        }
        if (omega_p) HIP_TRY(hipMemcpy(t->dOmegaP.p, op.data(), (size_t)p.m * 4, hipMemcpyHostToDevice));
        if (omega_q) HIP_TRY(hipMemcpy(t->dOmegaQ.p, oq.data(), (size_t)p.n * 4, hipMemcpyHostToDevice));
        if (g_host_init == 0) {
            // the reference's single minstd_rand0 stream, entered per row by skip-ahead (prep.hip)
            mfx::init_factors_device(t->dOmegaP.p, p.m, t->dOmegaQ.p, p.n, p.p_at.empty() ? nullptr : p.p_at.data(),
                                     p.q_at.empty() ? nullptr : p.q_at.data(), p.k, p.ka, t->cu_count, t->stream,
                                     t->dP, t->dQ);
        } else {
            std::vector<float> P, Q;
            mfx::init_factors(p, omega_p ? op.data() : nullptr, omega_q ? oq.data() : nullptr, P, Q, g_host_threads);
            HIP_TRY(hipMemcpy(t->dP, P.data(), P.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(t->dQ, Q.data(), Q.size() * 4, hipMemcpyHostToDevice));
        }
        // PG, QG <- 1 (reference mf/mf.cpp:2835)
        HIP_TRY(mfx::launch_fill(t->dPG, 2LL * p.m, 1.0f, grid_for(2LL * p.m, t->cu_count), t->stream));
        HIP_TRY(mfx::launch_fill(t->dQG, 2LL * p.n, 1.0f, grid_for(2LL * p.n, t->cu_count), t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        t->model_ready = true;
        t->epochs_done = 0;
        return MFX_OK;
    } catch (const std::bad_alloc &e) {
        return fail(MFX_E_NOMEM, e.what());
    } catch (const std::exception &e) {
        return fail(MFX_E_HIP, e.what());
    }
}

int mfx_trainer_init_model(mfx_trainer *t, const int *omega_q_override)
{
    return mfx_trainer_init_model_counts(t, nullptr, omega_q_override);
}

static hipEvent_t next_event(mfx_trainer *t)
{
    if (t->ev_used == t->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        t->ev_pool.push_back(e);
    }
    return t->ev_pool[t->ev_used++];
}

int mfx_trainer_epoch_part(mfx_trainer *t, int slow_only, void *stream_v, int part, int nparts)
{
    if (!t) return fail(MFX_E_ARG, "null trainer");
    if (!t->model_ready) return fail(MFX_E_STATE, "model not initialised");
    HIP_TRY(hipSetDevice(t->device));
    hipStream_t s = stream_v ? (hipStream_t)stream_v : t->stream;
    const mfx::Plan &p = t->plan;
    const int ns = p.ns;
    if (nparts < 1 || nparts > ns || part < 0 || part >= nparts)
        return fail(MFX_E_ARG, "epoch part out of range (1 <= nparts <= stripes)");
    if (part == 0) {
        // one launch checks the cursors of the epoch before (sticky flag, read by verify_rounds) and zeroes
        // the loss sums and the task cursors
        HIP_TRY(mfx::launch_epoch_reset(t->dLossP, t->dSlotStateP, t->dSlotPtr.p, t->dSlotWgPtr.p, ns * ns,
                                        t->cursors_live ? 1 : 0, t->dSticky.p, s));
    }

    mfx::RoundArgs a;
    a.own_rows = p.owner_is_q ? t->dQ : t->dP;
    a.gat_rows = p.owner_is_q ? t->dP : t->dQ;
    a.own_acc = p.owner_is_q ? t->dQG : t->dPG;
    a.gat_acc = p.owner_is_q ? t->dPG : t->dQG;
    a.hot_acc = t->dHotAcc.p;
    a.entries = t->dEntries.p;
    a.tasks = t->dTasks.p;
    a.wg_tasks = t->dWgTasks.p;
    a.wg_visits = t->dWgVisits.p;
    a.loss = t->dLossP;
    a.own_begin = t->dOwnBegin.p;
    a.gat_begin = t->dGatBegin.p;
    a.lambda_own = p.owner_is_q ? t->lambda_q : t->lambda_p;
    a.lambda_gat = p.owner_is_q ? t->lambda_p : t->lambda_q;
    a.eta = t->opt.eta;
    a.rk1 = t->rk1;
    a.ka = p.ka;
    a.slow_only = slow_only ? 1 : 0;
    a.ns = ns;
    a.n_xcc = t->xcd_count;
    a.active_waves = t->waves_per_wg;
#ifdef MFX_STAMPS
    { // diagnostic build (`make diag`): where do the cycles of a wave go?  MFX_STAMPS_DUMP=1 prints and resets the sums
        static unsigned long long *g_stamps = nullptr;
        const size_t words = (size_t)65536 * 12; // 8 sums + 4 timeline words per wave
        if (!g_stamps) {
            (void)hipMalloc((void **)&g_stamps, words * sizeof(unsigned long long));
            (void)hipMemset(g_stamps, 0, words * sizeof(unsigned long long));
        }
        a.stamps = g_stamps;
        if (getenv("MFX_STAMPS_DUMP")) {
            std::vector<unsigned long long> h(words);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h.data(), g_stamps, words * 8, hipMemcpyDeviceToHost);
            unsigned long long s8[8] = {0};
            for (size_t i = 0; i < (size_t)65536 * 8; ++i) s8[i % 8] += h[i];
            { // timeline of the latest launch on the 100 MHz device clock, per XCC
                const unsigned long long *tl = h.data() + (size_t)65536 * 8;
                unsigned long long t0 = ~0ull;
                for (size_t w = 0; w < 65536; ++w)
                    if (tl[w * 4 + 2] && (tl[w * 4 + 3] & 0xFFFFF) > 0) t0 = std::min(t0, tl[w * 4]);
                for (unsigned long long x = 1; x <= 16; ++x) {
                    std::vector<unsigned long long> st, en;
                    unsigned long long steps = 0, burst = 0;
                    for (size_t w = 0; w < 65536; ++w)
                        if (tl[w * 4 + 2] == x && (tl[w * 4 + 3] & 0xFFFFF) > 0) {
                            st.push_back(tl[w * 4] - t0);
                            en.push_back(tl[w * 4 + 1] - t0);
                            steps += tl[w * 4 + 3] & 0xFFFFF;
                            burst += tl[w * 4 + 3] >> 20;
                        }
                    if (st.empty()) continue;
                    std::sort(st.begin(), st.end());
                    std::sort(en.begin(), en.end());
                    auto q = [](const std::vector<unsigned long long> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
                    fprintf(stderr, "timeline XCC%llu: %zu waves x %.1f steps (workgroup-task phase: %.0f cycles per wave) | start max %llu | end min %llu p10 %llu p50 %llu p90 %llu max %llu (x10 ns)\n",
                            x - 1, st.size(), (double)steps / st.size(), (double)burst / std::max<size_t>(1, st.size()), st.back(), en.front(), q(en, .1), q(en, .5), q(en, .9), en.back());
                }
            }
            fprintf(stderr, "stamps: waves*launches %llu steps %llu tasks %llu | per step: wait %.0f window %.0f rest %.0f cycles | "
                            "per task: fetch+stage %.0f | wave total %.0f cycles per launch\n",
                    s8[7], s8[4], s8[5], (double)s8[1] / s8[4], (double)s8[2] / s8[4], (double)s8[3] / s8[4],
                    (double)s8[0] / s8[5], (double)s8[6] / s8[7]);
            (void)hipMemset(g_stamps, 0, words * 8);
        }
    }
#endif
    memcpy(a.xcc_rank, t->xcc_rank, sizeof(a.xcc_rank));
    const int grid = t->xcd_count * t->wgs_grid; // workgroups are dealt round-robin over XCDs
    const int i_begin = (int)((long long)part * ns / nparts), i_end = (int)((long long)(part + 1) * ns / nparts);
    // timing: one event pair around the launches of this call (a pair per launch costs ~7 us each,
    // 6 % of a 123 us launch); mean launch time = bracket / launches, inter-launch gaps included
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (t->timing && i_end > i_begin) {
        e0 = next_event(t);
        e1 = next_event(t);
        if (!e0 || !e1) return fail(MFX_E_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(e0, s));
    }
    a.warm = t->warm;
    a.merge_back = p.merge_back ? 1 : 0;
    a.waves_per_xcd = t->wgs_per_xcd * t->waves_per_wg;
    for (int i = i_begin; i < i_end; ++i) {
        const int r = (int)((i + t->epochs_done) % ns); // rotate the starting round per epoch
        a.round = r;
        a.slot_task_ptr = t->dSlotPtr.p + (size_t)r * ns;
        a.slot_cursor = t->dSlotStateP + (size_t)r * ns;
        a.slot_wg_ptr = t->dSlotWgPtr.p + (size_t)r * ns;
        a.wg_cursor = t->dSlotStateP + (size_t)ns * ns + (size_t)r * ns;
        HIP_TRY(mfx::launch_sgd_round(p.lanes, a, grid, s));
        // the copies of the rows of this round that are split over several workgroups are folded into their rows
        if (p.round_hot.empty() || !p.round_hot[(size_t)r]) continue; // no such row in this round: nothing to fold
        HIP_TRY(mfx::launch_fold_hot(a.own_rows, a.own_acc, a.gat_rows, a.gat_acc, t->dHotAcc.p, t->dHotRow.p, (int)p.n_hot_slots,
                                     p.ka, a.eta, a.rk1, a.slow_only, s));
    }
    if (e1) {
        HIP_TRY(hipEventRecord(e1, s));
        t->timed_launches += i_end - i_begin;
    }
    if (part == nparts - 1) {
        t->epochs_done++;
        t->loss_pending = true;
        t->cursors_live = true;
    }
    return MFX_OK;
}

int mfx_trainer_epoch(mfx_trainer *t, int slow_only, void *stream_v)
{
    return mfx_trainer_epoch_part(t, slow_only, stream_v, 0, 1);
}

// Every task of the last epoch must have been handed out: cursor >= task count per block.
// Fails loudly if an XCD the probe saw received no workgroup in some launch.
// Earlier epochs are covered by the sticky flag that epoch_reset raises on the device before it zeroes the
// cursors, so an epoch nobody synchronised on is checked all the same.  Callers have synchronised the device.
static int verify_rounds(mfx_trainer *t)
{
    static const char *const msg = "a stripe block was left unprocessed: workgroup-to-XCD placement "
                                   "differs from the probe (set MFX_STRIPES / rerun)";
    int sticky = 0;
    HIP_TRY(hipMemcpy(&sticky, t->dSticky.p, sizeof(int), hipMemcpyDeviceToHost));
    if (sticky) return fail(MFX_E_STATE, msg);
    if (!t->loss_pending) return MFX_OK;
    const mfx::Plan &p = t->plan;
    const size_t nb = (size_t)p.ns * p.ns;
    std::vector<int> cur(2 * nb);
    HIP_TRY(hipMemcpy(cur.data(), t->dSlotStateP, cur.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nb; ++i)
        if (cur[i] < p.slot_task_ptr[i + 1] - p.slot_task_ptr[i] || cur[nb + i] < p.slot_wg_ptr[i + 1] - p.slot_wg_ptr[i])
            return fail(MFX_E_STATE, msg);
    t->loss_pending = false;
    return MFX_OK;
}

int mfx_trainer_sync(mfx_trainer *t)
{
    if (!t) return fail(MFX_E_ARG, "null trainer");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipDeviceSynchronize());
    return verify_rounds(t);
}

int mfx_trainer_last_loss(mfx_trainer *t, double *sum_sq)
{
    if (!t || !sum_sq) return fail(MFX_E_ARG, "null pointer");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipDeviceSynchronize());
    double part[mfx::LOSS_SLOTS];
    HIP_TRY(hipMemcpy(part, t->dLossP, sizeof(part), hipMemcpyDeviceToHost));
    t->last_loss = 0;
    for (int i = 0; i < mfx::LOSS_SLOTS; ++i) t->last_loss += part[i];
    *sum_sq = t->last_loss;
    return verify_rounds(t);
}

int mfx_trainer_reg2(mfx_trainer *t, double *reg)
{
    if (!t || !reg) return fail(MFX_E_ARG, "null pointer");
    if (!t->model_ready) return fail(MFX_E_STATE, "model not initialised");
    HIP_TRY(hipSetDevice(t->device));
    const mfx::Plan &p = t->plan;
    double h[2] = {0, 0};
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(t->dScalars.p + 1, 0, 2 * sizeof(double)));
    HIP_TRY(mfx::launch_reg2(t->dP, t->dOmegaP.p, p.m, p.ka, t->dScalars.p + 1,
                             grid_for(p.m, t->cu_count), t->stream));
    HIP_TRY(mfx::launch_reg2(t->dQ, t->dOmegaQ.p, p.n, p.ka, t->dScalars.p + 2,
                             grid_for(p.n, t->cu_count), t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    HIP_TRY(hipMemcpy(h, t->dScalars.p + 1, 2 * sizeof(double), hipMemcpyDeviceToHost));
    *reg = t->lambda_p * h[0] + t->lambda_q * h[1];
    return MFX_OK;
}

int mfx_trainer_rmse(mfx_trainer *t, double *rmse)
{
    if (!t || !rmse) return fail(MFX_E_ARG, "null pointer");
    double s = 0;
    int rc = mfx_trainer_sq_err(t, &s);
    if (rc) return rc;
    *rmse = std::sqrt(s / (double)t->plan.nnz);
    return MFX_OK;
}

int mfx_trainer_sq_err(mfx_trainer *t, double *sum_sq)
{
    double *rmse = sum_sq;
    if (!t || !rmse) return fail(MFX_E_ARG, "null pointer");
    if (!t->model_ready) return fail(MFX_E_STATE, "model not initialised");
    HIP_TRY(hipSetDevice(t->device));
    const mfx::Plan &p = t->plan;
    HIP_TRY(hipDeviceSynchronize());
    if (int rcv = verify_rounds(t)) return rcv;
    HIP_TRY(hipMemset(t->dScalars.p + 3, 0, sizeof(double)));
    const float *own = p.owner_is_q ? t->dQ : t->dP, *gat = p.owner_is_q ? t->dP : t->dQ;
    HIP_TRY(mfx::launch_sq_err_entries(p.lanes, own, gat, t->dEntries.p, t->n_entries, p.ka,
                                       t->dScalars.p + 3, t->cu_count * 8, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    double s = 0;
    HIP_TRY(hipMemcpy(&s, t->dScalars.p + 3, sizeof(double), hipMemcpyDeviceToHost));
    *rmse = s * (double)p.scale * (double)p.scale;
    return MFX_OK;
}

int mfx_trainer_info(mfx_trainer *t, mfx_info *o)
{
    if (!t || !o) return fail(MFX_E_ARG, "null pointer");
    const mfx::Plan &p = t->plan;
    memset(o, 0, sizeof(*o));
    o->m = p.m;
    o->n = p.n;
    o->k = p.k;
    o->k_aligned = p.ka;
    o->nnz = p.nnz;
    o->avg = p.avg;
    o->std_dev = p.std_dev;
    o->scale = p.scale;
    o->lambda_p_scaled = t->lambda_p;
    o->lambda_q_scaled = t->lambda_q;
    o->stripes = p.ns;
    o->lanes_per_rating = p.lanes;
    o->ratings_per_wave = p.groups;
    o->owner_is_q = p.owner_is_q ? 1 : 0;
    o->n_entries = t->n_entries;
    o->n_tasks = t->n_tasks;
    o->n_hot_rows = p.n_hot_rows;
    o->n_wg_tasks = t->n_wg_tasks;
    o->n_wg_visits = t->n_wg_visits;
    o->n_hot_slots = p.n_hot_slots;
    o->hot_acc_bytes = (long long)t->dHotAcc.n * 4;
    o->waves_per_wg = t->waves_per_wg;
    o->hot_len = p.hot_len;
    o->merge_back = p.merge_back ? 1 : 0;
    o->grid_wg_per_cu = (t->wgs_grid + std::max(1, t->cu_count / t->xcd_count) - 1) / std::max(1, t->cu_count / t->xcd_count);
    o->cu_count = t->cu_count;
    o->xcd_count = t->xcd_count;
    o->wg_per_cu = t->wg_per_cu;
    o->dP = t->dP;
    o->dQ = t->dQ;
    o->dPG = t->dPG;
    o->dQG = t->dQG;
    o->bytes_per_rating = 16.0 * p.ka + 44.0;
    return MFX_OK;
}

int mfx_trainer_maps(mfx_trainer *t, int *p_map, int *q_map)
{
    if (!t) return fail(MFX_E_ARG, "null trainer");
    if (p_map) memcpy(p_map, t->plan.p_map.data(), (size_t)t->plan.m * 4);
    if (q_map) memcpy(q_map, t->plan.q_map.data(), (size_t)t->plan.n * 4);
    return MFX_OK;
}

int mfx_trainer_get_model(mfx_trainer *t, float *P, float *Q, float *PG, float *QG)
{
    if (!t) return fail(MFX_E_ARG, "null trainer");
    if (!t->model_ready) return fail(MFX_E_STATE, "model not initialised");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipDeviceSynchronize());
    if (int rcv = verify_rounds(t)) return rcv;
    const mfx::Plan &p = t->plan;
    if (P) HIP_TRY(hipMemcpy(P, t->dP, (size_t)p.m * p.ka * 4, hipMemcpyDeviceToHost));
    if (Q) HIP_TRY(hipMemcpy(Q, t->dQ, (size_t)p.n * p.ka * 4, hipMemcpyDeviceToHost));
    if (PG) HIP_TRY(hipMemcpy(PG, t->dPG, (size_t)p.m * 8, hipMemcpyDeviceToHost));
    if (QG) HIP_TRY(hipMemcpy(QG, t->dQG, (size_t)p.n * 8, hipMemcpyDeviceToHost));
    return MFX_OK;
}

int mfx_trainer_set_model(mfx_trainer *t, const float *P, const float *Q, const float *PG,
                          const float *QG)
{
    if (!t) return fail(MFX_E_ARG, "null trainer");
    HIP_TRY(hipSetDevice(t->device));
    int rc = ensure_model_storage(t);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    const mfx::Plan &p = t->plan;
    if (P) HIP_TRY(hipMemcpy(t->dP, P, (size_t)p.m * p.ka * 4, hipMemcpyHostToDevice));
    if (Q) HIP_TRY(hipMemcpy(t->dQ, Q, (size_t)p.n * p.ka * 4, hipMemcpyHostToDevice));
    if (PG) HIP_TRY(hipMemcpy(t->dPG, PG, (size_t)p.m * 8, hipMemcpyHostToDevice));
    if (QG) HIP_TRY(hipMemcpy(t->dQG, QG, (size_t)p.n * 8, hipMemcpyHostToDevice));
    t->model_ready = true;
    return MFX_OK;
}

// Fingerprint of the internal layout a raw model (mfx_trainer_get_model) is expressed in: sizes, padded width,
// stripe count, owner side and both id maps.  A checkpoint carries it; a trainer built with another stripe count,
// id layout or data refuses the state instead of training on rows that mean something else.
int mfx_trainer_layout_fingerprint(mfx_trainer *t, unsigned long long *fp)
{
    if (!t || !fp) return fail(MFX_E_ARG, "null pointer");
    const mfx::Plan &p = t->plan;
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](unsigned long long v) { h = (h ^ v) * 1099511628211ull; };
    mix((unsigned long long)p.m);
    mix((unsigned long long)p.n);
    mix((unsigned long long)p.ka);
    mix((unsigned long long)p.ns);
    mix(p.owner_is_q ? 1ull : 0ull);
    for (int v : p.p_map) mix((unsigned long long)(unsigned)v);
    for (int v : p.q_map) mix((unsigned long long)(unsigned)v);
    *fp = h;
    return MFX_OK;
}

long long mfx_trainer_epochs_done(mfx_trainer *t) { return t ? t->epochs_done : -1; }

int mfx_trainer_set_epochs_done(mfx_trainer *t, long long epochs)
{
    if (!t || epochs < 0) return fail(MFX_E_ARG, "bad argument");
    t->epochs_done = epochs; // the first round of an epoch rotates with this count (mfx_trainer_epoch_part)
    return MFX_OK;
}

int mfx_trainer_plan_copy(mfx_trainer *t, void *entries, void *tasks, long long *slot_task_ptr)
{
    if (!t) return fail(MFX_E_ARG, "null trainer");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipDeviceSynchronize());
    const mfx::Plan &p = t->plan;
    if (entries) HIP_TRY(hipMemcpy(entries, t->dEntries.p, (size_t)t->n_entries * sizeof(mfx::EntryD), hipMemcpyDeviceToHost));
    if (tasks) HIP_TRY(hipMemcpy(tasks, t->dTasks.p, (size_t)t->n_tasks * sizeof(mfx::TaskDescD), hipMemcpyDeviceToHost));
    if (slot_task_ptr) memcpy(slot_task_ptr, p.slot_task_ptr.data(), p.slot_task_ptr.size() * sizeof(long long));
    return MFX_OK;
}

int mfx_trainer_plan_copy_wg(mfx_trainer *t, void *wg_tasks, void *wg_visits, long long *slot_wg_ptr)
{
    if (!t) return fail(MFX_E_ARG, "null trainer");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipDeviceSynchronize());
    const mfx::Plan &p = t->plan;
    if (wg_tasks && t->n_wg_tasks) HIP_TRY(hipMemcpy(wg_tasks, t->dWgTasks.p, (size_t)t->n_wg_tasks * sizeof(mfx::WgTaskD), hipMemcpyDeviceToHost));
    if (wg_visits && t->n_wg_visits) HIP_TRY(hipMemcpy(wg_visits, t->dWgVisits.p, (size_t)t->n_wg_visits * sizeof(mfx::WgVisitD), hipMemcpyDeviceToHost));
    if (slot_wg_ptr) memcpy(slot_wg_ptr, p.slot_wg_ptr.data(), p.slot_wg_ptr.size() * sizeof(long long));
    return MFX_OK;
}

int mfx_trainer_timing_enable(mfx_trainer *t, int on)
{
    if (!t) return fail(MFX_E_ARG, "null trainer");
    t->timing = on != 0;
    t->ev_used = 0;
    t->timed_launches = 0;
    return MFX_OK;
}

int mfx_trainer_timing_read(mfx_trainer *t, long long *launches, double *total_ms)
{
    if (!t || !launches || !total_ms) return fail(MFX_E_ARG, "null pointer");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipDeviceSynchronize());
    double tot = 0;
    for (size_t i = 0; i + 1 < t->ev_used; i += 2) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, t->ev_pool[i], t->ev_pool[i + 1]));
        tot += ms;
    }
    *launches = t->timed_launches;
    *total_ms = tot;
    t->ev_used = 0;
    t->timed_launches = 0;
    return MFX_OK;
}

int mfx_trainer_export(mfx_trainer *t, float *arr, long long len)
{
    if (!t || !arr) return fail(MFX_E_ARG, "null pointer");
    if (!t->model_ready) return fail(MFX_E_STATE, "model not initialised");
    const mfx::Plan &p = t->plan;
    const long long pn = (long long)p.m * p.k, qn = (long long)p.n * p.k;
    if (len != pn + qn + 5) return fail(MFX_E_ARG, "model array length must be 5+(m+n)*k");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipDeviceSynchronize());
    if (int rcv = verify_rounds(t)) return rcv;
    DevBuf<float> tmp;
    HIP_TRY(tmp.alloc((size_t)(pn + qn)));
    const int do_scale = p.scale != 1.0f; // scale_model returns early on 1.0 (mf.cpp:531-532)
    const float f = std::sqrt(p.scale);
    HIP_TRY(mfx::launch_export(t->dP, t->dPmap.p, p.m, p.k, p.ka, f, do_scale, tmp.p,
                               grid_for(pn, t->cu_count), t->stream));
    HIP_TRY(mfx::launch_export(t->dQ, t->dQmap.p, p.n, p.k, p.ka, f, do_scale, tmp.p + pn,
                               grid_for(qn, t->cu_count), t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    HIP_TRY(hipMemcpy(arr + 5, tmp.p, (size_t)(pn + qn) * 4, hipMemcpyDeviceToHost));
    float b = p.avg / p.scale; // init_model's b (mf.cpp:3015), then scale_model (mf.cpp:536)
    if (do_scale) b *= p.scale;
    arr[0] = 0.0f; // P_L2_MFR
    arr[1] = (float)p.m;
    arr[2] = (float)p.n;
    arr[3] = (float)p.k;
    arr[4] = b;
    return MFX_OK;
}

static int parse_header(const float *a, long long len, int &m, int &n, int &k, float &b)
{
    if (!a || len < 5) return fail(MFX_E_ARG, "model array too short");
    m = (int)a[1]; // array_to_model, reference mf/mf.cpp:3455-3459
    n = (int)a[2];
    k = (int)a[3];
    b = a[4];
    if (m < 0 || n < 0 || k < 0 || len != (long long)m * k + (long long)n * k + 5)
        return fail(MFX_E_ARG, "model array length does not match its header");
    return MFX_OK;
}

// ---- the model array of utility_predict, optionally kept on the device -------------------------------------
// The reference rebuilds the model from the float array on every call (array_to_model, mf/mf.cpp:3444-3481), and by
// default so does this library: one H2D copy of the whole array per call (384 MB for configs[2], 15 ms of PCIe).  A PHP
// request loop calls utility_predict again and again with the SAME array; a caller that wants the array to stay
// resident between calls opts in -- mfx_predict_cache_enable(1), or MFX_PREDICT_CACHE=1 in the environment -- and
// thereby promises to call mfx_predict_cache_drop() after changing an array in place.  The reuse is keyed by host
// pointer, length, header and a checksum over 16 K evenly spaced words: a safety net, not a proof -- hashing every
// word would cost more than the copy it saves, which is why the reuse is not the default.
namespace {
struct ModelCache {
    std::mutex mu;
    bool enabled = g_predict_cache != 0;
    const float *host = nullptr;
    long long len = 0;
    unsigned long long sum = 0;
    float header[5] = {0, 0, 0, 0, 0};
    int device = -1;          // device that `dev` and `stream` belong to
    float *dev = nullptr;
    hipStream_t stream = nullptr;
    long long uploads = 0, hits = 0;
} g_model_cache;

unsigned long long sample_sum(const float *a, long long len)
{
    const long long samples = 16384, step = len > samples ? len / samples : 1;
    unsigned long long h = 1469598103934665603ull;
    for (long long i = 0; i < len; i += step) {
        unsigned w;
        memcpy(&w, a + i, 4);
        h = (h ^ w) * 1099511628211ull;
    }
    unsigned w;
    memcpy(&w, a + len - 1, 4);
    return (h ^ w) * 1099511628211ull;
}

// device copy of the model array on the current device; *stream = the stream the copy was made on
int resident_model(const float *model_arr, long long model_len, float **d_model, hipStream_t *stream)
{
    ModelCache &c = g_model_cache;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (c.device != dev) { // buffer and stream belong to one device: start over on another
        if (c.dev) {
            if (c.device >= 0) (void)hipSetDevice(c.device);
            (void)hipFree(c.dev);
            if (c.stream) (void)hipStreamDestroy(c.stream);
            (void)hipSetDevice(dev);
        }
        c.dev = nullptr;
        c.stream = nullptr;
        c.host = nullptr;
        c.device = dev;
    }
    if (!c.stream) HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    const unsigned long long sum = c.enabled ? sample_sum(model_arr, model_len) : 0ull;
    if (c.enabled && c.dev && c.host == model_arr && c.len == model_len && c.sum == sum &&
        memcmp(c.header, model_arr, sizeof(c.header)) == 0) {
        c.hits++;
    } else {
        if (c.dev && c.len != model_len) {
            (void)hipFree(c.dev);
            c.dev = nullptr;
        }
        if (!c.dev) HIP_TRY(hipMalloc((void **)&c.dev, (size_t)model_len * 4));
        c.host = nullptr; // not valid until the copy is through
        c.len = model_len;
        HIP_TRY(hipMemcpyAsync(c.dev, model_arr, (size_t)model_len * 4, hipMemcpyHostToDevice, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
        c.host = model_arr;
        c.sum = sum;
        memcpy(c.header, model_arr, sizeof(c.header));
        c.uploads++;
    }
    *d_model = c.dev;
    *stream = c.stream;
    return MFX_OK;
}
} // namespace

void mfx_predict_cache_enable(int on)
{
    std::lock_guard<std::mutex> lock(g_model_cache.mu);
    g_model_cache.enabled = on != 0;
    g_model_cache.host = nullptr;
}

void mfx_predict_cache_drop(void)
{
    std::lock_guard<std::mutex> lock(g_model_cache.mu);
    g_model_cache.host = nullptr; // the next call uploads again (the buffer itself is reused)
}

void mfx_predict_cache_stats(long long *uploads, long long *hits)
{
    std::lock_guard<std::mutex> lock(g_model_cache.mu);
    if (uploads) *uploads = g_model_cache.uploads;
    if (hits) *hits = g_model_cache.hits;
}

int mfx_predict_array(const float *model_arr, long long model_len, const float *pairs,
                      long long npairs, float *out)
{
    int m, n, k;
    float b;
    int rc = parse_header(model_arr, model_len, m, n, k, b);
    if (rc) return rc;
    if (npairs <= 0) return MFX_OK;
    if (!pairs || !out) return fail(MFX_E_ARG, "null pointer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MFX_E_HIP, "no HIP device: the MI355X path cannot run (there is no CPU fallback)");
    hipDeviceProp_t prop;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    std::lock_guard<std::mutex> lock(g_model_cache.mu);
    float *dModel = nullptr;
    hipStream_t s = nullptr;
    if (int rc2 = resident_model(model_arr, model_len, &dModel, &s)) return rc2;
    DevBuf<float> dPairs, dOut;
    HIP_TRY(dPairs.alloc((size_t)npairs * 2));
    HIP_TRY(dOut.alloc((size_t)npairs));
    HIP_TRY(hipMemcpyAsync(dPairs.p, pairs, (size_t)npairs * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(mfx::launch_predict(dModel, m, n, k, b, dPairs.p, npairs, dOut.p,
                                grid_for(npairs * 16, prop.multiProcessorCount), s));
    HIP_TRY(hipMemcpyAsync(out, dOut.p, (size_t)npairs * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return MFX_OK;
}

int mfx_rmse_array(const float *model_arr, long long model_len, const mfx_node *R, long long nnz,
                   double *rmse)
{
    int m, n, k;
    float b;
    int rc = parse_header(model_arr, model_len, m, n, k, b);
    if (rc) return rc;
    if (!rmse) return fail(MFX_E_ARG, "null pointer");
    if (nnz == 0) { // calc_rmse, reference mf/mf.cpp:4318-4319
        *rmse = 0;
        return MFX_OK;
    }
    if (!R) return fail(MFX_E_ARG, "null pointer");
    hipDeviceProp_t prop;
    int dev = 0, ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MFX_E_HIP, "no HIP device: the MI355X path cannot run (there is no CPU fallback)");
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    std::lock_guard<std::mutex> lock(g_model_cache.mu);
    float *dModel = nullptr;
    hipStream_t st = nullptr;
    if (int rc2 = resident_model(model_arr, model_len, &dModel, &st)) return rc2;
    DevBuf<mfx::EntryD> dR;
    DevBuf<double> dS;
    HIP_TRY(dR.alloc((size_t)nnz));
    HIP_TRY(dS.alloc(1));
    HIP_TRY(hipMemcpyAsync(dR.p, R, (size_t)nnz * 12, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(dS.p, 0, sizeof(double), st));
    HIP_TRY(mfx::launch_sq_err_nodes(dModel, m, n, k, b, dR.p, nnz, dS.p,
                                     grid_for(nnz * 16, prop.multiProcessorCount), st));
    double s = 0;
    HIP_TRY(hipMemcpyAsync(&s, dS.p, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *rmse = std::sqrt(s / (double)nnz);
    return MFX_OK;
}

struct mfx_hostplan {
    mfx::Plan plan;
};

int mfx_hostplan_build(const mfx_node *R, long long nnz, int m, int n, const mfx_options *opt,
                       mfx_hostplan **out)
{
    if (!out || !opt) return fail(MFX_E_ARG, "null pointer");
    *out = nullptr;
    if (int rc0 = check_options(*opt)) return rc0;
    if (!R || nnz <= 0 || m <= 0 || n <= 0) return fail(MFX_E_EMPTY, "train on an empty training set");
    mfx_hostplan *h = new (std::nothrow) mfx_hostplan();
    if (!h) return fail(MFX_E_NOMEM, "out of host memory");
    try {
        int wpw = 4, wgs = 1;
        const int stripes = choose_stripes(*opt, nnz, m, n, 8, 32, &wgs, &wpw); // an MI355X: 8 XCDs of 32 CUs
        mfx::build_plan((const mfx::Node *)R, nnz, m, n, plan_config(*opt, stripes, wgs, wpw, wgs_grid_for(*opt, wgs, wpw, 32)), h->plan);
    } catch (const std::bad_alloc &) {
        delete h;
        return fail(MFX_E_NOMEM, "out of host memory while building the plan");
    } catch (const std::exception &e) {
        delete h;
        return fail(MFX_E_ARG, e.what());
    }
    *out = h;
    return MFX_OK;
}

int mfx_stripes_for(const mfx_options *opt, long long nnz, int m, int n)
{
    if (!opt || nnz <= 0 || m <= 0 || n <= 0) return fail(MFX_E_ARG, "bad argument");
    if (int rc0 = check_options(*opt)) return rc0;
    int wpw = 4, wgs = 1;
    return choose_stripes(*opt, nnz, m, n, 8, 32, &wgs, &wpw); // an MI355X: 8 XCDs of 32 CUs
}

int mfx_hostplan_view(const mfx_hostplan *h, mfx_plan_view *v)
{
    if (!h || !v) return fail(MFX_E_ARG, "null pointer");
    const mfx::Plan &p = h->plan;
    memset(v, 0, sizeof(*v));
    v->m = p.m;
    v->n = p.n;
    v->k = p.k;
    v->k_aligned = p.ka;
    v->stripes = p.ns;
    v->lanes_per_rating = p.lanes;
    v->ratings_per_wave = p.groups;
    v->owner_is_q = p.owner_is_q;
    v->nnz = p.nnz;
    v->n_entries = (long long)p.entries.size();
    v->n_tasks = (long long)p.tasks.size();
    v->n_padding = p.n_padding;
    v->n_hot_rows = p.n_hot_rows;
    v->avg = p.avg;
    v->std_dev = p.std_dev;
    v->scale = p.scale;
    v->inv_scale = p.inv_scale;
    v->p_map = p.p_map.data();
    v->q_map = p.q_map.data();
    v->omega_p = p.omega_p.data();
    v->omega_q = p.omega_q.data();
    v->entries = p.entries.data();
    v->tasks = p.tasks.data();
    v->slot_task_ptr = p.slot_task_ptr.data();
    v->p_begin = p.p_begin.data();
    v->q_begin = p.q_begin.data();
    v->n_hot_slots = p.n_hot_slots;
    v->wg_tasks = p.wg_tasks.data();
    v->wg_visits = p.wg_visits.data();
    v->slot_wg_ptr = p.slot_wg_ptr.data();
    v->n_wg_tasks = (long long)p.wg_tasks.size();
    v->n_wg_visits = (long long)p.wg_visits.size();
    v->waves_per_wg = p.waves_per_wg;
    v->hot_len = p.hot_len;
    v->hot_rows = p.hot_rows.data();
    v->merge_back = p.merge_back ? 1 : 0;
    return MFX_OK;
}

int mfx_hostplan_init_factors(const mfx_hostplan *h, float *P, float *Q)
{
    if (!h || !P || !Q) return fail(MFX_E_ARG, "null pointer");
    try {
        std::vector<float> vp, vq;
        mfx::init_factors(h->plan, nullptr, nullptr, vp, vq, g_host_threads);
        memcpy(P, vp.data(), vp.size() * sizeof(float));
        memcpy(Q, vq.data(), vq.size() * sizeof(float));
        return MFX_OK;
    } catch (const std::exception &e) {
        return fail(MFX_E_NOMEM, e.what());
    }
}

void mfx_hostplan_destroy(mfx_hostplan *h) { delete h; }

int mfx_synth_host(unsigned long long seed, unsigned long long shard, long long first,
                   long long count, int m, int n, mfx_node *out)
{
    if (!out || count < 0 || m <= 0 || n <= 0) return fail(MFX_E_ARG, "bad argument");
    mfx::parallel_ranges(count, 0, [&](long long b, long long e, int) {
        for (long long i = b; i < e; ++i) {
            mfx::SynthNode s = mfx::synth_rating(seed, shard, first + i, m, n);
            out[i].u = s.u;
            out[i].v = s.v;
            out[i].r = s.r;
        }
    });
    return MFX_OK;
}

int mfx_triplets_to_device(const float *triplets, long long count, int device, void **d_nodes, int *m, int *n)
{
    if (!triplets || count <= 0 || !d_nodes || !m || !n) return fail(MFX_E_ARG, "bad argument");
    *d_nodes = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MFX_E_HIP, "no HIP device: the MI355X path cannot run (there is no CPU fallback)");
    if (device >= ndev) return fail(MFX_E_ARG, "device ordinal out of range");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    DevBuf<float> dTri;
    DevBuf<int> dMn;
    void *nodes = nullptr;
    HIP_TRY(dTri.alloc((size_t)count * 3));
    HIP_TRY(dMn.alloc(3));
    HIP_TRY(hipMalloc(&nodes, (size_t)count * sizeof(mfx_node)));
    int mn[3] = {0, 0, 0};
    hipError_t e = hipMemcpy(dTri.p, triplets, (size_t)count * 3 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(dMn.p, 0, sizeof(mn));
    if (e == hipSuccess) e = mfx::launch_triplets(dTri.p, count, nodes, dMn.p, grid_for(count, prop.multiProcessorCount), nullptr);
    if (e == hipSuccess) e = hipMemcpy(mn, dMn.p, sizeof(mn), hipMemcpyDeviceToHost);
    if (e != hipSuccess || mn[2]) {
        (void)hipFree(nodes);
        if (e != hipSuccess) return fail(MFX_E_HIP, hipGetErrorString(e));
        return fail(MFX_E_ARG, "negative id in the triplets");
    }
    *d_nodes = nodes;
    *m = mn[0];
    *n = mn[1];
    return MFX_OK;
}

int mfx_selftest_visibility(int rounds, int *result5)
{
    if (rounds < 1 || !result5) return fail(MFX_E_ARG, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MFX_E_HIP, "no HIP device: the MI355X path cannot run (there is no CPU fallback)");
    hipDeviceProp_t prop;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    DevBuf<int> words; // ticket, flag, ack (each on a line of its own), out[5]
    DevBuf<float> row;
    HIP_TRY(words.alloc(32 * 4));
    HIP_TRY(row.alloc(64));
    HIP_TRY(hipMemset(words.p, 0, 32 * 4 * sizeof(int)));
    HIP_TRY(hipMemset(row.p, 0, 64 * sizeof(float)));
    HIP_TRY(mfx::launch_visibility_probe(words.p, row.p, words.p + 32, words.p + 64, rounds, words.p + 96,
                                         prop.multiProcessorCount, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(result5, words.p + 96, 5 * sizeof(int), hipMemcpyDeviceToHost));
    return MFX_OK;
}

void mfx_device_free(void *p)
{
    if (p) (void)hipFree(p);
}

int mfx_synth_device(unsigned long long seed, unsigned long long shard, long long first,
                     long long count, int m, int n, void *out_dev, void *stream)
{
    if (!out_dev || count < 0 || m <= 0 || n <= 0) return fail(MFX_E_ARG, "bad argument");
    if (count == 0) return MFX_OK;
    hipDeviceProp_t prop;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    HIP_TRY(mfx::launch_synth(seed, shard, first, count, m, n, out_dev,
                              grid_for(count, prop.multiProcessorCount), (hipStream_t)stream));
    return MFX_OK;
}

} // extern "C"
#pragma GCC visibility pop
