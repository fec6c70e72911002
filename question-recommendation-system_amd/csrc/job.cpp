// job.cpp -- mfx_job_*: one training job over G devices of one node, inside ONE host process.
//
// The reference's utility_train is a blocking call whose parallelism is the std::thread workers of
// fpsg_core (reference mf/mf.cpp:2837-2846), entered from php_mf/mfWarp.cpp:12-16.  Here the same
// blocking call can drive G MI355X: SURVEY.md 8(e)'s process model -- one host process, G devices,
// one stream and one RCCL communicator per device (ncclCommInitAll), single node.
//
// Scheme (the one multi.py runs with one process per GPU): ratings are sharded by user
// range, so P rows have one writer and never travel; the item factors Q (+ their Adagrad slots) are
// cut into S = G item slots that travel round the ring of devices.  At step t device g trains only
// the ratings whose item lies in slot (g + t) mod S, then hands that slot to device g-1
// (ncclSend / ncclRecv, point to point over xGMI).  Every row has one writer at any time -- the
// reference scheduler's rule (no two live blocks share a stripe, mf.cpp:133-141) carried across
// devices -- so the result is ordinary SGD, not an average of replicas.
//
// RCCL is bound at run time (dlopen of librccl.so, declarations from <rccl/rccl.h>): a process that
// never asks for more than one device -- the PHP extension's default -- does not load it.
// Logical devices that map to the SAME physical device (tests on a one-GPU box) exchange their
// slots with device-to-device copies instead; that path exists for rehearsal, not for speed.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mfx.h"
#include "plan.hpp"

namespace {

thread_local std::string g_job_err;
int jfail(int code, const std::string &m)
{
    g_job_err = m;
    return code;
}
#define JOB_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return jfail(MFX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define JOB_TRY(expr)                                                                          \
    do {                                                                                      \
        int rc_ = (expr);                                                                     \
        if (rc_ != MFX_OK) return jfail(rc_, std::string(#expr) + ": " + mfx_last_error());   \
    } while (0)

// ---- RCCL, bound at run time ---------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
    bool load()
    {
        if (lib) return true;
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "");
            return false;
        }
#define BIND(field, sym)                                               \
    field = (decltype(field))dlsym(lib, sym);                          \
    if (!field) {                                                      \
        why = std::string("librccl.so lacks ") + sym;                  \
        return false;                                                  \
    }
        BIND(CommInitAll, "ncclCommInitAll")
        BIND(CommDestroy, "ncclCommDestroy")
        BIND(GroupStart, "ncclGroupStart")
        BIND(GroupEnd, "ncclGroupEnd")
        BIND(Send, "ncclSend")
        BIND(Recv, "ncclRecv")
        BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
        return true;
    }
} g_rccl;

struct Dev { // one logical device of the job
    int device = 0;                     // HIP ordinal
    hipStream_t stream = nullptr;
    int lo = 0, hi = 0;                 // user range [lo, hi)
    std::vector<mfx_trainer *> slot;    // S slot trainers over shared P (local users) and one Q slot each
    float *dP = nullptr, *dPG = nullptr, *dQS = nullptr; // P, its accumulators, the S slots [rows | accumulators]
    long long nnz = 0;
};

} // namespace

struct mfx_job {
    int G = 1, S = 1, m = 0, n = 0, k = 0, ka = 0, seg = 0;
    long long nnz = 0, slot_elems = 0, steps = 0;
    float avg = 0, std_dev = 0, scale = 1;
    bool staged = false;                // logical devices share a physical one: slots move by device-to-device copies
    mfx_trainer *single = nullptr;      // G == 1: the plain trainer, nothing else
    std::vector<Dev> dev;
    std::vector<ncclComm_t> comm;
    std::vector<std::vector<int>> q_map; // per slot: original item id (inside the slot) -> row
    std::vector<int> cnt_q;             // global item counts (seg * S)
    ~mfx_job()
    {
        if (single) mfx_trainer_destroy(single);
        for (Dev &d : dev) {
            (void)hipSetDevice(d.device);
            for (mfx_trainer *t : d.slot)
                if (t) mfx_trainer_destroy(t);
            if (d.dP) (void)hipFree(d.dP);
            if (d.dPG) (void)hipFree(d.dPG);
            if (d.dQS) (void)hipFree(d.dQS);
            if (d.stream) (void)hipStreamDestroy(d.stream);
        }
        for (ncclComm_t c : comm)
            if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
    }
};

#pragma GCC visibility push(default)
extern "C" {

const char *mfx_job_last_error(void) { return g_job_err.c_str(); }

// The ring schedule as a pure function (unit-tested on the CPU): at global step `step`, logical device `device` of
// `n_devices` trains *slot_trained; before it does, it sends the slot it trained at step-1 to *send_to and receives
// *recv_slot -- the slot it is about to train -- from *recv_from (step 0: nothing moves, *send_to = *recv_from = -1).
int mfx_job_schedule(int n_devices, int step, int device, int *slot_trained, int *send_slot, int *send_to, int *recv_slot,
                     int *recv_from)
{
    if (n_devices < 1 || device < 0 || device >= n_devices || step < 0) return jfail(MFX_E_ARG, "bad argument");
    const int S = n_devices;
    const int now = (device + step) % S;
    if (slot_trained) *slot_trained = now;
    const bool moves = n_devices > 1 && step > 0;
    if (send_slot) *send_slot = moves ? (device + step - 1) % S : -1;
    if (send_to) *send_to = moves ? (device - 1 + n_devices) % n_devices : -1;
    if (recv_slot) *recv_slot = moves ? now : -1;
    if (recv_from) *recv_from = moves ? (device + 1) % n_devices : -1;
    return MFX_OK;
}

int mfx_job_create(const mfx_node *R, long long nnz, int m, int n, const mfx_options *opt_in, int n_devices,
                   const int *device_ids, mfx_job **out)
{
    if (!out) return jfail(MFX_E_ARG, "null output handle");
    *out = nullptr;
    if (!R || nnz <= 0 || m <= 0 || n <= 0 || !opt_in) return jfail(MFX_E_EMPTY, "train on an empty training set");
    if (n_devices < 1 || n_devices > 64) return jfail(MFX_E_ARG, "device count out of range");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0)
        return jfail(MFX_E_HIP, "no HIP device: the MI355X path cannot run (there is no CPU fallback)");
    std::unique_ptr<mfx_job> j(new mfx_job());
    const int G = n_devices;
    j->G = G;
    j->S = G;
    j->m = m;
    j->n = n;
    j->nnz = nnz;
    mfx_options opt = *opt_in;
    j->k = opt.k;
    if (G == 1) { // nothing to shard: the plain trainer (bit for bit what mfx_trainer_* gives)
        if (device_ids) opt.device = device_ids[0];
        JOB_TRY(mfx_trainer_create(R, nnz, m, n, &opt, &j->single));
        JOB_TRY(mfx_trainer_init_model(j->single, nullptr));
        *out = j.release();
        return MFX_OK;
    }
    std::vector<int> ids(G);
    for (int g = 0; g < G; ++g) {
        ids[g] = device_ids ? device_ids[g] : g;
        if (ids[g] < 0 || ids[g] >= have) return jfail(MFX_E_ARG, "device ordinal out of range (fewer GPUs visible than asked for)");
    }
    {
        std::vector<int> u(ids);
        std::sort(u.begin(), u.end());
        j->staged = std::adjacent_find(u.begin(), u.end()) != u.end();
    }
    // ---- host: one common scale, global item counts, the user ranges, the slot of every rating ----
    const int S = j->S;
    int seg = (n + S - 1) / S;
    seg += (8 - seg % 8) % 8; // rows per slot: a multiple of 8 keeps every slot base 32-byte aligned
    j->seg = seg;
    // user ranges of equal RATING mass (consecutive ids): equal counts would give the device that holds the popular users a
    // fifth more ratings than the mean on a skewed stream, and the ring runs at the pace of the slowest device
    std::vector<int> ubound((size_t)G + 1, 0);
    {
        std::vector<long long> cnt_u((size_t)m, 0);
        for (long long i = 0; i < nnz; ++i) {
            const mfx_node &x = R[i];
            if (x.u < 0 || x.u >= m || x.v < 0 || x.v >= n) return jfail(MFX_E_ARG, "rating with id outside [0,m) x [0,n)");
            cnt_u[(size_t)x.u]++;
        }
        long long cum = 0;
        int g = 1;
        for (int u = 0; u < m && g < G; ++u) {
            cum += cnt_u[(size_t)u];
            while (g < G && cum >= nnz * g / G) ubound[(size_t)g++] = u + 1;
        }
        for (; g < G; ++g) ubound[(size_t)g] = m;
        ubound[(size_t)G] = m;
        for (int i = 1; i <= G; ++i) // (strictly increasing while users last, whatever the head rows weigh)
            ubound[(size_t)i] = std::min(m, std::max(ubound[(size_t)i], std::min(m, ubound[(size_t)i - 1] + 1)));
        ubound[(size_t)G] = m;
    }
    int useg = 1; // the longest range (stripe count of the job)
    for (int g = 0; g < G; ++g) useg = std::max(useg, ubound[(size_t)g + 1] - ubound[(size_t)g]);
    double s1 = 0, s2 = 0;
    j->cnt_q.assign((size_t)seg * S, 0);
    std::vector<std::vector<mfx_node>> part((size_t)G * S); // (device, slot) -> ratings with local ids
    std::vector<std::vector<int>> cnt_p(G);
    for (int g = 0; g < G; ++g) cnt_p[g].assign((size_t)(ubound[(size_t)g + 1] - ubound[(size_t)g]), 0);
    for (long long i = 0; i < nnz; ++i) {
        const mfx_node &x = R[i];
        s1 += (double)x.r;
        s2 += (double)x.r * x.r;
        const int g = (int)(std::upper_bound(ubound.begin() + 1, ubound.end(), x.u) - (ubound.begin() + 1)), s = x.v / seg;
        mfx_node y = {x.u - ubound[(size_t)g], x.v - s * seg, x.r};
        part[(size_t)g * S + s].push_back(y);
        cnt_p[g][y.u]++;
        j->cnt_q[x.v]++;
    }
    const double ex = s1 / (double)nnz, ex2 = s2 / (double)nnz;
    j->avg = (float)ex;
    j->std_dev = (float)std::sqrt(ex2 - ex * ex);
    long long smallest = nnz;
    for (const auto &p : part) smallest = std::min<long long>(smallest, (long long)p.size());
    if (smallest == 0) return jfail(MFX_E_ARG, "some device holds no rating for some item slot: fewer devices or more data");
    opt.use_stats = 1;
    opt.wide = 1; // slot trainers are small and skewed: wide launches (mfx_options.wide; parity of such shards: tests/test_gpu_multi.py)
    opt.stats_avg = j->avg;
    opt.stats_std = j->std_dev;
    if (opt.stripes <= 0) { // ONE stripe count per job (the id layout depends on it): from the smallest piece
        const int st = mfx_stripes_for(&opt, smallest, useg, seg);
        if (st <= 0) return jfail(st, mfx_last_error());
        opt.stripes = st;
    }
    // ---- devices: slot trainers over shared P, one Q slot each ----
    j->dev.resize(G);
    j->q_map.assign(S, std::vector<int>());
    for (int g = 0; g < G; ++g) {
        Dev &d = j->dev[g];
        d.device = ids[g];
        d.lo = ubound[(size_t)g];
        d.hi = ubound[(size_t)g + 1];
        JOB_HIP(hipSetDevice(d.device));
        JOB_HIP(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
        d.slot.assign(S, nullptr);
        const int mg = d.hi - d.lo;
        for (int s = 0; s < S; ++s) {
            const std::vector<mfx_node> &rs = part[(size_t)g * S + s];
            mfx_options o = opt;
            o.device = d.device;
            JOB_TRY(mfx_trainer_create_layout(rs.data(), nullptr, (long long)rs.size(), mg, seg, &o, cnt_p[g].data(),
                                              j->cnt_q.data() + (size_t)s * seg, &d.slot[s]));
            d.nnz += (long long)rs.size();
        }
        mfx_info info;
        JOB_TRY(mfx_trainer_info(d.slot[0], &info));
        j->ka = info.k_aligned;
        j->scale = info.scale;
        j->slot_elems = (long long)seg * (j->ka + 2);
        JOB_HIP(hipMalloc((void **)&d.dP, (size_t)mg * j->ka * sizeof(float)));
        JOB_HIP(hipMalloc((void **)&d.dPG, (size_t)mg * 2 * sizeof(float)));
        JOB_HIP(hipMalloc((void **)&d.dQS, (size_t)S * j->slot_elems * sizeof(float)));
        for (int s = 0; s < S; ++s) {
            float *q = d.dQS + (size_t)s * j->slot_elems;
            JOB_TRY(mfx_trainer_bind_model(d.slot[s], d.dP, q, d.dPG, q + (size_t)seg * j->ka));
        }
        for (int s = 0; s < S; ++s) // (P is written S times with the same values: same counts, same stream)
            JOB_TRY(mfx_trainer_init_model_counts(d.slot[s], cnt_p[g].data(), j->cnt_q.data() + (size_t)s * seg));
        if (g == 0)
            for (int s = 0; s < S; ++s) {
                j->q_map[s].resize(seg);
                std::vector<int> pm(mg);
                JOB_TRY(mfx_trainer_maps(d.slot[s], pm.data(), j->q_map[s].data()));
            }
    }
    if (!j->staged) {
        if (!g_rccl.load()) return jfail(MFX_E_HIP, "RCCL: " + g_rccl.why);
        j->comm.assign(G, nullptr);
        ncclResult_t r = g_rccl.CommInitAll(j->comm.data(), G, ids.data());
        if (r != ncclSuccess) return jfail(MFX_E_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
    }
    *out = j.release();
    return MFX_OK;
}

// One epoch = S steps of the ring.
int mfx_job_epoch(mfx_job *j, int slow_only)
{
    if (!j) return jfail(MFX_E_ARG, "null job");
    if (j->single) {
        JOB_TRY(mfx_trainer_epoch(j->single, slow_only, nullptr));
        return MFX_OK;
    }
    const int G = j->G, S = j->S;
    for (int t = 0; t < S; ++t, ++j->steps) {
        // the slot trained at the step before goes to the left neighbour, which trains it now
        if (j->steps > 0) {
            if (j->staged) { // rehearsal on one physical device: device-to-device copies, blocking
                for (int g = 0; g < G; ++g) JOB_HIP(hipStreamSynchronize(j->dev[g].stream));
                for (int g = 0; g < G; ++g) {
                    const int send_slot = (int)((g + j->steps - 1) % S), to = (g - 1 + G) % G;
                    JOB_HIP(hipSetDevice(j->dev[to].device));
                    JOB_HIP(hipMemcpyAsync(j->dev[to].dQS + (size_t)send_slot * j->slot_elems, j->dev[g].dQS + (size_t)send_slot * j->slot_elems,
                                           (size_t)j->slot_elems * sizeof(float), hipMemcpyDeviceToDevice, j->dev[to].stream));
                }
                for (int g = 0; g < G; ++g) JOB_HIP(hipStreamSynchronize(j->dev[g].stream));
            } else {
                ncclResult_t r = g_rccl.GroupStart();
                for (int g = 0; g < G && r == ncclSuccess; ++g) {
                    const long long st = j->steps;
                    const int send_slot = (int)((g + st - 1) % S), recv_slot = (int)((g + st) % S);
                    const int to = (g - 1 + G) % G, from = (g + 1) % G;
                    Dev &d = j->dev[g];
                    r = g_rccl.Send(d.dQS + (size_t)send_slot * j->slot_elems, (size_t)j->slot_elems, ncclFloat, to, j->comm[g], d.stream);
                    if (r == ncclSuccess)
                        r = g_rccl.Recv(d.dQS + (size_t)recv_slot * j->slot_elems, (size_t)j->slot_elems, ncclFloat, from, j->comm[g], d.stream);
                }
                ncclResult_t r2 = g_rccl.GroupEnd();
                if (r != ncclSuccess || r2 != ncclSuccess)
                    return jfail(MFX_E_HIP, std::string("RCCL slot exchange: ") + g_rccl.GetErrorString(r != ncclSuccess ? r : r2));
            }
        }
        for (int g = 0; g < G; ++g) { // every device trains the slot it holds now (asynchronous launches)
            const int s = (int)((g + j->steps) % S);
            JOB_TRY(mfx_trainer_epoch(j->dev[g].slot[s], slow_only, (void *)j->dev[g].stream));
        }
    }
    return MFX_OK;
}

// Online sum of squared errors of the last epoch over all ratings of the job (scaled units) and the common scale.
int mfx_job_last_loss(mfx_job *j, double *sum_sq, float *scale)
{
    if (!j || !sum_sq) return jfail(MFX_E_ARG, "null pointer");
    *sum_sq = 0;
    if (j->single) {
        mfx_info info;
        JOB_TRY(mfx_trainer_last_loss(j->single, sum_sq));
        JOB_TRY(mfx_trainer_info(j->single, &info));
        if (scale) *scale = info.scale;
        return MFX_OK;
    }
    for (Dev &d : j->dev)
        for (mfx_trainer *t : d.slot) {
            double x = 0;
            JOB_TRY(mfx_trainer_last_loss(t, &x));
            *sum_sq += x;
        }
    if (scale) *scale = j->scale;
    return MFX_OK;
}

int mfx_job_sync(mfx_job *j)
{
    if (!j) return jfail(MFX_E_ARG, "null job");
    if (j->single) {
        JOB_TRY(mfx_trainer_sync(j->single));
        return MFX_OK;
    }
    for (Dev &d : j->dev) {
        JOB_HIP(hipSetDevice(d.device));
        JOB_HIP(hipStreamSynchronize(d.stream));
        for (mfx_trainer *t : d.slot) JOB_TRY(mfx_trainer_sync(t));
    }
    return MFX_OK;
}

// which logical device holds the latest version of slot s after j->steps steps
static int slot_holder(const mfx_job *j, int s)
{
    if (j->steps == 0) return 0; // (every device starts from the same initial slots)
    // device g trained slot (g + steps - 1) mod S last
    return (int)(((long long)s - (j->steps - 1)) % j->S + j->S) % j->S;
}

// Training RMSE over all ratings of the job (calc_rmse formula, mf.cpp:4316-4331): every device's slot trainers on
// the latest version of their slots.
int mfx_job_rmse(mfx_job *j, double *rmse)
{
    if (!j || !rmse) return jfail(MFX_E_ARG, "null pointer");
    if (j->single) {
        JOB_TRY(mfx_trainer_rmse(j->single, rmse));
        return MFX_OK;
    }
    JOB_TRY(mfx_job_sync(j));
    std::vector<float> host((size_t)j->slot_elems);
    double sse = 0;
    for (int s = 0; s < j->S; ++s) { // bring the latest copy of the slot to every device (through the host: off the hot path)
        const int h = slot_holder(j, s);
        JOB_HIP(hipSetDevice(j->dev[h].device));
        JOB_HIP(hipMemcpy(host.data(), j->dev[h].dQS + (size_t)s * j->slot_elems, host.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int g = 0; g < j->G; ++g) {
            if (g == h) continue;
            JOB_HIP(hipSetDevice(j->dev[g].device));
            JOB_HIP(hipMemcpy(j->dev[g].dQS + (size_t)s * j->slot_elems, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    for (Dev &d : j->dev)
        for (mfx_trainer *t : d.slot) {
            double x = 0;
            JOB_TRY(mfx_trainer_sq_err(t, &x));
            sse += x;
        }
    *rmse = std::sqrt(sse / (double)j->nnz);
    return MFX_OK;
}

// scale_model + shrink_model + shuffle_model + model_to_array (mf.cpp:529-553, 1057-1074, 1027-1055, 3415-3441) over the
// pieces of the job: [fun, m, n, k, b, P (m*k), Q (n*k)] in original ids.
int mfx_job_export(mfx_job *j, float *arr, long long len)
{
    if (!j || !arr) return jfail(MFX_E_ARG, "null pointer");
    if (j->single) {
        JOB_TRY(mfx_trainer_export(j->single, arr, len));
        return MFX_OK;
    }
    const long long pn = (long long)j->m * j->k, qn = (long long)j->n * j->k;
    if (len != pn + qn + 5) return jfail(MFX_E_ARG, "model array length must be 5+(m+n)*k");
    JOB_TRY(mfx_job_sync(j));
    const int k = j->k, ka = j->ka;
    const bool do_scale = j->scale != 1.0f; // scale_model returns early on 1.0 (mf.cpp:531-532)
    const float f = std::sqrt(j->scale);
    for (Dev &d : j->dev) { // P: every device's users, through the user map of its trainers
        const int mg = d.hi - d.lo;
        std::vector<float> P((size_t)mg * ka);
        std::vector<int> pm(mg), qm(j->seg);
        JOB_HIP(hipSetDevice(d.device));
        JOB_HIP(hipMemcpy(P.data(), d.dP, P.size() * sizeof(float), hipMemcpyDeviceToHost));
        JOB_TRY(mfx_trainer_maps(d.slot[0], pm.data(), qm.data()));
        for (int u = 0; u < mg; ++u)
            for (int c = 0; c < k; ++c) {
                const float x = P[(size_t)pm[u] * ka + c];
                arr[5 + (size_t)(d.lo + u) * k + c] = do_scale ? x * f : x;
            }
    }
    std::vector<float> Qs((size_t)j->slot_elems);
    for (int s = 0; s < j->S; ++s) {
        const int h = slot_holder(j, s);
        JOB_HIP(hipSetDevice(j->dev[h].device));
        JOB_HIP(hipMemcpy(Qs.data(), j->dev[h].dQS + (size_t)s * j->slot_elems, Qs.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int v = 0; v < j->seg; ++v) {
            const long long id = (long long)s * j->seg + v;
            if (id >= j->n) break;
            for (int c = 0; c < k; ++c) {
                const float x = Qs[(size_t)j->q_map[s][v] * ka + c];
                arr[5 + pn + (size_t)id * k + c] = do_scale ? x * f : x;
            }
        }
    }
    float b = j->avg / j->scale; // init_model's b (mf.cpp:3015), then scale_model (mf.cpp:536)
    if (do_scale) b *= j->scale;
    arr[0] = 0.0f; // P_L2_MFR
    arr[1] = (float)j->m;
    arr[2] = (float)j->n;
    arr[3] = (float)j->k;
    arr[4] = b;
    return MFX_OK;
}

void mfx_job_destroy(mfx_job *j) { delete j; }

} // extern "C"
#pragma GCC visibility pop
