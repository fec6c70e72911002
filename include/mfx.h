/* mfx.h -- thin HIP C-ABI of the MI355X-native SGD matrix-factorisation trainer.
 *
 * Plain C: pointers, sizes, an opaque handle.  No torch / C++ types.  Every entry
 * returns 0 on success and a negative mfx_status on failure (message via
 * mfx_last_error()); nothing throws or aborts across this boundary.
 *
 * Where an entry replaces something in the reference it says so
 * (paths relative to /root/reference).  The reference has no device boundary at
 * all: these entries are what mf/mf.cpp's fpsg()/fpsg_core() pipeline
 * (mf.cpp:2774-3042) looks like once the per-rating loop (mf.cpp:1201-1238,
 * 1462-1548, 1720-1728) lives in HBM.  include/mf.h keeps the reference's own
 * C++ surface on top of them; INTEGRATION.md shows the bindings.
 */
#ifndef MFX_H
#define MFX_H

#ifdef __cplusplus
extern "C" {
#endif

#define MFX_ABI_VERSION 1

enum mfx_status {
    MFX_OK = 0,
    MFX_E_ARG = -1,      /* bad argument (check_parameter, mf.cpp:3115-3184)          */
    MFX_E_EMPTY = -2,    /* empty training set (mf.cpp:2792-2796)                     */
    MFX_E_HIP = -3,      /* HIP runtime error / no gfx950 device                      */
    MFX_E_NOMEM = -4,
    MFX_E_UNSUPPORTED = -5, /* k too large for the kernel family, lambda1 != 0, nmf   */
    MFX_E_STATE = -6
};

/* mf_node, mf/mf.h:36-41: 12 bytes, no padding. */
typedef struct mfx_node { int u; int v; float r; } mfx_node;

typedef struct mfx_trainer mfx_trainer;

typedef struct mfx_options {
    int k;             /* latent factors (mf_parameter.k, mf/mf.h:54); 1 .. 1024       */
    float lambda_p2;   /* mf/mf.h:59                                                   */
    float lambda_q2;   /* mf/mf.h:61                                                   */
    float eta;         /* mf/mf.h:62                                                   */
    int device;        /* HIP ordinal; -1 = current device                             */
    int stripes;       /* stripe count per side = launches per epoch; 0 = one per XCD  */
    int wg_per_cu;     /* resident 256-thread workgroups per CU; 0 = auto              */
    int task_steps;    /* ratings per lane-group per task; 0 = auto                    */
    int no_swap;       /* 1: leave the heavy rows of the GATHERED side on the lock-free side.  Default 0: their ratings run in
                          workgroup tasks with the roles swapped (DESIGN.md "Heavy rows"; tests and A/B runs set it -- with it the
                          popular rows of that side lose most of their accumulator growth, and streams with a heavy head
                          can overflow at k <= 16: measured on the bench stream, profiles/experiments/r03_k_sweep_100M.log) */
    int rk_mode;       /* 0: 1/8 for both accumulator slots (SSE build as shipped,
                          mf.cpp:1233-1234); 1: 1/(k_a-8) for slot 1 (mf.cpp:1314-1315) */
    int owner_side;    /* 0 auto (side with fewer rows), 1 users (P), 2 items (Q)      */
    int identity_maps; /* id layout: 0 mass-balanced stripes (default), 1 identity (tests),
                          2 the reference's shuffle + equal ranges (mf.cpp:1009-1017, 802) */
    int use_stats;     /* 1: take mean / std-dev from the next two fields instead of
                          collect_info (mf.cpp:462-484) -- a problem split over several
                          trainers or ranks must be scaled by ONE common figure        */
    float stats_avg;
    float stats_std;
    int conflict_div;  /* the lock-free side: at most (rows of one stripe of it) / conflict_div ratings in flight per XCD.
                          0 = default 32.  Two lists that read-modify-write one row at the same time keep one of the two
                          steps; the share of steps lost that way is about 2 x in flight / rows, and so is the price in
                          final RMSE (uniform 100 k x 50 k ids, 12 epochs: +3.4 % at 12, +1.5 % at 32, +1.0 % at 48 -- and the
                          epoch takes 0.74 / 1.9 / 2.8 ms).  Large problems are not touched by it (the occupancy cap binds
                          first); a caller who wants round 2's speed on 10 M-rating problems sets 12.                     */
    int wide;          /* 1: a WIDE launch -- where the concurrency cap holds a launch below what the chip runs anyway, the idle
                          workgroups take the heavy rows (the lists of one heavy row and of its copies never meet each other on a
                          row of the other side).  Twice to four times the speed on skewed 10 M-rating problems (configs[1] 3.4 ->
                          1.8 ms per epoch at -0.6 % / -0.1 %), but the rows of the other side are updated that much more often per
                          unit of time and lose accordingly more updates: +6.0 % final RMSE on a Zipf(1.1) law at k = 64 where the
                          default stays at +2.4 %.  Default 0.  (DESIGN.md "Wide launches")                                        */
} mfx_options;

typedef struct mfx_info {
    int m, n, k, k_aligned;
    long long nnz;
    float avg, std_dev, scale;       /* collect_info, mf.cpp:462-484; scale mf.cpp:2999  */
    float lambda_p_scaled, lambda_q_scaled; /* mf.cpp:2805-2806                         */
    int stripes, lanes_per_rating, ratings_per_wave;
    int owner_is_q;
    long long n_entries;             /* nnz + padding slots                              */
    long long n_tasks;
    long long n_hot_rows;            /* visits of heavy rows (workgroup visits)          */
    int cu_count, xcd_count, wg_per_cu;
    void *dP, *dQ, *dPG, *dQG;       /* device pointers (internal ids, k_aligned stride) */
    double bytes_per_rating;         /* algorithmic: 16*k_aligned + 44    */
    long long n_wg_tasks, n_wg_visits; /* workgroup tasks (heavy rows) and their visits  */
    long long n_hot_slots;           /* rows split over several workgroups somewhere     */
    long long hot_acc_bytes;         /* HBM held by their combine slots                  */
    int waves_per_wg, hot_len;       /* waves of a workgroup that take work; a row with more ratings in a block is heavy */
    int merge_back;                  /* 1: visits write back memory-now + their change (rows spend much of a launch in registers) */
    int grid_wg_per_cu;              /* workgroups per CU a launch starts (> wg_per_cu: a wide launch, the rest run the heavy rows) */
} mfx_info;

int mfx_abi_version(void);
const char *mfx_last_error(void);
int mfx_device_count(void);
void mfx_default_options(mfx_options *opt);

/* Pre-processing + upload.  Replaces fpsg() up to init_model (mf.cpp:2972-3016):
 * collect_info, gen_random_map, shuffle_problem, scale_problem, and a GPU stripe/task
 * layout in place of grid_problem.  R is borrowed host memory (copy_data semantics). */
int mfx_trainer_create(const mfx_node *R_host, long long nnz, int m, int n,
                       const mfx_options *opt, mfx_trainer **out);
/* Same, ratings already resident in HBM (device pointer). */
int mfx_trainer_create_device(const void *R_dev, long long nnz, int m, int n,
                              const mfx_options *opt, mfx_trainer **out);
/* Same, with the id layout taken from the caller's row counts (per ORIGINAL id, m and n ints; NULL = the
 * data's own) instead of this trainer's ratings: trainers that share factor rows -- the stripe trainers of
 * one rank share P, the ranks of a job exchange Q stripes (multi.py) -- must place every id in the same
 * row, and they do when they are given the same counts.  Exactly one of R_host / R_dev is non-NULL. */
int mfx_trainer_create_layout(const mfx_node *R_host, const void *R_dev, long long nnz, int m, int n,
                              const mfx_options *opt, const int *layout_cnt_p, const int *layout_cnt_q,
                              mfx_trainer **out);
void mfx_trainer_destroy(mfx_trainer *t);
/* Stripe count (= launches per epoch) a trainer would choose for a problem of this size on an MI355X
 * (8 XCDs of 32 CUs); > 0, or a negative mfx_status.  Trainers that share factor rows must be given ONE
 * stripe count (mfx_options.stripes): the id layout depends on it.  multi.py takes it from the smallest
 * piece of the job. */
int mfx_stripes_for(const mfx_options *opt, long long nnz, int m, int n);

/* Use caller-owned device buffers for the factors (k_aligned stride, internal ids):
 * P m*k_a, Q n*k_a, PG 2m, QG 2n floats.  Lets a host framework hand the same memory
 * to RCCL.  Must precede mfx_trainer_init_model. */
int mfx_trainer_bind_model(mfx_trainer *t, void *dP, void *dQ, void *dPG, void *dQG);

/* init_model (mf.cpp:952-1007) + the accumulator fill (mf.cpp:2835), bit-identical to
 * the reference's stream.  omega_q_override (host, n ints in ORIGINAL item ids, may be NULL)
 * replaces the local item counts when several ranks share Q. */
int mfx_trainer_init_model(mfx_trainer *t, const int *omega_q_override);
/* Same with both count vectors given (host, ORIGINAL ids, m resp. n ints, either may be NULL =
 * local counts): rows with a zero count start as NaN, the others draw from the stream. */
int mfx_trainer_init_model_counts(mfx_trainer *t, const int *omega_p, const int *omega_q);

/* One SGD epoch = `stripes` kernel launches on `stream` (hipStream_t as void*, NULL =
 * the trainer's own stream).  slow_only = 1 reproduces epoch 0 (mf.cpp:2834, 1230-1231).
 * Asynchronous. */
int mfx_trainer_epoch(mfx_trainer *t, int slow_only, void *stream);
/* The same epoch in `nparts` pieces (rounds [part*stripes/nparts, (part+1)*stripes/nparts)), so a
 * multi-GPU host can exchange the replicated factors more than once per epoch.  Call the parts in
 * order 0..nparts-1. */
int mfx_trainer_epoch_part(mfx_trainer *t, int slow_only, void *stream, int part, int nparts);
int mfx_trainer_sync(mfx_trainer *t);

/* Online sum of squared errors of the last finished epoch in scaled units
 * (the Scheduler::get_loss figure, mf.cpp:237-241).  Synchronises. */
int mfx_trainer_last_loss(mfx_trainer *t, double *sum_sq);
/* calc_reg2 (mf.cpp:608-633) on the device, scaled units. */
int mfx_trainer_reg2(mfx_trainer *t, double *reg);
/* Training-set RMSE of the current factors in original rating units
 * (calc_rmse formula, mf.cpp:4316-4331).  Synchronises. */
int mfx_trainer_rmse(mfx_trainer *t, double *rmse);
/* The sum behind it, in original rating units squared (to pool several trainers / ranks). */
int mfx_trainer_sq_err(mfx_trainer *t, double *sum_sq);

int mfx_trainer_info(mfx_trainer *t, mfx_info *info);
/* host copies of the id permutations (gen_random_map, mf.cpp:1009-1017): m and n ints */
int mfx_trainer_maps(mfx_trainer *t, int *p_map, int *q_map);
/* the stripe/task layout as it sits in HBM (tests): n_entries*12 B, n_tasks*16 B, stripes^2+1 longs */
int mfx_trainer_plan_copy(mfx_trainer *t, void *entries, void *tasks, long long *slot_task_ptr);
/* ... and its workgroup tasks: n_wg_tasks*24 B, n_wg_visits*24 B, stripes^2+1 longs (mfx_plan_view says what they hold) */
int mfx_trainer_plan_copy_wg(mfx_trainer *t, void *wg_tasks, void *wg_visits, long long *slot_wg_ptr);
/* raw factors in internal layout, for tests and checkpoints */
int mfx_trainer_get_model(mfx_trainer *t, float *P, float *Q, float *PG, float *QG);
int mfx_trainer_set_model(mfx_trainer *t, const float *P, const float *Q, const float *PG,
                          const float *QG);
/* A raw model is expressed in the trainer's INTERNAL layout (id maps, stripe count, padded width): a checkpoint
 * must carry the fingerprint of that layout and the number of epochs done (the first round of an epoch rotates
 * with it), and a restoring trainer must compare before mfx_trainer_set_model (the Python binding's
 * Trainer.checkpoint / Trainer.restore do). */
int mfx_trainer_layout_fingerprint(mfx_trainer *t, unsigned long long *fp);
long long mfx_trainer_epochs_done(mfx_trainer *t);
int mfx_trainer_set_epochs_done(mfx_trainer *t, long long epochs);

/* HIP-event timing of the epoch launches (events on the launch stream).  After
 * mfx_trainer_sync: number of launches timed since enable and their summed duration. */
int mfx_trainer_timing_enable(mfx_trainer *t, int on);
int mfx_trainer_timing_read(mfx_trainer *t, long long *launches, double *total_ms);

/* scale_model + shrink_model + shuffle_model + model_to_array
 * (mf.cpp:529-553, 1057-1074, 1027-1055, 3415-3441): writes
 * [fun,m,n,k,b,P(m*k),Q(n*k)] to host memory; len must equal 5+(m+n)*k. */
int mfx_trainer_export(mfx_trainer *t, float *model_arr, long long len);

/* utility_predict's loop (mf.cpp:3537-3568, mf_predict 4295-4314) batched on the
 * device: pairs = (u,v) as floats, out = float[npairs].  Host buffers. */
int mfx_predict_array(const float *model_arr, long long model_len, const float *pairs,
                      long long npairs, float *out);
/* By default the model array is uploaded on every call, like the reference's array_to_model (mf.cpp:3444-3481).
 * mfx_predict_cache_enable(1) (or MFX_PREDICT_CACHE=1 in the environment) keeps the array of the last
 * mfx_predict_array / mfx_rmse_array call resident in HBM: a call with the same host pointer, length, header and
 * sampled checksum then skips the upload.  The checksum reads 16 K words, not all of them: a caller that enables the
 * reuse must call mfx_predict_cache_drop() after changing an array in place.  _stats counts uploads and hits. */
void mfx_predict_cache_enable(int on);
void mfx_predict_cache_drop(void);
void mfx_predict_cache_stats(long long *uploads, long long *hits);
/* calc_rmse (mf.cpp:4316-4331) of a facade array on host ratings, on the device. */
int mfx_rmse_array(const float *model_arr, long long model_len, const mfx_node *R,
                   long long nnz, double *rmse);

/* Host-only view of the pre-processing result (no device needed): lets the CPU test
 * suite check the id maps, row counts, statistics, initial factors and the stripe/task
 * layout against the oracle.  Pointers stay valid until mfx_hostplan_destroy. */
typedef struct mfx_hostplan mfx_hostplan;
typedef struct mfx_plan_view {
    int m, n, k, k_aligned, stripes, lanes_per_rating, ratings_per_wave, owner_is_q;
    long long nnz, n_entries, n_tasks, n_padding, n_hot_rows;
    float avg, std_dev, scale, inv_scale;
    const int *p_map, *q_map, *omega_p, *omega_q;
    const void *entries;            /* {uint32 own|boundary<<31, int32 gat(-1 = pad), float r} */
    const void *tasks;              /* {uint64 entry_off, uint32 nsteps, uint32 pad}            */
    const long long *slot_task_ptr; /* stripes*stripes+1, ordered (round, slot)                */
    const int *p_begin, *q_begin;   /* stripes+1 internal-id boundaries of the user / item stripes */
    long long n_hot_slots;          /* rows that are split over several workgroups somewhere (combine slots of the kernel) */
    /* workgroup tasks: the heavy rows (more than hot_len ratings in a block, either side).  A task = nvisits visits run one
     * after the other by the waves_per_wg x ratings_per_wave lists of ONE workgroup on one LDS copy of the row; its entries
     * are stored wave-major (wave w: [off + w*nsteps*G, off + (w+1)*nsteps*G), step-major inside).                        */
    const void *wg_tasks;           /* {uint64 off, uint32 nsteps, visit0, nvisits, swapped}                              */
    const void *wg_visits;          /* {uint32 row, nsteps, len, info = copies << 1 | swapped, slot, pad}                 */
    const long long *slot_wg_ptr;   /* stripes*stripes+1, ordered (round, slot)                                           */
    long long n_wg_tasks, n_wg_visits;
    int waves_per_wg, hot_len;
    const int *hot_rows;            /* combine slot -> internal row | side << 31 (1 = the plan's gathered side)           */
    int merge_back;                 /* 1: visits write back what memory holds at their end plus what they changed          */
} mfx_plan_view;
int mfx_hostplan_build(const mfx_node *R_host, long long nnz, int m, int n,
                       const mfx_options *opt, mfx_hostplan **out);
int mfx_hostplan_view(const mfx_hostplan *h, mfx_plan_view *view);
/* init_model (mf.cpp:952-1007): P m*k_aligned, Q n*k_aligned floats, internal ids */
int mfx_hostplan_init_factors(const mfx_hostplan *h, float *P, float *Q);
void mfx_hostplan_destroy(mfx_hostplan *h);

/* read_triplet (mf/mf.cpp:3367-3394) without the host pass: `count` float triples (u, v, r) in host memory are
 * uploaded once and turned into an mfx_node array in HBM (ids truncated like the reference's (mf_int) cast);
 * *m = max u + 1, *n = max v + 1.  A negative id fails with MFX_E_ARG.  Feed the array to
 * mfx_trainer_create_device; release it with mfx_device_free. */
int mfx_triplets_to_device(const float *triplets, long long count, int device, void **d_nodes, int *m, int *n);
void mfx_device_free(void *p);

/* Self-test of the memory behaviour the lock-free side of the kernel relies on: a row stored by one CU is seen by the
 * non-temporal loads of another CU of the same XCD (kernels.hip: visibility_probe).  result5 = {rounds completed,
 * stale rows seen, polls that ran out, CU of the writer, CU of the reader}; a healthy device gives {rounds, 0, 0, a, b}. */
int mfx_selftest_visibility(int rounds, int *result5);

/* ---- one job over G devices of one node, inside this process (csrc/job.cpp) --------------------------------
 * What utility_train's worker threads are in the reference (std::thread, mf/mf.cpp:2837-2846, entered from
 * php_mf/mfWarp.cpp:12-16): the parallelism behind ONE blocking call.  Ratings are sharded by user range over the
 * devices (P rows never travel); the item factors are cut into G slots that go round the ring of devices --
 * ncclSend / ncclRecv on one stream and one RCCL communicator per device (ncclCommInitAll), point to point over
 * xGMI -- so every row has one writer at any time: ordinary SGD, the reference scheduler's rule (mf.cpp:133-141)
 * across devices.  RCCL is bound at run time (dlopen), only when n_devices > 1.  device_ids: HIP ordinals (NULL = 0 ..
 * n_devices-1); ordinals that repeat (tests on a one-GPU box) make the slots move by device-to-device copies
 * instead of RCCL.  n_devices = 1 is the plain trainer.  mf::utility_train takes this path when MFX_DEVICES > 1.
 * G > 1 over RCCL has not run on hardware yet (one GPU per test box): unmeasured. */
typedef struct mfx_job mfx_job;
int mfx_job_create(const mfx_node *R_host, long long nnz, int m, int n, const mfx_options *opt, int n_devices,
                   const int *device_ids, mfx_job **out);
int mfx_job_epoch(mfx_job *j, int slow_only);   /* G steps of the ring; asynchronous */
int mfx_job_sync(mfx_job *j);
int mfx_job_last_loss(mfx_job *j, double *sum_sq, float *scale);
int mfx_job_rmse(mfx_job *j, double *rmse);
int mfx_job_export(mfx_job *j, float *model_arr, long long len);
void mfx_job_destroy(mfx_job *j);
const char *mfx_job_last_error(void);
/* The ring schedule as a pure function (unit-tested on the CPU): at global step `step` device `device` trains
 * *slot_trained; before that it sends *send_slot (the slot it trained at step-1) to *send_to and receives *recv_slot
 * -- the one it trains now -- from *recv_from (-1 at step 0 or with one device: nothing moves). */
int mfx_job_schedule(int n_devices, int step, int device, int *slot_trained, int *send_slot, int *send_to, int *recv_slot,
                     int *recv_from);

/* Deterministic synthetic ratings: integer-only generator, identical on
 * host and device.  Writes ratings [first, first+count) of shard `shard` of problem `seed`:
 * a shard is one GPU's user range (m users of its own, the n items shared by all shards);
 * shard 0 alone is the single-GPU problem. */
int mfx_synth_host(unsigned long long seed, unsigned long long shard, long long first,
                   long long count, int m, int n, mfx_node *out);
int mfx_synth_device(unsigned long long seed, unsigned long long shard, long long first,
                     long long count, int m, int n, void *out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MFX_H */
