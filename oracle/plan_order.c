/* plan_order.c -- the oracle's per-rating arithmetic (orc_sgd_one, pinned bit-exact to the reference)
 * applied in the ORDER OF THE GPU PLAN.  TEST INFRASTRUCTURE ONLY (see the header of mf_oracle.c).
 *
 * The GPU trainer visits the ratings in another order than the reference (stripe rounds instead of the
 * block scheduler of reference mf/mf.cpp:49-312; the lists of a block advance side by side instead of one
 * thread walking a sorted block, mf.cpp:1201-1238; a heavy row's ratings are dealt over all lists of one
 * workgroup).  SGD is order dependent, so a difference in final RMSE between the GPU path and orc_train
 * mixes two things: arithmetic and order.  This file separates them: it walks the plan's own entry stream
 * (mfx_plan_view: entries, wave tasks, workgroup tasks and their visits) on one CPU thread -- rounds in the
 * kernel's order, every list of a block in lock step, owner rows of the wave tasks held in a private copy for
 * the length of a visit exactly as the kernel holds them in registers, the heavy row of a workgroup visit in
 * ONE copy that all lists of the workgroup add to, exactly as the kernel keeps it in LDS -- with the oracle's
 * update.  What is left between this and the GPU is the lock-free execution itself.
 *
 * mode says what happens to the heavy rows (workgroup visits):
 *   1  as the kernel: one copy per workgroup visit, every wave of the workgroup reads it once per step and its G
 *      lists add what they change; a row split over several workgroups is folded with fold_hot_rows' formula
 *   2  the sequential meaning of the order: every rating of a heavy row updates the row in memory at once, and the G lists
 *      of a wave take their turn one after the other (modes 1 and 3: one step of a wave is one burst of loads followed by
 *      one burst of stores, as on the GPU)
 *   3  (study) as 1, but a split row becomes the MEAN of its copies
 *   4  (study) as 1, but every rating sees the copy as the rating before it left it (no W x G ratings in flight)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mf_oracle.h"

typedef struct { uint32_t own; int32_t gat; float r; } pl_entry;
typedef struct { uint64_t off; uint32_t nsteps; uint32_t pad; } pl_task;
typedef struct { uint64_t off; uint32_t nsteps; uint32_t visit0; uint32_t nvisits; uint32_t swapped; } pl_wgtask;
typedef struct { uint32_t row; uint32_t nsteps; uint32_t len; uint32_t info; uint32_t slot; uint32_t pad; } pl_wgvisit;

#define IDMASK 0x3FFFFFFFu
#define GAT_ID 0x3FFFFFFF  /* bit 30 of gat: the row is read, not written (plan.hpp ENTRY_READ_ONLY) */
#define GAT_RO 0x40000000

typedef struct {
    uint32_t cur; /* owner row of the visit in progress, 0xFFFFFFFF = none */
    float og[2], og0[2];
    float *o;     /* private copy of the owner row (ka floats), followed by the state the visit started from */
} pl_list;

typedef struct {
    int visit;      /* index of the visit in progress inside the task, -1 before the first */
    uint32_t until; /* first step after it */
    float *copy;    /* ka floats + 2 accumulator slots: the "LDS" copy; then the same again: the state it started from */
    float tsum0, tsum;
} pl_wg;

static float damp(float S) { return S > 1e-3f ? (1.0f - expf(-S)) / S : 1.0f - 0.5f * S; }

int orc_plan_order_train(const void *entries_v, const void *tasks_v, const long long *slot_task_ptr, const void *wg_tasks_v,
                         const void *wg_visits_v, const long long *slot_wg_ptr, const unsigned *hot_rows, int ns, int G, int W,
                         int ka, int owner_is_q, float *P, float *Q, float *PG, float *QG, long long n_hot_slots,
                         float lambda_p, float lambda_q, float eta, int epochs, int first_epoch, int mode,
                         int rsqrt_mode, int rk_mode, double *epoch_loss)
{
    const pl_entry *entries = (const pl_entry *)entries_v;
    const pl_task *tasks = (const pl_task *)tasks_v;
    const pl_wgtask *wgt = (const pl_wgtask *)wg_tasks_v;
    const pl_wgvisit *wgv = (const pl_wgvisit *)wg_visits_v;
    float *own_rows = owner_is_q ? Q : P, *gat_rows = owner_is_q ? P : Q;
    float *own_acc = owner_is_q ? QG : PG, *gat_acc = owner_is_q ? PG : QG;
    const float lam_own = owner_is_q ? lambda_q : lambda_p, lam_gat = owner_is_q ? lambda_p : lambda_q;
    float *const own_rows_ = own_rows, *const gat_rows_ = gat_rows, *const own_acc_ = own_acc, *const gat_acc_ = gat_acc;
    const float lam_own_ = lam_own, lam_gat_ = lam_gat;
    const float rk1 = (rk_mode == ORC_RK_AS_BUILT || ka == 8) ? 0.125f : 1.0f / (float)(ka - 8);
    const size_t nslots = (size_t)(n_hot_slots > 0 ? n_hot_slots : 1);
    float *hot_acc = (float *)calloc(nslots * (size_t)(ka + 5), sizeof(float)); /* row, 2 acc, errors, ratings, copies */
    long long max_tasks = 0, max_wg = 0;
    int i, ep, merge_back = 0;
    pl_list *lists;
    pl_wg *wgs;
    float *copies, *wcopies, *snap = (float *)malloc(sizeof(float) * (size_t)(ka + 2)), *tmp = (float *)malloc(sizeof(float) * (size_t)(ka + 2));
    float *gsnap = (float *)malloc(sizeof(float) * (size_t)(ka + 2) * (size_t)G);
    /* plans that run heavy rows of the GATHERED side with the roles swapped: those visits read-modify-write owner rows
       while wave tasks hold them in registers, so the wave tasks (and one-copy workgroup visits) write back by merging */
    merge_back = (mode >> 8) & 1; /* (the plan says whether its visits write back by merging: mfx_plan_view.merge_back) */
    mode &= 0xFF;
    for (i = 0; i < ns * ns; i++) {
        if (slot_task_ptr[i + 1] - slot_task_ptr[i] > max_tasks) max_tasks = slot_task_ptr[i + 1] - slot_task_ptr[i];
        if (slot_wg_ptr[i + 1] - slot_wg_ptr[i] > max_wg) max_wg = slot_wg_ptr[i + 1] - slot_wg_ptr[i];
    }
    lists = (pl_list *)malloc((size_t)(max_tasks > 0 ? max_tasks : 1) * G * sizeof(pl_list));
    copies = (float *)malloc((size_t)(max_tasks > 0 ? max_tasks : 1) * G * 2 * ka * sizeof(float));
    wgs = (pl_wg *)malloc((size_t)(max_wg > 0 ? max_wg : 1) * sizeof(pl_wg));
    wcopies = (float *)malloc((size_t)(max_wg > 0 ? max_wg : 1) * 2 * (size_t)(ka + 2) * sizeof(float));

    for (ep = first_epoch; ep < first_epoch + epochs; ep++) {
        const int slow = ep == 0;
        double loss = 0;
        int ri;
        for (ri = 0; ri < ns; ri++) {
            const int r = (ri + ep) % ns; /* the kernel rotates the first round per epoch (trainer.cpp) */
            int s;
            for (s = 0; s < ns; s++) {
                const long long bi = (long long)r * ns + s;
                const long long tbeg = slot_task_ptr[bi], tend = slot_task_ptr[bi + 1];
                const long long wbeg = slot_wg_ptr[bi], wend = slot_wg_ptr[bi + 1];
                const long long nl = (tend - tbeg) * G;
                long long l, step, maxsteps = 0, t;
                for (l = 0; l < nl; l++) {
                    lists[l].cur = 0xFFFFFFFFu;
                    lists[l].o = copies + l * 2 * ka;
                }
                for (t = wbeg; t < wend; t++) {
                    wgs[t - wbeg].visit = -1;
                    wgs[t - wbeg].until = 0;
                    wgs[t - wbeg].copy = wcopies + (t - wbeg) * 2 * (ka + 2);
                    wgs[t - wbeg].tsum = 0;
                    if (wgt[t].nsteps > maxsteps) maxsteps = wgt[t].nsteps;
                }
                for (t = tbeg; t < tend; t++)
                    if (tasks[t].nsteps > maxsteps) maxsteps = tasks[t].nsteps;
                for (step = 0; step <= maxsteps * (getenv("ORC_SEQ_PHASES") ? 2 : 1) + 1; step++) {
                    const int seq = getenv("ORC_SEQ_PHASES") != NULL;
                    const long long wstep = step, tstep = seq ? step - maxsteps - 1 : step;
                    /* ---- workgroup tasks: the heavy rows ---- */
                    for (t = wbeg; t < wend && wstep <= maxsteps; t++) {
                        pl_wg *Wg = &wgs[t - wbeg];
                        const pl_wgtask *T = &wgt[t];
                        const int sw = (int)T->swapped;
                        float *o_rows = sw ? gat_rows : own_rows, *o_acc = sw ? gat_acc : own_acc;
                        float *g_rows = sw ? own_rows : gat_rows, *g_acc = sw ? own_acc : gat_acc;
                        const float lo = sw ? lam_gat : lam_own, lg = sw ? lam_own : lam_gat;
                        int w, g, d;
                        if (step > T->nsteps) continue;
                        if ((uint32_t)step == Wg->until) { /* a visit ends here (and the next one starts) */
                            if (Wg->visit >= 0 && mode != 2) {
                                const pl_wgvisit *V = &wgv[T->visit0 + Wg->visit];
                                const unsigned ncop = V->info >> 1;
                                float *rowp = o_rows + (size_t)V->row * ka, *accp = o_acc + (size_t)V->row * 2;
                                const float *c = Wg->copy, *c0 = Wg->copy + ka + 2;
                                if (ncop <= 1 && merge_back) {
                                    for (d = 0; d < ka; d++) rowp[d] += c[d] - c0[d];
                                    accp[0] += c[ka] - c0[ka];
                                    accp[1] += c[ka + 1] - c0[ka + 1];
                                } else if (ncop <= 1) { /* the one copy of the row is written back (like the owner row of a wave task) */
                                    memcpy(rowp, c, sizeof(float) * ka);
                                    accp[0] = c[ka];
                                    accp[1] = c[ka + 1];
                                } else {
                                    float *slot = hot_acc + (size_t)V->slot * (ka + 5);
                                    for (d = 0; d < ka; d++) slot[d] += c[d] - c0[d];
                                    slot[ka] += c[ka] - c0[ka];
                                    slot[ka + 1] += c[ka + 1] - c0[ka + 1];
                                    slot[ka + 2] += Wg->tsum - Wg->tsum0;
                                    slot[ka + 3] += (float)V->len;
                                    slot[ka + 4] += 1.0f;
                                }
                            }
                            if ((uint32_t)step >= T->nsteps) {
                                Wg->until = 0xFFFFFFFFu;
                                continue;
                            }
                            Wg->visit++;
                            {
                                const pl_wgvisit *V = &wgv[T->visit0 + Wg->visit];
                                Wg->until = (uint32_t)step + V->nsteps;
                                memcpy(Wg->copy, o_rows + (size_t)V->row * ka, sizeof(float) * ka);
                                Wg->copy[ka] = o_acc[(size_t)V->row * 2];
                                Wg->copy[ka + 1] = o_acc[(size_t)V->row * 2 + 1];
                                memcpy(Wg->copy + ka + 2, Wg->copy, sizeof(float) * (ka + 2));
                                Wg->tsum0 = Wg->tsum;
                            }
                        }
                        if ((uint32_t)step >= T->nsteps) continue;
                        {
                            const pl_wgvisit *V = &wgv[T->visit0 + Wg->visit];
                            for (w = 0; w < W; w++) {
                                const pl_entry *eb_raw = entries + T->off + (uint64_t)w * T->nsteps * G + (uint64_t)step * G;
                                pl_entry eb[64];
                                int ro[64];
                                for (g = 0; g < G; g++) {
                                    eb[g] = eb_raw[g];
                                    ro[g] = eb[g].gat >= 0 && (eb[g].gat & GAT_RO) != 0;
                                    if (eb[g].gat >= 0) eb[g].gat &= GAT_ID;
                                }
                                /* the wave reads the copy once per step; its G lists add what they change.  The rows of the
                                   other side are read by all G lists before any of them writes (one step of a wave = one
                                   burst of loads, then one burst of stores): of two lists that hold the same row in the
                                   same step, the later one's store wins */
                                if (mode != 2) memcpy(snap, Wg->copy, sizeof(float) * (ka + 2));
                                {
                                    /* all lists of the wave on the SAME row of the other side: their changes to it are summed
                                       (row and accumulators), as the kernel does across the wave */
                                    int nact_ = 0, same = 1, first = -1;
                                    const int own_seq = mode == 4 || mode == 5, gat_seq = mode == 4 || mode == 6;
                                    for (g = 0; g < G; g++)
                                        if (eb[g].gat >= 0) {
                                            if (first < 0) first = g;
                                            else if (eb[g].gat != eb[first].gat) same = 0;
                                            nact_++;
                                        }
                                        same = same && nact_ > 1 && mode != 2 && !gat_seq;
                                    if (mode != 2 && !gat_seq)
                                        for (g = 0; g < G; g++)
                                            if (eb[g].gat >= 0) {
                                                memcpy(gsnap + (size_t)g * (ka + 2), g_rows + (size_t)eb[g].gat * ka, sizeof(float) * ka);
                                                memcpy(gsnap + (size_t)g * (ka + 2) + ka, g_acc + (size_t)eb[g].gat * 2, sizeof(float) * 2);
                                            }
                                    /* The G lists of the wave step the heavy row from the same snapshot; their steps are summed.  Where the
                                       lists pull the same way (the same pair repeated, or simply the common direction of all rows) the sum
                                       overshoots what G ratings one after the other would do; with S = sum over the lists of step size x
                                       curvature, a sequential pass contracts by exp(-S) where the sum contracts by S: the summed step is
                                       scaled by (1 - exp(-S)) / S -- the fold's own 1-D model, one wave at a time (1 while S is small). */
                                    float wdamp = 1.0f;
                                    if (!own_seq && mode != 2) {
                                        float S = 0.0f;
                                        const float e0_ = eta / sqrtf(snap[ka]), e1_ = eta / sqrtf(snap[ka + 1]);
                                        for (g = 0; g < G; g++)
                                            if (eb[g].gat >= 0) {
                                                const float *gr_ = g_rows + (size_t)eb[g].gat * ka;
                                                for (d = 0; d < (slow ? 8 : ka); d++) S += (d < 8 ? e0_ : e1_) * gr_[d] * gr_[d];
                                            }
                                        wdamp = damp(S);
                                    }
                                    for (g = 0; g < G; g++) {
                                        const pl_entry e = eb[g];
                                        float err;
                                        if (e.gat < 0) continue;
                                        if (own_seq) memcpy(snap, Wg->copy, sizeof(float) * (ka + 2));
                                        if (ro[g] && (mode == 2 || gat_seq)) { /* read, not written: work on a scratch copy of that row */
                                            memcpy(gsnap + (size_t)g * (ka + 2), g_rows + (size_t)e.gat * ka, sizeof(float) * ka);
                                            memcpy(gsnap + (size_t)g * (ka + 2) + ka, g_acc + (size_t)e.gat * 2, sizeof(float) * 2);
                                        }
                                        if (mode == 2) {
                                            float *gr2 = ro[g] ? gsnap + (size_t)g * (ka + 2) : g_rows + (size_t)e.gat * ka;
                                            float *ga2 = ro[g] ? gr2 + ka : g_acc + (size_t)e.gat * 2;
                                            err = orc_sgd_one(o_rows + (size_t)V->row * ka, gr2, o_acc + (size_t)V->row * 2, ga2, e.r, ka, lo, lg, eta, slow,
                                                              rsqrt_mode, rk_mode);
                                        } else {
                                            float *gr = gat_seq && !ro[g] ? g_rows + (size_t)e.gat * ka : gsnap + (size_t)g * (ka + 2);
                                            float *ga = gat_seq && !ro[g] ? g_acc + (size_t)e.gat * 2 : gr + ka;
                                            memcpy(tmp, snap, sizeof(float) * (ka + 2));
                                            err = orc_sgd_one(tmp, gr, tmp + ka, ga, e.r, ka, lo, lg, eta, slow, rsqrt_mode, rk_mode);
                                            for (d = 0; d < ka; d++) Wg->copy[d] += wdamp * (tmp[d] - snap[d]);
                                            for (d = ka; d < ka + 2; d++) Wg->copy[d] += tmp[d] - snap[d];
                                        }
                                        Wg->tsum += err * err;
                                        loss += (double)(err * err);
                                    }
                                    if (same && ro[first]) {
                                        /* (read-only: nothing is written) */
                                    } else if (same) {
                                        float *rowm = g_rows + (size_t)eb[first].gat * ka, *accm = g_acc + (size_t)eb[first].gat * 2;
                                        for (g = 0; g < G; g++)
                                            if (eb[g].gat >= 0) { /* gsnap held the old value plus this list's change: add the change */
                                                const float *gs = gsnap + (size_t)g * (ka + 2);
                                                if (g == first) continue;
                                                for (d = 0; d < ka; d++) gsnap[(size_t)first * (ka + 2) + d] += gs[d] - rowm[d];
                                                gsnap[(size_t)first * (ka + 2) + ka] += gs[ka] - accm[0];
                                                gsnap[(size_t)first * (ka + 2) + ka + 1] += gs[ka + 1] - accm[1];
                                            }
                                        memcpy(rowm, gsnap + (size_t)first * (ka + 2), sizeof(float) * ka);
                                        memcpy(accm, gsnap + (size_t)first * (ka + 2) + ka, sizeof(float) * 2);
                                    } else if (mode != 2 && !gat_seq)
                                        for (g = 0; g < G; g++)
                                            if (eb[g].gat >= 0 && !ro[g]) {
                                                memcpy(g_rows + (size_t)eb[g].gat * ka, gsnap + (size_t)g * (ka + 2), sizeof(float) * ka);
                                                memcpy(g_acc + (size_t)eb[g].gat * 2, gsnap + (size_t)g * (ka + 2) + ka, sizeof(float) * 2);
                                            }
                                }
                            }
                        }
                    }
                    /* ---- wave tasks: the ordinary rows ---- */
                    for (t = tbeg; t < tend && tstep >= 0; t++) {
                        int g;
                        const long long step = tstep;
                        const int ended = step >= tasks[t].nsteps;
                        int acts[64], ros[64];
                        pl_entry es[64];
                        /* role 1 (tasks[t].pad): the lists are visits of rows of the plan's GATHERED side, roles swapped */
                        const int tsw = (int)tasks[t].pad;
                        float *own_rows = tsw ? gat_rows_ : own_rows_, *gat_rows = tsw ? own_rows_ : gat_rows_;
                        float *own_acc = tsw ? gat_acc_ : own_acc_, *gat_acc = tsw ? own_acc_ : gat_acc_;
                        const float lam_own = tsw ? lam_gat_ : lam_own_, lam_gat = tsw ? lam_own_ : lam_gat_;
                        if (step > tasks[t].nsteps) continue;
                        /* one step of a wave: visits are switched, then all G lists read their row of the other side, then all
                           write it (of two lists that hold the same row in the same step, the later one's store wins) */
                        for (g = 0; g < G; g++) {
                            pl_list *L = &lists[(t - tbeg) * G + g];
                            pl_entry e;
                            int act, newvisit;
                            uint32_t id;
                            if (ended) { /* the list is through: close the visit it holds */
                                act = 0;
                                id = 0x7FFFFFFFu;
                                newvisit = L->cur != 0xFFFFFFFFu;
                                e.own = 0;
                                e.gat = -1;
                                e.r = 0;
                            } else {
                                e = entries[tasks[t].off + (uint64_t)step * G + g];
                                act = e.gat >= 0;
                                id = e.own & IDMASK;
                                newvisit = act && (e.own >> 31) && id != L->cur;
                            }
                            acts[g] = act;
                            ros[g] = act && (e.gat & GAT_RO) != 0;
                            if (act) e.gat &= GAT_ID;
                            es[g] = e;
                            if (newvisit) {
                                if (L->cur != 0xFFFFFFFFu) { /* close_visit: the row is written back */
                                    float *rowp = own_rows + (size_t)L->cur * ka, *accp = own_acc + (size_t)L->cur * 2;
                                    if (merge_back) { /* ... as what memory holds NOW plus what the visit changed */
                                        int d;
                                        for (d = 0; d < ka; d++) rowp[d] += L->o[d] - L->o[ka + d];
                                        accp[0] += L->og[0] - L->og0[0];
                                        accp[1] += L->og[1] - L->og0[1];
                                    } else {
                                        memcpy(rowp, L->o, sizeof(float) * ka);
                                        accp[0] = L->og[0];
                                        accp[1] = L->og[1];
                                    }
                                }
                                if (ended) {
                                    L->cur = 0xFFFFFFFFu;
                                    continue;
                                }
                                L->cur = id;
                                memcpy(L->o, own_rows + (size_t)id * ka, sizeof(float) * ka);
                                memcpy(L->o + ka, L->o, sizeof(float) * ka);
                                L->og[0] = L->og0[0] = own_acc[(size_t)id * 2];
                                L->og[1] = L->og0[1] = own_acc[(size_t)id * 2 + 1];
                            }
                        }
                        if (ended) continue;
                        for (g = 0; g < G; g++)
                            if (acts[g]) {
                                memcpy(gsnap + (size_t)g * (ka + 2), gat_rows + (size_t)es[g].gat * ka, sizeof(float) * ka);
                                memcpy(gsnap + (size_t)g * (ka + 2) + ka, gat_acc + (size_t)es[g].gat * 2, sizeof(float) * 2);
                            }
                        for (g = 0; g < G; g++)
                            if (acts[g]) {
                                pl_list *L = &lists[(t - tbeg) * G + g];
                                const int inplace = (mode == 2 || mode == 4 || mode == 6) && !ros[g];
                                float *gr = inplace ? gat_rows + (size_t)es[g].gat * ka : gsnap + (size_t)g * (ka + 2);
                                float *ga = inplace ? gat_acc + (size_t)es[g].gat * 2 : gr + ka;
                                const float err = orc_sgd_one(L->o, gr, L->og, ga, es[g].r, ka, lam_own, lam_gat, eta, slow, rsqrt_mode, rk_mode);
                                loss += (double)(err * err);
                            }
                        if (mode != 2 && mode != 4 && mode != 6)
                            for (g = 0; g < G; g++)
                                if (acts[g] && !ros[g]) {
                                    memcpy(gat_rows + (size_t)es[g].gat * ka, gsnap + (size_t)g * (ka + 2), sizeof(float) * ka);
                                    memcpy(gat_acc + (size_t)es[g].gat * 2, gsnap + (size_t)g * (ka + 2) + ka, sizeof(float) * 2);
                                }
                    }
                }
            }
            /* ---- behind the round: fold the rows that were split over several workgroups (fold_hot_rows) ---- */
            for (i = 0; i < (int)n_hot_slots && mode != 2; i++) {
                float *slot = hot_acc + (size_t)i * (ka + 5);
                const float n = slot[ka + 4];
                int d;
                if (!(n > 0.0f)) continue;
                {
                    const unsigned hr = hot_rows[i];
                    float *rowp = ((hr >> 31) ? gat_rows : own_rows) + (size_t)(hr & 0x7FFFFFFFu) * ka;
                    float *accp = ((hr >> 31) ? gat_acc : own_acc) + (size_t)(hr & 0x7FFFFFFFu) * 2;
                    const float A0 = fmaxf(slot[ka], 0.0f), A1 = fmaxf(slot[ka + 1], 0.0f), E = slot[ka + 2], N = slot[ka + 3];
                    const float rn = 1.0f / n;
                    const float r00 = sqrtf(accp[0]), r01 = sqrtf(accp[1]);
                    const float ts0 = 1.0f / (sqrtf(accp[0] + A0) + r00), ts1 = 1.0f / (sqrtf(accp[1] + A1) + r01);
                    const float tc0 = 1.0f / (sqrtf(accp[0] + A0 * rn) + r00), tc1 = 1.0f / (sqrtf(accp[1] + A1 * rn) + r01);
                    const float cq = E > 0.0f ? 2.0f * eta * N / E : 0.0f;
                    const float c0 = cq * A0 * 8.0f, c1 = cq * A1 / rk1;
                    const float Sseq = ts0 * c0 + ts1 * c1, Sch = (tc0 * c0 + tc1 * c1) * rn;
                    const float phi = damp(Sseq) / damp(Sch);
                    const float sc0 = mode == 3 ? rn : phi * ts0 / tc0, sc1 = mode == 3 ? rn : phi * ts1 / tc1;
                    for (d = 0; d < (slow ? 8 : ka); d++) rowp[d] += (d >= 8 ? sc1 : sc0) * slot[d];
                    accp[0] += A0;
                    accp[1] += A1;
                    memset(slot, 0, sizeof(float) * (size_t)(ka + 5));
                }
            }
        }
        if (epoch_loss) epoch_loss[ep - first_epoch] = loss;
    }
    free(lists);
    free(copies);
    free(wgs);
    free(wcopies);
    free(snap);
    free(tmp);
    free(gsnap);
    free(hot_acc);
    return 0;
}
