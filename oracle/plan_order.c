/* plan_order.c -- the oracle's per-rating arithmetic (orc_sgd_one, pinned bit-exact to the reference)
 * applied in the ORDER OF THE GPU PLAN.  TEST INFRASTRUCTURE ONLY (see the header of mf_oracle.c).
 *
 * The GPU trainer visits the ratings in another order than the reference (stripe rounds instead of the
 * block scheduler of reference mf/mf.cpp:49-312; the lists of a block advance side by side instead of one
 * thread walking a sorted block, mf.cpp:1201-1238; a heavy owner row is cut into chains).  SGD is order
 * dependent, so a difference in final RMSE between the GPU path and orc_train mixes two things: arithmetic
 * and order.  This file separates them: it walks the plan's own entry stream (mfx_plan_view: entries, tasks,
 * slot_task_ptr) on one CPU thread -- rounds in the kernel's order, the lists of a block in lock step, owner
 * rows held in a private copy for the length of a visit exactly as the kernel holds them in registers -- with
 * the oracle's update.  What is left between this and the GPU is the lock-free execution itself.
 *
 * chain_mode says what happens to an owner row that the plan cut into chains:
 *   0  every chain works on a private copy, the last one to end overwrites the row (round 1's kernel)
 *   1  private copies, folded when the last chain of the launch ends (kernels.hip "hot chains", same formula)
 *   2  no private copies for chains: every chain updates the row in memory at once (what one thread walking the
 *      interleaved lists would do -- the sequential meaning of this order)
 *   6  (study) private copies; chain number idx of a row starts with its accumulators advanced by idx x the growth one
 *      chain of that row showed at the row's fold before (so the step sizes along the visit fall as they would
 *      sequentially); summed change damped as in mode 1 without the step-size ratio
 *   3  private copies; every chain's change is weighted by what a sequential pass would have left of it -- the
 *      step-size ratio of its place in the visit and exp(-contraction of the chains behind it) -- and the weighted
 *      changes are summed
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mf_oracle.h"

typedef struct { uint32_t own; int32_t gat; float r; } pl_entry;
typedef struct { uint64_t off; uint32_t nsteps; uint32_t pad; } pl_task;

typedef struct {
    uint32_t cur;      /* owner row of the visit in progress, 0xFFFFFFFF = none */
    int hot_n;         /* > 0: the visit is one of hot_n chains */
    uint32_t hot_h;    /* combine slot | chain length << 20 */
    float e0;          /* list's squared-error sum when the chain began */
    float tsum;
    float og[2];
    float goff[2];     /* chain_mode 6: accumulator advance this chain started with */
    float *o;          /* private copy of the owner row (ka floats) */
    int shared;        /* chain_mode 2: the visit works on the row in memory */
} pl_list;

static float damp(float S) { return S > 1e-3f ? (1.0f - expf(-S)) / S : 1.0f - 0.5f * S; }

int orc_plan_order_train(const void *entries_v, const void *tasks_v, const long long *slot_task_ptr, int ns, int G,
                         int ka, int owner_is_q, float *P, float *Q, float *PG, float *QG, long long n_hot_slots,
                         float lambda_p, float lambda_q, float eta, int epochs, int first_epoch, int chain_mode,
                         int rsqrt_mode, int rk_mode, double *epoch_loss)
{
    const pl_entry *entries = (const pl_entry *)entries_v;
    const pl_task *tasks = (const pl_task *)tasks_v;
    float *own_rows = owner_is_q ? Q : P, *gat_rows = owner_is_q ? P : Q;
    float *own_acc = owner_is_q ? QG : PG, *gat_acc = owner_is_q ? PG : QG;
    const float lam_own = owner_is_q ? lambda_q : lambda_p, lam_gat = owner_is_q ? lambda_p : lambda_q;
    const float rk1 = (rk_mode == ORC_RK_AS_BUILT || ka == 8) ? 0.125f : 1.0f / (float)(ka - 8);
    float *hot_growth = (float *)calloc((size_t)(n_hot_slots > 0 ? n_hot_slots : 1) * 2, sizeof(float));
    float *hot_acc = (float *)calloc((size_t)(n_hot_slots > 0 ? n_hot_slots : 1) * (size_t)(ka + 4), sizeof(float));
    int *hot_done = (int *)calloc((size_t)(n_hot_slots > 0 ? n_hot_slots : 1), sizeof(int));
    long long max_tasks = 0;
    int i, ep;
    const double study_pow = getenv("ORC_STUDY_POW") ? atof(getenv("ORC_STUDY_POW")) : 0.0;
    const double study_reg = getenv("ORC_STUDY_REG") ? atof(getenv("ORC_STUDY_REG")) : 0.0;
    const int study_dump = getenv("ORC_STUDY_DUMP") ? atoi(getenv("ORC_STUDY_DUMP")) : 0;
    const int study_dump_minn = getenv("ORC_STUDY_DUMP_MINN") ? atoi(getenv("ORC_STUDY_DUMP_MINN")) : 0;
    int dumped = 0;
    const double study_blend = getenv("ORC_STUDY_BLEND") ? atof(getenv("ORC_STUDY_BLEND")) : 0.0;
    const double study_smul = getenv("ORC_STUDY_SMUL") ? atof(getenv("ORC_STUDY_SMUL")) : 0.7; /* = HOT_S_GAIN of kernels.hpp */
    const double study_rgain = getenv("ORC_STUDY_RGAIN") ? atof(getenv("ORC_STUDY_RGAIN")) : 1.0;
    const int study_nform = getenv("ORC_STUDY_NFORM") ? atoi(getenv("ORC_STUDY_NFORM")) : 2; /* as the kernel: gain * (n / n0 + 1)^npow */
    const double study_n0 = getenv("ORC_STUDY_N0") ? atof(getenv("ORC_STUDY_N0")) : 2.0; /* = HOT_S_N0 of kernels.hpp */
    const double study_npow = getenv("ORC_STUDY_NPOW") ? atof(getenv("ORC_STUDY_NPOW")) : 0.5; /* = HOT_S_POW of kernels.hpp */
    const int study_avg = getenv("ORC_STUDY_AVG") != NULL;
    const double study_gain = getenv("ORC_STUDY_GAIN") ? atof(getenv("ORC_STUDY_GAIN")) : 1.0;
    pl_list *lists;
    float *copies;
    for (i = 0; i < ns * ns; i++)
        if (slot_task_ptr[i + 1] - slot_task_ptr[i] > max_tasks)
            max_tasks = slot_task_ptr[i + 1] - slot_task_ptr[i];
    lists = (pl_list *)malloc((size_t)max_tasks * G * sizeof(pl_list));
    copies = (float *)malloc((size_t)max_tasks * G * ka * sizeof(float));

    for (ep = first_epoch; ep < first_epoch + epochs; ep++) {
        const int slow = ep == 0;
        double loss = 0;
        int ri;
        for (ri = 0; ri < ns; ri++) {
            const int r = (ri + ep) % ns; /* the kernel rotates the first round per epoch (trainer.cpp) */
            int s;
            for (s = 0; s < ns; s++) {
                const long long tbeg = slot_task_ptr[(long long)r * ns + s], tend = slot_task_ptr[(long long)r * ns + s + 1];
                const long long nl = (tend - tbeg) * G;
                long long l, step, maxsteps = 0;
                for (l = 0; l < nl; l++) {
                    lists[l].cur = 0xFFFFFFFFu;
                    lists[l].hot_n = 0;
                    lists[l].tsum = 0;
                    lists[l].o = copies + l * ka;
                    lists[l].shared = 0;
                }
                for (l = tbeg; l < tend; l++)
                    if (tasks[l].nsteps > maxsteps)
                        maxsteps = tasks[l].nsteps;
                for (step = 0; step <= maxsteps; step++) {
                    long long t;
                    for (t = tbeg; t < tend; t++) {
                        int g;
                        const int ended = step >= tasks[t].nsteps;
                        if (step > tasks[t].nsteps)
                            continue;
                        for (g = 0; g < G; g++) {
                            pl_list *L = &lists[(t - tbeg) * G + g];
                            pl_entry e;
                            int act, hdr, newvisit;
                            uint32_t id;
                            if (ended) { /* the list is through: close the visit it holds */
                                e.own = 0x80000000u | 0x7FFFFFFFu;
                                e.gat = -1;
                                e.r = 0;
                                act = hdr = 0;
                                id = 0x7FFFFFFFu;
                                newvisit = L->cur != 0xFFFFFFFFu;
                            } else {
                                e = entries[tasks[t].off + (uint64_t)step * G + g];
                                act = e.gat >= 0;
                                hdr = e.gat < -1;
                                id = e.own & 0x7FFFFFFFu;
                                newvisit = (act || hdr) && (e.own >> 31) && (id != L->cur || hdr);
                            }
                            if (newvisit) {
                                if (L->cur != 0xFFFFFFFFu && !L->shared) { /* close_visit */
                                    float *rowp = own_rows + (size_t)L->cur * ka, *accp = own_acc + (size_t)L->cur * 2;
                                    int d;
                                    if (L->hot_n != 0 && chain_mode == 6) {
                                        float *slot = hot_acc + (size_t)(L->hot_h & 0xFFFFFu) * (ka + 4);
                                        const int nch = L->hot_n & 0x7FFF;
                                        for (d = 0; d < ka; d++)
                                            slot[d] += L->o[d] - rowp[d];
                                        slot[ka] += L->og[0] - accp[0] - L->goff[0];
                                        slot[ka + 1] += L->og[1] - accp[1] - L->goff[1];
                                        slot[ka + 2] += L->tsum - L->e0;
                                        slot[ka + 3] += (float)(L->hot_h >> 20);
                                        if (++hot_done[L->hot_h & 0xFFFFFu] == nch) {
                                            const float A0 = slot[ka], A1 = slot[ka + 1], E = slot[ka + 2], N = slot[ka + 3];
                                            const float ts0 = 1.0f / (sqrtf(accp[0] + A0) + sqrtf(accp[0])), ts1 = 1.0f / (sqrtf(accp[1] + A1) + sqrtf(accp[1]));
                                            const float cq = E > 0.0f ? 2.0f * eta * N / E : 0.0f;
                                            const float Sseq = ts0 * cq * A0 * 8.0f + ts1 * cq * A1 / rk1;
                                            const float phi = damp((float)study_smul * Sseq) / damp((float)study_smul * Sseq / (float)nch);
                                            float *gr = hot_growth + (size_t)(L->hot_h & 0xFFFFFu) * 2;
                                            for (d = 0; d < ka; d++)
                                                rowp[d] += phi * slot[d];
                                            accp[0] += A0;
                                            accp[1] += A1;
                                            gr[0] = A0 / (float)nch;
                                            gr[1] = A1 / (float)nch;
                                            memset(slot, 0, sizeof(float) * (ka + 4));
                                            hot_done[L->hot_h & 0xFFFFFu] = 0;
                                        }
                                    } else if (L->hot_n != 0 && (chain_mode == 4 || chain_mode == 5)) {
                                        /* 4: rows are folded as a plain damped sum (phi of mode 1 without the step-size ratio);
                                           5: only the accumulator growth is summed */
                                        float *slot = hot_acc + (size_t)(L->hot_h & 0xFFFFFu) * (ka + 4);
                                        const int nch = L->hot_n & 0x7FFF;
                                        if (chain_mode == 4)
                                            for (d = 0; d < ka; d++)
                                                slot[d] += L->o[d] - rowp[d];
                                        else {
                                            slot[ka] += L->og[0] - accp[0];
                                            slot[ka + 1] += L->og[1] - accp[1];
                                        }
                                        if (++hot_done[L->hot_h & 0xFFFFFu] == nch) {
                                            if (chain_mode == 4)
                                                for (d = 0; d < ka; d++)
                                                    rowp[d] += slot[d] / (float)nch * (float)study_gain;
                                            else {
                                                accp[0] += slot[ka];
                                                accp[1] += slot[ka + 1];
                                            }
                                            memset(slot, 0, sizeof(float) * (ka + 4));
                                            hot_done[L->hot_h & 0xFFFFFu] = 0;
                                        }
                                    } else if (L->hot_n != 0 && chain_mode == 3) {
                                        const int nch = L->hot_n & 0x7FFF, idx = L->hot_n >> 16;
                                        float *slot = hot_acc + (size_t)(L->hot_h & 0xFFFFFu) * (ka + 4);
                                        const float g0[2] = {accp[0], accp[1]};
                                        const float Ac[2] = {L->og[0] - g0[0], L->og[1] - g0[1]};
                                        const float Ec = L->tsum - L->e0, Nc = (float)(L->hot_h >> 20);
                                        const float rk[2] = {0.125f, rk1};
                                        float w[2], R = 0.0f;
                                        int j;
                                        for (j = 0; j < 2; j++) {
                                            /* step size of this place in the visit over the step size the chain used */
                                            w[j] = (sqrtf(g0[j] + Ac[j]) + sqrtf(g0[j])) /
                                                   (sqrtf(g0[j] + (idx + 1) * Ac[j]) + sqrtf(g0[j] + idx * Ac[j]));
                                            /* contraction of the chains behind this one: curvature x integrated step size */
                                            if (Ec > 0.0f && Ac[j] > 0.0f)
                                                R += (Ac[j] / (rk[j] * Ec)) * eta * (Nc / Ac[j]) * 2.0f *
                                                     (sqrtf(g0[j] + nch * Ac[j]) - sqrtf(g0[j] + (idx + 1) * Ac[j]));
                                        }
                                        for (d = 0; d < ka; d++)
                                            slot[d] += w[d >= 8] * expf(-(float)study_rgain * R) * (L->o[d] - rowp[d]);
                                        slot[ka] += Ac[0];
                                        slot[ka + 1] += Ac[1];
                                        if (++hot_done[L->hot_h & 0xFFFFFu] == nch) {
                                            for (d = 0; d < ka; d++)
                                                rowp[d] += slot[d];
                                            accp[0] += slot[ka];
                                            accp[1] += slot[ka + 1];
                                            memset(slot, 0, sizeof(float) * (ka + 4));
                                            hot_done[L->hot_h & 0xFFFFFu] = 0;
                                        }
                                    } else if (L->hot_n == 0 || chain_mode == 0) {
                                        memcpy(rowp, L->o, sizeof(float) * ka);
                                        accp[0] = L->og[0];
                                        accp[1] = L->og[1];
                                    } else {
                                        float *slot = hot_acc + (size_t)(L->hot_h & 0xFFFFFu) * (ka + 4);
                                        for (d = 0; d < ka; d++)
                                            slot[d] += L->o[d] - rowp[d];
                                        slot[ka] += L->og[0] - accp[0];
                                        slot[ka + 1] += L->og[1] - accp[1];
                                        slot[ka + 2] += L->tsum - L->e0;
                                        slot[ka + 3] += (float)(L->hot_h >> 20);
                                        if (++hot_done[L->hot_h & 0xFFFFFu] == (L->hot_n & 0x7FFF)) { /* fold */
                                            const float A0 = slot[ka], A1 = slot[ka + 1], E = slot[ka + 2], N = slot[ka + 3];
                                            const float rn = 1.0f / (float)(L->hot_n & 0x7FFF);
                                            const float r00 = sqrtf(accp[0]), r01 = sqrtf(accp[1]);
                                            const float ts0 = 1.0f / (sqrtf(accp[0] + A0) + r00), ts1 = 1.0f / (sqrtf(accp[1] + A1) + r01);
                                            const float tc0 = 1.0f / (sqrtf(accp[0] + A0 * rn) + r00), tc1 = 1.0f / (sqrtf(accp[1] + A1 * rn) + r01);
                                            const float cq = E > 0.0f ? 2.0f * eta * N / E : 0.0f;
                                            const float c0 = cq * A0 * 8.0f, c1 = cq * A1 / rk1;
                                            const float reg = (float)study_reg * 2.0f * eta * N * lam_own / (float)ka; /* curvature of the L2 term */
                                            const float Sseq = ts0 * (c0 + 8 * reg) + ts1 * (c1 + (ka - 8) * reg),
                                                        Sch = (tc0 * (c0 + 8 * reg) + tc1 * (c1 + (ka - 8) * reg)) * rn;
                                            const float nn = (float)(L->hot_n & 0x7FFF);
                                            const float geff = (float)study_smul * (study_nform == 1 ? fmaxf(1.0f, nn / (float)study_n0)
                                                                                    : study_nform == 2 ? powf(nn / (float)study_n0 + 1.0f, (float)study_npow) : 1.0f);
                                            const float phi = damp(geff * Sseq) / damp(geff * Sch);
                                            if (study_dump && ep == study_dump && (L->hot_n & 0x7FFF) >= study_dump_minn && dumped < 4000 && (dumped++ % 4) == 0)
                                                fprintf(stderr, "fold ep %d n %d N %.0f G0 %.1f %.1f A %.2f %.2f E %.1f Sseq %.3f Sch %.3f phi %.3f sc %.3f %.3f\n", ep,
                                                        L->hot_n & 0x7FFF, N, accp[0], accp[1], A0, A1, E, Sseq, Sch, phi, ts0 / tc0, ts1 / tc1);
                                            for (d = 0; d < ka; d++)
                                                rowp[d] += (study_blend > 0 ? (float)study_blend * rn + (1.0f - (float)study_blend) * phi * (d >= 8 ? ts1 / tc1 : ts0 / tc0) : study_avg ? rn * (float)study_gain : study_pow > 0 ? powf(d >= 8 ? ts1 / tc1 : ts0 / tc0, (float)study_pow) : phi * (d >= 8 ? ts1 / tc1 : ts0 / tc0)) * slot[d];
                                            accp[0] += A0;
                                            accp[1] += A1;
                                            memset(slot, 0, sizeof(float) * (ka + 4));
                                            hot_done[L->hot_h & 0xFFFFFu] = 0;
                                        }
                                    }
                                }
                                if (ended) {
                                    L->cur = 0xFFFFFFFFu;
                                    continue;
                                }
                                L->hot_n = hdr ? ((-e.gat - 1) & 0x7FFF) | ((-e.gat - 1) >> 15 << 16) : 0; /* chains | index << 16 */
                                memcpy(&L->hot_h, &e.r, 4);
                                L->e0 = L->tsum;
                                L->shared = hdr && chain_mode == 2;
                                L->cur = id;
                                if (!L->shared) {
                                    memcpy(L->o, own_rows + (size_t)id * ka, sizeof(float) * ka);
                                    L->og[0] = own_acc[(size_t)id * 2];
                                    L->og[1] = own_acc[(size_t)id * 2 + 1];
                                    L->goff[0] = L->goff[1] = 0.0f;
                                    if (hdr && chain_mode == 6) {
                                        const int idx = L->hot_n >> 16;
                                        const float *gr = hot_growth + (size_t)(L->hot_h & 0xFFFFFu) * 2;
                                        L->goff[0] = idx * gr[0];
                                        L->goff[1] = idx * gr[1];
                                        L->og[0] += L->goff[0];
                                        L->og[1] += L->goff[1];
                                    }
                                }
                            }
                            if (act) {
                                /* study modes: 4 = private row copy but the accumulators live in memory; 5 = the reverse */
                                const int hotc = L->hot_n != 0;
                                float *orow = (L->shared || (hotc && chain_mode == 5)) ? own_rows + (size_t)id * ka : L->o;
                                float *oacc = (L->shared || (hotc && chain_mode == 4)) ? own_acc + (size_t)id * 2 : L->og;
                                float *grow = gat_rows + (size_t)e.gat * ka, *gacc = gat_acc + (size_t)e.gat * 2;
                                float err;
                                /* orc_sgd_one(p, q, ...): p takes lambda_p; pass the owner as "p" with its own lambda */
                                err = orc_sgd_one(orow, grow, oacc, gacc, e.r, ka, lam_own, lam_gat, eta, slow, rsqrt_mode, rk_mode);
                                L->tsum += err * err;
                                loss += (double)(err * err);
                            }
                        }
                    }
                }
            }
        }
        if (epoch_loss)
            epoch_loss[ep - first_epoch] = loss;
    }
    free(lists);
    free(copies);
    free(hot_growth);
    free(hot_acc);
    free(hot_done);
    return 0;
}
