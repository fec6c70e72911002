/*
 * mf_oracle.h -- interface of the CPU restatement in mf_oracle.c.
 * TEST INFRASTRUCTURE ONLY (see the header of mf_oracle.c).
 */
#ifndef MF_ORACLE_H
#define MF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* struct mf_node, /root/reference/mf/mf.h:36-41 */
typedef struct { int u; int v; float r; } orc_node;

/* struct mf_model, mf.h:70-79, after shrink/un-permute: k un-padded, original ids */
typedef struct { int fun, m, n, k; float b; float *P; float *Q; } orc_model;

enum { ORC_RSQRT_SSE = 0, ORC_RSQRT_EXACT = 1 };
enum { ORC_RK_AS_BUILT = 0, ORC_RK_FAST = 1 }; /* SURVEY.md 3.4, quirk Q1 */

typedef struct {
    int k, nr_bins, nr_iters;
    float lambda_p2, lambda_q2, eta;
    int rsqrt_mode; /* ORC_RSQRT_* */
    int rk_mode;    /* ORC_RK_*    */
} orc_param;

typedef struct { int32_t r[31]; int f, b; } orc_glibc_rand_t;

float orc_canon_float(uint32_t *state);
void orc_glibc_srand(orc_glibc_rand_t *g, unsigned seed);
int orc_glibc_rand(orc_glibc_rand_t *g);
void orc_gen_random_map(int size, int *map);

void orc_read_triplet(const float *tri, int triplet_num, orc_node *R, int *m, int *n);
void orc_collect_info(const orc_node *R, long long nnz, float *avg, float *std_dev);
void orc_grid_problem(orc_node *R, long long nnz, int m, int n, int nr_bins,
                      long long *ptrs, int *omega_p, int *omega_q);
int orc_k_aligned(int k);
void orc_init_model(int m, int n, int k, const int *omega_p, const int *omega_q,
                    float *P, float *Q);

float orc_rsqrt_probe(float x);
float orc_sgd_one(float *p, float *q, float *pG, float *qG, float r, int ka,
                  float lambda_p, float lambda_q, float eta, int slow_only,
                  int rsqrt_mode, int rk_mode);

orc_param orc_default_param(void);
int orc_train(const orc_node *R, long long nnz, int m, int n, const orc_param *prm,
              orc_model *out, double *tr_rmse, double *obj);
/* order study (mf_oracle.c: orc_train_order): 0 everywhere = the reference's order */
typedef struct {
    int block_order; /* 0 the reference's scheduler; 1 cyclic rounds (s, (s+r) mod bins); 2 the same, first round rotated per epoch */
    int sort_side;   /* 0 as the reference (by user when m > n); 1 by item, then user; 2 by user, then item */
    int lists;       /* > 1: a block's ratings are dealt over this many lists that advance in lock step */
} orc_order;
int orc_train_order(const orc_node *R, long long nnz, int m, int n, const orc_param *prm,
                    orc_model *out, double *tr_rmse, double *obj, const orc_order *ord);
/* plan_order.c: orc_sgd_one over the GPU plan's own entry stream (see the header of that file) */
int orc_plan_order_train(const void *entries, const void *tasks, const long long *slot_task_ptr, const void *wg_tasks,
                         const void *wg_visits, const long long *slot_wg_ptr, const unsigned *hot_rows, int ns, int G, int W,
                         int ka, int owner_is_q, float *P, float *Q, float *PG, float *QG, long long n_hot_slots,
                         float lambda_p, float lambda_q, float eta, int epochs, int first_epoch, int mode,
                         int rsqrt_mode, int rk_mode, double *epoch_loss);
void orc_free_model(orc_model *mdl);

float orc_predict(const orc_model *mdl, int u, int v);
double orc_rmse(const orc_node *R, long long nnz, const orc_model *mdl);
float *orc_model_to_array(const orc_model *mdl, int *lens);
float *orc_utility_train(const float *train, int triplets, double p_l2, double q_l2,
                         int k, int iters, double eta, int *lens);
float *orc_utility_predict(const float *test, int pairs, const float *model_arr, int model_len);
void orc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
