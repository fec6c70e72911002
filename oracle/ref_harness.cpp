// ref_harness.cpp -- extern "C" doorway into the reference itself, for tests only.
//
// Built ONLY in the development container, where /root/reference exists, by
// oracle/Makefile: it includes the reference's own mf/mf.h from where it lies and
// links oracle/_ref/libmf_ref.so (the reference's mf/mf.cpp compiled with the
// reference's flags, mf/CMakeLists.txt:10,12).  No reference source is copied
// into this repository; the outputs stay in oracle/_ref/ (git-ignored).
//
// Driving rules from SURVEY.md 8c: mf::mf_train with quiet=true (the literal
// utility_train entry, quiet=false, deadlocks on small inputs -- quirk Q2), the
// facade's parameter overrides (mf.cpp:3508-3513) applied by hand.
#include <cstdlib>
#include <cstring>
#include "mf.h"

extern "C" {

// Train with the reference and hand back the facade array [fun,m,n,k,b,P,Q]
// (same layout as model_to_array, mf.cpp:3415-3441).  Returns malloc'd floats.
float *ref_train_array(const mf::mf_node *R, long long nnz, int m, int n, int k,
                       int nr_threads, int nr_bins, int nr_iters, float lambda_p2,
                       float lambda_q2, float eta, long long *lens)
{
    mf::mf_problem prob;
    prob.m = m;
    prob.n = n;
    prob.nnz = nnz;
    prob.R = const_cast<mf::mf_node *>(R);
    mf::mf_parameter param = mf::mf_get_default_param();
    param.k = k;
    param.nr_threads = nr_threads;
    param.nr_bins = nr_bins;
    param.nr_iters = nr_iters;
    param.lambda_p2 = lambda_p2;
    param.lambda_q2 = lambda_q2;
    param.eta = eta;
    param.quiet = true;
    mf::mf_model *model = mf::mf_train(&prob, param);
    if (model == nullptr) {
        *lens = 0;
        return nullptr;
    }
    long long pn = (long long)model->m * model->k, qn = (long long)model->n * model->k;
    float *out = (float *)malloc(sizeof(float) * (size_t)(pn + qn + 5));
    out[0] = (float)model->fun;
    out[1] = (float)model->m;
    out[2] = (float)model->n;
    out[3] = (float)model->k;
    out[4] = model->b;
    memcpy(out + 5, model->P, sizeof(float) * (size_t)pn);
    memcpy(out + 5 + pn, model->Q, sizeof(float) * (size_t)qn);
    *lens = pn + qn + 5;
    mf::mf_destroy_model(&model);
    return out;
}

// mf::calc_rmse (mf.cpp:4316-4331) of a facade array on a problem.
double ref_rmse_array(const mf::mf_node *R, long long nnz, int m, int n, float *arr)
{
    mf::mf_problem prob;
    prob.m = m;
    prob.n = n;
    prob.nnz = nnz;
    prob.R = const_cast<mf::mf_node *>(R);
    mf::mf_model model;
    model.fun = (int)arr[0];
    model.m = (int)arr[1];
    model.n = (int)arr[2];
    model.k = (int)arr[3];
    model.b = arr[4];
    model.P = arr + 5;
    model.Q = arr + 5 + (long long)model.m * model.k;
    return mf::calc_rmse(&prob, &model);
}

// mf::utility_predict (mf.cpp:3537-3568) as it stands.
float *ref_utility_predict(float *test, int pairs, float *model_arr, int model_len)
{
    return mf::utility_predict(test, pairs, model_arr, model_len);
}

// Wall-clock legs for bench.py's cpu_baseline: seconds for mf_train at a given
// iteration count (the iteration-delta method of SURVEY.md 8d is applied by the caller).
double ref_time_train(const mf::mf_node *R, long long nnz, int m, int n, int k,
                      int nr_threads, int nr_bins, int nr_iters, float lambda_p2,
                      float lambda_q2, float eta, double *rmse_out);

// mf::cos_similarity (mf.cpp:3591-3683) and mf::DINA (mf.cpp:3685-4109) as they stand.  DINA draws its start
// values from the process-global rand() (mf.cpp:3759): call it in a fresh process for a defined result.
float *ref_cos_similarity(int item_id, float *q_arr, int q_arr_num) { return mf::cos_similarity(item_id, q_arr, q_arr_num); }
int *ref_DINA(float *q_arr, int q_triplet_num, float *x_arr, int x_triplet_num, int iterators)
{
    return mf::DINA(q_arr, q_triplet_num, x_arr, x_triplet_num, iterators);
}

void ref_free(void *p) { free(p); }
}

#include <chrono>
extern "C" double ref_time_train(const mf::mf_node *R, long long nnz, int m, int n, int k,
                                 int nr_threads, int nr_bins, int nr_iters,
                                 float lambda_p2, float lambda_q2, float eta,
                                 double *rmse_out)
{
    mf::mf_problem prob;
    prob.m = m;
    prob.n = n;
    prob.nnz = nnz;
    prob.R = const_cast<mf::mf_node *>(R);
    mf::mf_parameter param = mf::mf_get_default_param();
    param.k = k;
    param.nr_threads = nr_threads;
    param.nr_bins = nr_bins;
    param.nr_iters = nr_iters;
    param.lambda_p2 = lambda_p2;
    param.lambda_q2 = lambda_q2;
    param.eta = eta;
    param.quiet = true;
    auto t0 = std::chrono::steady_clock::now();
    mf::mf_model *model = mf::mf_train(&prob, param);
    auto t1 = std::chrono::steady_clock::now();
    if (model == nullptr)
        return -1.0;
    if (rmse_out)
        *rmse_out = mf::calc_rmse(&prob, model);
    mf::mf_destroy_model(&model);
    return std::chrono::duration<double>(t1 - t0).count();
}
