// mf_api.cpp -- namespace mf: the C++ surface an unchanged libphp_mf.so / mfTest binds
// (include/mf.h), implemented over the mfx_* HIP C-ABI.
//
// Mirrors the reference's "utility" facade and the few LIBMF entry points it calls
// (reference mf/mf.cpp:3307-3568, 4143-4331, 4538-4557).  Same argument meaning and
// return conventions; the differences are deliberate and listed in DESIGN.md:
//   * nothing throws or dereferences null across the boundary: failures return
//     nullptr / lens = 0 (the reference crashes, SURVEY.md 8b "Errors");
//   * the worker loop cannot dead-lock at shutdown (reference quirk Q2);
//   * nr_threads / nr_bins are accepted and ignored: the schedule is the device's.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mf.h"
#include "../../include/mfx.h"

namespace mf {

namespace {

// One training / prediction call at a time per process: the facade is entered from PHP
// request threads (ZTS build, reference php_mf/CMakeLists.txt:12).
std::mutex g_gpu_mutex;

float *alloc_aligned_floats(long long count)
{
    // 32-byte aligned like the reference's model storage (mf.cpp:936-950); freed with free()
    void *p = nullptr;
    if (count <= 0) count = 1;
    if (posix_memalign(&p, 32, (size_t)count * sizeof(float)) != 0) return nullptr;
    return (float *)p;
}

bool check_parameter(const mf_parameter &param)
{
    // reference mf/mf.cpp:3115-3184, restricted to what this path supports
    if (param.fun != P_L2_MFR) {
        std::cerr << "unknown loss function" << std::endl;
        return false;
    }
    if (param.k < 1) {
        std::cerr << "number of factors must be greater than zero" << std::endl;
        return false;
    }
    if (param.nr_threads < 1) {
        std::cerr << "number of threads must be greater than zero" << std::endl;
        return false;
    }
    if (param.nr_bins < 1 || param.nr_bins < param.nr_threads) {
        std::cerr << "number of bins must be greater than number of threads" << std::endl;
        return false;
    }
    if (param.nr_iters < 1) {
        std::cerr << "number of iterations must be greater than zero" << std::endl;
        return false;
    }
    if (param.lambda_p1 < 0 || param.lambda_p2 < 0 || param.lambda_q1 < 0 || param.lambda_q2 < 0) {
        std::cerr << "regularization coefficient must be non-negative" << std::endl;
        return false;
    }
    if (param.eta <= 0) {
        std::cerr << "learning rate must be greater than zero" << std::endl;
        return false;
    }
    if (param.lambda_p1 != 0 || param.lambda_q1 != 0 || param.do_nmf) {
        std::cerr << "L1 regularization and NMF are not available on the GPU path" << std::endl;
        return false;
    }
    return true;
}

// Train on the device and export the facade array; returns 0 or an mfx_status.
// R_dev: the ratings as nodes in HBM (mfx_triplets_to_device) instead of tr->R.
// out_malloc: export straight into a malloc'd array (*out_malloc, *out_len) instead of `arr`.
int train_to_array(const mf_problem *tr, const mf_parameter &param, std::vector<float> &arr,
                   const void *R_dev = nullptr, float **out_malloc = nullptr, long long *out_len = nullptr)
{
    mfx_options opt;
    mfx_default_options(&opt);
    opt.k = param.k;
    opt.lambda_p2 = param.lambda_p2;
    opt.lambda_q2 = param.lambda_q2;
    opt.eta = param.eta;

    mfx_trainer *t = nullptr;
    int rc = R_dev ? mfx_trainer_create_device(R_dev, tr->nnz, tr->m, tr->n, &opt, &t)
                   : mfx_trainer_create((const mfx_node *)tr->R, tr->nnz, tr->m, tr->n, &opt, &t);
    if (rc != MFX_OK) return rc;
    struct Guard {
        mfx_trainer *t;
        ~Guard() { mfx_trainer_destroy(t); }
    } guard{t};
    if ((rc = mfx_trainer_init_model(t, nullptr)) != MFX_OK) return rc;
    mfx_info info;
    mfx_trainer_info(t, &info);

    if (!param.quiet) // progress table header, reference mf/mf.cpp:2818-2832
        printf("%4s%13s%13s\n", "iter", "tr_rmse", "obj");
    for (int iter = 0; iter < param.nr_iters; ++iter) {
        // epoch 0 updates the first 8 factors only (slow_only, mf.cpp:2834, 2910-2911)
        if ((rc = mfx_trainer_epoch(t, iter == 0, nullptr)) != MFX_OK) return rc;
        if (!param.quiet) { // reference mf/mf.cpp:2852-2908
            double loss = 0, reg = 0;
            if ((rc = mfx_trainer_last_loss(t, &loss)) != MFX_OK) return rc;
            if ((rc = mfx_trainer_reg2(t, &reg)) != MFX_OK) return rc;
            const double s = info.scale;
            double tr_rmse = std::sqrt(loss / (double)info.nnz * s * s);
            double obj = reg * s * s + loss * (double)(float)(info.scale * info.scale);
            printf("%4d%13.4f%13.4e\n", iter, tr_rmse, obj);
            fflush(stdout);
        }
    }
    long long len = 5 + ((long long)info.m + info.n) * info.k;
    if (out_malloc) {
        float *buf = (float *)malloc(sizeof(float) * (size_t)len); // caller frees with free()
        if (!buf) return MFX_E_NOMEM;
        rc = mfx_trainer_export(t, buf, len);
        if (rc != MFX_OK) {
            free(buf);
            return rc;
        }
        *out_malloc = buf;
        *out_len = len;
        return MFX_OK;
    }
    arr.resize((size_t)len);
    return mfx_trainer_export(t, arr.data(), len);
}

} // namespace

mf_parameter mf_get_default_param()
{
    mf_parameter param; // reference mf/mf.cpp:4538-4557
    param.fun = P_L2_MFR;
    param.k = 8;
    param.nr_threads = 12;
    param.nr_bins = 20;
    param.nr_iters = 20;
    param.lambda_p1 = 0.0f;
    param.lambda_p2 = 0.1f;
    param.lambda_q1 = 0.0f;
    param.lambda_q2 = 0.1f;
    param.eta = 0.1f;
    param.do_nmf = false;
    param.quiet = false;
    param.copy_data = true;
    return param;
}

mf_model *mf_train_with_validation(mf_problem const *tr, mf_problem const *va, mf_parameter param)
{
    try {
        if (!check_parameter(param)) return nullptr;
        if (tr == nullptr || tr->R == nullptr || tr->nnz <= 0) {
            std::cout << "warning: train on an empty training set" << std::endl; // mf.cpp:2794
            return nullptr;
        }
        if (va != nullptr && va->nnz != 0) {
            std::cerr << "validation sets are not supported on the GPU path" << std::endl;
            return nullptr;
        }
        std::lock_guard<std::mutex> lock(g_gpu_mutex);
        std::vector<float> arr;
        int rc = train_to_array(tr, param, arr);
        if (rc != MFX_OK) {
            std::cerr << "mf_train: " << mfx_last_error() << std::endl;
            return nullptr;
        }
        mf_model *model = new mf_model;
        model->fun = (mf_int)arr[0];
        model->m = tr->m;
        model->n = tr->n;
        model->k = param.k;
        model->b = arr[4];
        long long pn = (long long)model->m * model->k, qn = (long long)model->n * model->k;
        model->P = alloc_aligned_floats(pn);
        model->Q = alloc_aligned_floats(qn);
        if (!model->P || !model->Q) {
            mf_destroy_model(&model);
            return nullptr;
        }
        memcpy(model->P, arr.data() + 5, (size_t)pn * sizeof(float));
        memcpy(model->Q, arr.data() + 5 + pn, (size_t)qn * sizeof(float));
        return model;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return nullptr;
    } catch (...) {
        return nullptr;
    }
}

mf_model *mf_train(mf_problem const *prob, mf_parameter param)
{
    return mf_train_with_validation(prob, nullptr, param);
}

void mf_destroy_model(mf_model **model)
{
    if (model == nullptr || *model == nullptr) return;
    free((*model)->P);
    free((*model)->Q);
    delete *model;
    *model = nullptr;
}

mf_float mf_predict(mf_model const *model, mf_int u, mf_int v)
{
    // reference mf/mf.cpp:4295-4314
    if (model == nullptr) return std::numeric_limits<mf_float>::quiet_NaN();
    if (u < 0 || u >= model->m || v < 0 || v >= model->n) return model->b;
    const mf_float *p = model->P + (mf_long)u * model->k;
    const mf_float *q = model->Q + (mf_long)v * model->k;
    mf_float z = 0.0f;
    for (mf_int d = 0; d < model->k; ++d) z += p[d] * q[d];
    if (std::isnan(z)) z = model->b;
    return z;
}

namespace {
void model_header(const mf_model *model, float *a)
{
    a[0] = (float)model->fun; // reference mf/mf.cpp:3427-3431
    a[1] = (float)model->m;
    a[2] = (float)model->n;
    a[3] = (float)model->k;
    a[4] = model->b;
}
} // namespace

mf_double calc_rmse(mf_problem *prob, mf_model *model)
{
    try {
        if (prob == nullptr || model == nullptr) return std::numeric_limits<double>::quiet_NaN();
        if (prob->nnz == 0) return 0;
        long long pn = (long long)model->m * model->k, qn = (long long)model->n * model->k;
        std::vector<float> arr((size_t)(pn + qn + 5));
        model_header(model, arr.data());
        memcpy(arr.data() + 5, model->P, (size_t)pn * sizeof(float));
        memcpy(arr.data() + 5 + pn, model->Q, (size_t)qn * sizeof(float));
        std::lock_guard<std::mutex> lock(g_gpu_mutex);
        double rmse = 0;
        if (mfx_rmse_array(arr.data(), (long long)arr.size(), (const mfx_node *)prob->R, prob->nnz,
                           &rmse) != MFX_OK) {
            std::cerr << "calc_rmse: " << mfx_last_error() << std::endl;
            return std::numeric_limits<double>::quiet_NaN();
        }
        return rmse;
    } catch (...) {
        return std::numeric_limits<double>::quiet_NaN();
    }
}

// ---- float-array facade ---------------------------------------------------------------

// MFX_DEVICES=g (read once): the blocking call shards the job over g GPUs of the node (mfx_job_*, csrc/job.cpp) -- the
// place the reference's worker threads have (mf.cpp:2837-2846).  Fewer GPUs visible than asked for: the call fails.
static const int g_devices = []() {
    const char *s = getenv("MFX_DEVICES");
    return (s && *s) ? atoi(s) : 1;
}();

static float *utility_train_job(float *train_data, int count, const mf_parameter &param, int &lens)
{
    std::vector<mfx_node> R((size_t)count);
    int m = 0, n = 0;
    for (int i = 0; i < count; ++i) { // read_triplet, reference mf/mf.cpp:3367-3394
        R[i].u = (int)train_data[3 * (size_t)i];
        R[i].v = (int)train_data[3 * (size_t)i + 1];
        R[i].r = train_data[3 * (size_t)i + 2];
        if (R[i].u < 0 || R[i].v < 0) return nullptr;
        m = std::max(m, R[i].u + 1);
        n = std::max(n, R[i].v + 1);
    }
    const long long total = 5 + ((long long)m + n) * (long long)param.k;
    if (total > 2147483647LL) return nullptr;
    mfx_options opt;
    mfx_default_options(&opt);
    opt.k = param.k;
    opt.lambda_p2 = param.lambda_p2;
    opt.lambda_q2 = param.lambda_q2;
    opt.eta = param.eta;
    std::lock_guard<std::mutex> lock(g_gpu_mutex);
    mfx_job *job = nullptr;
    if (mfx_job_create(R.data(), count, m, n, &opt, g_devices, nullptr, &job) != MFX_OK) {
        std::cerr << "utility_train (MFX_DEVICES=" << g_devices << "): " << mfx_job_last_error() << std::endl;
        return nullptr;
    }
    struct Guard {
        mfx_job *j;
        ~Guard() { mfx_job_destroy(j); }
    } guard{job};
    if (!param.quiet) printf("%4s%13s\n", "iter", "tr_rmse"); // (the objective column needs one pass over every slot's rows: left out)
    for (int iter = 0; iter < param.nr_iters; ++iter) {
        if (mfx_job_epoch(job, iter == 0) != MFX_OK) {
            std::cerr << "utility_train: " << mfx_job_last_error() << std::endl;
            return nullptr;
        }
        if (!param.quiet) {
            double loss = 0;
            float scale = 1;
            if (mfx_job_last_loss(job, &loss, &scale) != MFX_OK) return nullptr;
            printf("%4d%13.4f\n", iter, std::sqrt(loss / (double)count * (double)scale * scale));
            fflush(stdout);
        }
    }
    float *buf = (float *)malloc(sizeof(float) * (size_t)total); // caller frees with free()
    if (!buf) return nullptr;
    if (mfx_job_export(job, buf, total) != MFX_OK) {
        std::cerr << "utility_train: " << mfx_job_last_error() << std::endl;
        free(buf);
        return nullptr;
    }
    lens = (int)total;
    return buf;
}

float *utility_train(float *train_data, int train_triplet_num, double p_l2, double q_l2, int k,
                     int iters, double eta, int &lens)
{
    lens = 0;
    try {
        if (train_data == nullptr || train_triplet_num <= 0) return nullptr;
        if (g_devices > 1) {
            mf_parameter param = mf_get_default_param(); // reference mf/mf.cpp:3508-3513
            param.lambda_p2 = (mf_float)p_l2;
            param.lambda_q2 = (mf_float)q_l2;
            param.k = k;
            param.nr_iters = iters;
            param.eta = (mf_float)eta;
            if (!check_parameter(param)) return nullptr;
            return utility_train_job(train_data, train_triplet_num, param, lens);
        }
        // read_triplet, reference mf/mf.cpp:3367-3394, on the device: the float triples are uploaded once and
        // become nodes in HBM (64-bit index: no overflow past 715 M); m, n = largest ids + 1
        mf_problem tr;
        tr.m = 0;
        tr.n = 0;
        tr.nnz = train_triplet_num;
        tr.R = nullptr;
        void *dR = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_gpu_mutex);
            if (mfx_triplets_to_device(train_data, train_triplet_num, -1, &dR, &tr.m, &tr.n) != MFX_OK) {
                std::cerr << "utility_train: " << mfx_last_error() << std::endl;
                return nullptr;
            }
        }
        struct FreeDev {
            void *p;
            ~FreeDev() { mfx_device_free(p); }
        } free_dr{dR};

        mf_parameter param = mf_get_default_param(); // reference mf/mf.cpp:3508-3513
        param.lambda_p2 = (mf_float)p_l2;
        param.lambda_q2 = (mf_float)q_l2;
        param.k = k;
        param.nr_iters = iters;
        param.eta = (mf_float)eta;
        if (!check_parameter(param)) return nullptr;

        long long total = 5 + ((long long)tr.m + tr.n) * (long long)k;
        if (total > 2147483647LL) return nullptr; // lens is an int (reference mf/mf.cpp:3424-3425)

        std::vector<float> unused;
        float *result = nullptr; // caller frees with free() (reference model_to_array, mf/mf.cpp:3426)
        long long len = 0;
        {
            std::lock_guard<std::mutex> lock(g_gpu_mutex);
            int rc = train_to_array(&tr, param, unused, dR, &result, &len);
            if (rc != MFX_OK) {
                std::cerr << "utility_train: " << mfx_last_error() << std::endl;
                return nullptr;
            }
        }
        lens = (int)len;
        return result;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return nullptr;
    } catch (...) {
        return nullptr;
    }
}

float *utility_predict(float *test_arr, int test_triplet_num, float *model_arr, int model_arr_len)
{
    try {
        if (test_arr == nullptr || model_arr == nullptr || test_triplet_num < 0) return nullptr;
        float *out = (float *)malloc(sizeof(float) * (size_t)(test_triplet_num > 0 ? test_triplet_num : 1));
        if (!out) return nullptr;
        std::lock_guard<std::mutex> lock(g_gpu_mutex);
        // array_to_model's length check (reference mf/mf.cpp:3463-3467) lives in mfx_predict_array
        if (mfx_predict_array(model_arr, model_arr_len, test_arr, test_triplet_num, out) != MFX_OK) {
            std::cerr << "utility_predict: " << mfx_last_error() << std::endl;
            free(out);
            return nullptr;
        }
        return out;
    } catch (...) {
        return nullptr;
    }
}

// ---- text formats (reference mf/mf.cpp:4143-4225) and mf_my_train (3397-3413) ---------

mf_problem read_problem(char const *path)
{
    mf_problem prob;
    prob.m = 0;
    prob.n = 0;
    prob.nnz = 0;
    prob.R = nullptr;
    if (!path) return prob;
    std::ifstream f(path);
    if (!f.is_open()) return prob;
    std::vector<mf_node> rows;
    mf_node N;
    while (f >> N.u >> N.v >> N.r) { // one "u v r" per line
        if (N.u + 1 > prob.m) prob.m = N.u + 1;
        if (N.v + 1 > prob.n) prob.n = N.v + 1;
        rows.push_back(N);
    }
    prob.nnz = (mf_long)rows.size();
    prob.R = new mf_node[rows.size() ? rows.size() : 1];
    if (!rows.empty()) memcpy(prob.R, rows.data(), rows.size() * sizeof(mf_node));
    return prob;
}

mf_int mf_save_model(mf_model const *model, char const *path)
{
    if (model == nullptr || path == nullptr) return 1;
    std::ofstream f(path);
    if (!f.is_open()) return 1;
    f << "f " << model->fun << std::endl;
    f << "m " << model->m << std::endl;
    f << "n " << model->n << std::endl;
    f << "k " << model->k << std::endl;
    f << "b " << model->b << std::endl;
    for (int side = 0; side < 2; ++side) {
        const mf_float *base = side == 0 ? model->P : model->Q;
        const mf_int rows = side == 0 ? model->m : model->n;
        const char prefix = side == 0 ? 'p' : 'q';
        for (mf_int i = 0; i < rows; ++i) {
            const mf_float *row = base + (mf_long)i * model->k;
            f << prefix << i << " ";
            if (std::isnan(row[0])) { // unseen rows are written as "F" and zeros
                f << "F ";
                for (mf_int d = 0; d < model->k; ++d) f << 0 << " ";
            } else {
                f << "T ";
                for (mf_int d = 0; d < model->k; ++d) f << row[d] << " ";
            }
            f << std::endl;
        }
    }
    return 0;
}

mf_int mf_my_train(char const *tr_path, char const *model_path)
{
    try {
        mf_problem tr = read_problem(tr_path);
        mf_parameter param = mf_get_default_param();
        param.nr_iters = 40;
        mf_model *model = mf_train_with_validation(&tr, nullptr, param);
        delete[] tr.R;
        if (model == nullptr) return -1;
        mf_int status = mf_save_model(model, model_path);
        mf_destroy_model(&model);
        return status;
    } catch (...) {
        return -1;
    }
}

} // namespace mf
