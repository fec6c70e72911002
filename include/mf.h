// mf.h -- public interface of the MI355X-native libmf.so (namespace mf).
//
// This header is the lower face of the drop-in boundary: an unchanged
// libphp_mf.so / mfTest from the reference binds these C++ (Itanium-mangled)
// symbols, because the reference keeps its `extern "C"` commented out
// (reference mf/mf.h:19-20,156).  Every declaration below names the reference
// declaration it replaces; signatures and struct layouts are identical so the
// mangled names and the ABI match:
//
//   _ZN2mf13utility_trainEPfiddiidRi   mf::utility_train     (ref mf/mf.h:95-102)
//   _ZN2mf15utility_predictEPfiS0_i    mf::utility_predict   (ref mf/mf.h:104-107)
//   _ZN2mf11mf_my_trainEPKcS1_         mf::mf_my_train       (ref mf/mf.h:93)
//   _ZN2mf14cos_similarityEiPfi        mf::cos_similarity    (ref mf/mf.h:109)
//   _ZN2mf4DINAEPfiS0_ii               mf::DINA              (ref mf/mf.h:111)
//
// Only what the SGD training / prediction path needs is declared; the
// reference's on-disk, cross-validation and ranking-metric entry points are out
// of scope.
#ifndef MFX_MF_H
#define MFX_MF_H

#if defined(__GNUC__) && __GNUC__ >= 4
#define MF_API __attribute__((visibility("default")))
#else
#define MF_API
#endif

namespace mf {

typedef float mf_float;
typedef double mf_double;
typedef int mf_int;
typedef long long mf_long;

// loss selector; only P_L2_MFR is reachable through the facade (ref mf/mf.h:31-32)
enum { P_L2_MFR = 0 };

// one rating, 12 bytes, no padding (ref mf/mf.h:36-41)
struct mf_node {
    mf_int u;
    mf_int v;
    mf_float r;
};

// borrowed COO matrix (ref mf/mf.h:43-49)
struct mf_problem {
    mf_int m;
    mf_int n;
    mf_long nnz;
    struct mf_node *R;
};

// hyper-parameters, passed by value (ref mf/mf.h:51-66)
struct mf_parameter {
    mf_int fun;
    mf_int k;
    mf_int nr_threads;  // accepted for ABI compatibility; the GPU schedule ignores it
    mf_int nr_bins;     // accepted for ABI compatibility; the GPU stripe grid is sized by the device
    mf_int nr_iters;
    mf_float lambda_p1; // must be 0 on this path
    mf_float lambda_p2;
    mf_float lambda_q1; // must be 0 on this path
    mf_float lambda_q2;
    mf_float eta;
    bool do_nmf;        // must be false on this path
    bool quiet;
    bool copy_data;
};

// trained factors: P is m x k, Q is n x k, row-major, original ids (ref mf/mf.h:70-79)
struct mf_model {
    mf_int fun;
    mf_int m;
    mf_int n;
    mf_int k;
    mf_float b;
    mf_float *P;
    mf_float *Q;
};

// ref mf/mf.h:68, defaults mf/mf.cpp:4538-4557
MF_API struct mf_parameter mf_get_default_param();

// ref mf/mf.h:87-89 (mf.cpp:3362-3365): train on the GPU, model returned on the host
MF_API struct mf_model *mf_train(struct mf_problem const *prob, struct mf_parameter param);

// ref mf/mf.h:117-120 (mf.cpp:3307-3332); `va` must be null or empty on this path
MF_API struct mf_model *mf_train_with_validation(struct mf_problem const *tr,
                                                 struct mf_problem const *va,
                                                 struct mf_parameter param);

// ref mf/mf.h:85 (mf.cpp:4280-4293)
MF_API void mf_destroy_model(struct mf_model **model);

// ref mf/mf.h:137 (mf.cpp:4295-4314): host-side single prediction
MF_API mf_float mf_predict(struct mf_model const *model, mf_int u, mf_int v);

// ref mf/mf.h:139 (mf.cpp:4316-4331): batched on the GPU
MF_API mf_double calc_rmse(mf_problem *prob, mf_model *model);

// ref mf/mf.h:81,83 (mf.cpp:4143-4182, 4184-4225): text formats used by mf_my_train
MF_API mf_problem read_problem(char const *path);
MF_API mf_int mf_save_model(struct mf_model const *model, char const *path);

// ---- the float-array facade the PHP extension imports ----

// ref mf/mf.h:95-102 (mf.cpp:3483-3535).  train_data = (u,v,r) float triplets;
// returns malloc'd [fun,m,n,k,b,P...,Q...] and its length in `lens`.
// On any failure returns NULL with lens = 0 and never throws.
MF_API float *utility_train(float *train_data, int train_triplet_num, double p_l2,
                            double q_l2, int k, int iters, double eta, int &lens);

// ref mf/mf.h:104-107 (mf.cpp:3537-3568).  test_arr = (u,v) float pairs; returns
// malloc'd float[test_triplet_num]; NULL when model_arr_len does not match the header.
MF_API float *utility_predict(float *test_arr, int test_triplet_num, float *model_arr,
                              int model_arr_len);

// ref mf/mf.h:93 (mf.cpp:3397-3413): text file in, text model out, 40 iterations
MF_API mf_int mf_my_train(char const *tr_path, char const *model_path);

// ref mf/mf.h:109,111 (mf.cpp:3591-3683, 3685-4109): off-path symbols the
// extension imports; host-side, present so the library binds.
MF_API float *cos_similarity(int item_id, float *q_arr, int q_arr_num);
MF_API int *DINA(float *q_arr, int q_triplet_num, float *x_arr, int x_triplet_num,
                 int iterators);

} // namespace mf

#endif // MFX_MF_H
