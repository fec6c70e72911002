// php_shim.cpp -- the extern "C" upper face of the boundary (include/mfwarp.h): five thunks
// from the C names the Zend glue calls (reference php_mf/php_mf.c:36-40, declared in
// php_mf/mfWarp.h:6-10) to the C++ facade in namespace mf.  Built into lib/libmfwarp.so so
// the upper face can be exercised on its own; tests/test_abi.py also links the reference's
// unchanged php_mf/mfWarp.cpp against lib/libmf.so.
#include "../../include/mf.h"
#include "../../include/mfwarp.h"

namespace {
// the facade returns the model length through an int&; at the C boundary it is an int*
inline float *train_thunk(float *triplets, int count, double lp, double lq, int k, int iters,
                          double eta, int *out_len)
{
    int len = 0;
    float *model = mf::utility_train(triplets, count, lp, lq, k, iters, eta, len);
    if (out_len != nullptr) *out_len = len;
    return model;
}
} // namespace

extern "C" {

int php_mf_my_train(char *ratings_file, char *model_file)
{
    return mf::mf_my_train(ratings_file, model_file);
}

float *php_utility_train(float *triplets, int count, double lp, double lq, int k, int iters,
                         double eta, int *out_len)
{
    return train_thunk(triplets, count, lp, lq, k, iters, eta, out_len);
}

float *php_utility_predict(float *pairs, int count, float *model, int model_len)
{
    return mf::utility_predict(pairs, count, model, model_len);
}

float *php_cos_similarity(int item, float *q_triplets, int count)
{
    return mf::cos_similarity(item, q_triplets, count);
}

int *php_DINA(float *q_triplets, int q_count, float *x_triplets, int x_count, int em_iters)
{
    return mf::DINA(q_triplets, q_count, x_triplets, x_count, em_iters);
}

} // extern "C"
