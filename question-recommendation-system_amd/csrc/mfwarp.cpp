// mfwarp.cpp -- extern "C" shim php_* -> mf::* (include/mfwarp.h).
// Own equivalent of reference php_mf/mfWarp.cpp:3-34; the reference's unchanged file also
// links against this libmf.so (tests/test_abi.py builds it from /root/reference when present).
#include "../../include/mf.h"
#include "../../include/mfwarp.h"

extern "C" {

int php_mf_my_train(char *tr_path, char *model_path) { return mf::mf_my_train(tr_path, model_path); }

float *php_utility_train(float *train_data, int train_triplet_num, double p_l2, double q_l2, int k,
                         int iters, double eta, int *lens)
{
    int n = 0;
    float *r = mf::utility_train(train_data, train_triplet_num, p_l2, q_l2, k, iters, eta, n);
    if (lens) *lens = n;
    return r;
}

float *php_utility_predict(float *test_arr, int test_triplet_num, float *model_arr, int model_arr_len)
{
    return mf::utility_predict(test_arr, test_triplet_num, model_arr, model_arr_len);
}

float *php_cos_similarity(int item_id, float *q_arr, int q_arr_num)
{
    return mf::cos_similarity(item_id, q_arr, q_arr_num);
}

int *php_DINA(float *q_arr, int q_triplet_num, float *x_arr, int x_triplet_num, int iterators)
{
    return mf::DINA(q_arr, q_triplet_num, x_arr, x_triplet_num, iterators);
}
}
