/* mfwarp.h -- the extern "C" upper face of the boundary.
 *
 * Mirrors reference php_mf/mfWarp.h:6-10 (definitions php_mf/mfWarp.cpp:3-34),
 * which php_mf/php_mf.c:36-40 re-declares as plain C externs.  Same names,
 * argument order and meaning; outputs are malloc'd by the callee.
 */
#ifndef MFX_MFWARP_H
#define MFX_MFWARP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ref mfWarp.h:6  -> mf::mf_my_train */
int php_mf_my_train(char *tr_path, char *model_path);
/* ref mfWarp.h:7  -> mf::utility_train (the int& becomes int*) */
float *php_utility_train(float *train_data, int train_triplet_num, double p_l2, double q_l2,
                         int k, int iters, double eta, int *lens);
/* ref mfWarp.h:8  -> mf::utility_predict */
float *php_utility_predict(float *test_arr, int test_triplet_num, float *model_arr,
                           int model_arr_len);
/* ref mfWarp.h:9  -> mf::cos_similarity */
float *php_cos_similarity(int item_id, float *q_arr, int q_arr_num);
/* ref mfWarp.h:10 -> mf::DINA */
int *php_DINA(float *q_arr, int q_triplet_num, float *x_arr, int x_triplet_num, int iterators);

#ifdef __cplusplus
}
#endif
#endif
