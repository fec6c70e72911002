// extras.cpp -- the two off-path symbols an unchanged libphp_mf.so imports from libmf.so
// besides the trainer: mf::cos_similarity and mf::DINA (reference mf/mf.h:109,111;
// bodies mf/mf.cpp:3591-3683 and 3685-4109).  They are small dense host algorithms
// (O(items^2) and O(2^items)), unrelated to the SGD path, kept on the host and written
// here from their observable behaviour so the library binds and answers the same way.
// Differences, on purpose: matrices are zero-initialised instead of left uninitialised
// for cells no triplet names, indices are bounds-checked, and DINA's slip/guess start
// values come from a private generator instead of the process-global rand().
#include <cmath>
#include <cstdlib>
#include <vector>

#include "../../include/mf.h"
#include "rng.hpp"

namespace mf {

namespace {

// dense int matrix from (row, col, value) float triplets; dims = max index + 1
struct Dense {
    int rows = 0, cols = 0;
    std::vector<int> a;
    int &at(int r, int c) { return a[(size_t)r * cols + c]; }
    int get(int r, int c) const { return a[(size_t)r * cols + c]; }
};

bool densify(const float *tri, int count, Dense &d, int fill)
{
    if (!tri || count <= 0) return false;
    for (int j = 0; j < count; ++j) { // read_triplet semantics, reference mf/mf.cpp:3367-3394
        int r = (int)tri[3 * j], c = (int)tri[3 * j + 1];
        if (r < 0 || c < 0) return false;
        if (r + 1 > d.rows) d.rows = r + 1;
        if (c + 1 > d.cols) d.cols = c + 1;
    }
    d.a.assign((size_t)d.rows * d.cols, fill);
    for (int j = 0; j < count; ++j) d.at((int)tri[3 * j], (int)tri[3 * j + 1]) = (int)tri[3 * j + 2];
    return true;
}

} // namespace

float *cos_similarity(int item_id, float *q_arr, int q_arr_num)
{
    try {
        Dense q;
        if (!densify(q_arr, q_arr_num, q, 0)) return nullptr;
        if (item_id < 0 || item_id >= q.rows) return nullptr;
        const int items = q.rows, dims = q.cols;
        std::vector<float> sim(items), ids(items);
        int self = 0;
        for (int d = 0; d < dims; ++d) self += q.get(item_id, d) * q.get(item_id, d);
        for (int i = 0; i < items; ++i) {
            int dot = 0, norm = 0;
            for (int d = 0; d < dims; ++d) {
                dot += q.get(item_id, d) * q.get(i, d);
                norm += q.get(i, d) * q.get(i, d);
            }
            sim[i] = (float)(dot / (std::sqrt((double)self) * std::sqrt((double)norm)));
            ids[i] = (float)i;
        }
        // descending exchange sort, same visiting order as the reference so ties (and
        // NaNs from all-zero rows) land where they do there (mf.cpp:3646-3661)
        for (int i = 0; i + 1 < items; ++i)
            for (int j = i + 1; j < items; ++j)
                if (sim[i] < sim[j]) {
                    std::swap(sim[i], sim[j]);
                    std::swap(ids[i], ids[j]);
                }
        float *result = (float *)calloc((size_t)items, sizeof(float));
        if (!result) return nullptr;
        for (int i = 0; i < items; ++i) result[i] = ids[i];
        return result;
    } catch (...) {
        return nullptr;
    }
}

int *DINA(float *q_arr, int q_triplet_num, float *x_arr, int x_triplet_num, int iterators)
{
    try {
        Dense q, x;
        if (!densify(q_arr, q_triplet_num, q, 0)) return nullptr;
        if (!densify(x_arr, x_triplet_num, x, -1)) return nullptr;
        const int items = q.rows, skills = q.cols, users = x.rows;
        if (items > 24) return nullptr; // the state space is 2^items (mf.cpp:3764)
        const int states = 1 << items;

        // skill pattern of state s: the `skills` low bits of (states-1-s), MSB first
        // (convert(), mf.cpp:3570-3589)
        auto bit = [&](int s, int k) {
            int v = states - 1 - s, shift = skills - 1 - k;
            return shift < 31 ? (v >> shift) & 1 : 0;
        };
        // mastered[j][s]: every cell of item j's Q row that equals the pattern, counted
        // against the number of skills the item needs (mf.cpp:3829-3841)
        std::vector<int> need(items, 0);
        for (int j = 0; j < items; ++j)
            for (int k = 0; k < skills; ++k) need[j] += q.get(j, k) == 1;
        std::vector<char> mastered((size_t)items * states);
        for (int j = 0; j < items; ++j)
            for (int s = 0; s < states; ++s) {
                int same = 0;
                for (int k = 0; k < skills; ++k) same += q.get(j, k) == bit(s, k);
                mastered[(size_t)j * states + s] = same == need[j];
            }
        auto answer = [&](int u, int j) { return j < x.cols ? x.get(u, j) : -1; };

        mfx::GlibcRand rng(1);
        std::vector<float> slip(items), guess(items);
        for (int j = 0; j < items; ++j) {
            slip[j] = rng.next() % 100 / (float)100;
            guess[j] = rng.next() % 100 / (float)100;
        }
        std::vector<float> prior(states, (float)(1.0 / states));
        std::vector<float> post((size_t)users * states, 1.0f); // carried across iterations
        std::vector<float> r_state((size_t)items * states), i_state(states);

        for (int iter = 1; iter < iterators; ++iter) {
            // E step: likelihood of each state per user, times prior, normalised
            for (int u = 0; u < users; ++u)
                for (int j = 0; j < items; ++j) {
                    int ans = answer(u, j);
                    if (ans == -1) continue;
                    for (int s = 0; s < states; ++s) {
                        bool ok = mastered[(size_t)j * states + s];
                        float f = ans == 1 ? (ok ? 1 - slip[j] : guess[j])
                                           : (ok ? slip[j] : 1 - guess[j]);
                        post[(size_t)u * states + s] *= f;
                    }
                }
            for (int u = 0; u < users; ++u) {
                float *row = &post[(size_t)u * states];
                for (int s = 0; s < states; ++s) row[s] *= prior[s];
                float sum = 0;
                for (int s = 0; s < states; ++s) sum += row[s];
                for (int s = 0; s < states; ++s) row[s] = (float)(row[s] / sum);
            }
            // expected counts
            for (int j = 0; j < items; ++j)
                for (int s = 0; s < states; ++s) {
                    float acc = 0;
                    for (int u = 0; u < users; ++u) {
                        int ans = answer(u, j);
                        if (ans != -1) acc += post[(size_t)u * states + s] * ans;
                    }
                    r_state[(size_t)j * states + s] = acc;
                }
            for (int s = 0; s < states; ++s) {
                float acc = 0;
                for (int u = 0; u < users; ++u) acc += post[(size_t)u * states + s];
                i_state[s] = acc;
            }
            // M step: slip, guess, prior (mf.cpp:3996-4011)
            for (int j = 0; j < items; ++j) {
                float r0 = 0, r1 = 0, i0 = 0, i1 = 0;
                for (int s = 0; s < states; ++s) {
                    if (mastered[(size_t)j * states + s]) {
                        r1 += r_state[(size_t)j * states + s];
                        i1 += i_state[s];
                    } else {
                        r0 += r_state[(size_t)j * states + s];
                        i0 += i_state[s];
                    }
                }
                slip[j] = (i1 - r1) / i1;
                guess[j] = r0 / i0;
            }
            for (int s = 0; s < states; ++s) prior[s] = i_state[s] / users;
        }

        int *res = (int *)malloc(sizeof(int) * (size_t)(users > 0 ? users : 1) * (size_t)(skills > 0 ? skills : 1));
        if (!res) return nullptr;
        for (int u = 0; u < users; ++u) {
            const float *row = &post[(size_t)u * states];
            int best = 0;
            for (int s = 0; s < states; ++s)
                if (row[best] < row[s]) best = s;
            for (int k = 0; k < skills; ++k) res[(size_t)u * skills + k] = bit(best, k);
        }
        return res;
    } catch (...) {
        return nullptr;
    }
}

} // namespace mf
