// plan.cpp -- see plan.hpp.
#include "plan.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <queue>
#include <stdexcept>
#include <thread>

#include "knobs.hpp"
#include "rng.hpp"

namespace mfx {

int lanes_for(int ka)
{
    int need = (ka + 3) / 4, l = 2;
    while (l < need && l < 64) l <<= 1; // (k_a > 256: a whole wavefront per rating, several float4 per lane)
    return l;
}

void parallel_ranges(long long n, int threads,
                     const std::function<void(long long, long long, int)> &fn)
{
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    if (n < 4096 || threads == 1) {
        fn(0, n, 0);
        return;
    }
    std::vector<std::thread> pool;
    long long chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        long long b = t * chunk, e = std::min(n, b + chunk);
        if (b >= e) break;
        pool.emplace_back(fn, b, e, t);
    }
    for (auto &th : pool) th.join();
}

void gen_random_map(int size, std::vector<int> &map)
{
    // srand(0); iota; std::random_shuffle  (reference mf/mf.cpp:1009-1017)
    GlibcRand rng(0);
    map.resize(size);
    for (int i = 0; i < size; ++i) map[i] = i;
    for (int i = 1; i < size; ++i) {
        int j = rng.next() % (i + 1);
        if (i != j) std::swap(map[i], map[j]);
    }
}

// mean / standard deviation in double (reference mf/mf.cpp:462-484).  Fixed 1M-rating
// chunks summed in order, so the result does not depend on the thread count.
static void collect_info(const Node *R, long long nnz, int threads, float &avg, float &sd)
{
    const long long CH = 1 << 20;
    long long nch = (nnz + CH - 1) / CH;
    std::vector<double> s1(nch), s2(nch);
    parallel_ranges(nch, threads, [&](long long b, long long e, int) {
        for (long long c = b; c < e; ++c) {
            double a = 0, q = 0;
            long long hi = std::min(nnz, (c + 1) * CH);
            for (long long i = c * CH; i < hi; ++i) {
                a += (double)R[i].r;
                q += (double)R[i].r * R[i].r;
            }
            s1[c] = a;
            s2[c] = q;
        }
    });
    double ex = 0, ex2 = 0;
    for (long long c = 0; c < nch; ++c) {
        ex += s1[c];
        ex2 += s2[c];
    }
    ex /= (double)nnz;
    ex2 /= (double)nnz;
    avg = (float)ex;
    sd = (float)std::sqrt(ex2 - ex * ex);
}

namespace {

struct Rat { uint32_t own; uint32_t gat; float r; }; // (primary, secondary) of the sort: owner/gathered id, swapped for heavy gathered rows

// Pack the visits of the wave tasks (ordinary rows: at most hot_len ratings in the block) into `ntasks` tasks of G lists of
// equal load: longest-processing-time-first, then every list runs its visits in ascending row id.
void pack_class(const std::vector<Visit> &visits, size_t vbeg, size_t vend, int G, int target,
                BlockPack &out, long long force_ntasks = 0)
{
    if (vbeg >= vend) return;
    const int list_order = knob_int("MFX_LIST_ORDER", 1);
    long long L = 0;
    for (size_t i = vbeg; i < vend; ++i) L += visits[i].len;
    target = std::max<long long>(target, visits[vbeg].len); // visits are sorted, longest first
    long long ntasks = (L + (long long)G * target - 1) / ((long long)G * target);
    if (force_ntasks > 0) ntasks = force_ntasks;
    if (ntasks < 1) ntasks = 1;
    const long long NG = ntasks * G;
    // Longest-processing-time-first: every visit goes to the list with the smallest load (ties: smallest list index).
    // A binary min-heap of (load << 32 | list) keys, all distinct; the top is replaced and sifted down (no pop + push).
    std::vector<uint64_t> heap((size_t)NG);
    for (long long b = 0; b < NG; ++b) heap[(size_t)b] = (uint64_t)b; // ascending = a valid heap
    std::vector<uint32_t> load(NG, 0);
    // the visits of a list as a chain of indices (head / tail per list, next per visit), in packing order
    const uint32_t NIL = 0xFFFFFFFFu;
    std::vector<uint32_t> head((size_t)NG, NIL), tail((size_t)NG, NIL), next(vend - vbeg, NIL), count((size_t)NG, 0);
    for (size_t vi = vbeg; vi < vend; ++vi) {
        const uint64_t top = heap[0];
        const uint32_t lst = (uint32_t)top;
        if (tail[lst] == NIL) head[lst] = (uint32_t)(vi - vbeg);
        else next[tail[lst]] = (uint32_t)(vi - vbeg);
        tail[lst] = (uint32_t)(vi - vbeg);
        ++count[lst];
        const uint64_t key = top + ((uint64_t)visits[vi].len << 32);
        load[lst] = (uint32_t)(key >> 32);
        size_t i = 0;
        const size_t n = (size_t)NG;
        for (;;) { // sift the new key down
            size_t c = 2 * i + 1;
            if (c >= n) break;
            if (c + 1 < n && heap[c + 1] < heap[c]) ++c;
            if (heap[c] >= key) break;
            heap[i] = heap[c];
            i = c;
        }
        heap[i] = key;
    }
    std::vector<uint32_t> lv; // the visits of the list being emitted
    std::vector<uint32_t> order(NG);
    for (long long b = 0; b < NG; ++b) order[b] = (uint32_t)b;
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t a, uint32_t b) { return load[a] > load[b]; });
    for (long long t = 0; t < ntasks; ++t) {
        uint32_t nsteps = load[order[t * G]];
        if (nsteps == 0) break; // the remaining lists are empty
        TaskDesc td;
        td.off = out.n_entries;
        td.nsteps = nsteps;
        td.pad = 0;
        const uint64_t base = out.n_entries;
        out.n_entries += (uint64_t)nsteps * G;
        for (int g = 0; g < G; ++g) {
            uint32_t lst = order[t * G + g];
            uint32_t step = 0;
            // Which visits a list holds comes from the longest-first packing above (equal loads); the ORDER they are
            // run in does not matter for the load.  As packed (round 1) every launch did all its heavy rows first and
            // all its light rows last -- a systematic order the reference does not have (it walks a block sorted by
            // row id, mf.cpp:843-852), worth -0.6 % (20 M sample) to -1.9 % (configs[2]) of final RMSE by itself in the
            // order emulation.  Lists therefore run their visits by owner row id (the reference's own
            // order inside a block; experiment knob MFX_LIST_ORDER: 0 as packed, 2 scattered by a hash).
            lv.clear();
            lv.reserve(count[lst]);
            for (uint32_t x = head[lst]; x != NIL; x = next[x]) lv.push_back((uint32_t)(x + vbeg));
            if (list_order == 1)
                std::sort(lv.begin(), lv.end(), [&](uint32_t a, uint32_t b) { return visits[a].own < visits[b].own; });
            else if (list_order == 2)
                std::sort(lv.begin(), lv.end(), [&](uint32_t a, uint32_t b) {
                    const uint32_t ha = visits[a].own * 2654435761u, hb = visits[b].own * 2654435761u;
                    return ha != hb ? ha < hb : a < b;
                });
            for (uint32_t vi : lv) {
                const Visit &v = visits[vi];
                out.places.push_back({v.start, base + (uint64_t)step * G + g, v.len, 1u | 0x40000000u});
                step += v.len;
            }
            out.padding += nsteps - step;
        }
        out.tasks.push_back(td);
    }
}

// number of workgroups a heavy visit is split over: one, unless it holds more than a workgroup does in the launch
inline long long visit_copies(uint32_t len, long long WGL, long long T)
{
    const long long steps = ((long long)len + WGL - 1) / WGL;
    return std::max<long long>(1, (steps + T - 1) / T);
}

struct Piece { uint32_t vi; uint32_t idx; uint32_t steps; uint32_t len; uint64_t start; };

} // namespace

// How a block is cut: T = steps one wave does if the block is dealt out evenly over the `waves` the concurrency cap allows;
// Th[role] = steps of one piece of a heavy row (a piece = what one workgroup does of it), wgs[role] = workgroup tasks the
// role's heavy rows may fill.
//
// A heavy row is cut into pieces of T steps -- unless the block's heavy work does not fit the workgroups there are at T
// steps each (a block that is little else than one row: one GPU's slot of a strong split holds such blocks).  Then the
// pieces are cut longer, so that they fill the workgroups evenly: 31 pieces of T steps on 30 workgroups make one
// workgroup -- and the launch -- take twice as long.
//
// WIDE launches (wgs_hw > waves / W: the concurrency cap holds the launch below what the chip can run).  The cap exists
// because two lists that read-modify-write one row at the same time lose an update.  The W x G lists of a heavy row -- and
// those of all its copies -- hold DISJOINT ranges of the other side's rows (its ratings are sorted by that id and dealt
// in contiguous runs): they never meet each other on a row, and for any row of the other side the whole heavy row counts
// as ONE list.  So the heavy rows do not take their workgroups out of the capped ones: they run on the workgroups the cap
// leaves idle, cut so that they are done when the ordinary rows are (never finer than that: fewer copies fold better).
struct BlockShape { long long T; long long Th[2]; long long wgs[2]; long long waves_main; bool wide; };
static BlockShape block_shape(const std::vector<Visit> &raw, int G, int W, int waves, int wgs_hw, int hot_len)
{
    const long long WGL = (long long)W * G;
    BlockShape sh;
    long long hs[2] = {0, 0}, L_main = 0;
    for (const Visit &v : raw) {
        if ((long long)v.len <= hot_len) L_main += v.len;
        else hs[v.swapped ? 1 : 0] += ((long long)v.len + WGL - 1) / WGL;
    }
    const long long wgs_cap = std::max(1, waves / std::max(1, W));
    const long long hw = std::max<long long>(wgs_hw, wgs_cap);
    sh.wide = hw > wgs_cap;
    const long long hsum = hs[0] + hs[1];
    // A step of a workgroup task costs more than a step of a wave task: the lists' changes of the heavy row are summed
    // across the wave and added to LDS with atomics.  The weights are where the epoch time of configs[1] (k = 32) and
    // configs[2] (k = 64) is flat in them (profiles/experiments/r03_wide_launches.log, last section: 16 .. 20 and 10 .. 12
    // tenths); at k = 128 the two kinds of step measured equal.  (A cost model for the balance of a launch only; no result
    // depends on it.)
    const long long c10 = knob_int("MFX_WG_COST10", G >= 8 ? 18 : G == 4 ? 11 : 10); // cost of a workgroup step in tenths of a wave step
    // n workgroups for the heavy rows, the others (at most the capped ones) for the ordinary rows: the launch takes
    // max(heavy steps / n, ordinary ratings / lists of the others).  The smallest n that reaches the minimum: fewer copies of
    // a row fold better.
    long long best_n = 0, best_x = 0;
    auto ord_steps = [&](long long n) {
        const long long w = std::min(wgs_cap, hw - n);
        return w > 0 ? (L_main + w * WGL - 1) / (w * WGL) : (L_main ? (long long)1 << 40 : 0);
    };
    if (hsum == 0) {
        best_n = 0;
        best_x = ord_steps(0);
    } else {
        const long long n_max = L_main > 0 ? std::max<long long>(1, hw - 1) : hw;
        best_n = 1;
        best_x = std::max((hsum * c10 + 9) / 10, ord_steps(1));
        for (long long n = 2; n <= n_max; ++n) {
            const long long x = std::max((hsum * c10 + 10 * n - 1) / (10 * n), ord_steps(n));
            if (x < best_x) {
                best_x = x;
                best_n = n;
            }
        }
    }
    sh.T = std::max<long long>(1, best_x);
    sh.waves_main = std::max<long long>(std::min<long long>(W, waves), W * std::min(wgs_cap, hw - best_n));
    for (int role = 0; role < 2; ++role) {
        sh.wgs[role] = hsum ? std::max<long long>(1, (best_n * hs[role] + hsum / 2) / hsum) : 1;
        // (at least 16 steps a piece in a wide launch: a visit costs two barriers and the staging of its row)
        sh.Th[role] = std::max<long long>(std::max<long long>(sh.T * 10 / c10, sh.wide ? 16 : 1), hs[role] ? (hs[role] + sh.wgs[role] - 1) / sh.wgs[role] : 1);
    }
    return sh;
}

// Cut one (owner-stripe, gather-stripe) block, given as its visits, into workgroup tasks (heavy rows) and wave tasks.
static void pack_block(std::vector<Visit> &raw, int G, int W, int waves, int wgs_hw, int target, int hot_len, int tasks_per_wave,
                       const std::vector<int> &slot_own, const std::vector<int> &slot_gat, BlockPack &out)
{
    if (raw.empty()) return;
    const long long WGL = (long long)W * G;
    const BlockShape shape = block_shape(raw, G, W, waves, wgs_hw, hot_len);
    const long long *const Th = shape.Th;
    std::vector<Visit> ordinary[2]; // the ordinary ones, by role
    ordinary[0].reserve(raw.size());
    std::vector<Piece> pieces[2]; // by role
    for (size_t i = 0; i < raw.size(); ++i) {
        const Visit &v = raw[i];
        if ((long long)v.len <= hot_len) {
            ordinary[v.swapped ? 1 : 0].push_back(v);
            continue;
        }
        out.hot++;
        const long long steps = ((long long)v.len + WGL - 1) / WGL, nc = visit_copies(v.len, WGL, Th[v.swapped ? 1 : 0]);
        const long long per = (steps + nc - 1) / nc; // steps per copy
        uint32_t idx = 0;
        for (long long s0 = 0; s0 < steps; s0 += per, ++idx) {
            const long long s1 = std::min(steps, s0 + per);
            const uint64_t first = (uint64_t)s0 * WGL;
            const uint32_t len = (uint32_t)(std::min<uint64_t>(v.len, (uint64_t)s1 * WGL) - first);
            pieces[v.swapped ? 1 : 0].push_back({(uint32_t)i, idx, (uint32_t)(s1 - s0), len, v.start + first});
        }
    }
    // workgroup tasks, per role: as many as the heavy work fills at T steps each, the pieces dealt longest first
    int wgs_used = 0;
    for (int role = 0; role < 2; ++role) {
        std::vector<Piece> &pc = pieces[role];
        if (pc.empty()) continue;
        long long steps = 0;
        for (const Piece &q : pc) steps += q.steps;
        long long nt = std::max<long long>(1, (steps + Th[role] - 1) / Th[role]);
        nt = std::min<long long>(nt, (long long)pc.size());
        nt = std::min<long long>(nt, std::max<long long>(1, shape.wgs[role]));
        std::stable_sort(pc.begin(), pc.end(), [](const Piece &a, const Piece &b) { return a.steps > b.steps; });
        std::vector<long long> load;
        std::vector<std::vector<uint32_t>> of;
        for (;;) { // longest piece first into the emptiest task; one more task while the fullest is far above the mean
            load.assign((size_t)nt, 0);
            of.assign((size_t)nt, std::vector<uint32_t>());
            for (uint32_t i = 0; i < pc.size(); ++i) {
                size_t best = 0;
                for (size_t t = 1; t < (size_t)nt; ++t)
                    if (load[t] < load[best]) best = t;
                of[best].push_back(i);
                load[best] += pc[i].steps;
            }
            const long long top = *std::max_element(load.begin(), load.end());
            if (top * 4 <= std::max(Th[role], (steps + nt - 1) / nt) * 5 || nt >= (long long)pc.size() || nt >= shape.wgs[role] + (shape.wide ? 2 : 0)) break;
            ++nt;
        }
        for (size_t t = 0; t < (size_t)nt; ++t) {
            std::vector<uint32_t> &lst = of[t];
            // by row id, like the lists of the wave tasks (the reference walks a block sorted by row, mf.cpp:843-852)
            std::sort(lst.begin(), lst.end(), [&](uint32_t a, uint32_t b) {
                const Visit &va = raw[pc[a].vi], &vb = raw[pc[b].vi];
                return va.own != vb.own ? va.own < vb.own : pc[a].idx < pc[b].idx;
            });
            WgTask wt;
            wt.off = out.n_entries;
            wt.nsteps = (uint32_t)load[t];
            wt.visit0 = (uint32_t)out.wg_visits.size();
            wt.nvisits = (uint32_t)lst.size();
            wt.swapped = (uint32_t)role;
            const uint64_t wave_stride = (uint64_t)wt.nsteps * G;
            out.n_entries += wave_stride * W;
            uint32_t step0 = 0;
            long long rated = 0;
            for (uint32_t pi : lst) {
                const Piece &q = pc[pi];
                const Visit &v = raw[q.vi];
                const long long nc = visit_copies(v.len, WGL, Th[role]);
                WgVisitRec rec;
                rec.v.row = v.own;
                rec.v.nsteps = q.steps;
                rec.v.len = q.len;
                rec.v.info = ((uint32_t)nc << 1) | (uint32_t)role;
                rec.slot = 0;
                rec.pad = 0;
                if (nc > 1) {
                    const int sl = (role ? slot_gat : slot_own)[v.own];
                    if (sl < 0) throw std::logic_error("split row without a combine slot");
                    rec.slot = (uint32_t)sl;
                    out.folds = true;
                }
                out.wg_visits.push_back(rec);
                // The piece's ratings (sorted by the other side's id) are dealt over the W x G lists in CONTIGUOUS runs of
                // `steps` ratings, not round-robin: ratings of one pair -- the synthetic streams repeat pairs, a heavy user
                // rates a heavy item thousands of times -- then sit one behind the other in ONE list, where the second
                // sees what the first did; side by side in one step they would all read the other row before any of them
                // wrote it.
                for (long long l = 0; l < WGL; ++l) {
                    const long long first = l * (long long)q.steps;
                    if (first >= (long long)q.len) break;
                    const uint32_t cnt = (uint32_t)std::min<long long>(q.steps, (long long)q.len - first);
                    const long long w = l / G, g = l % G;
                    const uint64_t dst = wt.off + (uint64_t)w * wave_stride + (uint64_t)step0 * G + (uint64_t)g;
                    const uint32_t fl = 1u | (role ? 0x80000000u : 0u);
                    // Every heavy row cuts the other side's id range into W x G runs the same way, and its lists walk their
                    // runs upwards at the same pace: list l of one heavy row and list l of another would reach a row they have
                    // in common at nearly the same step, launch after launch.  Each row therefore starts its runs at its own
                    // point (a hash of the row id) and wraps round.
                    const uint32_t at = (uint32_t)(((uint64_t)(v.own * 2654435761u) * cnt) >> 32);
                    out.places.push_back({q.start + (uint64_t)first + at, dst, cnt - at, fl});
                    if (at) out.places.push_back({q.start + (uint64_t)first, dst + (uint64_t)(cnt - at) * G, at, fl});
                }
                step0 += q.steps;
                rated += q.len;
            }
            out.padding += (long long)(wave_stride * W) - rated;
            out.wg_tasks.push_back(wt);
        }
        wgs_used += (int)nt;
    }
    // The ordinary rows: wave tasks, one class per role.  The waves that the workgroup tasks leave are dealt over the two roles
    // in proportion to their ratings.
    const int waves_main = (int)shape.waves_main; // the waves the heavy rows leave, at most the capped ones
    (void)wgs_used;
    long long L_role[2] = {0, 0};
    for (int role = 0; role < 2; ++role)
        for (const Visit &v : ordinary[role]) L_role[role] += v.len;
    for (int role = 0; role < 2; ++role) {
        std::vector<Visit> &visits = ordinary[role];
        if (visits.empty()) continue;
        // longest first, equal lengths in their given order: a counting sort (lengths are at most hot_len)
        uint32_t max_len = 0;
        for (const Visit &v : visits) max_len = std::max(max_len, v.len);
        {
            std::vector<size_t> at((size_t)max_len + 2, 0);
            for (const Visit &v : visits) at[(size_t)(max_len - v.len) + 1]++;
            for (size_t i = 1; i < at.size(); ++i) at[i] += at[i - 1];
            std::vector<Visit> sorted(visits.size());
            for (const Visit &v : visits) sorted[at[(size_t)(max_len - v.len)]++] = v;
            visits.swap(sorted);
        }
        const size_t t_first = out.tasks.size();
        if (tasks_per_wave > 0) { // equal-load tasks, (tasks per wave) x (waves of this role) of them
            long long wr = waves_main;
            if (L_role[1 - role] > 0) {
                // the waves are dealt over the two roles so that the longer of the two kinds of task is as short as it gets
                const long long tot = L_role[0] + L_role[1];
                long long w0 = std::max<long long>(1, std::min<long long>(waves_main - 1, waves_main * L_role[0] / tot));
                auto longest = [&](long long a0) { return std::max((L_role[0] + a0 - 1) / a0, (L_role[1] + (waves_main - a0) - 1) / (waves_main - a0)); };
                if (w0 + 1 <= waves_main - 1 && longest(w0 + 1) < longest(w0)) ++w0;
                wr = role == 0 ? w0 : waves_main - w0;
            }
            pack_class(visits, 0, visits.size(), G, 8, out, (long long)tasks_per_wave * wr);
        } else {
            pack_class(visits, 0, visits.size(), G, std::max(8, target), out); // explicit task_steps (tests): tasks of that size
        }
        for (size_t t = t_first; t < out.tasks.size(); ++t) out.tasks[t].pad = (uint32_t)role;
        if (role) // (the lists of these tasks are visits of rows of the gathered side)
            for (Placement &pl : out.places)
                if (pl.dst >= out.tasks[t_first].off && (pl.stride_flags & 0x40000000u)) pl.stride_flags |= 0x80000000u;
    }
}

void plan_sizes(long long nnz, int NB, int G, const PlanConfig &cfg, int &target, int &hot_len)
{
    const long long per_wave = nnz / ((long long)NB * G * std::max(1, cfg.waves_per_stripe));
    target = cfg.task_steps;
    if (target <= 0) target = (int)std::min<long long>(64, std::max<long long>(8, per_wave / 2));
    // A row with more ratings in a block than hot_len goes to a workgroup task.  A list must not outlast what one wave
    // does in the launch, or it sets the launch time: hot_len = per-wave load, within [48,128] (48: below it the ordinary
    // rows of small blocks would all become heavy).
    hot_len = knob_int("MFX_HOT_LEN", (int)std::min<long long>(128, std::max<long long>(48, per_wave)));
    if (hot_len < 8) hot_len = 8;
    // A launch of ONE workgroup per XCD (tiny problems) has nobody to run workgroup tasks beside the wave tasks: the
    // workgroup would do all heavy rows of a block first and all ordinary rows after them, an order the reference does
    // not have (+2 % final RMSE on 2000 x 1500 problems).  There a heavy row simply is one long list of a wave task --
    // every rating sees the one before it -- and the launch takes as long as that list; nobody times tiny problems.
    if (cfg.waves_per_stripe <= 4 && cfg.task_steps <= 0) hot_len = 0x3FFFFFFF; // (an explicit task size keeps the workgroup tasks: tests)
    if (k_aligned(cfg.k) > 256) hot_len = 0x3FFFFFFF; // rows wider than one float4 per lane: wave tasks only (kernels.hip sgd_round_wide)
}

void plan_hot_gathered(const PlanConfig &cfg, Plan &p)
{
    // A gathered row is read-modified-written by whatever lists hold one of its ratings at the moment; a row with more
    // ratings in a block than one list holds is, on average, in several lists at once and loses most of its updates (and
    // of its accumulator growth: its steps stay too large -- round 2 traced configs[2]'s -2 % to exactly these rows).
    // Such rows are taken out of the lock-free side: their ratings are grouped by THEM and run in workgroup tasks with the
    // roles swapped (mfx_options.no_swap switches it off).  Those visits read-modify-write owner rows while a wave task may
    // hold them in registers, so with such a plan every visit writes back what memory holds at its end plus what it
    // changed, not its copy (kernels.hip: merge_back; without that an owner row kept for most of a launch -- 20 k x 400 k:
    // 80 % of the time -- lost what the swapped visits did to it: +4.0 % against +1.0 % in the order emulation).
    // The test uses the row's global count (a block holds about 1/NS of it: the stripes have equal
    // mass), so it can be made per rating before anything is sorted: swapped iff the gathered row is heavy and heavier
    // than the rating's owner row.
    int target, hot_len;
    plan_sizes(p.nnz, p.ns * p.ns, p.groups, cfg, target, hot_len);
    p.hot_len = hot_len;
    p.waves_per_wg = std::max(1, std::min(4, cfg.waves_per_wg));
    const std::vector<int> &og = p.owner_is_q ? p.omega_p : p.omega_q;
    p.hot_gat.assign(og.size(), 0);
    p.swap_heavy = cfg.swap_heavy;
    if (!cfg.swap_heavy) return; // (mfx_options.no_swap)
    const long long thr = (long long)hot_len * p.ns;
    p.heavy_thr = thr;
    for (size_t i = 0; i < og.size(); ++i) p.hot_gat[i] = (long long)og[i] > thr ? 1 : 0;
}

void finish_plan(std::vector<std::vector<Visit>> &block_visits, const PlanConfig &cfg, Plan &p,
                 PlaceVec &places, int threads)
{
    const int NS = p.ns, NB = NS * NS, G = p.groups;
    int target, hot_len;
    plan_sizes(p.nnz, NB, G, cfg, target, hot_len);
    const int W = p.waves_per_wg, waves = std::max(1, cfg.waves_per_stripe);
    const int wgs_hw = std::max(cfg.wgs_hw, waves / std::max(1, W)); // workgroups per XCD the launch really has
    const bool timing = env_int_raw("MFX_PLAN_TIMING", 0) != 0;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "mfx plan:   %-26s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // Tasks per wave.  The ordinary rows of a block are cut into (tasks per wave) x (waves left to them) tasks of equal load
    // (longest-first packing), handed out through the block's cursor.  One task per wave -- every hand-over costs a
    // chain of misses, and longer lists keep the order closer to the reference's walk through a block (final RMSE
    // 0.4 .. 0.6 points closer to the oracle on configs[1] and configs[2] at no cost in time,
    // profiles/experiments/r02_tasks_per_wave.log) -- except for wide rows (k_a >= 128: two ratings per wave step) with a
    // long share per wave, where two tasks let the waves that run ahead take up the slack (3 .. 15 % of the epoch time;
    // round 1: profiles/experiments/r01_task_sweep*.log).  An explicit task_steps (tests) selects tasks of that size.
    const long long per_wave = p.nnz / ((long long)NB * G * waves);
    int tasks_per_wave = cfg.task_steps > 0 ? 0 : knob_int("MFX_ONE_TASK", (per_wave < 128 || G >= 4) ? 1 : 2);
    if (tasks_per_wave > 1 && cfg.wgs_hw * std::max(1, cfg.waves_per_wg) > std::max(1, cfg.waves_per_stripe))
        tasks_per_wave = 1; // a wide launch: the number of wave tasks IS the concurrency cap (plan.cpp block_shape)
    // combine slots: one per row that is split over several workgroups in some block (row -> slot, -1 = none), both sides
    const int n_own = p.owner_is_q ? p.n : p.m, n_gat = p.owner_is_q ? p.m : p.n;
    std::vector<int> slot_own((size_t)n_own, -1), slot_gat((size_t)n_gat, -1);
    p.n_hot_slots = 0;
    p.hot_rows.clear();
    {
        std::vector<std::vector<uint32_t>> cut(NB); // row | side << 31
        std::atomic<int> nb(0);
        auto scan = [&]() {
            for (;;) {
                const int b = nb.fetch_add(1);
                if (b >= NB) break;
                const BlockShape sh = block_shape(block_visits[b], G, W, waves, wgs_hw, hot_len);
                for (const Visit &v : block_visits[b])
                    if ((long long)v.len > hot_len && visit_copies(v.len, (long long)W * G, sh.Th[v.swapped ? 1 : 0]) > 1)
                        cut[b].push_back(v.own | (v.swapped ? 0x80000000u : 0u));
            }
        };
        std::vector<std::thread> pool;
        const int nt = std::max(1, std::min(threads, NB));
        for (int t = 0; t < nt; ++t) pool.emplace_back(scan);
        for (auto &th : pool) th.join();
        for (int b = 0; b < NB; ++b)
            for (uint32_t key : cut[b]) {
                int &sl = (key >> 31) ? slot_gat[key & 0x7FFFFFFFu] : slot_own[key & 0x7FFFFFFFu];
                if (sl < 0) {
                    sl = (int)p.n_hot_slots++;
                    p.hot_rows.push_back((int)key);
                }
            }
    }
    lap("combine slots");
    std::vector<BlockPack> packs(NB);
    {
        std::vector<int> blocks(NB);
        for (int b = 0; b < NB; ++b) blocks[b] = b;
        std::sort(blocks.begin(), blocks.end(), [&](int a, int b) {
            return block_visits[a].size() > block_visits[b].size();
        });
        std::atomic<int> next(0);
        std::exception_ptr first_error; // an exception must not leave a worker thread (std::terminate): rethrown after the join
        std::mutex err_mu;
        auto work = [&]() {
            for (;;) {
                int idx = next.fetch_add(1);
                if (idx >= NB) break;
                try {
                    pack_block(block_visits[blocks[idx]], G, W, waves, wgs_hw, target, hot_len, tasks_per_wave, slot_own, slot_gat,
                               packs[blocks[idx]]);
                } catch (...) {
                    std::lock_guard<std::mutex> lock(err_mu);
                    if (!first_error) first_error = std::current_exception();
                }
            }
        };
        std::vector<std::thread> pool;
        int nt = std::max(1, std::min(threads, NB));
        for (int t = 0; t < nt; ++t) pool.emplace_back(work);
        for (auto &th : pool) th.join();
        if (first_error) std::rethrow_exception(first_error);
    }
    lap("pack blocks (threads)");
    // concatenate in (round, slot) order: round r, slot s -> owner stripe s,
    // gathered stripe (s + r) mod NS, so the NS slots of a round are stripe-disjoint
    size_t tot_t = 0, tot_p = 0, tot_w = 0, tot_v = 0;
    for (auto &o : packs) {
        tot_t += o.tasks.size();
        tot_p += o.places.size();
        tot_w += o.wg_tasks.size();
        tot_v += o.wg_visits.size();
        p.n_hot_rows += o.hot;
        p.n_padding += o.padding;
    }
    p.round_hot.assign((size_t)NS, 0);
    p.slot_task_ptr.assign((size_t)NB + 1, 0);
    p.slot_wg_ptr.assign((size_t)NB + 1, 0);
    // bases of every (round, slot) piece, then the pieces are copied side by side
    std::vector<uint64_t> e_base((size_t)NB + 1, 0);
    std::vector<size_t> t_base((size_t)NB + 1, 0), p_base((size_t)NB + 1, 0), w_base((size_t)NB + 1, 0), v_base((size_t)NB + 1, 0);
    for (int r = 0; r < NS; ++r)
        for (int s = 0; s < NS; ++s) {
            const BlockPack &o = packs[s * NS + (s + r) % NS];
            const size_t i = (size_t)r * NS + s;
            e_base[i + 1] = e_base[i] + o.n_entries;
            t_base[i + 1] = t_base[i] + o.tasks.size();
            p_base[i + 1] = p_base[i] + o.places.size();
            w_base[i + 1] = w_base[i] + o.wg_tasks.size();
            v_base[i + 1] = v_base[i] + o.wg_visits.size();
            if (o.folds) p.round_hot[r] = 1;
            p.slot_task_ptr[i + 1] = (long long)t_base[i + 1];
            p.slot_wg_ptr[i + 1] = (long long)w_base[i + 1];
        }
    p.tasks.resize(tot_t);
    p.wg_tasks.resize(tot_w);
    p.wg_visits.resize(tot_v);
    places.resize(tot_p);
    {
        std::atomic<int> nb(0);
        auto copy = [&]() {
            for (;;) {
                const int i = nb.fetch_add(1);
                if (i >= NB) break;
                const int r = i / NS, s = i % NS;
                BlockPack &o = packs[s * NS + (s + r) % NS];
                const uint64_t eb = e_base[i];
                TaskDesc *td = p.tasks.data() + t_base[i];
                for (size_t j = 0; j < o.tasks.size(); ++j) {
                    td[j] = o.tasks[j];
                    td[j].off += eb;
                }
                WgTask *wt = p.wg_tasks.data() + w_base[i];
                for (size_t j = 0; j < o.wg_tasks.size(); ++j) {
                    wt[j] = o.wg_tasks[j];
                    wt[j].off += eb;
                    wt[j].visit0 += (uint32_t)v_base[i];
                }
                std::copy(o.wg_visits.begin(), o.wg_visits.end(), p.wg_visits.begin() + (long long)v_base[i]);
                Placement *pl = places.data() + p_base[i];
                for (size_t j = 0; j < o.places.size(); ++j) {
                    pl[j] = o.places[j];
                    pl[j].dst += eb;
                }
                std::vector<TaskDesc>().swap(o.tasks);
                std::vector<Placement>().swap(o.places);
                std::vector<WgTask>().swap(o.wg_tasks);
                std::vector<WgVisitRec>().swap(o.wg_visits);
            }
        };
        std::vector<std::thread> pool;
        const int nt = std::max(1, std::min(threads, NB));
        for (int t = 0; t < nt; ++t) pool.emplace_back(copy);
        for (auto &th : pool) th.join();
    }
    const uint64_t ebase = e_base[NB];
    p.n_entries = (long long)ebase;
    // With roles per task a row can be read-modified-written by a visit of the other role while a wave task holds it in
    // registers; the visit then has to write back what memory holds at its end plus what it changed (one more read of the
    // row per visit: +21 % traffic on configs[2]).  How often that matters is the share of the launch a row spends in
    // registers: the mean visit length over the steps a wave does.  20 k x 400 k: 80 % (+4.0 % final RMSE without the merge);
    // configs[2]: 3 % -- there the plain store stays (what is lost is 3 % of the steps of the swapped ratings).
    {
        long long nv = 0, lv = 0;
        for (const TaskDesc &t : p.tasks) lv += (long long)t.nsteps * G;
        for (int b = 0; b < NB; ++b) nv += (long long)block_visits[b].size();
        const double mean_visit = nv > 0 ? (double)p.nnz / (double)nv : 0.0;
        // (steps of a wave in a launch: its tasks one after the other)
        const double steps = p.tasks.empty() ? 1.0 : (double)lv / (double)G / (double)p.tasks.size() * (double)std::max(1, tasks_per_wave);
        p.merge_back = p.swap_heavy && mean_visit > 0.05 * steps;
    }
    lap("concatenate");
}

void plan_header(long long nnz, int m, int n, const PlanConfig &cfg, Plan &p)
{
    if (nnz <= 0 || m <= 0 || n <= 0) throw std::invalid_argument("empty problem");
    if (cfg.k < 1) throw std::invalid_argument("number of factors must be greater than zero");
    p.m = m;
    p.n = n;
    p.k = cfg.k;
    p.ka = k_aligned(cfg.k);
    p.nnz = nnz;
    p.ns = std::max(1, cfg.stripes);
    p.lanes = cfg.lanes;
    p.groups = 64 / cfg.lanes;
    p.owner_is_q = cfg.owner_side == 0 ? (m >= n) : cfg.owner_side == 2;
    if ((long long)p.ns * p.ns > 65535) throw std::invalid_argument("too many stripes");
}

void plan_scale(Plan &p)
{
    p.scale = std::max((float)1e-4, p.std_dev); // reference mf/mf.cpp:2999
    p.inv_scale = (float)1.0 / p.scale;         // reference mf/mf.cpp:3010
}

namespace {

void equal_ranges(int size, int ns, std::vector<int> &begin)
{
    const int seg = (size + ns - 1) / ns; // reference seg_p / seg_q, mf.cpp:802-803
    begin.resize(ns + 1);
    for (int s = 0; s <= ns; ++s) begin[s] = (int)std::min<long long>((long long)s * seg, size);
}

// Stripes of (nearly) equal rating mass instead of equal row count.  The reference shuffles the
// ids (gen_random_map) and cuts equal ranges, which balances its grid only statistically; one
// row that holds a few per cent of all ratings (a Zipf head) then makes its stripe, and every
// block of it, that much heavier -- and a round takes as long as its heaviest block.  Here the few
// heavy rows (more than 1/16 of a stripe's share) are dealt out longest-first, and the others,
// taken in the reference's shuffled order, are cut where the cumulative mass reaches each
// stripe's share.
void balanced_map(int size, int ns, const int *cnt, std::vector<int> &map, std::vector<int> &begin,
                  std::vector<int> &at)
{
    std::vector<int> shuffled;
    gen_random_map(size, shuffled); // original id -> position in the reference's order
    std::vector<int> order(size);   // position -> original id
    for (int i = 0; i < size; ++i) order[shuffled[i]] = i;
    long long total = 0;
    for (int i = 0; i < size; ++i) total += cnt[i];
    const long long heavy_min = std::max<long long>(2, total / ((long long)ns * 16));
    std::vector<int> heavy;
    for (int i = 0; i < size; ++i)
        if (cnt[i] >= heavy_min) heavy.push_back(i);
    std::stable_sort(heavy.begin(), heavy.end(), [&](int a, int b) { return cnt[a] > cnt[b]; });
    std::vector<long long> load(ns, 0);
    std::vector<std::vector<int>> heavy_of(ns);
    std::vector<char> is_heavy(size, 0);
    for (int i : heavy) is_heavy[i] = 1;
    for (int i : heavy) {
        int best = 0;
        for (int s = 1; s < ns; ++s)
            if (load[s] < load[best]) best = s;
        heavy_of[best].push_back(i);
        load[best] += cnt[i];
    }
    map.assign(size, -1);
    begin.assign(ns + 1, 0);
    int next_id = 0, pos = 0;
    long long light_done = 0, light_want = 0; // cumulative light mass: placed / wanted up to this stripe
    for (int s = 0; s < ns; ++s) {
        begin[s] = next_id;
        for (int i : heavy_of[s]) map[i] = next_id++;
        light_want += (total * (s + 1)) / ns - (total * s) / ns - load[s];
        while (pos < size) {
            const int i = order[pos];
            if (is_heavy[i]) { ++pos; continue; } // placed with its stripe
            if (s < ns - 1 && light_done >= light_want && cnt[i] > 0) break;
            map[i] = next_id++;
            light_done += cnt[i];
            ++pos;
        }
    }
    begin[ns] = next_id;
    at.resize(size);
    for (int i = 0; i < size; ++i) at[shuffled[i]] = map[i];
}

} // namespace

void plan_maps(const PlanConfig &cfg, Plan &p, const int *cnt_p, const int *cnt_q)
{
    p.p_at.clear();
    p.q_at.clear();
    if (cfg.map_mode == 1) {
        p.p_map.resize(p.m);
        p.q_map.resize(p.n);
        for (int i = 0; i < p.m; ++i) p.p_map[i] = i;
        for (int i = 0; i < p.n; ++i) p.q_map[i] = i;
        equal_ranges(p.m, p.ns, p.p_begin);
        equal_ranges(p.n, p.ns, p.q_begin);
    } else if (cfg.map_mode == 2) {
        std::thread tq([&] { gen_random_map(p.n, p.q_map); });
        gen_random_map(p.m, p.p_map);
        tq.join();
        equal_ranges(p.m, p.ns, p.p_begin);
        equal_ranges(p.n, p.ns, p.q_begin);
    } else {
        const int *lp = cfg.layout_cnt_p ? cfg.layout_cnt_p : cnt_p, *lq = cfg.layout_cnt_q ? cfg.layout_cnt_q : cnt_q;
        std::thread tq([&] { balanced_map(p.n, p.ns, lq, p.q_map, p.q_begin, p.q_at); });
        balanced_map(p.m, p.ns, lp, p.p_map, p.p_begin, p.p_at);
        tq.join();
    }
    p.omega_p.assign(p.m, 0); // omega, mf.cpp:815-816
    p.omega_q.assign(p.n, 0);
    for (int i = 0; i < p.m; ++i) p.omega_p[p.p_map[i]] = cnt_p[i];
    for (int i = 0; i < p.n; ++i) p.omega_q[p.q_map[i]] = cnt_q[i];
}

void build_plan(const Node *R, long long nnz, int m, int n, const PlanConfig &cfg, Plan &p)
{
    plan_header(nnz, m, n, cfg, p);
    int threads = cfg.threads > 0 ? cfg.threads : (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;

    if (cfg.use_stats) {
        p.avg = cfg.stats_avg;
        p.std_dev = cfg.stats_std;
    } else {
        collect_info(R, nnz, threads, p.avg, p.std_dev);
    }
    plan_scale(p);

    // validate ids and count the ratings of every row (omega, mf.cpp:815-816)
    const int NS = p.ns;
    const int NB = NS * NS;
    std::vector<int> cnt_p(m, 0), cnt_q(n, 0);
    for (long long i = 0; i < nnz; ++i) {
        if (R[i].u < 0 || R[i].u >= m || R[i].v < 0 || R[i].v >= n)
            throw std::invalid_argument("rating with id outside [0,m) x [0,n)");
        cnt_p[R[i].u]++;
        cnt_q[R[i].v]++;
    }
    plan_maps(cfg, p, cnt_p.data(), cnt_q.data());
    plan_hot_gathered(cfg, p);

    // relabel (shuffle_problem, mf.cpp:775-791), scale (mf.cpp:517-527) and bucket by (block, role): the ratings of a
    // heavy gathered row are keyed by that row first (plan_hot_gathered)
    const int *own_begin = p.owner_is_q ? p.q_begin.data() : p.p_begin.data();
    const int *gat_begin = p.owner_is_q ? p.p_begin.data() : p.q_begin.data();
    const int *omega_own = p.owner_is_q ? p.omega_q.data() : p.omega_p.data();
    const int *omega_gat = p.owner_is_q ? p.omega_p.data() : p.omega_q.data();
    std::vector<Rat> rat(nnz);
    std::vector<uint32_t> blk(nnz); // block * 2 + swapped
    const bool do_scale = p.inv_scale != 1.0f;
    parallel_ranges(nnz, threads, [&](long long b, long long e, int) {
        for (long long i = b; i < e; ++i) {
            uint32_t u = (uint32_t)p.p_map[R[i].u], v = (uint32_t)p.q_map[R[i].v];
            const uint32_t own = p.owner_is_q ? v : u, gat = p.owner_is_q ? u : v;
            const bool sw = p.swap_heavy && omega_gat[gat] > omega_own[own];
            Rat x;
            x.own = sw ? gat : own;
            x.gat = sw ? own : gat;
            x.r = do_scale ? R[i].r * p.inv_scale : R[i].r;
            rat[i] = x;
            blk[i] = (uint32_t)(stripe_of(own_begin, NS, own) * NS + stripe_of(gat_begin, NS, gat)) * 2u + (sw ? 1u : 0u);
        }
    });
    std::vector<long long> bptr(2 * NB + 1, 0);
    for (long long i = 0; i < nnz; ++i) bptr[blk[i] + 1]++;
    for (int b = 0; b < 2 * NB; ++b) bptr[b + 1] += bptr[b];
    std::vector<Rat> sorted(nnz);
    {
        std::vector<long long> cur(bptr.begin(), bptr.end() - 1);
        for (long long i = 0; i < nnz; ++i) sorted[cur[blk[i]]++] = rat[i];
    }
    std::vector<Rat>().swap(rat);
    std::vector<uint32_t>().swap(blk);

    // per-bucket sort by (primary, secondary) -- stable, so equal pairs keep their input order, the
    // same order the device path's radix sort leaves them in -- and the visit table
    std::vector<std::vector<Visit>> block_visits(NB);
    {
        std::atomic<int> next(0);
        auto work = [&]() {
            for (;;) {
                int b = next.fetch_add(1);
                if (b >= NB) break;
                std::vector<Visit> &vs = block_visits[b];
                for (int sw = 0; sw < 2; ++sw) {
                    Rat *beg = sorted.data() + bptr[2 * b + sw];
                    const long long L = bptr[2 * b + sw + 1] - bptr[2 * b + sw];
                    std::stable_sort(beg, beg + L, [](const Rat &x, const Rat &y) {
                        return x.own != y.own ? x.own < y.own : x.gat < y.gat;
                    });
                    for (long long i = 0; i < L;) {
                        long long j = i;
                        while (j < L && beg[j].own == beg[i].own) ++j;
                        vs.push_back({beg[i].own, (uint32_t)(j - i), (uint64_t)(bptr[2 * b + sw] + i), (uint32_t)sw});
                        i = j;
                    }
                }
            }
        };
        std::vector<std::thread> pool;
        int nt = std::min(threads, NB);
        for (int t = 0; t < nt; ++t) pool.emplace_back(work);
        for (auto &th : pool) th.join();
    }

    PlaceVec places;
    finish_plan(block_visits, cfg, p, places, threads);

    // write the entries: a placement puts `len` sorted ratings (stride apart) into one lane-group list (stride G)
    p.entries.assign((size_t)p.n_entries, Entry{0u, -1, 0.0f});
    const int G = p.groups;
    parallel_ranges((long long)places.size(), threads, [&](long long b, long long e, int) {
        for (long long i = b; i < e; ++i) {
            const Placement &pl = places[i];
            const uint32_t stride = pl.stride_flags & 0xFFFFu;
            const uint32_t flags = (pl.stride_flags >> 31) ? ENTRY_SWAPPED : 0u;
            const bool visit_start = ((pl.stride_flags >> 30) & 1u) != 0;
            // the (block, role) bucket of the sorted array this placement reads from
            const size_t bk = (size_t)(std::upper_bound(bptr.begin(), bptr.end(), (long long)pl.src) - bptr.begin()) - 1;
            const uint64_t bk_lo = (uint64_t)bptr[bk], bk_hi = (uint64_t)bptr[bk + 1];
            for (uint32_t x = 0; x < pl.len; ++x) {
                const uint64_t si = pl.src + (uint64_t)x * stride;
                const Rat &rr = sorted[si];
                Entry &en = p.entries[pl.dst + (uint64_t)x * G];
                en.own = rr.own | flags | ((visit_start && x == 0) ? 0x80000000u : 0u);
                en.gat = (int32_t)rr.gat;
                // A long run of ONE pair of two heavy rows.  The lighter row of the pair lives in the LDS of its own workgroup
                // visits during the launch; a run that read-modify-writes it in memory as well is a second stream that
                // converges by itself, and the two changes are added: the row overshoots (eta = 0.2 overflowed that way).
                // Inside such a run the rating moves the heavier row only (ENTRY_READ_ONLY).  (Both neighbours belong to the
                // same block and role: a pair does not repeat across them.)
                if (p.swap_heavy && (flags ? (long long)omega_own[rr.gat] > p.heavy_thr : p.hot_gat[rr.gat] != 0) &&
                    si >= bk_lo + RUN_READ_ONLY / 2 && si + RUN_READ_ONLY / 2 < bk_hi) {
                    const Rat &lo = sorted[si - RUN_READ_ONLY / 2], &hi = sorted[si + RUN_READ_ONLY / 2];
                    if (lo.own == rr.own && lo.gat == rr.gat && hi.own == rr.own && hi.gat == rr.gat) en.gat |= ENTRY_READ_ONLY;
                }
                en.r = rr.r;
            }
        }
    });
}

void init_factors(const Plan &p, const int *omega_p_override, const int *omega_q_override,
                  std::vector<float> &P, std::vector<float> &Q, int threads)
{
    // One minstd_rand0 stream, P rows then Q rows in the REFERENCE's row order (plan.p_at/q_at),
    // k draws per seen row scaled by sqrt(1/k); unseen rows NaN; padding zero (mf.cpp:952-1007).
    const int k = p.k, ka = p.ka;
    const float s = (float)std::sqrt(1.0 / k);
    P.assign((size_t)p.m * ka, 0.0f);
    Q.assign((size_t)p.n * ka, 0.0f);
    const int *op = omega_p_override ? omega_p_override : p.omega_p.data();
    const int *oq = omega_q_override ? omega_q_override : p.omega_q.data();
    auto row_at = [&](long long i) -> long long { // position in the reference's order -> internal row
        if (i < p.m) return p.p_at.empty() ? i : p.p_at[i];
        return p.q_at.empty() ? i - p.m : p.q_at[i - p.m];
    };
    // stream position of every row = k * (seen rows before it)
    const long long rows = (long long)p.m + p.n;
    std::vector<uint64_t> pos((size_t)rows);
    uint64_t seen = 0;
    for (long long i = 0; i < rows; ++i) {
        pos[i] = seen;
        seen += (i < p.m ? op[row_at(i)] : oq[row_at(i)]) > 0;
    }
    parallel_ranges(rows, threads, [&](long long b, long long e, int) {
        Minstd0 gen(Minstd0::jump(1u, pos[b] * (uint64_t)k));
        for (long long i = b; i < e; ++i) {
            const bool isP = i < p.m;
            const long long row = row_at(i);
            float *dst = (isP ? P.data() : Q.data()) + row * ka;
            const bool seen_row = (isP ? op[row] : oq[row]) > 0;
            if (seen_row)
                for (int d = 0; d < k; ++d) dst[d] = (float)(gen.unit() * s);
            else
                for (int d = 0; d < k; ++d) dst[d] = std::numeric_limits<float>::quiet_NaN();
        }
    });
}

} // namespace mfx
