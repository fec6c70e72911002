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
#include <queue>
#include <stdexcept>
#include <thread>

#include "rng.hpp"

namespace mfx {

int lanes_for(int ka)
{
    int need = (ka + 3) / 4, l = 2;
    while (l < need) l <<= 1;
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

struct Rat { uint32_t own; uint32_t gat; float r; };

// Pack one class of visits into tasks whose lane-group lists hold about `target` ratings:
// longest-processing-time-first into G*ntasks lists, lists of similar load share a task.
void pack_class(const std::vector<Visit> &visits, size_t vbeg, size_t vend, int G, int target,
                BlockPack &out, long long force_ntasks = 0)
{
    if (vbeg >= vend) return;
    static const int list_order = getenv("MFX_LIST_ORDER") ? atoi(getenv("MFX_LIST_ORDER")) : 1;
    // load of a visit in steps: its ratings, plus the header entry of a hot chain
    auto steps_of = [](const Visit &v) { return v.len + (v.nch ? 1u : 0u); };
    long long L = 0;
    for (size_t i = vbeg; i < vend; ++i) L += steps_of(visits[i]);
    target = std::max<long long>(target, steps_of(visits[vbeg])); // visits are sorted, longest first
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
        const uint64_t key = top + ((uint64_t)steps_of(visits[vi]) << 32);
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
            // order emulation.  MFX_LIST_ORDER: 0 as packed, 1 by owner row id (default: the reference's own
            // order inside a block; +0.9 % epoch time on configs[2]), 2 scattered by a hash (same fit, +3.3 % time).
            lv.clear();
            lv.reserve(count[lst]);
            for (uint32_t x = head[lst]; x != NIL; x = next[x]) lv.push_back((uint32_t)(x + vbeg));
            if (list_order == 1)
                std::sort(lv.begin(), lv.end(), [&](uint32_t a, uint32_t b) {
                    return visits[a].own != visits[b].own ? visits[a].own < visits[b].own : visits[a].idx < visits[b].idx;
                });
            else if (list_order == 2)
                std::sort(lv.begin(), lv.end(), [&](uint32_t a, uint32_t b) {
                    const uint32_t ha = (visits[a].own * 2654435761u) ^ (visits[a].idx * 40503u), hb = (visits[b].own * 2654435761u) ^ (visits[b].idx * 40503u);
                    return ha != hb ? ha < hb : a < b;
                });
            for (uint32_t vi : lv) {
                const Visit &v = visits[vi];
                if (v.nch) { // hot chain: a header entry first
                    out.headers.push_back({base + (uint64_t)step * G + g, v.own, v.nch, v.hot | (v.len << 20), v.idx});
                    ++step;
                }
                out.places.push_back({v.start, base + (uint64_t)step * G + g, v.len, 0u});
                step += v.len;
            }
            out.padding += nsteps - step;
        }
        out.tasks.push_back(td);
    }
}

} // namespace

// Cut one (owner-stripe, gather-stripe) block, given as its visits, into wavefront tasks.
void pack_visits(std::vector<Visit> &raw, int G, int target, int hot_len, BlockPack &out, int one_task_waves,
                 const std::vector<int> *hot_slot_of_row)
{
    if (raw.empty()) return;
    // A visit (all ratings of one owner row in this block) longer than hot_len is cut into
    // chains that may run in different lane groups, each on its own register copy of the owner row;
    // when the last chain of the launch ends, the chains' changes are folded into the row (kernels.hip,
    // "hot chains"; DESIGN.md "Hot rows").  No list is longer than the longest visit, so hot_len also
    // bounds the longest task of the launch.
    std::vector<Visit> visits;
    visits.reserve(raw.size());
    long long L = 0;
    for (const Visit &v : raw) {
        L += v.len;
        if ((long long)v.len > hot_len) {
            long long nch = ((long long)v.len + hot_len - 1) / hot_len;
            if (nch > 32767) nch = 32767; // the header entry counts chains in 15 bits: a monster row gets longer chains instead
            long long per = ((long long)v.len + nch - 1) / nch;
            long long made = 0;
            for (long long s = 0; s < v.len; s += per) ++made;
            const uint32_t slot = hot_slot_of_row ? (uint32_t)(*hot_slot_of_row)[v.own] : 0u;
            if (made >= (1 << 15) || per >= (1 << 12))
                throw std::invalid_argument("one row holds more than 134 M ratings of a block: more stripes (mfx_options.stripes)");
            for (long long s = 0; s < v.len; s += per)
                visits.push_back({v.own, (uint32_t)std::min<long long>(per, v.len - s), v.start + (uint64_t)s,
                                  hot_slot_of_row ? (uint32_t)made : 0u, slot, (uint32_t)(s / per)});
            out.hot++;
        } else {
            visits.push_back(v);
        }
    }
    // longest first, equal lengths in their given order: a counting sort (lengths are at most hot_len unless a monster
    // row got longer chains), the comparison sort otherwise
    uint32_t max_len = 0;
    for (const Visit &v : visits) max_len = std::max(max_len, v.len);
    if (max_len <= 65536u) {
        std::vector<size_t> at((size_t)max_len + 2, 0);
        for (const Visit &v : visits) at[(size_t)(max_len - v.len) + 1]++;
        for (size_t i = 1; i < at.size(); ++i) at[i] += at[i - 1];
        std::vector<Visit> sorted(visits.size());
        for (const Visit &v : visits) sorted[at[(size_t)(max_len - v.len)]++] = v;
        visits.swap(sorted);
    } else {
        std::stable_sort(visits.begin(), visits.end(),
                         [](const Visit &a, const Visit &b) { return a.len > b.len; });
    }

    // Graded task sizes: the first half of the work goes into full-size tasks, then a
    // quarter at half size, ... so the waves that drain the block's queue last are holding
    // short tasks (the launch ends when the slowest wave does).
    if (one_task_waves > 0) { // equal-load tasks, (tasks per wave) x (waves of the block's XCD) of them
        pack_class(visits, 0, visits.size(), G, 8, out, one_task_waves);
        return;
    }
    const double frac[4] = {0.5, 0.75, 0.875, 1.0};
    const char *ge = getenv("MFX_GRADES"); // experiment knob: number of size classes (1..4)
    const int grades = ge && *ge ? std::max(1, std::min(4, atoi(ge))) : 4;
    size_t vbeg = 0;
    long long acc = 0;
    for (int c = 0; c < grades; ++c) {
        size_t vend = vbeg;
        if (c == grades - 1) {
            vend = visits.size();
        } else {
            while (vend < visits.size() && acc < (long long)(frac[c] * (double)L)) acc += visits[vend++].len;
        }
        pack_class(visits, vbeg, vend, G, std::max(8, target >> c), out);
        vbeg = vend;
    }
}

void plan_sizes(long long nnz, int NB, int G, const PlanConfig &cfg, int &target, int &hot_len)
{
    // Full-size tasks hold about half of what one wave does in a launch (same T for every
    // block: a block that is heavier than average must not get longer tasks, its XCD would
    // finish the round last); the graded tail in pack_visits keeps the end short.
    const long long per_wave = nnz / ((long long)NB * G * std::max(1, cfg.waves_per_stripe));
    target = cfg.task_steps;
    if (target <= 0) target = (int)std::min<long long>(64, std::max<long long>(8, per_wave / 2));
    // A hot owner row is cut into chains of at most hot_len ratings.  Longer chains keep more of
    // the row's updates (measured: 128 vs 64 is worth 2-3 % RMSE on small problems,
    // profiles/experiments/r01_hot_chain_length.log), but a list must not outlast what one wave
    // does in the launch, or it sets the launch time: hot_len = per-wave load, within [48,128] (48: below it the
    // ordinary rows of small blocks get cut too -- one rank of N=8: RMSE 0.730 at 38, 0.698 at 48).
    const char *he = getenv("MFX_HOT_LEN"); // experiment knob
    hot_len = he && *he ? std::max(8, atoi(he))
                        : (int)std::min<long long>(128, std::max<long long>(48, per_wave));
}

void finish_plan(std::vector<std::vector<Visit>> &block_visits, const PlanConfig &cfg, Plan &p,
                 PlaceVec &places, int threads)
{
    const int NS = p.ns, NB = NS * NS, G = p.groups;
    int target, hot_len;
    plan_sizes(p.nnz, NB, G, cfg, target, hot_len);
    const bool timing = getenv("MFX_PLAN_TIMING") && atoi(getenv("MFX_PLAN_TIMING")) != 0;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "mfx plan:   %-26s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // Tasks per wave.  A block is cut into (tasks per wave) x (waves of its XCD) tasks of equal load
    // (longest-first packing), handed out through the block's cursor.  One task per wave -- every hand-over costs a
    // chain of misses, and longer lists keep the order closer to the reference's walk through a block (final RMSE
    // 0.4 .. 0.6 points closer to the oracle on configs[1] and configs[2] at no cost in time,
    // profiles/experiments/r02_tasks_per_wave.log) -- except for wide rows (k_a >= 128: two ratings per wave step) with a
    // long share per wave, where two tasks let the waves that run ahead take up the slack (3 .. 15 % of the epoch time;
    // round 1: profiles/experiments/r01_task_sweep*.log).
    // MFX_ONE_TASK=0 selects the older graded sizes (T, T/2, T/4, T/8 from task_steps).
    const long long per_wave = p.nnz / ((long long)NB * G * std::max(1, cfg.waves_per_stripe));
    const char *ot = getenv("MFX_ONE_TASK");
    const int tasks_per_wave = ot && *ot ? atoi(ot) : cfg.task_steps > 0 ? 0 : (per_wave < 128 || G >= 4) ? 1 : 2;
    const int one_task = tasks_per_wave * std::max(1, cfg.waves_per_stripe);
    // combine slots: one per owner row that is cut into chains in some block (row -> slot, -1 = none)
    std::vector<int> hot_slot((size_t)(p.owner_is_q ? p.n : p.m), -1);
    p.n_hot_slots = 0;
    p.hot_rows.clear();
    {
        // rows cut in each block (found in parallel), numbered in block order, then visit order
        std::vector<std::vector<uint32_t>> cut(NB);
        std::atomic<int> nb(0);
        auto scan = [&]() {
            for (;;) {
                const int b = nb.fetch_add(1);
                if (b >= NB) break;
                for (const Visit &v : block_visits[b])
                    if ((long long)v.len > hot_len) cut[b].push_back(v.own);
            }
        };
        std::vector<std::thread> pool;
        const int nt = std::max(1, std::min(threads, NB));
        for (int t = 0; t < nt; ++t) pool.emplace_back(scan);
        for (auto &th : pool) th.join();
        for (int b = 0; b < NB; ++b)
            for (uint32_t own : cut[b])
                if (hot_slot[own] < 0) {
                    hot_slot[own] = (int)p.n_hot_slots++;
                    p.hot_rows.push_back((int)own);
                }
    }
    if (p.n_hot_slots >= (1 << 20) || hot_len >= (1 << 12))
        throw std::invalid_argument("too many hot rows / too long chains for the header entry format");
    // experiment knob: round 1's behaviour (chains overwrite the row, the last writer wins: no headers, no fold)
    const char *lww = getenv("MFX_HOT_LWW");
    const bool hot_lww = lww && *lww && atoi(lww) != 0;
    lap("hot slots");
    std::vector<BlockPack> packs(NB);
    {
        std::vector<int> blocks(NB);
        for (int b = 0; b < NB; ++b) blocks[b] = b;
        std::sort(blocks.begin(), blocks.end(), [&](int a, int b) {
            return block_visits[a].size() > block_visits[b].size();
        });
        std::atomic<int> next(0);
        auto work = [&]() {
            for (;;) {
                int idx = next.fetch_add(1);
                if (idx >= NB) break;
                pack_visits(block_visits[blocks[idx]], G, target, hot_len, packs[blocks[idx]], one_task,
                            hot_lww ? nullptr : &hot_slot);
            }
        };
        std::vector<std::thread> pool;
        int nt = std::max(1, std::min(threads, NB));
        for (int t = 0; t < nt; ++t) pool.emplace_back(work);
        for (auto &th : pool) th.join();
    }
    lap("pack blocks (threads)");
    // concatenate in (round, slot) order: round r, slot s -> owner stripe s,
    // gathered stripe (s + r) mod NS, so the NS slots of a round are stripe-disjoint
    size_t tot_t = 0, tot_p = 0;
    for (auto &o : packs) {
        tot_t += o.tasks.size();
        tot_p += o.places.size();
        p.n_hot_rows += o.hot;
        p.n_padding += o.padding;
    }
    p.round_hot.assign((size_t)NS, 0);
    p.slot_task_ptr.assign((size_t)NB + 1, 0);
    // bases of every (round, slot) piece, then the pieces are copied side by side
    std::vector<uint64_t> e_base((size_t)NB + 1, 0);
    std::vector<size_t> t_base((size_t)NB + 1, 0), p_base((size_t)NB + 1, 0), h_base((size_t)NB + 1, 0);
    for (int r = 0; r < NS; ++r)
        for (int s = 0; s < NS; ++s) {
            const BlockPack &o = packs[s * NS + (s + r) % NS];
            const size_t i = (size_t)r * NS + s;
            e_base[i + 1] = e_base[i] + o.n_entries;
            t_base[i + 1] = t_base[i] + o.tasks.size();
            p_base[i + 1] = p_base[i] + o.places.size();
            h_base[i + 1] = h_base[i] + o.headers.size();
            if (!o.headers.empty()) p.round_hot[r] = 1;
            p.slot_task_ptr[i + 1] = (long long)t_base[i + 1];
        }
    p.tasks.resize(tot_t);
    places.resize(tot_p);
    p.headers.resize(h_base[NB]);
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
                Placement *pl = places.data() + p_base[i];
                for (size_t j = 0; j < o.places.size(); ++j) {
                    pl[j] = o.places[j];
                    pl[j].dst += eb;
                }
                HeaderRec *hd = p.headers.data() + h_base[i];
                for (size_t j = 0; j < o.headers.size(); ++j) {
                    hd[j] = o.headers[j];
                    hd[j].dst += eb;
                }
                std::vector<TaskDesc>().swap(o.tasks);
                std::vector<Placement>().swap(o.places);
                std::vector<HeaderRec>().swap(o.headers);
            }
        };
        std::vector<std::thread> pool;
        const int nt = std::max(1, std::min(threads, NB));
        for (int t = 0; t < nt; ++t) pool.emplace_back(copy);
        for (auto &th : pool) th.join();
    }
    const uint64_t ebase = e_base[NB];
    p.n_entries = (long long)ebase;
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

    // relabel (shuffle_problem, mf.cpp:775-791), scale (mf.cpp:517-527) and bucket by block
    const int *own_begin = p.owner_is_q ? p.q_begin.data() : p.p_begin.data();
    const int *gat_begin = p.owner_is_q ? p.p_begin.data() : p.q_begin.data();
    std::vector<Rat> rat(nnz);
    std::vector<uint16_t> blk(nnz);
    const bool do_scale = p.inv_scale != 1.0f;
    parallel_ranges(nnz, threads, [&](long long b, long long e, int) {
        for (long long i = b; i < e; ++i) {
            uint32_t u = (uint32_t)p.p_map[R[i].u], v = (uint32_t)p.q_map[R[i].v];
            Rat x;
            x.own = p.owner_is_q ? v : u;
            x.gat = p.owner_is_q ? u : v;
            x.r = do_scale ? R[i].r * p.inv_scale : R[i].r;
            rat[i] = x;
            blk[i] = (uint16_t)(stripe_of(own_begin, NS, x.own) * NS + stripe_of(gat_begin, NS, x.gat));
        }
    });
    std::vector<long long> bptr(NB + 1, 0);
    for (long long i = 0; i < nnz; ++i) bptr[blk[i] + 1]++;
    for (int b = 0; b < NB; ++b) bptr[b + 1] += bptr[b];
    std::vector<Rat> sorted(nnz);
    {
        std::vector<long long> cur(bptr.begin(), bptr.end() - 1);
        for (long long i = 0; i < nnz; ++i) sorted[cur[blk[i]]++] = rat[i];
    }
    std::vector<Rat>().swap(rat);
    std::vector<uint16_t>().swap(blk);

    // per-block sort by (owner, gathered) -- stable, so equal pairs keep their input order, the
    // same order the device path's radix sort leaves them in -- and the visit table
    std::vector<std::vector<Visit>> block_visits(NB);
    {
        std::atomic<int> next(0);
        auto work = [&]() {
            for (;;) {
                int b = next.fetch_add(1);
                if (b >= NB) break;
                Rat *beg = sorted.data() + bptr[b];
                const long long L = bptr[b + 1] - bptr[b];
                std::stable_sort(beg, beg + L, [](const Rat &x, const Rat &y) {
                    return x.own != y.own ? x.own < y.own : x.gat < y.gat;
                });
                std::vector<Visit> &vs = block_visits[b];
                for (long long i = 0; i < L;) {
                    long long j = i;
                    while (j < L && beg[j].own == beg[i].own) ++j;
                    vs.push_back({beg[i].own, (uint32_t)(j - i), (uint64_t)(bptr[b] + i)});
                    i = j;
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

    // write the entries: a placement puts `len` consecutive sorted ratings into one lane-group
    // list (stride G), the first one flagged "owner row changes here"
    p.entries.assign((size_t)p.n_entries, Entry{0u, -1, 0.0f});
    const int G = p.groups;
    parallel_ranges((long long)places.size(), threads, [&](long long b, long long e, int) {
        for (long long i = b; i < e; ++i) {
            const Placement &pl = places[i];
            for (uint32_t x = 0; x < pl.len; ++x) {
                const Rat &rr = sorted[pl.src + x];
                Entry &en = p.entries[pl.dst + (uint64_t)x * G];
                en.own = rr.own | (x == 0 ? 0x80000000u : 0u);
                en.gat = (int32_t)rr.gat;
                en.r = rr.r;
            }
        }
    });
    for (const HeaderRec &h : p.headers) p.entries[h.dst] = header_entry(h);
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
