// rng.hpp -- private generators of the host pipeline.
//
// The reference seeds its id permutation with the process-global srand(0)/rand()
// and draws the initial factors from std::default_random_engine
// (reference mf/mf.cpp:1009-1017, 971-995).  To hand back the same permutation
// and the same initial model without touching process-global state (the facade
// is called on a PHP request thread, SURVEY.md 8b "Threading"), the published
// algorithms behind those library calls are implemented here as plain structs.
#pragma once
#include <cmath>
#include <cstdint>

namespace mfx {

// glibc random(), TYPE_3: additive feedback r[i] = r[i-3] + r[i-31] over 31 words,
// seeded by the Lehmer sequence 16807*x mod (2^31-1), first 310 outputs discarded.
struct GlibcRand {
    int32_t w[31];
    int front, back;
    explicit GlibcRand(unsigned seed)
    {
        if (seed == 0) seed = 1;
        int64_t x = (int32_t)seed;
        w[0] = (int32_t)x;
        for (int i = 1; i < 31; ++i) {
            int64_t hi = x / 127773, lo = x % 127773;
            x = 16807 * lo - 2836 * hi;
            if (x < 0) x += 2147483647;
            w[i] = (int32_t)x;
        }
        front = 3;
        back = 0;
        for (int i = 0; i < 310; ++i) (void)next();
    }
    int next()
    {
        uint32_t s = (uint32_t)w[front] + (uint32_t)w[back];
        w[front] = (int32_t)s;
        if (++front == 31) front = 0;
        if (++back == 31) back = 0;
        return (int)(s >> 1);
    }
};

// std::minstd_rand0 and the float in [0,1) libstdc++ derives from one draw of it
// (generate_canonical<float,24>: (x-1)/2^31 in float, kept below 1).
struct Minstd0 {
    uint32_t s;
    explicit Minstd0(uint32_t seed = 1) : s(seed) {}
    uint32_t next()
    {
        s = (uint32_t)(((uint64_t)s * 16807u) % 2147483647u);
        return s;
    }
    float unit()
    {
        float f = (float)(next() - 1u) / 2147483648.0f;
        return f < 1.0f ? f : std::nextafter(1.0f, 0.0f);
    }
    // state after `steps` further draws: s * 16807^steps mod (2^31-1)
    static uint32_t jump(uint32_t s, uint64_t steps)
    {
        uint64_t base = 16807, acc = 1;
        const uint64_t M = 2147483647ull;
        while (steps) {
            if (steps & 1) acc = acc * base % M;
            base = base * base % M;
            steps >>= 1;
        }
        return (uint32_t)((uint64_t)s * acc % M);
    }
};

} // namespace mfx
