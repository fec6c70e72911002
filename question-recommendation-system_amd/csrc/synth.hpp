// synth.hpp -- deterministic synthetic rating stream (SURVEY.md 8d "Synthetic inputs").
//
// Planted rank-16 model with noise, ids half uniform / half Zipf(0.8) over an affine
// permutation, every id covered at least once.  Counter-based (splitmix64 keyed by
// seed, stream and index) and INTEGER-ONLY up to the final int->float conversion, so
// the host build and the gfx950 build emit the same bits and any shard of the stream
// can be produced independently.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define MFX_HD __host__ __device__
#else
#define MFX_HD
#endif

namespace mfx {

struct SynthNode { int u; int v; float r; };

MFX_HD inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

MFX_HD inline uint64_t synth_hash(uint64_t seed, uint64_t stream, uint64_t idx)
{
    return splitmix64(splitmix64(seed ^ (stream * 0xD1B54A32D192ED03ull)) + idx);
}

// sum of four 16-bit uniforms, centred: mean 0, std 65536/sqrt(3) = 37837.2
MFX_HD inline int64_t irwin4(uint64_t h)
{
    return (int64_t)((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48)) -
           2 * 65535 - 0; // [-131070, 131070]
}

// multiplier of the affine id permutation: a prime that does not divide `dim`
MFX_HD inline uint64_t synth_mult(uint64_t dim)
{
    const uint64_t primes[4] = {2654435761ull, 1000003ull, 7919ull, 104729ull};
    for (int i = 0; i < 4; ++i)
        if (dim % primes[i] != 0) return primes[i] % dim == 0 ? 1 : primes[i];
    return 1;
}

// id in [0, dim): bit 63 of h picks uniform or Zipf(0.8); Zipf rank = dim * x^5 with
// x = 32-bit fraction (density ~ rank^-0.8), then the affine permutation.
MFX_HD inline int synth_id(uint64_t h, uint64_t dim, uint64_t mult, uint64_t add)
{
    uint64_t x = h & 0xFFFFFFFFull;
    uint64_t rank;
    if (h >> 63) {
        uint64_t y = x;
        for (int i = 0; i < 4; ++i) y = (y * x) >> 32;
        rank = (y * dim) >> 32;
    } else {
        rank = (x * dim) >> 32;
    }
    return (int)((rank * (mult % dim) + add) % dim);
}

// Rating i of shard `shard`.  A shard is one user range of a larger problem (one per GPU):
// its m users are distinct from every other shard's, the n items and their planted factors
// are shared by all shards.  shard 0 alone is the single-GPU problem.
MFX_HD inline SynthNode synth_rating(uint64_t seed, uint64_t shard, int64_t i, int m, int n)
{
    SynthNode N;
    const uint64_t dseed = seed ^ (shard * 0xA24BAED4963EE407ull); // per-shard draw streams
    const uint64_t um = (uint64_t)m, un = (uint64_t)n;
    const int64_t cover = m > n ? m : n;
    if (i < cover) { // coverage pass: every user and every item at least once
        N.u = (int)((uint64_t)i % um);
        N.v = (int)(((uint64_t)i * (synth_mult(un) % un) + 12345u) % un);
    } else {
        N.u = synth_id(synth_hash(dseed, 1, (uint64_t)i), um, synth_mult(um), 17);
        N.v = synth_id(synth_hash(dseed, 2, (uint64_t)i), un, synth_mult(un), 29);
    }
    int64_t dot = 0;
    for (int d = 0; d < 16; ++d) {
        int64_t a = irwin4(synth_hash(seed, 3, (shard * um + (uint64_t)N.u) * 16 + d));
        int64_t b = irwin4(synth_hash(seed, 4, (uint64_t)N.v * 16 + d));
        dot += a * b;
    }
    // entries have std 0.5 (variance 1/sqrt(16)); Q20 fixed point:
    //   dot_real = dot * 0.25/37837.2^2 ; noise_real = 0.5 * nz/37837.2
    const int64_t K1 = 3145878; // round(0.25/37837.2^2 * 2^54)
    const int64_t K2 = 908116;  // round(0.5/37837.2 * 2^36)
    int64_t nz = irwin4(synth_hash(dseed, 5, (uint64_t)i));
    int64_t q20 = (3ll << 20) + ((dot * K1) >> 34) + ((nz * K2) >> 16);
    if (q20 < (1ll << 20)) q20 = 1ll << 20;
    if (q20 > (5ll << 20)) q20 = 5ll << 20;
    N.r = (float)q20 * (1.0f / 1048576.0f);
    return N;
}

} // namespace mfx
