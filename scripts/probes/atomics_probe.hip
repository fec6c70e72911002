// Throughput probe: read-modify-write of random 256-byte rows by 16-lane groups, the update written as
//   0 one 16-byte store per lane            (what sgd_round does on the lock-free side)
//   1 four float atomics per lane, lane-major addresses (lane*16 + i*4), scope = workgroup
//   2 the same, scope = agent
//   3 four float atomics per lane, instruction-major addresses (i*64 + lane*4): one instruction = 64 contiguous bytes per group
//   4 as 3, scope = agent
//   5 one packed 2 x f32 atomic? (not on gfx950 for global memory: skipped)
// Rows are drawn from the partition of the wave's own XCD (HW_REG_XCC_ID), as in the trainer.
// build: hipcc -O3 --offload-arch=gfx950 atomics_probe.hip -o atomics_probe ; run: ./atomics_probe [rows_per_xcd] [steps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7u; }

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* tab, unsigned rows_per_xcd, int steps, unsigned long long* sink)
{
    const unsigned lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const unsigned xcc = xcc_id();
    unsigned long long s = (blockIdx.x * 256ull + threadIdx.x / 16) * 0x9E3779B97F4A7C15ull + 12345;
    float acc = 0.f;
    float* base = tab + (size_t)xcc * rows_per_xcd * 64;
    for (int it = 0; it < steps; ++it) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        unsigned row = (unsigned)((s >> 33) % rows_per_xcd);
        float* r = base + (size_t)row * 64;
        float4 x = *reinterpret_cast<const float4*>(r + sub * 4);
        float d = 1e-6f * (x.x + x.y + x.z + x.w) + 1e-7f;
        acc += d;
        if (MODE == 0) {
            x.x += d; x.y += d; x.z += d; x.w += d;
            *reinterpret_cast<float4*>(r + sub * 4) = x;
        } else if (MODE == 1) {
            for (int i = 0; i < 4; ++i) __hip_atomic_fetch_add(r + sub * 4 + i, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 2) {
            for (int i = 0; i < 4; ++i) __hip_atomic_fetch_add(r + sub * 4 + i, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 3) {
            for (int i = 0; i < 4; ++i) __hip_atomic_fetch_add(r + i * 16 + sub, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 4) {
            for (int i = 0; i < 4; ++i) __hip_atomic_fetch_add(r + i * 16 + sub, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 5) {
            for (int i = 0; i < 4; ++i) unsafeAtomicAdd(r + i * 16 + sub, d);
        }
        (void)grp;
    }
    if (acc == 123.456f) sink[0] = 1;
}

template <int MODE>
static int run(const char* name, float* tab, unsigned rows, int steps, unsigned long long* sink, int wgs)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(256), 0, 0, tab, rows, steps / 10, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(256), 0, 0, tab, rows, steps, sink);
    CHECK(hipEventRecord(b)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    double rows_done = (double)wgs * 16 * steps;
    printf("  %-58s %8.3f ms  %7.2f G rows/s  %7.1f GB/s of row updates\n", name, ms, rows_done / ms / 1e6, rows_done * 256 / ms / 1e6);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv)
{
    int steps = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned long long* sink; CHECK(hipMalloc(&sink, 8));
    for (unsigned rows : {2048u, 16384u, 62500u, 1000000u}) {
        if (argc > 1 && atoi(argv[1]) > 0) rows = atoi(argv[1]);
        float* tab; size_t bytes = (size_t)rows * 8 * 256;
        CHECK(hipMalloc(&tab, bytes)); CHECK(hipMemset(tab, 0, bytes));
        for (int wgs : {512, 2048}) {
            printf("rows per XCD %u (%.1f MB per XCD), %d workgroups of 4 waves, %d steps\n", rows, rows * 256 / 1e6, wgs, steps);
            if (run<0>("0 store 16 B per lane", tab, rows, steps, sink, wgs)) return 1;
            if (run<1>("1 4 x atomic f32, lane-major, workgroup scope", tab, rows, steps, sink, wgs)) return 1;
            if (run<2>("2 4 x atomic f32, lane-major, agent scope", tab, rows, steps, sink, wgs)) return 1;
            if (run<3>("3 4 x atomic f32, instruction-major (64 B contiguous), wg scope", tab, rows, steps, sink, wgs)) return 1;
            if (run<4>("4 the same, agent scope", tab, rows, steps, sink, wgs)) return 1;
            if (run<5>("5 unsafeAtomicAdd, instruction-major", tab, rows, steps, sink, wgs)) return 1;
        }
        CHECK(hipFree(tab));
        if (argc > 1 && atoi(argv[1]) > 0) break;
    }
    return 0;
}
