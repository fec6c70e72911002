// Do v_permlane16_swap / v_permlane32_swap (gfx950) give the xor-16 / xor-32 butterfly that ds_bpermute (__shfl_xor) gives?
// build: hipcc -O3 --offload-arch=gfx950 permlane_probe.hip -o permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2v __attribute__((ext_vector_type(2)));
// (inline asm: with the builtin and the same value for both operands hipcc 7.2 folds the two results into one)
__device__ inline float xor16_sum(float x)
{
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ inline float xor32_sum(float x)
{
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__global__ void k(const float *in, float *o16, float *o32, float *r16, float *r32)
{
    const float x = in[threadIdx.x];
    o16[threadIdx.x] = xor16_sum(x);
    o32[threadIdx.x] = xor32_sum(x);
    r16[threadIdx.x] = x + __shfl_xor(x, 16);
    r32[threadIdx.x] = x + __shfl_xor(x, 32);
}
int main()
{
    float h[64], *d, *o;
    for (int i = 0; i < 64; ++i) h[i] = 1.0f / (float)(i + 3) + (float)(i * i) * 0.37f;
    hipMalloc(&d, 256); hipMalloc(&o, 4 * 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, o + 64, o + 128, o + 192);
    float r[256];
    hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    int bad16 = 0, bad32 = 0;
    for (int i = 0; i < 64; ++i) { bad16 += r[i] != r[128 + i]; bad32 += r[64 + i] != r[192 + i]; }
    for (int i = 0; i < 64; i += 5) printf("lane %2d x %.3f | p16 %.3f ref %.3f | p32 %.3f ref %.3f\n", i, h[i], r[i], r[128+i], r[64+i], r[192+i]);
    printf("permlane16_swap vs shfl_xor 16: %d lanes differ; permlane32_swap vs shfl_xor 32: %d lanes differ\n", bad16, bad32);
    return bad16 + bad32 != 0;
}
