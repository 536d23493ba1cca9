// exec_mask.hip -- does a wave64 VALU instruction issue faster when only the low 32 / 16 lanes are active?
// (If it did, compacting the few lanes that still walk at the tail of a round would pay.)  One MI355X, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(float *out, int iters, float a, float b, int active)
{
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i;
    if ((int)(threadIdx.x & 63) < active) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    const int cus = 256, threads = 256, blocks = cus * 4, iters = 20000;
    float *out;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int active : {64, 48, 32, 16, 8, 1}) {
        k<<<blocks, threads>>>(out, 100, 1.0001f, 0.5f, active);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<blocks, threads>>>(out, iters, 1.0001f, 0.5f, active);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)blocks * (threads / 64) * iters * 32;
        printf("active lanes %2d (the low ones): %.3f ms  %.2f cyc/instr/SIMD @2.4GHz\n", active, ms, ms * 1e-3 * 2.4e9 / (instr / (cus * 4.0)));
    }
    return 0;
}
