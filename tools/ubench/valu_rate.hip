// valu_rate.hip -- microbenchmark: how many lane-FMAs per second do v_fma_f32 / v_pk_fma_f32 sustain on
// gfx950 at 1, 2, 4 waves per SIMD?  Decides whether the sphere filter should be packed or not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, int iters, float a, float b)
{
    float x[16];
    float2v y[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = float2v{x[2 * i], x[2 * i + 1]};
    float2v a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a2), "v"(b2));
        } else if (MODE == 2) {  // fma with an SGPR operand
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(b));
        } else if (MODE == 3) {  // v_max3 + cmp mix
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        } else if (MODE == 4) {  // v_sub_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int waves_per_simd, int lanes_per_instr)
{
    int cus = 256;
    int threads = 256;                       // 4 waves per block = 1 per SIMD
    int blocks = cus * waves_per_simd;
    float *out;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * (threads / 64) * iters * (MODE == 1 ? 8 : 16);
    double laneops = instr * 64 * lanes_per_instr;
    // cycles per wave-instruction per SIMD assuming 2.4 GHz
    double simd_instr = instr / (cus * 4.0);
    printf("%-22s waves/SIMD %d: %.3f ms  %.2f T lane-ops/s  %.2f cyc/instr/SIMD @2.4GHz\n", name, waves_per_simd, ms,
           laneops / ms / 1e9, ms * 1e-3 * 2.4e9 / simd_instr);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w, 1);
        run<1>("v_pk_fma_f32", w, 2);
        run<2>("v_fma_f32 (sgpr src)", w, 1);
        run<3>("v_max3_f32", w, 1);
        run<4>("v_sub_f32", w, 1);
    }
    return 0;
}
