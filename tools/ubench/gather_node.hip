// gather_node.hip -- what does a per-lane walk's node fetch cost the CU's L1, and does it matter HOW the 64 bytes are asked for?
// A per-lane walk reads its 64-byte node as four global_load_dwordx4 of a lane-private address: 64 lanes x 4 pieces, every
// piece of 16 bytes a request of its own.  Variants, all fetching the same 64 x 64 B per wave-"visit" from an L2-resident array
// of nodes, with dependent addresses (the next node index comes out of the loaded words, as in a walk):
//   A  per lane: 4 x dwordx4 at node[idx[lane]] + 0/16/32/48                         (what the kernels do)
//   B  quads:    4 x dwordx4, instruction k serves rays 16k..16k+15: lane l reads piece (l & 3) of node[idx[16k + (l >> 2)]]
//                -- four neighbouring lanes ask for 64 contiguous bytes -- then 16 ds_bpermute_b32 bring a ray's pieces to its lane
//   C  as B without the redistribution (the floor of the fetch alone)
//   D  LDS: the nodes in LDS, per lane 4 x ds_read_b128 at random node addresses   (an LDS copy of the tree's top)
// 16 waves per CU (4 workgroups of 256), `active` of 64 lanes walking.  One MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr int kNodes = 4096;                    // 256 KB of 64-byte nodes: L2-resident, far beyond one L1

__device__ __forceinline__ uint32_t mixu(uint32_t x) { x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15; return x; }

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(const float4 *__restrict__ nodes, uint32_t *out, int iters, int active)
{
    __shared__ float4 lds_nodes[MODE == 3 ? 4 * 512 : 4];          // D: 512 nodes = 32 KB per workgroup
    const uint32_t lane = threadIdx.x & 63u;
    if (MODE == 3) {
        for (uint32_t i = threadIdx.x; i < 4u * 512u; i += 256u) lds_nodes[i] = nodes[i];
        __syncthreads();
    }
    uint32_t idx = mixu(blockIdx.x * 256u + threadIdx.x) % kNodes, acc = 0;
    const bool on = (int)lane < active;
    for (int it = 0; it < iters; ++it) {
        float4 h0, h1, h2, h3;
        if (MODE == 0) {
            if (on) { const float4 *p = nodes + 4 * (size_t)idx; h0 = p[0]; h1 = p[1]; h2 = p[2]; h3 = p[3]; }
        } else if (MODE == 3) {
            if (on) { const float4 *p = lds_nodes + 4 * (size_t)(idx & 511u); h0 = p[0]; h1 = p[1]; h2 = p[2]; h3 = p[3]; }
        } else {
            // instruction k: lane l fetches piece (l & 3) of the node of ray-lane 16k + (l >> 2)
            float4 q[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const uint32_t src = 16u * kk + (lane >> 2);
                const uint32_t nidx = (uint32_t)__shfl((int)idx, (int)src, 64);
                const bool src_on = (int)src < active;
                q[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (src_on) q[kk] = nodes[4 * (size_t)nidx + (lane & 3u)];
            }
            if (MODE == 1) {
                // ray-lane r needs piece p from lane 4 (r & 15) + p of instruction r >> 4
                const uint32_t kk = lane >> 4;
                float4 mine[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int from = (int)(4u * (lane & 15u) + p);
                    float4 v;
                    // (the instruction index differs per 16-lane group: select after the permute)
                    float4 c0, c1, c2, c3;
                    c0.x = __shfl(q[0].x, from, 64); c0.y = __shfl(q[0].y, from, 64); c0.z = __shfl(q[0].z, from, 64); c0.w = __shfl(q[0].w, from, 64);
                    c1.x = __shfl(q[1].x, from, 64); c1.y = __shfl(q[1].y, from, 64); c1.z = __shfl(q[1].z, from, 64); c1.w = __shfl(q[1].w, from, 64);
                    c2.x = __shfl(q[2].x, from, 64); c2.y = __shfl(q[2].y, from, 64); c2.z = __shfl(q[2].z, from, 64); c2.w = __shfl(q[2].w, from, 64);
                    c3.x = __shfl(q[3].x, from, 64); c3.y = __shfl(q[3].y, from, 64); c3.z = __shfl(q[3].z, from, 64); c3.w = __shfl(q[3].w, from, 64);
                    v = kk == 0 ? c0 : (kk == 1 ? c1 : (kk == 2 ? c2 : c3));
                    mine[p] = v;
                }
                h0 = mine[0]; h1 = mine[1]; h2 = mine[2]; h3 = mine[3];
            } else {
                h0 = q[0]; h1 = q[1]; h2 = q[2]; h3 = q[3];
            }
        }
        if (on || MODE == 2) {
            const uint32_t w = __float_as_uint(h0.x) ^ __float_as_uint(h1.y) ^ __float_as_uint(h2.z) ^ __float_as_uint(h3.w);
            acc += w;
            idx = mixu(w + idx + (uint32_t)it) % kNodes;               // the next node depends on what was loaded
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main()
{
    const int cus = 256, blocks = cus * 4, iters = 4000;
    std::vector<float4> h(4 * (size_t)kNodes);
    for (size_t i = 0; i < h.size(); ++i) { uint32_t v = (uint32_t)(i * 2654435761u); float f; std::memcpy(&f, &v, 4); h[i] = make_float4(f, f, f, f); }
    float4 *d;
    uint32_t *out;
    hipMalloc(&d, h.size() * sizeof(float4));
    hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[4] = { "A per lane, 4 x dwordx4", "B quads + 16 bpermute", "C quads, no redistribution", "D LDS, 4 x ds_read_b128" };
    for (int active : {64, 40, 24}) {
        for (int mode = 0; mode < 4; ++mode) {
            auto launch = [&](int n) {
                if (mode == 0) k<0><<<blocks, 256>>>(d, out, n, active);
                else if (mode == 1) k<1><<<blocks, 256>>>(d, out, n, active);
                else if (mode == 2) k<2><<<blocks, 256>>>(d, out, n, active);
                else k<3><<<blocks, 256>>>(d, out, n, active);
            };
            launch(50);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch(iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double visits = (double)blocks * 4 * iters;             // wave-visits
            printf("active %2d  %-28s %8.3f ms  %7.1f cycles per wave-visit per CU @2.4GHz  (%.2f lane-requests of 16 B per cycle per CU)\n",
                   active, names[mode], ms, ms * 1e-3 * 2.4e9 / (visits / cus), visits / cus * active * 4 / (ms * 1e-3 * 2.4e9));
        }
    }
    return 0;
}
