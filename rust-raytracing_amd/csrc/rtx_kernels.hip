// rtx_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the Scene::render hot path.
//
//   trace_exact_kernel  one thread per ray (pixel, sample); every shape test in f64 in the
//                       reference's operation order.  The parity kernel.
//   trace_mixed_kernel  persistent workgroups; each lane owns S ray slots whose f64 state lives in
//                       HBM structure-of-arrays; the sphere list is staged through LDS in chunks as
//                       16-byte f32 records and swept with a conservative f32 discriminant filter
//                       (9 VALU ops per sphere per ray, one broadcast ds_read_b128 shared by 64*S
//                       rays); candidates the filter cannot exclude are queued per slot in LDS and
//                       re-evaluated exactly in f64, so the result has the same bits as
//                       trace_exact_kernel.  Dead slots are refilled from a global ray queue with a
//                       wave ballot + mbcnt prefix sum (one atomic per workgroup per round).
//   resolve_kernel      left fold of a pixel's samples in sample order, then / rays_per_pixel.
//   quantize_kernel     render_to_image epilogue.
//
// Compiled with -ffp-contract=off; the f32 filter uses explicit fmaf.
#include "rtx_launch.h"
#include "rtx_traverse.h"

#include <cstdlib>

namespace rtx {

// ------------------------------------------------------------------------------------------
// wave helpers (wave = 64 lanes on gfx950)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ uint32_t mbcnt(unsigned long long mask)      // # set bits of mask below this lane
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ void flush_counters(Counters *ctr, unsigned long long segs, unsigned long long exact,
                                               unsigned long long filt)
{
    segs = wave_sum_u64(segs);
    exact = wave_sum_u64(exact);
    filt = wave_sum_u64(filt);
    if (lane_id() == 0) {
        uint32_t shard = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (kCounterShards - 1);
        if (segs) atomicAdd(&ctr[shard].segments, segs);
        if (exact) atomicAdd(&ctr[shard].exact_tests, exact);
        if (filt) atomicAdd(&ctr[shard].filter_tests, filt);
    }
}

// closest_object (scene.rs:243-251) over every shape of one type, exact f64.
__device__ __forceinline__ void closest_spheres_exact(const SceneView &sv, const RayX &rx, Hit &h)
{
    for (uint32_t k = 0; k < sv.n_spheres; ++k) {
        double t;
        if (sphere_distance(sv.spheres[k], rx, &t)) hit_consider(h, t, sv.sphere_id[k], 0, k);
    }
}

__device__ __forceinline__ void closest_planes_exact(const SceneView &sv, const RayX &rx, Hit &h)
{
    for (uint32_t k = 0; k < sv.n_planes; ++k) {
        double t;
        if (plane_distance(sv.planes[k], rx, &t)) hit_consider(h, t, sv.planes[k].id, 1, k);
    }
}

__device__ __forceinline__ void closest_tris_exact(const SceneView &sv, const RayX &rx, Hit &h)
{
    for (uint32_t k = 0; k < sv.n_tris; ++k) {
        double t;
        if (triangle_distance(sv.tris[k], rx, &t)) hit_consider(h, t, sv.tris[k].id, 2, k);
    }
}

// ------------------------------------------------------------------------------------------
// EXACT kernel
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trace_exact_kernel(const SceneView *__restrict__ svp, const RowsView *__restrict__ rvp,
                                                          double *__restrict__ samples, Counters *__restrict__ ctr,
                                                          uint64_t ray_offset)
{
    // scene and launch descriptors live in device memory (not in the kernarg segment): fields are
    // (re)loaded where they are used instead of pinning ~100 SGPRs for the whole kernel
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    uint64_t i = ray_offset + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long segs = 0;
    if (i < rv.n_rays) {
        uint32_t pl, s_local;
        ray_index_to_pixel(rv, i, pl, s_local);
        RayState r;
        gen_primary(sv, rv, pl, rv.sample_begin + s_local, r);
        if (sv.n_objects != 0) {                                          // scene.rs:224-226
            const uint64_t limit = sv.max_bounces + 1;                    // scene.rs:227
            for (uint64_t b = 0; b < limit; ++b) {
                if (light_is_zero(r)) break;                              // scene.rs:228
                RayX rx = make_rayx(r.pos, r.dir);
                Hit h;
                hit_init(h);
                ++segs;
                closest_spheres_exact(sv, rx, h);
                closest_planes_exact(sv, rx, h);
                closest_tris_exact(sv, rx, h);
                if (h.id == 0xFFFFFFFFu) break;                           // scene.rs:238
                advance_and_shade(sv, h, r);
            }
        }
        store_sample(samples, rv, i, r.result);
    }
    flush_counters(ctr, segs, segs * sv.n_objects, 0);
}

#ifdef RTX_LAB
// The EXACT kernel's loop once more, writing every segment down (rtx_debug_paths): step b of the path of local pixel pl, sample s
// goes to steps[(pl * n_samples + s) * max_steps + b] while b < max_steps; counts[] takes the path's number of segments.
__global__ __launch_bounds__(256) void trace_transcript_kernel(const SceneView *__restrict__ svp, const RowsView *__restrict__ rvp,
                                                               PathStep *__restrict__ steps, uint32_t *__restrict__ counts, uint32_t max_steps)
{
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rv.n_rays) return;
    uint32_t pl, s_local;
    ray_index_to_pixel(rv, i, pl, s_local);
    RayState r;
    gen_primary(sv, rv, pl, rv.sample_begin + s_local, r);
    PathStep *out = steps + ((size_t)pl * rv.n_samples + s_local) * max_steps;
    uint32_t n = 0;
    if (sv.n_objects != 0) {
        const uint64_t limit = sv.max_bounces + 1;
        for (uint64_t b = 0; b < limit; ++b) {
            if (light_is_zero(r)) break;
            RayX rx = make_rayx(r.pos, r.dir);
            Hit h;
            hit_init(h);
            closest_spheres_exact(sv, rx, h);
            closest_planes_exact(sv, rx, h);
            closest_tris_exact(sv, rx, h);
            const bool miss = h.id == 0xFFFFFFFFu;
            if (n < max_steps) {
                PathStep st;
                st.pos[0] = r.pos.x; st.pos[1] = r.pos.y; st.pos[2] = r.pos.z;
                st.dir[0] = r.dir.x; st.dir[1] = r.dir.y; st.dir[2] = r.dir.z;
                st.t = miss ? __builtin_inf() : h.t;
                st.object = miss ? -1ll : (long long)h.id;
                out[n] = st;
            }
            ++n;
            if (miss) break;
            advance_and_shade(sv, h, r);
        }
    }
    counts[(size_t)pl * rv.n_samples + s_local] = n;
}

hipError_t launch_trace_transcript(const SceneView *d_sv, const RowsView *d_rv, const RowsView &rv, PathStep *steps, uint32_t *counts,
                                   uint32_t max_steps, hipStream_t stream)
{
    if (rv.n_rays == 0) return hipSuccess;
    if (rv.n_rays > (1ull << 30)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(trace_transcript_kernel, dim3((uint32_t)((rv.n_rays + 255) / 256)), dim3(256), 0, stream, d_sv, d_rv, steps, counts, max_steps);
    return hipGetLastError();
}
#endif

// ------------------------------------------------------------------------------------------
// MIXED kernel
// ------------------------------------------------------------------------------------------
constexpr uint32_t kInvalid = 0xFFFFFFFFu;

// The f32 filter evaluates  D = ((c-p).d)^2 - (|c-p|^2 - r^2)  (a quarter of the reference's
// discriminant for |d| = 1, sphere.rs:22-25) in the expanded form
//     b = c.d - p.d            q = 2 c.p - p.p + E            D = b*b + (q - (c.c - r*r))
// with all points relative to the scene centre.  E bounds the f32 evaluation error, so a sphere the
// reference could report (discriminant > 1e-100) always has D_f32 >= 0 and is queued.
// Two spheres are evaluated per instruction (v_pk_fma_f32): LDS holds the records pair-interleaved,
//     float4 A = {x0, x1, y0, y1}   float4 B = {z0, z1, w0, w1},   w = |c|^2 - r^2
// and the slot holds {d.xyz, -p.d, 2p.xyz, -p.p + E}.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// D for the two spheres of one LDS pair record
__device__ __forceinline__ f32x2 filter_disc2(const float4 A, const float4 B, const FilterParams &f)
{
    const f32x2 x = {A.x, A.y}, y = {A.z, A.w}, z = {B.x, B.y}, w = {B.z, B.w};
    f32x2 b = __builtin_elementwise_fma(z, (f32x2){f.dz, f.dz}, (f32x2){f.npd, f.npd});
    b = __builtin_elementwise_fma(y, (f32x2){f.dy, f.dy}, b);
    b = __builtin_elementwise_fma(x, (f32x2){f.dx, f.dx}, b);
    f32x2 q = __builtin_elementwise_fma(z, (f32x2){f.p2z, f.p2z}, (f32x2){f.nppE, f.nppE});
    q = __builtin_elementwise_fma(y, (f32x2){f.p2y, f.p2y}, q);
    q = __builtin_elementwise_fma(x, (f32x2){f.p2x, f.p2x}, q);
    return __builtin_elementwise_fma(b, b, q - w);
}

// Per-slot data that lives between rounds, structure-of-arrays in HBM (slot = lane_gid + s * n_lanes):
//   state  [12][n_slots] f64   ray.position, ray.direction, resulting_color, light_color  (ray.rs:4-21)
//   fstate [16][n_slots] f32   the slot's sphere-filter (0-7) and triangle-filter (8-15) parameters
//   istate [ 3][n_slots] u32   local pixel, batch-local sample, bounce count
//   oflow  [24][n_slots] u32   candidates beyond the Q-entry LDS queue (a ray that collects more than Q + 24
//                              candidates falls back to the exhaustive exact sweep)
// Only a one-bit-per-slot `live` mask stays in registers across rounds, so the heavy f64 code (primary ray
// generation, exact tests, shading) exists once and loops over the slots instead of being unrolled S times.
constexpr int kOverflowQ = 24;             // candidates per slot beyond the LDS queue, kept in HBM (rare)
constexpr size_t kSlotBytes = 12 * sizeof(double) + 16 * sizeof(float) + (3 + kOverflowQ) * sizeof(uint32_t);

template <int S, int THREADS, int CHUNK, int Q, int WAVES_PER_EU>
__global__ __launch_bounds__(THREADS, WAVES_PER_EU) void trace_mixed_kernel(const SceneView *__restrict__ svp,
                                                              const RowsView *__restrict__ rvp, double *__restrict__ samples,
                                                              char *__restrict__ slot_mem, Counters *__restrict__ ctr,
                                                              unsigned long long *__restrict__ work_counter,
                                                              uint32_t verify, unsigned long long max_rounds)
{
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    constexpr int WAVES = THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *lds_sph = reinterpret_cast<float4 *>(smem);                                  // [CHUNK]
    uint32_t *lds_q = reinterpret_cast<uint32_t *>(smem + (size_t)CHUNK * 16);           // [Q][S][THREADS]
    uint32_t *lds_cnt = lds_q + (size_t)Q * S * THREADS;                                 // [S][THREADS]
    uint32_t *lds_misc = lds_cnt + (size_t)S * THREADS;                                  // [WAVES + 4]

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint64_t n_lanes = (uint64_t)gridDim.x * THREADS;         // SoA stride between the slots of one lane
    const uint64_t n_slots = n_lanes * S;                           // SoA stride between fields
    const uint64_t lane_gid = (uint64_t)blockIdx.x * THREADS + tid;
    double *state = reinterpret_cast<double *>(slot_mem);
    float *fstate = reinterpret_cast<float *>(state + 12 * n_slots);
    uint32_t *istate = reinterpret_cast<uint32_t *>(fstate + 16 * n_slots);
    uint32_t *oflow = istate + 3 * n_slots;
    const uint32_t ns = sv.n_spheres;
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;

    uint32_t live = 0;                           // bit s: slot s holds a ray
    unsigned long long segs = 0, exact = 0, mism = 0, rounds_swept = 0;

    // Every round either advances every live ray by one segment or finds the queue drained, so a
    // workgroup needs at most (rays / slots + 2) * (max_bounces + 1) rounds; max_rounds is that bound
    // (computed by the launcher).  Hitting it means a logic error: the workgroup leaves and flags it.
    for (unsigned long long round = 0;; ++round) {
        if (round >= max_rounds) {
            if (tid == 0) atomicAdd(&ctr[1].pad_, 1ull);
            break;
        }
        // ---- refill idle slots from the global ray queue: ballot + prefix sum, one atomic per workgroup
        uint32_t wave_need = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) wave_need += (uint32_t)__popcll(__ballot(((live >> s) & 1u) == 0u));
        if (lane == 0) lds_misc[wave] = wave_need;
        __syncthreads();
        if (tid == 0) {
            uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) total += lds_misc[w];
            unsigned long long base = 0;
            if (total) base = atomicAdd(work_counter, (unsigned long long)total);
            lds_misc[WAVES] = (uint32_t)base;
            lds_misc[WAVES + 1] = (uint32_t)(base >> 32);
            lds_misc[WAVES + 2] = 0;           // "some slot of this workgroup is live" (set below)
        }
        __syncthreads();
        {
            unsigned long long base = ((unsigned long long)lds_misc[WAVES + 1] << 32) | lds_misc[WAVES];
            for (uint32_t w = 0; w < wave; ++w) base += lds_misc[w];
#pragma unroll 1
            for (int s = 0; s < S; ++s) {
                const bool idle = ((live >> s) & 1u) == 0u;
                const unsigned long long m = __ballot(idle);          // same value as in the count above: bit s unchanged so far
                if (idle) {
                    const unsigned long long my = base + mbcnt(m);
                    if (my < rv.n_rays) {
                        uint32_t p, sl;
                        ray_index_to_pixel(rv, my, p, sl);
                        RayState r;
                        gen_primary(sv, rv, p, rv.sample_begin + sl, r);
                        if (sv.n_objects == 0) {                                  // scene.rs:224-226
                            store_sample(samples, rv, my, mk(0.0, 0.0, 0.0));
                        } else {
                            const uint64_t slot = lane_gid + (uint64_t)s * n_lanes;
                            double *st = state + slot;
                            st[0 * n_slots] = r.pos.x; st[1 * n_slots] = r.pos.y; st[2 * n_slots] = r.pos.z;
                            st[3 * n_slots] = r.dir.x; st[4 * n_slots] = r.dir.y; st[5 * n_slots] = r.dir.z;
                            st[6 * n_slots] = 0.0; st[7 * n_slots] = 0.0; st[8 * n_slots] = 0.0;
                            st[9 * n_slots] = 1.0; st[10 * n_slots] = 1.0; st[11 * n_slots] = 1.0;
                            FilterParams f;
                            filter_from_ray(sv, r.pos, r.dir, f);
                            float *fs = fstate + slot;
                            fs[0 * n_slots] = f.dx; fs[1 * n_slots] = f.dy; fs[2 * n_slots] = f.dz; fs[3 * n_slots] = f.npd;
                            fs[4 * n_slots] = f.p2x; fs[5 * n_slots] = f.p2y; fs[6 * n_slots] = f.p2z; fs[7 * n_slots] = f.nppE;
                    {
                        TriFilterParams tf;
                        if (sv.n_tri_filter != 0) tri_filter_from_ray(sv, r.pos, r.dir, tf); else tri_filter_idle(tf);
                        fs[8 * n_slots] = tf.dx; fs[9 * n_slots] = tf.dy; fs[10 * n_slots] = tf.dz; fs[11 * n_slots] = tf.npx;
                        fs[12 * n_slots] = tf.npy; fs[13 * n_slots] = tf.npz; fs[14 * n_slots] = tf.A;
                    }
                            uint32_t *is = istate + slot;
                            is[0 * n_slots] = p; is[1 * n_slots] = sl; is[2 * n_slots] = 0u;
                            live |= 1u << s;
                        }
                    }
                }
                base += (unsigned long long)__popcll(m);
            }
        }
        // ---- workgroup-uniform exit: nothing live here and the queue handed out nothing
        if (__ballot(live != 0u) != 0ull && lane == 0) lds_misc[WAVES + 2] = 1;
        __syncthreads();
        // the flag is cleared by thread 0 between the two refill barriers of the NEXT round, i.e. after
        // every wave has passed this read
        if (lds_misc[WAVES + 2] == 0) break;

        // ---- sweep the sphere list through LDS with the f32 filter
        FilterParams fp[S];
        uint32_t cnt[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            cnt[s] = 0;
            filter_idle(fp[s]);
            if ((live >> s) & 1u) {
                const float *fs = fstate + (lane_gid + (uint64_t)s * n_lanes);
                fp[s].dx = fs[0 * n_slots]; fp[s].dy = fs[1 * n_slots]; fp[s].dz = fs[2 * n_slots]; fp[s].npd = fs[3 * n_slots];
                fp[s].p2x = fs[4 * n_slots]; fp[s].p2y = fs[5 * n_slots]; fp[s].p2z = fs[6 * n_slots]; fp[s].nppE = fs[7 * n_slots];
            }
        }
        for (uint32_t c0 = 0; c0 < ns; c0 += CHUNK) {
            const uint32_t n = (ns - c0 < (uint32_t)CHUNK) ? ns - c0 : (uint32_t)CHUNK;
            const uint32_t n4 = (n + 3u) & ~3u;            // the device array is padded to a multiple of 4
            if (c0 != 0) __syncthreads();                  // everyone is done with the previous chunk
            for (uint32_t j = tid; j < n4; j += THREADS) lds_sph[j] = sv.sphere_f32[c0 + j];
            __syncthreads();
#pragma unroll 1
            for (uint32_t j = 0; j < n4; j += 4) {
                const float4 A0 = lds_sph[j], B0 = lds_sph[j + 1], A1 = lds_sph[j + 2], B1 = lds_sph[j + 3];
                f32x2 d[S][2];
                uint32_t sign_and = 0xFFFFFFFFu;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    d[s][0] = filter_disc2(A0, B0, fp[s]);
                    d[s][1] = filter_disc2(A1, B1, fp[s]);
                    sign_and &= __float_as_uint(d[s][0].x) & __float_as_uint(d[s][0].y) &
                                __float_as_uint(d[s][1].x) & __float_as_uint(d[s][1].y);
                }
                if ((int)sign_and >= 0) {                  // some D has its sign bit clear: D >= 0
#pragma unroll
                    for (int s = 0; s < S; ++s) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float dk = (k & 1) ? d[s][k >> 1].y : d[s][k >> 1].x;
                            if ((int)__float_as_uint(dk) >= 0) {
                                if (cnt[s] < (uint32_t)Q) lds_q[((size_t)cnt[s] * S + s) * THREADS + tid] = c0 + j + k;
                                else if (cnt[s] < (uint32_t)(Q + kOverflowQ))
                                    oflow[(size_t)(cnt[s] - Q) * n_slots + lane_gid + (uint64_t)s * n_lanes] = c0 + j + k;
                                cnt[s] += 1;                       // > Q + kOverflowQ marks overflow
                            }
                        }
                    }
                }
            }
        }
        // ---- the same for the triangles that can be hit at all: 32-byte records {n, n.(v0-c)} {cx, cy, hx, hy}
        if (sv.n_tri_filter != 0) {
            const uint32_t nt = sv.n_tri_filter;
            constexpr uint32_t TCHUNK = CHUNK / 2;                  // records per LDS load (2 float4 each)
            TriFilterParams tp[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                tri_filter_idle(tp[s]);
                if ((live >> s) & 1u) {
                    const float *fs = fstate + (lane_gid + (uint64_t)s * n_lanes);
                    tp[s].dx = fs[8 * n_slots]; tp[s].dy = fs[9 * n_slots]; tp[s].dz = fs[10 * n_slots]; tp[s].npx = fs[11 * n_slots];
                    tp[s].npy = fs[12 * n_slots]; tp[s].npz = fs[13 * n_slots]; tp[s].A = fs[14 * n_slots];
                }
            }
            for (uint32_t c0 = 0; c0 < nt; c0 += TCHUNK) {
                const uint32_t n = (nt - c0 < TCHUNK) ? nt - c0 : TCHUNK;
                if (c0 != 0 || ns != 0) __syncthreads();           // everyone is done with the previous LDS contents
                for (uint32_t j = tid; j < 2 * n; j += THREADS) lds_sph[j] = sv.tri_f32[2 * (size_t)c0 + j];
                __syncthreads();
#pragma unroll 1
                for (uint32_t j = 0; j < n; j += 2) {
                    const bool two = j + 1 < n;
                    const float4 A0 = lds_sph[2 * j], B0 = lds_sph[2 * j + 1];
                    const float4 A1 = two ? lds_sph[2 * j + 2] : A0, B1 = two ? lds_sph[2 * j + 3] : B0;
                    uint32_t sg[S][2];
                    uint32_t sign_and = 0xFFFFFFFFu;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        sg[s][0] = tri_filter_sign(A0, B0, tp[s]);
                        sg[s][1] = two ? tri_filter_sign(A1, B1, tp[s]) : 0x80000000u;
                        sign_and &= sg[s][0] & sg[s][1];
                    }
                    if ((int)sign_and >= 0) {
#pragma unroll
                        for (int s = 0; s < S; ++s) {
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                if ((int)sg[s][k] >= 0) {
                                    if (cnt[s] < (uint32_t)Q)
                                        lds_q[((size_t)cnt[s] * S + s) * THREADS + tid] = 0x80000000u | sv.tri_fidx[c0 + j + k];
                                    else if (cnt[s] < (uint32_t)(Q + kOverflowQ))
                                        oflow[(size_t)(cnt[s] - Q) * n_slots + lane_gid + (uint64_t)s * n_lanes] =
                                            0x80000000u | sv.tri_fidx[c0 + j + k];
                                    cnt[s] += 1;
                                }
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < S; ++s) lds_cnt[s * THREADS + tid] = cnt[s];
        rounds_swept += 1;

        // ---- exact f64 re-evaluation of the candidates, other shapes, shading (one code instance, loop over slots)
#pragma unroll 1
        for (int s = 0; s < S; ++s) {
            if (((live >> s) & 1u) == 0u) continue;
            const uint64_t slot = lane_gid + (uint64_t)s * n_lanes;
            double *st = state + slot;
            uint32_t *is = istate + slot;
            const uint32_t pl = is[0 * n_slots], smp = is[1 * n_slots], bnc = is[2 * n_slots];
            const uint32_t ncand = lds_cnt[s * THREADS + tid];
            RayState r;
            r.pos = mk(st[0 * n_slots], st[1 * n_slots], st[2 * n_slots]);
            r.dir = mk(st[3 * n_slots], st[4 * n_slots], st[5 * n_slots]);
            const RayX rx = make_rayx(r.pos, r.dir);
            Hit h;
            hit_init(h);
            ++segs;
            if (ncand > (uint32_t)(Q + kOverflowQ)) {
                // more candidates than the queue holds (many shapes along one line, or a pass-all filter): exact sweep
                closest_spheres_exact(sv, rx, h);
                closest_tris_exact(sv, rx, h);
                exact += ns + sv.n_tris;
            } else {
                for (uint32_t k = 0; k < ncand; ++k) {
                    const uint32_t idx = k < (uint32_t)Q ? lds_q[((size_t)k * S + s) * THREADS + tid]
                                                         : oflow[(size_t)(k - Q) * n_slots + slot];
                    double t;
                    if (idx & 0x80000000u) {                                  // a triangle candidate
                        const uint32_t ti = idx & 0x7FFFFFFFu;
                        if (ti < sv.n_tris && triangle_distance(sv.tris[ti], rx, &t)) hit_consider(h, t, sv.tris[ti].id, 2, ti);
                    } else if (idx < ns) {
                        if (sphere_distance(sv.spheres[idx], rx, &t)) hit_consider(h, t, sv.sphere_id[idx], 0, idx);
                    }
                }
                exact += ncand;
                if (verify) {            // debug: the filters must never lose the exact winner
                    Hit hv;
                    hit_init(hv);
                    closest_spheres_exact(sv, rx, hv);
                    closest_tris_exact(sv, rx, hv);
                    if (hv.id != h.id || (hv.id != kInvalid && hv.t != h.t)) {
                        ++mism;
                        if (verify > 1 && mism < 3)
                            printf("MISMATCH round %llu bnc %u pl %u blk %u tid %u s %d ncand %u h(%u,%g) hv(%u,%g) pos(%g,%g,%g) "
                                   "dir(%g,%g,%g)\n", round, bnc, pl, blockIdx.x, tid, s, ncand, h.id, h.t, hv.id, hv.t, r.pos.x,
                                   r.pos.y, r.pos.z, r.dir.x, r.dir.y, r.dir.z);
                    }
                }
            }
            closest_planes_exact(sv, rx, h);
            exact += sv.n_planes;

            r.result = mk(st[6 * n_slots], st[7 * n_slots], st[8 * n_slots]);
            bool done = true;
            if (h.id != kInvalid) {                                            // scene.rs:233-236
                r.light = mk(st[9 * n_slots], st[10 * n_slots], st[11 * n_slots]);
                const uint32_t k = fastdiv(pl, rv.div_width);
                const uint32_t x = pl - k * rv.width;
                const uint64_t pix = (uint64_t)image_row(rv, k) * rv.width + x;
                r.key = rng_key(sv.seed, pix, rv.sample_begin + smp);
                r.draw = 6u + 2u * bnc;
                r.bounce = bnc;
                advance_and_shade(sv, h, r);
                done = (r.bounce >= bounce_limit) || light_is_zero(r);         // scene.rs:227-228
                if (!done) {
                    st[0 * n_slots] = r.pos.x; st[1 * n_slots] = r.pos.y; st[2 * n_slots] = r.pos.z;
                    st[3 * n_slots] = r.dir.x; st[4 * n_slots] = r.dir.y; st[5 * n_slots] = r.dir.z;
                    st[6 * n_slots] = r.result.x; st[7 * n_slots] = r.result.y; st[8 * n_slots] = r.result.z;
                    st[9 * n_slots] = r.light.x; st[10 * n_slots] = r.light.y; st[11 * n_slots] = r.light.z;
                    is[2 * n_slots] = r.bounce;
                    FilterParams f;
                    filter_from_ray(sv, r.pos, r.dir, f);
                    float *fs = fstate + slot;
                    fs[0 * n_slots] = f.dx; fs[1 * n_slots] = f.dy; fs[2 * n_slots] = f.dz; fs[3 * n_slots] = f.npd;
                    fs[4 * n_slots] = f.p2x; fs[5 * n_slots] = f.p2y; fs[6 * n_slots] = f.p2z; fs[7 * n_slots] = f.nppE;
                    {
                        TriFilterParams tf;
                        if (sv.n_tri_filter != 0) tri_filter_from_ray(sv, r.pos, r.dir, tf); else tri_filter_idle(tf);
                        fs[8 * n_slots] = tf.dx; fs[9 * n_slots] = tf.dy; fs[10 * n_slots] = tf.dz; fs[11 * n_slots] = tf.npx;
                        fs[12 * n_slots] = tf.npy; fs[13 * n_slots] = tf.npz; fs[14 * n_slots] = tf.A;
                    }
                }
            }
            if (done) {
                store_sample(samples, rv, (uint64_t)smp * rv.npix + pl, r.result);      // = the ray's queue index (ray_index_to_pixel)
                live &= ~(1u << s);
            }
        }
    }
    flush_counters(ctr, segs, exact, rounds_swept * ((unsigned long long)ns + sv.n_tri_filter) * S);
    if (verify) {
        mism = wave_sum_u64(mism);
        if (lane == 0 && mism) atomicAdd(&ctr[0].pad_, mism);
    }
}

// ------------------------------------------------------------------------------------------
// resolve: avg() (scene.rs:253-259) = left fold from zeros (iter_ops.rs:4-8), then / len
// One thread per slot of a sample's queue order (tiles_x != 0: 8x8 pixel tiles, padding slots skipped); consecutive
// threads read consecutive 32-byte records {r, g, b, 0}, sample after sample.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resolve_kernel(const double *__restrict__ samples, double *__restrict__ acc,
                                                      double *__restrict__ out, uint32_t width, uint32_t n_rows, uint32_t tiles_x,
                                                      uint32_t per_sample, uint32_t n_samples,
                                                      double divisor, int first, int last)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= per_sample) return;
    uint32_t p = t;                                   // local pixel k * width + x
    if (tiles_x != 0u) {
        const uint32_t tile = t >> 6, j = t & 63u;
        const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const uint32_t x = tx * 8u + (j & 7u), k = ty * 8u + (j >> 3);
        if (x >= width || k >= n_rows) return;
        p = k * width + x;
    }
    double sx = 0.0, sy = 0.0, sz = 0.0;
    if (!first) { sx = acc[3 * (uint64_t)p]; sy = acc[3 * (uint64_t)p + 1]; sz = acc[3 * (uint64_t)p + 2]; }
    for (uint32_t s = 0; s < n_samples; ++s) {
        const double4 c = reinterpret_cast<const double4 *>(samples)[(uint64_t)s * per_sample + t];
        sx = sx + c.x; sy = sy + c.y; sz = sz + c.z;
    }
    if (last) {
        double *o = out + 3 * (uint64_t)p;
        o[0] = sx / divisor; o[1] = sy / divisor; o[2] = sz / divisor;
    } else {
        acc[3 * (uint64_t)p] = sx; acc[3 * (uint64_t)p + 1] = sy; acc[3 * (uint64_t)p + 2] = sz;
    }
}

// ------------------------------------------------------------------------------------------
// render_to_image epilogue (scene.rs:175-178): img[height-1-y][x] * 256, Rust `as u8`
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t rust_as_u8(double v)
{
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

__global__ __launch_bounds__(256) void quantize_kernel(const double *__restrict__ rgb, uint8_t *__restrict__ rgb8,
                                                       uint32_t width, uint32_t height)
{
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t npix = (uint64_t)width * height;
    if (p >= npix) return;
    uint32_t y = (uint32_t)(p / width);
    uint32_t x = (uint32_t)(p - (uint64_t)y * width);
    const double *c = rgb + 3 * ((uint64_t)(height - y - 1) * width + x);
    uint8_t *o = rgb8 + 3 * p;
    o[0] = rust_as_u8(c[0] * 256.0);
    o[1] = rust_as_u8(c[1] * 256.0);
    o[2] = rust_as_u8(c[2] * 256.0);
}

// render_to_image's `* 256` + saturating `as u8` (scene.rs:175-178) on the rows of a band, WITHOUT the flip (the gather
// epilogue puts rows where they belong): what a device sends over xGMI is then 3 bytes per pixel instead of 24.
__global__ __launch_bounds__(256) void quantize_values_kernel(const double *__restrict__ rgb, uint8_t *__restrict__ rgb8, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rgb8[i] = rust_as_u8(rgb[i] * 256.0);
}

// ------------------------------------------------------------------------------------------
// multi-device gather epilogue: parts[p] (cap_rows rows each) holds band p of n: the blocks of `block` image rows
// b = p, p + n, ... in order.  flip: output row height - 1 - y (render_to_image, scene.rs:176).
// ------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void deinterleave_kernel(const T *__restrict__ parts, T *__restrict__ full,
                                                           uint32_t width, uint32_t height, uint32_t n, uint32_t cap_rows,
                                                           uint32_t block, uint32_t flip)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one value of the frame
    const uint64_t row_vals = (uint64_t)width * 3;
    if (i >= row_vals * height) return;
    const uint32_t yo = (uint32_t)(i / row_vals);
    const uint64_t c = i - (uint64_t)yo * row_vals;
    const uint32_t y = flip ? height - 1u - yo : yo;
    const uint32_t b = y / block, p = b % n, k = (b / n) * block + (y - b * block);
    full[i] = parts[((uint64_t)p * cap_rows + k) * row_vals + c];
}

#ifdef RTX_LAB
__global__ void debug_math_kernel(int op, const double *a, const double *b, double *out, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r;
    switch (op) {
        case 0: r = a[i] / b[i]; break;
        case 1: r = sqrt(a[i]); break;
        case 2: r = sin(a[i]); break;
        case 3: r = cos(a[i]); break;
        case 4: case 5: { double sn, cs; sincos(a[i], &sn, &cs); r = (op == 4) ? sn : cs; } break;
        case 6:                                          // rtx_writelane (rtx_traverse.h): lane (int)b[1] of every wave takes (int)b[0]
            r = (double)rtx_writelane(__builtin_amdgcn_readfirstlane((int)b[0]), __builtin_amdgcn_readfirstlane((int)b[1]), (int)a[i]);
            break;
        case 9: case 10: { double sn, cs; sincos_2pi(a[i], &sn, &cs); r = (op == 9) ? sn : cs; } break;     // random_direction's sin / cos (rtx_math.h)
        default: {                                       // 7 / 8: the child sort's v_min_f64 / v_max_f64 on raw bit patterns
            double lo, hi;
            rtx_minmax_f64_bits(a[i], b[i], lo, hi);
            r = (op == 7) ? lo : hi;
        } break;
    }
    out[i] = r;
}
#endif

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
hipError_t launch_trace_exact(const SceneView *d_sv, const RowsView *d_rv, const RowsView &rv, double *samples,
                              Counters *counters, hipStream_t stream)
{
    const uint64_t per_launch = 1ull << 30;                 // rays per launch (grid.x stays < 2^31)
    for (uint64_t off = 0; off < rv.n_rays; off += per_launch) {
        uint64_t n = rv.n_rays - off < per_launch ? rv.n_rays - off : per_launch;
        uint32_t blocks = (uint32_t)((n + 255) / 256);
        hipLaunchKernelGGL(trace_exact_kernel, dim3(blocks), dim3(256), 0, stream, d_sv, d_rv, samples, counters, off);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

namespace {

// Launch shape of the MIXED kernel: 4 workgroups x 4 waves per CU, 2 slots per lane, 128 VGPRs, 34 KB LDS each.
// Other shapes (8 waves per workgroup, 1/3/4 slots per lane, 2048-record chunks, ...) were measured in round 1 and were
// all slower on C2 (0.67x - 0.95x); they are compiled only with -DRTX_MIXED_TUNING (then RTX_HIP_MIXED_VARIANT=n picks
// one, and tests/test_gpu_parity.py::test_mixed_tuning_variants checks each against the exact kernel), so the product
// binary holds no kernel the test-suite does not run.
struct MixVariant {
    int s, threads, chunk, q, blocks_per_cu;
    size_t lds;
    const void *fn;
};

template <int S, int T, int CH, int Q, int WPE, int BPC>
MixVariant make_variant()
{
    MixVariant v;
    v.s = S; v.threads = T; v.chunk = CH; v.q = Q; v.blocks_per_cu = BPC;
    v.lds = (size_t)CH * 16 + (size_t)Q * S * T * 4 + (size_t)S * T * 4 + (T / 64 + 4) * 4;
    v.fn = reinterpret_cast<const void *>(&trace_mixed_kernel<S, T, CH, Q, WPE>);
    return v;
}

const MixVariant &mix_variant()
{
    static const MixVariant table[] = {
        make_variant<2, 256, 1024, 8, 4, 4>(),      // 0: production
#ifdef RTX_MIXED_TUNING
        make_variant<2, 512, 2048, 8, 4, 2>(),      // 1: 2 workgroups x 8 waves per CU
        make_variant<3, 256, 1024, 8, 3, 3>(),      // 2: 3 slots per lane, 12 waves/CU
        make_variant<4, 256, 2048, 8, 2, 2>(),      // 3: 4 slots per lane, 8 waves/CU, 256 VGPRs
        make_variant<1, 512, 1024, 8, 8, 4>(),      // 4: 1 slot per lane, 32 waves/CU
        make_variant<2, 1024, 2048, 8, 4, 1>(),     // 5: one 1024-thread workgroup per CU
        make_variant<4, 256, 2048, 8, 4, 2>(),      // 6: as 3, 128 VGPRs
        make_variant<2, 256, 2048, 8, 4, 3>(),      // 7: as 0 with 2048-record chunks, 3 workgroups per CU
        make_variant<2, 128, 1024, 8, 4, 8>(),      // 8: 128-thread workgroups
#endif
    };
#ifdef RTX_MIXED_TUNING
    static const int idx = [] {                           // tuning builds only: the production binary holds one shape
        const char *e = getenv("RTX_HIP_MIXED_VARIANT");
        int i = e ? atoi(e) : 0;
        return (i < 0 || i >= (int)(sizeof(table) / sizeof(table[0]))) ? 0 : i;
    }();
    return table[idx];
#else
    return table[0];
#endif
}

}  // namespace

size_t mixed_state_bytes(int n_cus)
{
    const MixVariant &v = mix_variant();
    return (size_t)n_cus * v.blocks_per_cu * v.threads * v.s * kSlotBytes;
}

hipError_t launch_trace_mixed(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                              double *samples, double *state, Counters *counters, unsigned long long *work_counter,
                              int n_cus, bool verify, hipStream_t stream)
{
    const MixVariant &v = mix_variant();
    {   // per launch: the attribute is per device and this call may come from any thread / device
        hipError_t e = hipFuncSetAttribute(v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds);
        if (e != hipSuccess) return e;
    }
    const uint64_t slots_per_block = (uint64_t)v.threads * v.s;
    uint64_t want = (rv.n_rays + slots_per_block - 1) / slots_per_block;
    uint64_t cap = (uint64_t)n_cus * v.blocks_per_cu;
    uint32_t blocks = (uint32_t)(want < cap ? want : cap);
    if (blocks == 0) return hipSuccess;
    const unsigned long long limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0ull : sv.max_bounces + 1ull;
    unsigned long long max_rounds = (rv.n_rays / slots_per_block + 2ull) * limit + 4ull;
#ifdef RTX_MIXED_TUNING
    uint32_t verify_u = verify ? (getenv("RTX_HIP_DEBUG_PRINT") ? 2u : 1u) : 0u;
#else
    uint32_t verify_u = verify ? 1u : 0u;
#endif
    void *args[] = { (void *)&d_sv, (void *)&d_rv, (void *)&samples, (void *)&state, (void *)&counters, (void *)&work_counter,
                     (void *)&verify_u, (void *)&max_rounds };
    return hipLaunchKernel(v.fn, dim3(blocks), dim3(v.threads), args, v.lds, stream);
}

hipError_t launch_resolve(const double *samples, double *acc, double *out, const RowsView &rv, uint32_t per_sample,
                          uint64_t rays_per_pixel, bool first, bool last, hipStream_t stream)
{
    if (per_sample == 0) return hipSuccess;
    hipLaunchKernelGGL(resolve_kernel, dim3((per_sample + 255) / 256), dim3(256), 0, stream, samples, acc, out, rv.width, rv.n_rows,
                       rv.tiles_x, per_sample, rv.n_samples, (double)rays_per_pixel, first ? 1 : 0, last ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_quantize(const double *rgb, uint8_t *rgb8, uint32_t width, uint32_t height, hipStream_t stream)
{
    uint64_t npix = (uint64_t)width * height;
    if (npix == 0) return hipSuccess;
    hipLaunchKernelGGL(quantize_kernel, dim3((uint32_t)((npix + 255) / 256)), dim3(256), 0, stream, rgb, rgb8, width,
                       height);
    return hipGetLastError();
}

hipError_t launch_quantize_values(const double *rgb, uint8_t *rgb8, uint64_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(quantize_values_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, rgb, rgb8, n);
    return hipGetLastError();
}

hipError_t launch_deinterleave(const double *parts, double *full, uint32_t width, uint32_t height, uint32_t n, uint32_t cap_rows,
                               uint32_t block, hipStream_t stream)
{
    const uint64_t total = (uint64_t)width * height * 3;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(deinterleave_kernel<double>, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, parts, full, width,
                       height, n, cap_rows, block, 0u);
    return hipGetLastError();
}

hipError_t launch_deinterleave_u8(const uint8_t *parts, uint8_t *full, uint32_t width, uint32_t height, uint32_t n, uint32_t cap_rows,
                                  uint32_t block, bool flip, hipStream_t stream)
{
    const uint64_t total = (uint64_t)width * height * 3;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(deinterleave_kernel<uint8_t>, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, parts, full, width,
                       height, n, cap_rows, block, flip ? 1u : 0u);
    return hipGetLastError();
}

#ifdef RTX_LAB
hipError_t launch_debug_math(int op, const double *a, const double *b, double *out, uint64_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(debug_math_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, op, a, b, out, n);
    return hipGetLastError();
}
#endif

}  // namespace rtx
