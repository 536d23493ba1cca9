// rtx_wavefront.h -- what the two wavefront forms (rtx_wavefront.hip: pure triangle meshes; rtx_wavefront_spheres.hip:
// sphere trees) share: the ray state in HBM, the per-level queues, the wave-aggregated append.
#pragma once

#include "rtx_device.h"
#include "rtx_traverse.h"

#include <utility>

namespace rtx {

constexpr uint32_t kWfFallback = 0x80000000u;                     // cand.count flag: no f32 walk could finish this ray (origin out of any range, a full overflow list): the shade kernel tests every shape exactly
constexpr uint32_t kWfDead = 0x40000000u;                         // cand.count flag: no ray in this queue slot (the padding of partial tiles)
constexpr uint32_t kWfExtra = 0x20000000u;                        // cand.count flag: more candidates of this ray are in the level's overflow list
constexpr uint32_t kWfExtraCap = 1u << 16;                        // entries of that list (a handful per frame are used: C3 11 rays, C5 33)
constexpr uint32_t kWfLevelsPerSync = 16;                         // levels enqueued between two looks at the queue length from the host

struct WfRec {                        // 64 bytes: what the f32 walk needs of one segment
    float px, py, pz;                 // origin - scene centre
    float dx, dy, dz;                 // direction
    float ix, iy, iz, nx, ny, nz;     // Ray32: inv = fl(1/d), noi = fl(-o * inv)
    float best_up;                    // the self-hit's distance rounded up, or +inf; NaN: no f32 walk (see kWfFallback)
    float A;                          // tri_filter_from_ray's slack 64uS
    uint32_t ridx;                    // the ray (index in the launch's queue order)
    float slack;                      // Ray32S::e: 0 for an origin inside origin_limit
};
static_assert(sizeof(WfRec) == 64, "WfRec must be 64 bytes");

struct WfCand { uint32_t count; uint32_t e[7]; };                 // 32 bytes per queue position: e[0..5] candidates, e[6] = the ray (ridx)
static_assert(sizeof(WfCand) == 32, "WfCand must be 32 bytes");

// The f64 state of a level's rays, structure-of-arrays in QUEUE order (slot p of the level's queue): the shade kernel reads
// it with unit stride and writes the survivors' to the next level's set at their new slots.  (Indexed by ray instead,
// the survivors thin out level by level and every 8-byte access drags a mostly dead line along: measured 1.4 KB of HBM
// traffic per C3 segment in the shade kernel, which made it HBM-bound.)
struct WfRays {
    double *pos[3], *dir[3], *res[3], *lig[3];
    double *hit_t;                    // the pre-tested self-hit's distance, 0.0 = none            (meshes)
    uint32_t *left;                   // the triangle the ray just left (index in tris[]), kNone = none
};

struct WfState {                      // capacity n (the launch's rays), all on the device
    WfRays ray[2];                    // [0]: the level being processed, [1]: the next level's
    WfRec *rec[2];                    // likewise: the f32 records of the walk
    WfCand *cand;
    uint2 *extra;                     // the level's overflow list: {queue position, candidate} of walks whose LDS queue ran full
    unsigned long long *xcount;       // xcount[0]: its length
    unsigned long long *count;        // count[0]: this level's queue length, count[1]: the next level's (being appended to)
    unsigned long long *work;         // work[0]: the walk kernel's queue head for this level
    uint64_t n;
};

// Atomics on ONE address retire at ~13 ns each on this part whoever issues them (measured: a wave-aggregated append
// per 64 rays made level 0 of C2 take 5 ms for 25M rays), so the queues are touched once per BLOCK iteration when
// appending and once per few hundred records when taking.
//
// Block-aggregated append: every thread of the workgroup calls it the same number of times (`it` = the call's index);
// the threads that `want` get consecutive slots, one atomicAdd per call.  lds: 16 words.
__device__ __forceinline__ unsigned long long wf_append_block(unsigned long long *counter, bool want, uint32_t *lds, uint32_t it)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t *const b = lds + ((it & 1u) << 3);              // two buffers: a fast wave's next call must not overwrite what a slow one still reads
    const unsigned long long m = __ballot(want);
    if (lane == 0) b[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    const uint32_t c0 = b[0], c1 = b[1], c2 = b[2], c3 = b[3];
    if (tid == 0) {
        const uint32_t total = c0 + c1 + c2 + c3;
        const unsigned long long base = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
        b[4] = (uint32_t)base; b[5] = (uint32_t)(base >> 32);
    }
    __syncthreads();
    const unsigned long long base = ((unsigned long long)b[5] << 32) | b[4];
    const uint32_t before = (wave > 0u ? c0 : 0u) + (wave > 1u ? c1 : 0u) + (wave > 2u ? c2 : 0u);
    return base + before + bvh_mbcnt(m);
}

// A walk whose LDS candidate queue is full moves the live entries (t_lo <= best_up) to the level's overflow list and goes
// on with an empty queue; the shade kernel finds them by queue position.  Rare (a ray skimming many triangles none of
// which is a certain hit), so a lone atomic per flush is fine.  Returns false when the list is full (-> kWfFallback).
__device__ __forceinline__ bool wf_flush_to_extra(const WfState &st, uint32_t pos, uint32_t *lds_q, int queue, uint32_t tid,
                                                  uint32_t &qcnt, float best_up)
{
    uint32_t live = 0;
    for (int e = 0; e < queue; ++e)
        if ((uint32_t)e < qcnt && __uint_as_float(lds_q[(size_t)(queue + e) * kBvhThreads + tid]) <= best_up) live += 1;
    const unsigned long long base = atomicAdd(st.xcount, (unsigned long long)live);
    if (base + live > (unsigned long long)kWfExtraCap) {
        // the list is full.  The shade kernel scans entries [0, min(xcount, cap)): what this reservation covers of that range
        // stays unwritten by any flush, so it is marked as nobody's (the list is not cleared between levels or launches: a
        // stale entry could name a live queue position, an unwritten one index a null array)
        for (unsigned long long k = base; k < (unsigned long long)kWfExtraCap && k < base + live; ++k) st.extra[k] = make_uint2(kNone, 0u);
        return false;
    }
    uint32_t k = 0;
    for (int e = 0; e < queue; ++e) {
        if ((uint32_t)e < qcnt && __uint_as_float(lds_q[(size_t)(queue + e) * kBvhThreads + tid]) <= best_up) {
            st.extra[base + k] = make_uint2(pos, lds_q[(size_t)e * kBvhThreads + tid]);
            k += 1;
        }
    }
    qcnt = 0;
    return true;
}

// Taking records: a wave owns a chunk [next, end) of the level's queue (one atomicAdd of `grab` per chunk) and hands its
// records to the lanes that `want` one, in lane order.  Returns whether this lane got one (`my`).
struct WfChunk { unsigned long long next, end; bool drained; };

__device__ __forceinline__ bool wf_take(WfChunk &ch, unsigned long long *head, unsigned long long grab, unsigned long long n_queue,
                                        bool want, unsigned long long &my)
{
    const unsigned long long idle = __ballot(want);
    if (ch.next >= ch.end && !ch.drained) {
        unsigned long long base = 0;
        if ((threadIdx.x & 63u) == 0u) base = atomicAdd(head, grab);
        base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
               __builtin_amdgcn_readfirstlane((uint32_t)base);
        ch.next = base;
        ch.end = base + grab < n_queue ? base + grab : n_queue;
        if (base >= n_queue) { ch.drained = true; ch.next = ch.end = 0; }
    }
    my = ch.next + bvh_mbcnt(idle);
    const bool ok = want && my < ch.end;
    const unsigned long long taken = (unsigned long long)__popcll(idle);
    ch.next = ch.next + taken < ch.end ? ch.next + taken : ch.end;
    return ok;
}

// records a wave takes per atomic: ~8 chunks per wave, so that the last chunks decide little of the level's length
__device__ __forceinline__ unsigned long long wf_grab_size(unsigned long long n_queue)
{
    const unsigned long long g = n_queue / ((unsigned long long)gridDim.x * (blockDim.x >> 6) * 8ull);
    return g > 512ull ? 512ull : (g < 64ull ? 64ull : g);
}

// ---- host ------------------------------------------------------------------------------------------------------------------
// carve the state block of a launch of n rays (every array starts on a 256-byte boundary; wavefront_state_bytes() sizes it)
inline void wf_carve(void *state_mem, uint64_t n, WfState &st)
{
    char *p = static_cast<char *>(state_mem);
    auto take = [&](size_t bytes) { char *q = p; p += (bytes + 255) & ~(size_t)255; return q; };
    for (int s = 0; s < 2; ++s) {
        WfRays &r = st.ray[s];
        for (int k = 0; k < 3; ++k) r.pos[k] = reinterpret_cast<double *>(take(n * sizeof(double)));
        for (int k = 0; k < 3; ++k) r.dir[k] = reinterpret_cast<double *>(take(n * sizeof(double)));
        for (int k = 0; k < 3; ++k) r.res[k] = reinterpret_cast<double *>(take(n * sizeof(double)));
        for (int k = 0; k < 3; ++k) r.lig[k] = reinterpret_cast<double *>(take(n * sizeof(double)));
        r.hit_t = reinterpret_cast<double *>(take(n * sizeof(double)));
        r.left = reinterpret_cast<uint32_t *>(take(n * sizeof(uint32_t)));
    }
    st.rec[0] = reinterpret_cast<WfRec *>(take(n * sizeof(WfRec)));
    st.rec[1] = reinterpret_cast<WfRec *>(take(n * sizeof(WfRec)));
    st.cand = reinterpret_cast<WfCand *>(take(n * sizeof(WfCand)));
    // counters for one chunk of levels at a time (re-zeroed per chunk; the carried-over queue length is copied to slot 0)
    st.count = reinterpret_cast<unsigned long long *>(take((kWfLevelsPerSync + 2) * sizeof(unsigned long long)));
    st.work = reinterpret_cast<unsigned long long *>(take((kWfLevelsPerSync + 2) * sizeof(unsigned long long)));
    st.xcount = reinterpret_cast<unsigned long long *>(take((kWfLevelsPerSync + 2) * sizeof(unsigned long long)));
    st.extra = reinterpret_cast<uint2 *>(take((size_t)kWfExtraCap * sizeof(uint2)));
    st.n = n;
}

// The levels are enqueued without host round trips: every kernel reads its queue length from the device.  A level whose
// queue is empty costs two near-empty launches, so paths that may run for more than kWfLevelsPerSync levels are checked
// from the host every that many levels (max_bounces is 10 by default: one chunk).  generate(st) fills level 0's queue;
// level(sk, l) enqueues the walk and shade kernels of path level l on the state view sk (its count / work / rec
// pointers are the level's own).
template <class Generate, class Level>
inline hipError_t wf_run_levels(WfState st, uint32_t levels, hipStream_t stream, Generate generate, Level level_fn)
{
    const size_t counter_bytes = (kWfLevelsPerSync + 2) * sizeof(unsigned long long);
    hipError_t e = hipMemsetAsync(st.count, 0, counter_bytes, stream);
    if (e == hipSuccess) e = hipMemsetAsync(st.work, 0, counter_bytes, stream);
    if (e == hipSuccess) e = hipMemsetAsync(st.xcount, 0, counter_bytes, stream);
    if (e != hipSuccess) return e;
    if ((e = generate(st)) != hipSuccess) return e;
    uint32_t level = 0;                           // the path's level: draw / bounce indices
    while (level < levels) {
        const uint32_t chunk = levels - level < kWfLevelsPerSync ? levels - level : kWfLevelsPerSync;
        for (uint32_t k = 0; k < chunk; ++k) {
            // counters are indexed by the level's position in the chunk (k); the records by the parity of k as well
            WfState sk = st;
            sk.count = st.count + k; sk.work = st.work + k; sk.xcount = st.xcount + k;
            sk.rec[0] = st.rec[k & 1u]; sk.rec[1] = st.rec[(k + 1u) & 1u];
            sk.ray[0] = st.ray[k & 1u]; sk.ray[1] = st.ray[(k + 1u) & 1u];
            if ((e = level_fn(sk, level + k)) != hipSuccess) return e;
        }
        level += chunk;
        if (level >= levels) break;
        // more levels allowed than one chunk: stop when the queue ran empty, else carry the queue length over
        unsigned long long left = 0;
        if ((e = hipMemcpyAsync(&left, st.count + chunk, sizeof left, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        if (left == 0) break;
        if ((e = hipMemsetAsync(st.count, 0, counter_bytes, stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(st.work, 0, counter_bytes, stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(st.xcount, 0, counter_bytes, stream)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(st.count, &left, sizeof left, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;      // (`left` is a stack variable)
        if (chunk & 1u) { std::swap(st.rec[0], st.rec[1]); std::swap(st.ray[0], st.ray[1]); }                      // the next chunk's level 0 reads what this chunk's last level wrote
    }
    return hipSuccess;
}

}  // namespace rtx
