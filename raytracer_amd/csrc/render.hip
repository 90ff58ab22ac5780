// render.hip — RayTracer (Renderer.h:29) as a software wavefront path tracer for gfx950.
//
// One rt_render() call enqueues, per batch of B samples per pixel, a fixed sequence of persistent kernels
//   generate -> { ctl -> extend -> shade -> connect [-> gather] } x 6 -> accumulate
// on one HIP stream with no host round trip: queue lengths live in a device control block that each kernel
// reads, waves pull 64-item chunks from work counters, live paths are compacted with ballot + prefix
// appends (one atomic per wave per queue), and hits are binned per BSDF class so a shade wave runs one
// material's code.  Path state is one 64-B record per path in HBM (+16 B radiance), the film is a plain fp32 RGB sum.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "rt_shading.h"
#include "scene_dev.h"

#define RT_NCLASS 6
#ifndef RT_TRI_CLS
#define RT_TRI_CLS 1   // ExtendJob::retire files a hit by DScene::tri_cls (0: by the hit's tris record and its material)
#endif
#define RT_FLAG_CANHIT 0x100u
#define RT_FLAG_DEAD 0x200u

// Every counter that kernels hammer with atomics sits in a 128-B line of its own: atomics on one line serialise in its L2
// channel (~11 ns each).  With block-wide appends (one atomic per queue per 512 items) this measured neutral on path
// tracing; it keeps the work-fetch counter off the lines the appends hit.
struct alignas(128) Ctl {
    uint32_t n_active[2];
    uint32_t nf_shadow;   // a NEE request of this bounce carries a non-finite throughput (see k_shade)
    uint32_t pad0_[29];
    uint32_t n_shadow, pad1_[31];
    uint32_t n_gather, pad2_[31];
    uint32_t n_class[8][32];           // [class][0]
    unsigned long long work[4][16];    // [extend, shade, connect, gather][0]
    unsigned long long tot_closest, tot_any, pad3_[14];
    unsigned long long work_x[3][8][16];  // [extend|connect|gather][XCD shard], one 128-B line each
};

struct RBuf {  // device buffers of one renderer
    // path state, one 64-B record per path: [0] ray origin + rng key, [1] ray direction + rng counter, [2] hit (t, u, v, rank),
    // [3] throughput + flags.  After the first bounce paths are reached through the queues (scattered): as four SoA arrays a
    // shade lane touched four 64-B sectors for 64 B of state, as a record it touches one.
    float4* st;
    float4* rad;
    uint32_t* q_active[2];
    uint32_t* q_class;   // RT_NCLASS * cap
    uint32_t* cls;       // per position of the active queue: BSDF class of the hit extend found, RT_CLS_NONE when the path ended
    float4* shadow;      // 3 * cap
    float4* gather;      // 4 * cap (radiosity requests)
    float4* vpl;         // 4 * vpl_cap
    uint32_t cap, vpl_slots;
    uint32_t np_pad, tiles_x, w, h;
    uint32_t q0, nq;   // pixel slots [q0, q0 + nq) of every sample are this launch sequence's (the whole frame: 0, np_pad)
};

__device__ __forceinline__ unsigned long long fetch_chunk(unsigned long long* work) {
    unsigned long long base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(work, 64ULL);
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// A wave takes `grant` consecutive items per atomic.  One contended device-scope atomic costs ~11 ns (MI355X_MICROARCH.md
// "fanin"/"dequeue"), so 64-item grants alone would cap a kernel at ~5.8 G items/s: big launches take 256.  A launch ends
// with its last grants, though, and a 256-item grant is ~0.5 ms of work for one wave: with 2 M items (one sample per pixel)
// that is a third of the launch spent waiting for the waves that drew a second grant.  So the grant shrinks with the
// launch: about RT_GRANTS_PER_WAVE grants per wave, between RT_GRANT_MIN and 256 items.
#ifndef RT_GRANTS_PER_WAVE
#define RT_GRANTS_PER_WAVE 16u
#endif
#ifndef RT_GRANT_MIN
#define RT_GRANT_MIN 64u
#endif
#ifndef RT_XCD_SHARD
#define RT_XCD_SHARD 1
#endif
// XCD-aware work distribution: the item range is cut into 8 contiguous shards, one per XCD (HW_REG_XCC_ID); a wave
// drains its own XCD's shard first and then steals from the others.  Queues keep pixel/path order through
// compaction, so a shard is a spatially coherent part of the frame and its BVH working set stays in that XCD's 4 MB
// L2 instead of being replicated in all eight.  Placement only affects speed, never results.
struct XcdFetch {
    uint32_t xcc, step, per, grant;
    __device__ __forceinline__ void init(uint32_t n) {
        xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;  // HW_REG_XCC_ID[3:0]
        step = 0;
        const uint32_t waves = gridDim.x * (blockDim.x >> 6);
        const uint32_t want = n / (waves * RT_GRANTS_PER_WAVE);
        grant = 256u;
        while (grant > RT_GRANT_MIN && grant > want) grant >>= 1;
        per = (((n + 7u) / 8u + grant - 1u) / grant) * grant;
    }
    // returns false when every shard is exhausted; otherwise [base, end) with base a multiple of the grant
    __device__ __forceinline__ bool next(unsigned long long (*w)[16], uint32_t n, uint32_t& base, uint32_t& end) {
        while (step < 8) {
            uint32_t sh = RT_XCD_SHARD ? ((xcc + step) & 7u) : step;
            unsigned long long lo = (unsigned long long)sh * per, hi = lo + per;
            if (hi > n) hi = n;
            if (lo < hi) {
                unsigned long long b = 0;
                if ((threadIdx.x & 63) == 0) b = atomicAdd(&w[sh][0], (unsigned long long)grant);
                uint32_t b32 = __builtin_amdgcn_readfirstlane((unsigned)b);
                if (lo + b32 < hi) { base = (uint32_t)lo + b32; end = (uint32_t)hi; return true; }
            }
            step++;
        }
        return false;
    }
};
// Block-wide append to NQ queues: per-wave ballots, an LDS scan over the waves, ONE global atomic per queue per
// block iteration.  All threads of the block must call it (it contains barriers).  smem: NQ*(NW+1) words.
template <int NQ, int NW>
__device__ __forceinline__ void block_append(uint32_t* const (&counters)[NQ], const bool (&pred)[NQ], uint32_t (&slot)[NQ], uint32_t* smem) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t prefix[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        unsigned long long mask = __ballot(pred[q]);
        prefix[q] = (uint32_t)__popcll(mask & ((1ULL << lane) - 1ULL));
        if (lane == 0) smem[q * (NW + 1) + wave] = (uint32_t)__popcll(mask);
    }
    __syncthreads();
    if (threadIdx.x < NQ) {
        const int q = threadIdx.x;
        uint32_t sum = 0;
        for (int w = 0; w < NW; w++) { uint32_t c = smem[q * (NW + 1) + w]; smem[q * (NW + 1) + w] = sum; sum += c; }
        smem[q * (NW + 1) + NW] = sum ? atomicAdd(counters[q], sum) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; q++) slot[q] = smem[q * (NW + 1) + NW] + smem[q * (NW + 1) + wave] + prefix[q];
}
// ballot + prefix-sum append: one atomic per wave, slots handed out in lane order
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool pred) {
    unsigned long long mask = __ballot(pred);
    if (mask == 0) return 0;
    uint32_t lane = threadIdx.x & 63;
    uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(mask & ((1ULL << lane) - 1ULL));
}
__device__ __forceinline__ bool slot_pixel(const RBuf& R, uint32_t q, uint32_t& x, uint32_t& y) {
    uint32_t tile = q >> 6, in = q & 63;
    x = (tile % R.tiles_x) * 8 + (in & 7);
    y = (tile / R.tiles_x) * 8 + (in >> 3);
    return x < R.w && y < R.h;
}
__device__ __forceinline__ int class_of(uint32_t type) { return rt_class_of(type); }
static_assert(RT_MAT_MIRROR == 1 && RT_MAT_CONDUCTOR == 2 && RT_MAT_GLASS == 3 && RT_MAT_DIELECTRIC == 4 && RT_MAT_PLASTIC == 6,
              "rt_class_of (rt_common.h) switches on rtamd.h's RT_MAT_* values");
__device__ __forceinline__ f3 mul_point(const float* m, f3 v) {  // Core.h:302
    return mk3((v.x * m[0] + v.y * m[1] + v.z * m[2]) + m[3], (v.x * m[4] + v.y * m[5] + v.z * m[6]) + m[7],
               (v.x * m[8] + v.y * m[9] + v.z * m[10]) + m[11]);
}
__device__ __forceinline__ f3 mul_point_div(const float* m, f3 v) {  // Core.h:310
    f3 a = mul_point(m, v);
    float w = (m[12] * v.x) + (m[13] * v.y) + (m[14] * v.z) + m[15];
    w = 1.0f / w;
    return a * w;
}
__device__ __forceinline__ void camera_ray(const DCamera& c, float x, float y, f3& o, f3& dir) {  // Scene.h:47
    float xc = ((2 * x) / (c.width - 1)) - 1;
    float yc = 1 - ((2 * y) / (c.height - 1));
    f3 pcam = mul_point_div(c.invP, mk3(xc, yc, 0));
    f3 dcam = mul_point_div(c.cam, pcam);
    o = ld3(c.origin);
    dir = dcam - o;
}
__device__ __forceinline__ bool camera_project(const DCamera& c, f3 p, float& x, float& y) {  // Scene.h:65
    f3 pview = mul_point(c.cam2view, p);
    f3 pproj = mul_point_div(c.P, pview);
    x = (pproj.x + 1.0f) * 0.5f;
    y = (pproj.y + 1.0f) * 0.5f;
    if (x < 0 || x > 1.0f || y < 0 || y > 1.0f) return false;
    x = x * c.width;
    y = 1.0f - y;
    y = y * c.height;
    return true;
}

// ------------------------------------------------------------------------------------------------ control
// `first`: the queue is the identity over `first_count` slots of which `first_live` hold a real path
__global__ void k_ctl_reset(Ctl* ctl, int cur, int first, uint32_t first_count, uint32_t first_live) {
    ctl->tot_any += ctl->n_shadow;
    if (first) ctl->n_active[cur] = first_count;
    ctl->tot_closest += first ? first_live : ctl->n_active[cur];
    ctl->n_active[cur ^ 1] = 0;
    for (int i = 0; i < 8; i++) ctl->n_class[i][0] = 0;
    ctl->n_shadow = 0;
    ctl->n_gather = 0;
    ctl->nf_shadow = 0;
    for (int i = 0; i < 4; i++) ctl->work[i][0] = 0;
    for (int k = 0; k < 3; k++) for (int i = 0; i < 8; i++) ctl->work_x[k][i][0] = 0;
}
__global__ void k_ctl_final(Ctl* ctl) {
    ctl->tot_any += ctl->n_shadow;
    ctl->n_shadow = 0;
}

// ------------------------------------------------------------------------------------------------ generate
// pathTrace(px, py, sampler) (Renderer.h:203): jitter in [0,1) added to px = x + 0.5 (Renderer.h:527, :205)
// Four lanes per path, one per 16-B quarter of its 64-B record, so that a wave's store covers 1 KB of consecutive bytes (one
// lane per path wrote 16 B out of every 64: 2.3 times the duration).  The four lanes compute the same camera ray.
__global__ __launch_bounds__(256) void k_generate(DScene S, RBuf R, uint32_t n, uint64_t seed, uint64_t first_sample) {
    const uint32_t total = n * 4u;   // the host keeps n <= 64 M paths
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const uint32_t i = j >> 2, part = j & 3u;
        uint32_t s = i / R.nq, q = R.q0 + (i - s * R.nq);
        uint32_t x, y;
        float4 v = make_float4(0, 0, 0, 0);   // part 2: the hit slot
        if (!slot_pixel(R, q, x, y)) {
            if (part == 3u) v = make_float4(0, 0, 0, __uint_as_float(RT_FLAG_DEAD));
        } else {
            Rng rng; rng.key = rt_rng_key(seed, first_sample + s, y * R.w + x, 0); rng.ctr = 0;
            float px = x + 0.5f, py = y + 0.5f;
            float u1 = rt_rng_next(rng), u2 = rt_rng_next(rng);
            f3 o, d;
            camera_ray(S.cam, px + u1, py + u2, o, d);
            if (part == 0u) { v = make_float4(o.x, o.y, o.z, __uint_as_float(rng.key)); R.rad[i] = make_float4(0, 0, 0, 0); }
            else if (part == 1u) v = make_float4(d.x, d.y, d.z, __uint_as_float(rng.ctr));
            else if (part == 3u) v = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(RT_FLAG_CANHIT));
        }
        R.st[j] = v;
    }
}

// ------------------------------------------------------------------------------------------------ extend
// Scene::traverse for every live path + the two trivial outcomes of pathTrace handled in place:
// escaped rays add background->evaluate(dir) un-weighted (Renderer.h:254) and emitters add T*Le only when
// canHitLight (Renderer.h:221-231).  Everything else is binned by BSDF class for the shade kernel.
#define RT_CLS_NONE 0xFFu
struct XcdSource {   // rt_persistent_walk's Source over the XCD-sharded work counters
    XcdFetch xf;
    unsigned long long (*w)[16];
    uint32_t n;
    __device__ __forceinline__ bool next(uint32_t& begin, uint32_t& end) {
        uint32_t b, e;
        if (!xf.next(w, n, b, e)) return false;
        begin = b;
        end = (e - b) < xf.grant ? e : b + xf.grant;
        return true;
    }
};
template <bool LT>
struct ExtendJob {
    const DScene& S;
    const RBuf& R;
    const uint32_t* queue;
    int first;
    static constexpr uint32_t refill_idle = RT_REFILL_IDLE;
    __device__ __forceinline__ uint32_t path_of(uint32_t i) const { return first ? i : queue[i]; }
    __device__ __forceinline__ bool load(uint32_t i, f3& o, f3& d) {
        uint32_t path = path_of(i);
        if (__float_as_uint(R.st[(size_t)path * 4 + 3].w) & RT_FLAG_DEAD) { R.cls[i] = RT_CLS_NONE; return false; }
        float4 ro = R.st[(size_t)path * 4 + 0], rd = R.st[(size_t)path * 4 + 1];
        o = mk3(ro.x, ro.y, ro.z); d = mk3(rd.x, rd.y, rd.z);
        return true;
    }
    __device__ __forceinline__ void retire(uint32_t i, const Walk& w, bool) {
        const HitRec& h = w.best;
        uint32_t path = path_of(i);
        uint32_t cls = RT_CLS_NONE;
        if (h.t < RT_FLT_MAX) {
#if RT_TRI_CLS
            const uint32_t tc = S.tri_cls[h.rank];   // class | emitter bit: a byte of a 1-B-per-triangle array instead of a line of tris and one of mats
            const bool is_light = (tc & 0x80u) != 0u;
#else
            const DMaterial& m0 = S.mats[S.tris[h.rank].mat_gsign & 0x7FFFFFFFu];
            const uint32_t tc = (uint32_t)class_of(m0.type);
            const bool is_light = (m0.flags & RT_MF_LIGHT) != 0u;
#endif
            if (!LT && is_light) {
                float4 th = R.st[(size_t)path * 4 + 3];
                if (__float_as_uint(th.w) & RT_FLAG_CANHIT) {
                    const DMaterial& m = S.mats[S.tris[h.rank].mat_gsign & 0x7FFFFFFFu];
                    float4 L = R.rad[path];
                    col3 c = mkc(th.x, th.y, th.z) * mat_emission(m);
                    R.rad[path] = make_float4(L.x + c.r, L.y + c.g, L.z + c.b, L.w);
                }
            } else {
                R.st[(size_t)path * 4 + 2] = make_float4(h.t, h.u, h.v, __uint_as_float(h.rank));
                cls = LT ? 0u : (tc & 0x7Fu);
            }
        } else if (!LT) {
            float4 L = R.rad[path];
            float4 rd = R.st[(size_t)path * 4 + 1];  // not w.dir: a zero / untamed direction never initialises the walk
            col3 c = bg_evaluate(S, mk3(rd.x, rd.y, rd.z));
            R.rad[path] = make_float4(L.x + c.r, L.y + c.g, L.z + c.b, L.w);
        }
        R.cls[i] = cls;   // by queue position, not by path: k_bin then streams through it
    }
};
// TAG only gives the launches of a one-stream renderer (rt_renderer_set_streams) a kernel name of their own, so that a
// profiler's per-kernel average keeps the shared (two-stream) and the unshared launch durations apart.
template <bool LT, bool COUNT, int TAG = 0>
__global__ __launch_bounds__(RT_BLOCK, RT_TRAV_WAVES) void k_extend(DScene S, RBuf R, Ctl* ctl, int cur, int first, uint32_t* spillw, uint32_t* spill2,
                                                     DCounters* counters) {
    __shared__ __align__(16) uint32_t lds_stack[RT_STACK_WORDS];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x, gstride = gridDim.x * RT_BLOCK;
    RT_TRAV_STACK_INIT(st, lds_stack, spillw, gtid, gstride)
    const uint32_t n = ctl->n_active[cur];
    uint32_t cn = 0, ct = 0, cx = 0;
    ExtendJob<LT> job{S, R, R.q_active[cur], first};
    XcdSource src; src.xf.init(n); src.w = ctl->work_x[0]; src.n = n;
    rt_persistent_walk<false, COUNT>(S, job, src, 0, st, spill2 + gtid, gstride, cn, ct, cx, counters);
    if (COUNT) {
        atomicAdd(&counters->nodes, (unsigned long long)cn); atomicAdd(&counters->tris, (unsigned long long)ct);
        atomicAdd(&counters->exact, (unsigned long long)cx);
    }
}

// ------------------------------------------------------------------------------------------------ bin
// Material sorting: the hits extend left behind (R.cls per path) are binned into one queue per BSDF class, in
// queue order, so a shade wave runs one material's code.  Block-lockstep: one atomic per class per 512 paths.
#define RT_BIN_BLOCK 512
#ifndef RT_BIN_ITEMS
#define RT_BIN_ITEMS 4   // paths per thread and block iteration: a block moves 2048 paths per atomic and class
#endif
// One contended atomic on a counter costs ~11 ns whatever else the GPU does, so a launch over n paths cannot take less than
// n / (paths per atomic) x 11 ns per class: with 512 paths per atomic that floor (0.25 ms at 12 M paths) was the kernel's
// whole duration (PMC: VALU 0.04, TA 0.18 busy).  Slots keep queue order inside the block: (item, wave, lane).
__global__ __launch_bounds__(RT_BIN_BLOCK) void k_bin(RBuf R, Ctl* ctl, int cur, int first) {
    constexpr int NW = RT_BIN_BLOCK / 64, NV = NW * RT_BIN_ITEMS;   // waves, and (item, wave) pairs
    __shared__ uint32_t s_app[RT_NCLASS * (NV + 1)];
    const uint32_t n = ctl->n_active[cur];
    const uint32_t* queue = R.q_active[cur];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ULL << lane) - 1ULL;
    for (uint32_t base = blockIdx.x * (RT_BIN_BLOCK * RT_BIN_ITEMS); base < n; base += gridDim.x * (RT_BIN_BLOCK * RT_BIN_ITEMS)) {
        uint32_t path[RT_BIN_ITEMS], c[RT_BIN_ITEMS], prefix[RT_BIN_ITEMS];
#pragma unroll
        for (int k = 0; k < RT_BIN_ITEMS; k++) {
            const uint32_t i = base + (uint32_t)k * RT_BIN_BLOCK + threadIdx.x;
            path[k] = 0; c[k] = RT_CLS_NONE; prefix[k] = 0;
            if (i < n) { path[k] = first ? i : queue[i]; c[k] = R.cls[i]; }
        }
#pragma unroll
        for (int k = 0; k < RT_BIN_ITEMS; k++)
#pragma unroll
            for (int q = 0; q < RT_NCLASS; q++) {
                const unsigned long long m = __ballot(c[k] == (uint32_t)q);
                if (c[k] == (uint32_t)q) prefix[k] = (uint32_t)__popcll(m & lt);
                if (lane == 0) s_app[q * (NV + 1) + k * NW + (int)wave] = (uint32_t)__popcll(m);
            }
        __syncthreads();
        if (threadIdx.x < RT_NCLASS) {
            uint32_t* t = s_app + threadIdx.x * (NV + 1);
            uint32_t sum = 0;
            for (int v = 0; v < NV; v++) { const uint32_t cnt = t[v]; t[v] = sum; sum += cnt; }
            t[NV] = sum ? atomicAdd(&ctl->n_class[threadIdx.x][0], sum) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RT_BIN_ITEMS; k++)
            if (c[k] < (uint32_t)RT_NCLASS) {
                const uint32_t* t = s_app + c[k] * (NV + 1);
                R.q_class[(size_t)c[k] * R.cap + t[NV] + t[k * NW + (int)wave] + prefix[k]] = path[k];
            }
        __syncthreads();  // s_app is reused by the next iteration
    }
}

// ------------------------------------------------------------------------------------------------ shade
// Body of pathTrace after the hit (Renderer.h:232-252): NEE request (computeDirect, Renderer.h:155), depth
// test, Russian roulette, BSDF sample, throughput update, next ray.  RADIOSITY adds the VPL gather request of
// pathTraceRadiosity (Renderer.h:931).
#ifndef RT_SHADE_BLOCK
#define RT_SHADE_BLOCK 512
#endif
// ENV: the scene's light list holds an EnvironmentMap (its tabulated sampling costs 19 VGPRs; scenes without one run the lean instantiation)
template <bool RADIOSITY, bool ENV>
#ifndef RT_SHADE_OCC
#define RT_SHADE_OCC 2   // blocks of 512 per CU the register budget is set for
#endif
__global__ __launch_bounds__(RT_SHADE_BLOCK, RT_SHADE_OCC) void k_shade(DScene S, RBuf R, Ctl* ctl, int cur, int direct_only) {
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_app[3 * (RT_SHADE_BLOCK / 64 + 1)];
    uint32_t cnt[RT_NCLASS], total = 0;
#pragma unroll
    for (int c = 0; c < RT_NCLASS; c++) { cnt[c] = ctl->n_class[c][0]; total += cnt[c]; }
    for (;;) {
        // the block moves in lockstep: one work atomic and one atomic per output queue per 512 shading points
        if (threadIdx.x == 0) s_base = atomicAdd(&ctl->work[1][0], (unsigned long long)RT_SHADE_BLOCK);
        __syncthreads();
        unsigned long long base = s_base;
        if (base >= total) break;
        uint32_t g = (uint32_t)base + threadIdx.x;
        bool live = g < total;
        bool want_next = false, want_shadow = false, want_gather = false;
        uint32_t path = 0;
        float4 sh_a, sh_b, sh_c, ga, gb, gc, gd;
        if (live) {
            int c = 0; uint32_t idx = g;
#pragma unroll
            for (int k = 0; k < RT_NCLASS - 1; k++) if (c == k && idx >= cnt[k]) { idx -= cnt[k]; c = k + 1; }
            path = R.q_class[(size_t)c * R.cap + idx];
            float4 ro = R.st[(size_t)path * 4 + 0], rd = R.st[(size_t)path * 4 + 1], hh = R.st[(size_t)path * 4 + 2], th = R.st[(size_t)path * 4 + 3];
            f3 o = mk3(ro.x, ro.y, ro.z), d = mk3(rd.x, rd.y, rd.z);
            Rng rng; rng.key = __float_as_uint(ro.w); rng.ctr = __float_as_uint(rd.w);
            uint32_t flags = __float_as_uint(th.w);
            uint32_t depth = flags & 0xFFu;
            col3 T = mkc(th.x, th.y, th.z);
            Shading sd = shading_data(S, o, d, hh.x, hh.y, hh.z, __float_as_uint(hh.w));
            const DMaterial& m = S.mats[sd.mat];
            const bool specular = (m.flags & RT_MF_SPECULAR) != 0;
            // Upstream always forms pathThroughput * computeDirect(...) (Renderer.h:233), and computeDirect returns
            // Colour(0) for specular hits, G <= 0 and occluded samples: a channel of T that is inf/NaN (the rough
            // dielectric's half-vector draw, Materials.h:571, produces them) then turns the pixel NaN whatever the
            // light sample does.  nfb = those channels; 0 for every ordinary path.
            const uint32_t nfb = ((T.r - T.r) != 0.0f ? 1u : 0u) | ((T.g - T.g) != 0.0f ? 2u : 0u) | ((T.b - T.b) != 0.0f ? 4u : 0u);
            // --- computeDirect
            if (!specular && S.n_lights > 0) {
                float pmf = 1.f / S.n_lights;
                int li = (int)(S.n_lights * rt_rng_next(rng));
                li = li < (int)S.n_lights - 1 ? li : (int)S.n_lights - 1;
                DLight light = S.lights[li];
                float r1 = rt_rng_next(rng), r2 = rt_rng_next(rng);
                if (light.kind == 0) {
                    const TriRows lt = tri_rows(S.tris + light.tri);
                    col3 emitted = mat_emission(S.mats[tri_mat_gsign(lt) & 0x7FFFFFFFu]);
                    f3 p = tri_sample_point(lt, r1, r2);
                    float pdf = 1 / light.area;
                    f3 wi = p - sd.x;
                    float l = lensq3(wi);
                    wi = norm3(wi);
                    float G = (smaxf_w(dot3(wi, sd.sN), 0.0f) * smaxf_w(-dot3(wi, tri_gnormal(lt)), 0.0f)) / l;
                    if (G > 0) {
                        col3 cd = ((bsdf_eval(S, sd, wi) * emitted) * G) / (pmf * pdf);
                        col3 cc = T * cd;
                        want_shadow = true;
                        sh_a = make_float4(sd.x.x, sd.x.y, sd.x.z, __uint_as_float(path));
                        sh_b = make_float4(p.x, p.y, p.z, 0);
                        sh_c = make_float4(cc.r, cc.g, cc.b, __uint_as_float(nfb));
                    }
                } else if (light.kind == 1) {
                    f3 wi = uniform_sample_sphere(r1, r2);
                    float pdf = uniform_sphere_pdf();
                    col3 emitted = mkc(S.bg_colour[0], S.bg_colour[1], S.bg_colour[2]);
                    float G = smaxf_w(dot3(wi, sd.sN), 0.0f);
                    if (G > 0) {
                        col3 cd = ((bsdf_eval(S, sd, wi) * emitted) * G) / (pmf * pdf);
                        col3 cc = T * cd;
                        f3 p2 = sd.x + (wi * 10000.0f);
                        want_shadow = true;
                        sh_a = make_float4(sd.x.x, sd.x.y, sd.x.z, __uint_as_float(path));
                        sh_b = make_float4(p2.x, p2.y, p2.z, 0);
                        sh_c = make_float4(cc.r, cc.g, cc.b, __uint_as_float(nfb));
                    }
                }
                else if (ENV) {   // EnvironmentMap::sample (Lights.h:385-399): tabulated draw, read through Vec2::operator[]'s int (env_sample_dir)
                    float pdf;
                    f3 wi = env_sample_dir(S, r1, r2, pdf);
                    col3 emitted = bg_evaluate(S, wi);
                    float G = smaxf_w(dot3(wi, sd.sN), 0.0f);
                    if (G > 0) {
                        col3 cd = ((bsdf_eval(S, sd, wi) * emitted) * G) / (pmf * pdf);
                        col3 cc = T * cd;
                        // pdf is +inf for every draw whose table coordinates truncate to 0, i.e. the term is exactly 0 whether or not
                        // Scene::visible finds the sky: such a request is not queued (upstream traces it and adds 0)
                        if (!(cc.r == 0.0f && cc.g == 0.0f && cc.b == 0.0f)) {
                            f3 p2 = sd.x + (wi * 10000.0f);
                            want_shadow = true;
                            sh_a = make_float4(sd.x.x, sd.x.y, sd.x.z, __uint_as_float(path));
                            sh_b = make_float4(p2.x, p2.y, p2.z, 0);
                            sh_c = make_float4(cc.r, cc.g, cc.b, __uint_as_float(nfb));
                        }
                    }
                }
            }
            if (nfb) {
                if (want_shadow) atomicOr(&ctl->nf_shadow, 1u);   // k_connect adds T * 0 when the sample is occluded
                else { float4 L = R.rad[path]; R.rad[path] = make_float4(L.x + T.r * 0.0f, L.y + T.g * 0.0f, L.z + T.b * 0.0f, L.w); }
            }
            if (RADIOSITY && !specular) {
                want_gather = true;
                ga = make_float4(sd.x.x, sd.x.y, sd.x.z, __uint_as_float(path));
                gb = make_float4(sd.sN.x, sd.sN.y, sd.sN.z, sd.tu);
                gc = make_float4(sd.wo.x, sd.wo.y, sd.wo.z, sd.tv);
                gd = make_float4(T.r, T.g, T.b, __uint_as_float(sd.mat));
            }
            if (!direct_only && !(depth > RT_MAX_DEPTH)) {
                float lum = lumc(T);
                float rr = lum < 0.9f ? lum : 0.9f;  // min(pathThroughput.Lum(), 0.9f) (Renderer.h:237)
                if (rt_rng_next(rng) < rr) {
                    T = T / rr;
                    col3 f; float pdf;
                    f3 wi = bsdf_sample(S, sd, rng, f, pdf);
                    T = ((T * f) * fabsf(dot3(wi, sd.sN))) / pdf;
                    f3 no = sd.x + (wi * RT_EPSILON);
                    R.st[(size_t)path * 4 + 0] = make_float4(no.x, no.y, no.z, ro.w);
                    R.st[(size_t)path * 4 + 1] = make_float4(wi.x, wi.y, wi.z, __uint_as_float(rng.ctr));
                    R.st[(size_t)path * 4 + 3] = make_float4(T.r, T.g, T.b, __uint_as_float((depth + 1) | (specular ? RT_FLAG_CANHIT : 0u)));
                    want_next = true;
                }
            }
        }
        uint32_t* const counters[3] = { &ctl->n_shadow, &ctl->n_active[cur ^ 1], &ctl->n_gather };
        const bool pred[3] = { want_shadow, want_next, RADIOSITY && want_gather };
        uint32_t slot[3];
        block_append<3, RT_SHADE_BLOCK / 64>(counters, pred, slot, s_app);
        if (want_shadow) { float4* e = R.shadow + (size_t)slot[0] * 3; e[0] = sh_a; e[1] = sh_b; e[2] = sh_c; }
        if (want_next) R.q_active[cur ^ 1][slot[1]] = path;
        if (RADIOSITY && want_gather) { float4* e = R.gather + (size_t)slot[2] * 4; e[0] = ga; e[1] = gb; e[2] = gc; e[3] = gd; }
    }
}

// ------------------------------------------------------------------------------------------------ connect
// Scene::visible for every NEE request; visible ones add their (already throughput-weighted) term.
struct ConnectJob {
    const RBuf& R;
    const bool nf;   // some request of this launch has a non-finite throughput channel (Ctl::nf_shadow)
    static constexpr uint32_t refill_idle = RT_REFILL_IDLE_ANY;
    __device__ __forceinline__ bool load(uint32_t i, f3& p1, f3& p2) {
        const float4* e = R.shadow + (size_t)i * 3;
        float4 a = e[0], b = e[1];
        p1 = mk3(a.x, a.y, a.z); p2 = mk3(b.x, b.y, b.z);
        return true;
    }
    __device__ __forceinline__ void retire(uint32_t i, const Walk&, bool visible) {
        const float4* e = R.shadow + (size_t)i * 3;
        if (!visible) {
            if (!nf) return;
            const uint32_t bits = __float_as_uint(e[2].w);   // occluded: T * Colour(0), NaN in T's non-finite channels
            if (!bits) return;
            const uint32_t path = __float_as_uint(e[0].w);
            const float nan = __uint_as_float(0x7FC00000u);
            float4 L = R.rad[path];
            R.rad[path] = make_float4(L.x + ((bits & 1u) ? nan : 0.0f), L.y + ((bits & 2u) ? nan : 0.0f), L.z + ((bits & 4u) ? nan : 0.0f), L.w);
            return;
        }
        uint32_t path = __float_as_uint(e[0].w);
        float4 c = e[2];
        float4 L = R.rad[path];
        R.rad[path] = make_float4(L.x + c.x, L.y + c.y, L.z + c.z, L.w);
    }
};
template <bool COUNT>
__global__ __launch_bounds__(RT_BLOCK, RT_TRAV_WAVES) void k_connect(DScene S, RBuf R, Ctl* ctl, uint32_t* spillw, uint32_t* spill2, DCounters* counters) {
    __shared__ __align__(16) uint32_t lds_stack[RT_STACK_WORDS];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x, gstride = gridDim.x * RT_BLOCK;
    RT_TRAV_STACK_INIT(st, lds_stack, spillw, gtid, gstride)
    const uint32_t n = ctl->n_shadow;
    uint32_t cn = 0, ct = 0, cx = 0;
    ConnectJob job{R, ctl->nf_shadow != 0};
    XcdSource src; src.xf.init(n); src.w = ctl->work_x[1]; src.n = n;
    rt_persistent_walk<true, COUNT>(S, job, src, 0, st, spill2 + gtid, gstride, cn, ct, cx, counters);
    if (COUNT) {
        atomicAdd(&counters->any_nodes, (unsigned long long)cn); atomicAdd(&counters->any_tris, (unsigned long long)ct);
        atomicAdd(&counters->exact, (unsigned long long)cx);
    }
}

// ------------------------------------------------------------------------------------------------ film filters
// ImageFilter (Imaging.h:131-214) as Film::splat uses it (Imaging.h:224-247): weights over the (2 size + 1)^2 pixels
// around (int)x, (int)y, normalised by their in-bounds sum.  kind 0 = BoxFilter, size 0: upstream's default
// (Renderer.h:61) and the fast path everywhere.  The Gaussian receives its offsets truncated to int (gauss(int d, ..)),
// so it is a table over d; Mitchell-Netravali keeps upstream's branch conditions (the first polynomial only at x == 0,
// the second for 0 < x <= 1, no value for 1 < x < 2: 0 here) and its double-precision promotions.
struct DFilter {
    int kind;
    float g[4];                            // GaussianFilter::gauss(d) for d = -2, -1, 0, 1
    float a1, b1, c1, a2, b2, c2, d2;      // MitchellNetravaliFilter coefficients (Imaging.h:185-191)
    float w[25], wtot;                     // weights at pixel-centre offsets (j - 0.5, i - 0.5), row-major (i, j), and their full sum
};
__device__ __forceinline__ float filter_mn(const DFilter& F, float y) {
    const float x = fabsf(y);
    const double xd = (double)x;
    if (x <= 0.0f) return (float)((((double)F.a1 * (xd * xd * xd)) + ((double)F.b1 * (xd * xd)) + (double)F.c1) / 6);
    if (x <= 1.0f) return (float)((((double)F.a2 * (xd * xd * xd)) + ((double)F.b2 * (xd * xd)) + (double)(F.c2 * x) + (double)F.d2) / 6);
    return 0.0f;
}
__device__ __forceinline__ float filter_1d(const DFilter& F, float x) {   // filter(x, y) = filter_1d(x) * filter_1d(y) for both
    if (F.kind == 1) {
        int d = (int)x + 2;   // offsets lie in (-size - 1, size]
        d = d < 0 ? 0 : (d > 3 ? 3 : d);
        return F.g[d];
    }
    return filter_mn(F, x);
}
template <int SIZE>
__device__ __forceinline__ void film_splat_n(float* film, const DCamera& cam, const DFilter& F, float x, float y, col3 c) {
    const int bx = (int)x, by = (int)y;
    float wx[2 * SIZE + 1], wy[2 * SIZE + 1];
#pragma unroll
    for (int j = -SIZE; j <= SIZE; j++) { wx[j + SIZE] = filter_1d(F, (float)(bx + j) - x); wy[j + SIZE] = filter_1d(F, (float)(by + j) - y); }
    float total = 0.0f;
#pragma unroll
    for (int i = -SIZE; i <= SIZE; i++)
#pragma unroll
        for (int j = -SIZE; j <= SIZE; j++) {
            const int px = bx + j, py = by + i;
            if (px >= 0 && px < cam.iw && py >= 0 && py < cam.ih) total += wx[j + SIZE] * wy[i + SIZE];
        }
#pragma unroll
    for (int i = -SIZE; i <= SIZE; i++)
#pragma unroll
        for (int j = -SIZE; j <= SIZE; j++) {
            const int px = bx + j, py = by + i;
            if (px >= 0 && px < cam.iw && py >= 0 && py < cam.ih) {
                const float w = wx[j + SIZE] * wy[i + SIZE];
                float* f = film + ((size_t)py * cam.iw + px) * 3;
                const float vr = (c.r * w) / total, vg = (c.g * w) / total, vb = (c.b * w) / total;
                // adding +-0 changes nothing (most Mitchell-Netravali weights are exactly 0); NaN != 0, so those still land
                if (vr != 0.0f) atomicAdd(f + 0, vr);
                if (vg != 0.0f) atomicAdd(f + 1, vg);
                if (vb != 0.0f) atomicAdd(f + 2, vb);
            }
        }
}
// Film::splat with float atomics (light tracing, PSSMLT)
__device__ __forceinline__ void film_splat(float* film, const DCamera& cam, const DFilter& F, float x, float y, col3 c) {
    if (F.kind == 0) {
        const int bx = (int)x, by = (int)y;
        if (bx >= 0 && bx < cam.iw && by >= 0 && by < cam.ih) {
            float* f = film + ((size_t)by * cam.iw + bx) * 3;
            atomicAdd(f + 0, c.r); atomicAdd(f + 1, c.g); atomicAdd(f + 2, c.b);
        }
    } else if (F.kind == 1) film_splat_n<1>(film, cam, F, x, y, c);
    else film_splat_n<2>(film, cam, F, x, y, c);
}

// connectToCamera (Renderer.h:669-692): projection, visibility to the lens, We = 1/(Afilm cos^4), atomic splat
template <bool FILTERED>   // false: BoxFilter only (the filter code stays out of the persistent kernel's register budget)
struct CameraJob {
    const DScene& S;
    const RBuf& R;
    float* film;
    const DFilter& F;
    static constexpr uint32_t refill_idle = RT_REFILL_IDLE;
    __device__ __forceinline__ bool load(uint32_t i, f3& p1, f3& p2) {
        float4 a = R.shadow[(size_t)i * 3];
        float x, y;
        p1 = mk3(a.x, a.y, a.z);
        if (!camera_project(S.cam, p1, x, y)) return false;
        p2 = ld3(S.cam.origin);
        return true;
    }
    __device__ __forceinline__ void retire(uint32_t i, const Walk&, bool visible) {
        if (!visible) return;
        const DCamera& cam = S.cam;
        const float4* e = R.shadow + (size_t)i * 3;
        float4 a = e[0];
        f3 p = mk3(a.x, a.y, a.z), org = ld3(cam.origin);
        float x, y;
        if (!camera_project(cam, p, x, y)) return;
        f3 cd = norm3(p - org);
        float costheta = dot3(ld3(cam.view_dir), cd);
        if (costheta <= 0) return;
        // std::pow(costheta, 4) (Renderer.h:688) is pow(double, double).  A float's square is exact in double (48 bits) and the
        // square of that rounds once: the correctly rounded cos^4, which is what glibc's pow returns -- without the device
        // library's pow in a traversal kernel (it cost k_connect_camera 124-188 B of scratch)
        const double c2 = (double)costheta * (double)costheta;
        float we = (float)(1.f / (cam.Afilm * (c2 * c2)));
        float4 c = e[2];
        if (FILTERED) { film_splat(film, cam, F, x, y, mkc(c.x * we, c.y * we, c.z * we)); return; }
        int px = (int)x, py = (int)y;  // Film::splat with BoxFilter (Imaging.h:224)
        if (px >= 0 && px < cam.iw && py >= 0 && py < cam.ih) {
            float* f = film + ((size_t)py * cam.iw + px) * 3;
            atomicAdd(f + 0, c.x * we); atomicAdd(f + 1, c.y * we); atomicAdd(f + 2, c.z * we);
        }
    }
};
template <bool FILTERED>
__global__ __launch_bounds__(RT_BLOCK, RT_TRAV_WAVES) void k_connect_camera(DScene S, RBuf R, Ctl* ctl, float* film, DFilter F, uint32_t* spillw, uint32_t* spill2, DCounters* counters) {
    __shared__ __align__(16) uint32_t lds_stack[RT_STACK_WORDS];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x, gstride = gridDim.x * RT_BLOCK;
    RT_TRAV_STACK_INIT(st, lds_stack, spillw, gtid, gstride)
    const uint32_t n = ctl->n_shadow;
    uint32_t cn = 0, ct = 0, cx = 0;
    CameraJob<FILTERED> job{S, R, film, F};
    XcdSource src; src.xf.init(n); src.w = ctl->work_x[1]; src.n = n;
    rt_persistent_walk<true, false>(S, job, src, 0, st, spill2 + gtid, gstride, cn, ct, cx, counters);
}

// ------------------------------------------------------------------------------------------------ accumulate
// film->splat(px, py, col) with the size-0 box filter: film[(int)py * W + (int)px] += col (Imaging.h:224, Renderer.h:535)
__global__ __launch_bounds__(256) void k_accumulate(RBuf R, float* film, uint32_t batch) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < R.nq; q += gridDim.x * blockDim.x) {
        uint32_t x, y;
        if (!slot_pixel(R, R.q0 + q, x, y)) continue;
        float* f = film + ((size_t)y * R.w + x) * 3;
        float r = f[0], g = f[1], b = f[2];
        for (uint32_t s = 0; s < batch; s++) {
            float4 L = R.rad[(size_t)s * R.nq + q];
            r = r + L.x; g = g + L.y; b = b + L.z;
        }
        f[0] = r; f[1] = g; f[2] = b;
    }
}

// The same with a Gaussian / Mitchell-Netravali film: every sample is splatted at its pixel centre (Renderer.h:527-535), so
// the weights are one (2 size + 1)^2 table, and a pixel can gather what its neighbours' samples give it -- no atomics, fixed
// order.  A source pixel near the border divides by the sum of its in-bounds weights, in upstream's loop order.
__global__ __launch_bounds__(256) void k_accumulate_filtered(RBuf R, float* film, uint32_t batch, DFilter F) {
    const int size = F.kind == 1 ? 1 : 2, span = 2 * size + 1;
    const int W = (int)R.w, H = (int)R.h;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < R.np_pad; q += gridDim.x * blockDim.x) {
        uint32_t x, y;
        if (!slot_pixel(R, q, x, y)) continue;
        float* f = film + ((size_t)y * R.w + x) * 3;
        float r = f[0], g = f[1], b = f[2];
        for (int i = -size; i <= size; i++)
            for (int j = -size; j <= size; j++) {
                const int sx = (int)x - j, sy = (int)y - i;   // the source whose footprint offset (j, i) lands here
                if (sx < 0 || sx >= W || sy < 0 || sy >= H) continue;
                const float w = F.w[(i + size) * span + (j + size)];
                float total = F.wtot;
                if (sx < size || sx + size >= W || sy < size || sy + size >= H) {
                    total = 0.0f;
                    for (int i2 = -size; i2 <= size; i2++)
                        for (int j2 = -size; j2 <= size; j2++) {
                            const int px = sx + j2, py = sy + i2;
                            if (px >= 0 && px < W && py >= 0 && py < H) total += F.w[(i2 + size) * span + (j2 + size)];
                        }
                }
                const uint32_t sq = (((uint32_t)sy >> 3) * R.tiles_x + ((uint32_t)sx >> 3)) * 64u + (((uint32_t)sy & 7u) << 3) + ((uint32_t)sx & 7u);
                for (uint32_t s = 0; s < batch; s++) {
                    float4 L = R.rad[(size_t)s * R.np_pad + sq];
                    r = r + (L.x * w) / total; g = g + (L.y * w) / total; b = b + (L.z * w) / total;
                }
            }
        f[0] = r; f[1] = g; f[2] = b;
    }
}

// ------------------------------------------------------------------------------------------------ PSSMLT
// PSSMLTRender (Renderer.h:1013-1063) as upstream's sampler actually behaves (Sampling.h:60-67, :85-94, :118-139):
// PSS::next never re-reads a perturbed coordinate (currentIndex only resets on a large step, which also clears the
// vector), and pathTrace draws from PSSSampler::random, so every proposal is a fresh uniform (x, y) plus an
// independent path.  That makes one spp = 100 paths for b + 1 initial state + 100 000 proposals that can all be
// traced at once by the wavefront path tracer, followed by upstream's sequential accept/splat loop over their
// radiances -- one chain per spp, one wave per chain (k_mlt_chain).  Path p = chain * RT_MLT_PATHS + index (a wave reads
// 64 consecutive proposals); numbers: stream 2 / pixel = index for the path, stream 4 for the chain's two draws per proposal.
#define RT_MLT_M 100u             // Renderer.h:975
#define RT_MLT_PROPOSALS 100000u  // Renderer.h:357
#define RT_MLT_PATHS (RT_MLT_M + 1u + RT_MLT_PROPOSALS)
__global__ __launch_bounds__(256) void k_mlt_generate(DScene S, RBuf R, uint32_t n, uint32_t chains, uint64_t seed, uint64_t first_sample) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t c = i / RT_MLT_PATHS, index = i - c * RT_MLT_PATHS;
        Rng rng; rng.key = rt_rng_key(seed, first_sample + c, index, 2); rng.ctr = 0;
        float x = rt_rng_next(rng) * S.cam.width;   // genPathC / genPathP (Renderer.h:978-992)
        float y = rt_rng_next(rng) * S.cam.height;
        float u1 = rt_rng_next(rng), u2 = rt_rng_next(rng);
        f3 o, d;
        camera_ray(S.cam, x + u1, y + u2, o, d);    // pathTrace(px, py, sampler) (Renderer.h:203-207)
        R.st[(size_t)i * 4 + 0] = make_float4(o.x, o.y, o.z, __uint_as_float(rng.key));
        R.st[(size_t)i * 4 + 1] = make_float4(d.x, d.y, d.z, __uint_as_float(rng.ctr));
        R.st[(size_t)i * 4 + 3] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(RT_FLAG_CANHIT));
        R.rad[i] = make_float4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ float mlt_acceptance(col3 current, col3 proposed) {  // calculateAcceptance (Renderer.h:965-973)
    float cl = lumc(current), pl = lumc(proposed);
    if (cl <= 0.0f) return 1.0f;
    if (pl <= 0.0f) return 0.0f;
    float r = pl / cl;
    return (r < 1.0f) ? r : 1.0f;  // (std::min)(1.0f, r)
}
__device__ __forceinline__ void mlt_xy(const DScene& S, uint64_t seed, uint64_t sample, uint32_t index, float& x, float& y) {
    Rng rng; rng.key = rt_rng_key(seed, sample, index, 2); rng.ctr = 0;
    x = rt_rng_next(rng) * S.cam.width;
    y = rt_rng_next(rng) * S.cam.height;
}
// The accept/reject loop is sequential only in "which proposal is the current state": a proposal's radiance, film
// position and the chain's two draws for step i (counter-based: draws 2i and 2i+1 of stream 4) do not depend on it.  A wave
// therefore takes 64 consecutive steps at once: every lane evaluates its step against the wave's current state, the first
// accepting lane ends the run that state is valid for, becomes the new state, and the lanes behind it re-evaluate -- one
// round per accepted proposal instead of one trip through memory, RNG and atomics per step.
__global__ __launch_bounds__(64) void k_mlt_chain(DScene S, RBuf R, float* film, DFilter F, uint32_t chains, uint64_t seed, uint64_t first_sample) {
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    if (c >= chains) return;
    const uint64_t sample = first_sample + c;
    const float4* rad = R.rad + (size_t)c * RT_MLT_PATHS;
    const float largeStepProb = 0.3f;  // Renderer.h:1010
    float b = 0.0f;                    // getNormalization (Renderer.h:995-1006): summed in index order
#pragma unroll 10
    for (uint32_t i = 0; i < RT_MLT_M; i++) {
        float4 L = rad[i];
        b += lumc(mkc(L.x, L.y, L.z));
    }
    b /= (float)RT_MLT_M;
    if (b <= 0.0f) b = 1.0f;
    float4 L0 = rad[RT_MLT_M];
    col3 current = mkc(L0.x, L0.y, L0.z);   // wave-uniform state
    float cx, cy;
    mlt_xy(S, seed, sample, RT_MLT_M, cx, cy);
    const uint32_t chain_key = rt_rng_key(seed, sample, 0, 4);
    for (uint32_t i0 = 0; i0 < RT_MLT_PROPOSALS; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool valid = i < RT_MLT_PROPOSALS;
        float4 L = valid ? rad[RT_MLT_M + 1u + i] : make_float4(0, 0, 0, 0);
        const col3 proposed = mkc(L.x, L.y, L.z);
        Rng chain; chain.key = chain_key; chain.ctr = 2u * i;
        const bool largeStep = rt_rng_next(chain) < largeStepProb;
        const float u_accept = rt_rng_next(chain);
        float px, py;
        mlt_xy(S, seed, sample, RT_MLT_M + 1u + i, px, py);
        unsigned long long pending = __ballot(valid);
        float a = 0.0f, scx = cx, scy = cy;
        col3 scur = current;   // the state this lane's step sees
        while (pending) {
            const float aj = mlt_acceptance(current, proposed);
            const unsigned long long accept = __ballot(u_accept < aj) & pending;
            const uint32_t first = accept ? (uint32_t)__ffsll((long long)accept) - 1u : 63u;
            if (((pending >> lane) & 1ULL) && lane <= first) { a = aj; scur = current; scx = cx; scy = cy; }
            if (!accept) break;
            current = mkc(__shfl(proposed.r, first), __shfl(proposed.g, first), __shfl(proposed.b, first));
            cx = __shfl(px, first); cy = __shfl(py, first);
            pending = first == 63u ? 0ULL : (pending & ~((2ULL << first) - 1ULL));
        }
        if (valid) {
            if (largeStep) film_splat(film, S.cam, F, px, py, (proposed * a) / (b + largeStepProb * lumc(proposed)));
            else film_splat(film, S.cam, F, px, py, (proposed * a) / b);
            film_splat(film, S.cam, F, scx, scy, (scur * (1.0f - a)) / b);
        }
    }
}

// ------------------------------------------------------------------------------------------------ AOV / direct
// albedo(), viewNormals(), direct() (Renderer.h:257-296): one ray through the pixel centre (Renderer.h:590).
__global__ __launch_bounds__(RT_BLOCK) void k_aov(DScene S, RBuf R, Ctl* ctl, int mode, uint64_t seed, uint64_t first_sample, uint32_t n,
                                                  uint32_t* spillw, uint32_t* spill2) {
    __shared__ __align__(16) uint32_t lds_stack[RT_STACK_WORDS];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x, gstride = gridDim.x * RT_BLOCK;
    RT_TRAV_STACK_INIT(st, lds_stack, spillw, gtid, gstride)
    uint32_t cn = 0, ct = 0, cx = 0;
    for (;;) {
        unsigned long long base = fetch_chunk(&ctl->work[0][0]);
        if (base >= n) break;
        uint32_t i = (uint32_t)base + (threadIdx.x & 63);
        bool want_shadow = false;
        float4 sh_a, sh_b, sh_c;
        if (i < n) {
            uint32_t s = i / R.np_pad, q = i - s * R.np_pad, x, y;
            col3 out = mkc(0, 0, 0);
            if (slot_pixel(R, q, x, y)) {
                Rng rng; rng.key = rt_rng_key(seed, first_sample + s, y * R.w + x, 0); rng.ctr = 0;
                f3 o, d;
                camera_ray(S.cam, x + 0.5f, y + 0.5f, o, d);
                HitRec h;
                rt_closest<false>(S, o, d, 0, h, st, spill2 + gtid, gstride, cn, ct, cx);
                if (h.t < RT_FLT_MAX) {
                    Shading sd = shading_data(S, o, d, h.t, h.u, h.v, h.rank);
                    const DMaterial& m = S.mats[sd.mat];
                    if (mode == RT_MODE_AOV_NORMAL) out = mkc(fabsf(sd.sN.x), fabsf(sd.sN.y), fabsf(sd.sN.z));
                    else if (m.flags & RT_MF_LIGHT) out = mat_emission(m);
                    else if (mode == RT_MODE_AOV_ALBEDO) out = bsdf_eval(S, sd, mk3(0, 1, 0));
                    else if (!(m.flags & RT_MF_SPECULAR) && S.n_lights > 0) {  // computeDirect
                        float pmf = 1.f / S.n_lights;
                        int li = (int)(S.n_lights * rt_rng_next(rng));
                        li = li < (int)S.n_lights - 1 ? li : (int)S.n_lights - 1;
                        DLight light = S.lights[li];
                        float r1 = rt_rng_next(rng), r2 = rt_rng_next(rng);
                        if (light.kind == 0) {
                            const DTri& lt = S.tris[light.tri];
                            col3 emitted = mat_emission(S.mats[lt.mat_gsign & 0x7FFFFFFFu]);
                            f3 p = tri_sample_point(lt, r1, r2);
                            float pdf = 1 / light.area;
                            f3 wi = p - sd.x;
                            float l = lensq3(wi);
                            wi = norm3(wi);
                            float G = (smaxf_w(dot3(wi, sd.sN), 0.0f) * smaxf_w(-dot3(wi, tri_gnormal(lt)), 0.0f)) / l;
                            if (G > 0) {
                                col3 cd = ((bsdf_eval(S, sd, wi) * emitted) * G) / (pmf * pdf);
                                want_shadow = true;
                                sh_a = make_float4(sd.x.x, sd.x.y, sd.x.z, __uint_as_float(i));
                                sh_b = make_float4(p.x, p.y, p.z, 0);
                                sh_c = make_float4(cd.r, cd.g, cd.b, 0);
                            }
                        } else if (light.kind == 1) {
                            f3 wi = uniform_sample_sphere(r1, r2);
                            float G = smaxf_w(dot3(wi, sd.sN), 0.0f);
                            if (G > 0) {
                                col3 emitted = mkc(S.bg_colour[0], S.bg_colour[1], S.bg_colour[2]);
                                col3 cd = ((bsdf_eval(S, sd, wi) * emitted) * G) / (pmf * uniform_sphere_pdf());
                                f3 p2 = sd.x + (wi * 10000.0f);
                                want_shadow = true;
                                sh_a = make_float4(sd.x.x, sd.x.y, sd.x.z, __uint_as_float(i));
                                sh_b = make_float4(p2.x, p2.y, p2.z, 0);
                                sh_c = make_float4(cd.r, cd.g, cd.b, 0);
                            }
                        } else {   // EnvironmentMap::sample, as in k_shade
                            float pdf;
                            f3 wi = env_sample_dir(S, r1, r2, pdf);
                            float G = smaxf_w(dot3(wi, sd.sN), 0.0f);
                            if (G > 0) {
                                col3 cd = ((bsdf_eval(S, sd, wi) * bg_evaluate(S, wi)) * G) / (pmf * pdf);
                                if (!(cd.r == 0.0f && cd.g == 0.0f && cd.b == 0.0f)) {
                                    f3 p2 = sd.x + (wi * 10000.0f);
                                    want_shadow = true;
                                    sh_a = make_float4(sd.x.x, sd.x.y, sd.x.z, __uint_as_float(i));
                                    sh_b = make_float4(p2.x, p2.y, p2.z, 0);
                                    sh_c = make_float4(cd.r, cd.g, cd.b, 0);
                                }
                            }
                        }
                    }
                } else if (mode == RT_MODE_AOV_ALBEDO) out = bg_evaluate(S, d);
            }
            R.rad[i] = make_float4(out.r, out.g, out.b, 0);
        }
        uint32_t s1 = wave_append(&ctl->n_shadow, want_shadow);
        if (want_shadow) { float4* e = R.shadow + (size_t)s1 * 3; e[0] = sh_a; e[1] = sh_b; e[2] = sh_c; }
    }
}

// ------------------------------------------------------------------------------------------------ light tracing
// lightTrace (Renderer.h:695-719): sample light, position, direction; Le = evaluate(-wi)/pdfPosition; connect the
// light vertex to the camera; shoot the path from p (no epsilon offset, as upstream).
__device__ inline void light_emit(const DScene& S, Rng& rng, float N_vpl, bool for_vpl, f3& p, f3& wi, col3& Le, f3& ln, bool& ok) {
    ok = false;
    if (S.n_lights == 0) return;
    float pmf = 1.f / S.n_lights;
    int li = (int)(S.n_lights * rt_rng_next(rng));
    li = li < (int)S.n_lights - 1 ? li : (int)S.n_lights - 1;
    DLight light = S.lights[li];
    float pdfPos, pdfDir;
    float r1 = rt_rng_next(rng), r2 = rt_rng_next(rng);
    float r3 = rt_rng_next(rng), r4 = rt_rng_next(rng);
    col3 emitted;
    if (light.kind == 0) {
        const DTri& lt = S.tris[light.tri];
        p = tri_sample_point(lt, r1, r2);                              // samplePositionFromLight (Lights.h:66)
        pdfPos = 1 / light.area;
        f3 wl = cosine_sample_hemisphere(r3, r4);                      // sampleDirectionFromLight (Lights.h:70)
        pdfDir = cosine_hemisphere_pdf(wl);
        ln = tri_gnormal(lt);
        Frame3 fr = frame_from(ln);
        wi = to_world(fr, wl);
        emitted = mat_emission(S.mats[lt.mat_gsign & 0x7FFFFFFFu]);
        if (!for_vpl) {
            f3 mw = -wi;                                               // AreaLight::evaluate(-wi) (Lights.h:42)
            Le = (dot3(mw, ln) < 0) ? emitted : mkc(0, 0, 0);
            Le = Le / pdfPos;
        }
    } else {
        p = uniform_sample_sphere(r1, r2);                             // Lights.h:122, :327
        p = p * S.scene_radius;
        p = p + ld3(S.scene_centre);
        if (light.kind == 1) pdfPos = (float)(4 * RT_PI * S.scene_radius * S.scene_radius);  // sic
        else pdfPos = (float)(1.0f / (4 * RT_PI * (S.scene_radius * S.scene_radius)));
        if (light.kind == 1) { wi = uniform_sample_sphere(r3, r4); pdfDir = uniform_sphere_pdf(); }
        else wi = env_sample_dir(S, r3, r4, pdfDir);                   // sampleDirectionFromLight (Lights.h:411-424)
        emitted = mkc(S.bg_colour[0], S.bg_colour[1], S.bg_colour[2]);
        ln = -p;                                                       // normal(shadingData, p) = -p (Lights.h:114)
        if (!for_vpl) Le = bg_evaluate(S, -wi) / pdfPos;
    }
    if (for_vpl) {  // traceVPLs (Renderer.h:788-797): light->sample() draws two more numbers; emitted = what it returns as reflectedColour
        const float r5 = rt_rng_next(rng), r6 = rt_rng_next(rng);
        if (light.kind == 2) { float pdf_; emitted = bg_evaluate(S, env_sample_dir(S, r5, r6, pdf_)); }
        Le = (emitted * fabsf(dot3(wi, ln))) / (pmf * pdfPos * pdfDir * N_vpl);
    }
    ok = true;
}

// Queue appends are block-wide (one atomic per 512 paths): a per-wave atomic on one counter caps a kernel at ~6 G items/s.
#define RT_LT_BLOCK 512
__global__ __launch_bounds__(RT_LT_BLOCK) void k_lt_generate(DScene S, RBuf R, Ctl* ctl, uint32_t n, uint32_t npix, uint64_t seed, uint64_t first_sample) {
    __shared__ uint32_t s_app[RT_LT_BLOCK / 64 + 1];
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
        uint32_t i = base + threadIdx.x;
        bool ok = false;
        f3 p, wi, ln; col3 Le;
        if (i < n) {
            uint32_t s = i / npix, pix = i - s * npix;
            Rng rng; rng.key = rt_rng_key(seed, first_sample + s, pix, 1); rng.ctr = 0;
            light_emit(S, rng, 0.0f, false, p, wi, Le, ln, ok);
            if (ok) {
                R.st[(size_t)i * 4 + 0] = make_float4(p.x, p.y, p.z, __uint_as_float(rng.key));
                R.st[(size_t)i * 4 + 1] = make_float4(wi.x, wi.y, wi.z, __uint_as_float(rng.ctr));
                R.st[(size_t)i * 4 + 2] = make_float4(0, 0, 0, 0);   // the hit row too: a whole 64-B record per lane, no partial line for the memory side to merge
                R.st[(size_t)i * 4 + 3] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(0u));
                R.rad[i] = make_float4(Le.r, Le.g, Le.b, 0);
            } else R.st[(size_t)i * 4 + 3] = make_float4(0, 0, 0, __uint_as_float(RT_FLAG_DEAD));
        }
        // connectToCamera returns before Scene::visible when the point projects outside the film (Renderer.h:672): no request
        float px_, py_;
        const bool cam = ok && camera_project(S.cam, p, px_, py_);
        uint32_t* const counters[1] = { &ctl->n_shadow };
        const bool pred[1] = { cam };
        uint32_t slot[1];
        block_append<1, RT_LT_BLOCK / 64>(counters, pred, slot, s_app);
        const uint32_t s1 = slot[0];
        if (cam) { float4* e = R.shadow + (size_t)s1 * 3; e[0] = make_float4(p.x, p.y, p.z, 0); e[1] = make_float4(0, 0, 0, 0); e[2] = make_float4(Le.r, Le.g, Le.b, 0); }
    }
}

// lightTracePath (Renderer.h:722-767)
__global__ __launch_bounds__(RT_LT_BLOCK) void k_lt_shade(DScene S, RBuf R, Ctl* ctl, int cur) {
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_app[2 * (RT_LT_BLOCK / 64 + 1)];
    const uint32_t total = ctl->n_class[0][0];
    for (;;) {
        if (threadIdx.x == 0) s_base = atomicAdd(&ctl->work[1][0], (unsigned long long)RT_LT_BLOCK);
        __syncthreads();
        unsigned long long base = s_base;
        if (base >= total) break;
        uint32_t g = (uint32_t)base + threadIdx.x;
        bool live = g < total, want_next = false, cam = false;
        uint32_t path = 0;
        float4 sh_a, sh_c;
        if (live) {
            path = R.q_class[g];
            float4 ro = R.st[(size_t)path * 4 + 0], rd = R.st[(size_t)path * 4 + 1], hh = R.st[(size_t)path * 4 + 2], th = R.st[(size_t)path * 4 + 3], le4 = R.rad[path];
            f3 o = mk3(ro.x, ro.y, ro.z), d = mk3(rd.x, rd.y, rd.z);
            Rng rng; rng.key = __float_as_uint(ro.w); rng.ctr = __float_as_uint(rd.w);
            uint32_t depth = __float_as_uint(th.w) & 0xFFu;
            col3 T = mkc(th.x, th.y, th.z), Le = mkc(le4.x, le4.y, le4.z);
            Shading sd = shading_data(S, o, d, hh.x, hh.y, hh.z, __float_as_uint(hh.w));
            const DMaterial& m = S.mats[sd.mat];
            f3 wc = norm3(ld3(S.cam.origin) - sd.x);
            col3 c = (T * bsdf_eval(S, sd, wc)) * Le;
            sh_a = make_float4(sd.x.x, sd.x.y, sd.x.z, 0);
            sh_c = make_float4(c.r, c.g, c.b, 0);
            float px_, py_;
            cam = camera_project(S.cam, sd.x, px_, py_);   // off-film points never reach Scene::visible (Renderer.h:672)
            if (!(m.flags & RT_MF_LIGHT) && !(depth > RT_MAX_DEPTH)) {
                float lum = lumc(T);
                float rr = lum < 0.9f ? lum : 0.9f;
                if (rt_rng_next(rng) < rr) {
                    T = T / rr;
                    float r1 = rt_rng_next(rng), r2 = rt_rng_next(rng);
                    f3 nw = cosine_sample_hemisphere(r1, r2);
                    float pdf = cosine_hemisphere_pdf(wc);  // sic: world-space direction to the camera (Renderer.h:757)
                    nw = to_world(sd.frame, nw);
                    col3 f = bsdf_eval(S, sd, nw);
                    T = ((T * f) * fabsf(dot3(nw, sd.sN))) / pdf;
                    f3 no = sd.x + (nw * RT_EPSILON);
                    R.st[(size_t)path * 4 + 0] = make_float4(no.x, no.y, no.z, ro.w);
                    R.st[(size_t)path * 4 + 1] = make_float4(nw.x, nw.y, nw.z, __uint_as_float(rng.ctr));
                    R.st[(size_t)path * 4 + 3] = make_float4(T.r, T.g, T.b, __uint_as_float(depth + 1));
                    want_next = true;
                }
            }
        }
        uint32_t* const counters[2] = { &ctl->n_shadow, &ctl->n_active[cur ^ 1] };
        const bool pred[2] = { cam, want_next };
        uint32_t slot[2];
        block_append<2, RT_LT_BLOCK / 64>(counters, pred, slot, s_app);   // its barriers also fence s_base for the next round
        if (cam) { float4* e = R.shadow + (size_t)slot[0] * 3; e[0] = sh_a; e[1] = make_float4(0, 0, 0, 0); e[2] = sh_c; }
        if (want_next) R.q_active[cur ^ 1][slot[1]] = path;
    }
}

// ------------------------------------------------------------------------------------------------ instant radiosity
// traceVPLs / VPLTracePath (Renderer.h:772-863): one lane per light sample, <= 6 records each, fixed slots so the
// gather order is reproducible.  record: a=(x, mat|-1) b=(sN, tu) c=(wo, tv) d=(Le, valid)
__global__ __launch_bounds__(64) void k_vpl_trace(DScene S, RBuf R, int n_vpl, uint64_t seed, uint32_t* spill2, Ctl* ctl) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t gstride = gridDim.x * blockDim.x;
    if (i >= (uint32_t)n_vpl) return;
    float4* rec = R.vpl + (size_t)i * 6 * 4;
    for (int k = 0; k < 6; k++) rec[k * 4 + 3] = make_float4(0, 0, 0, 0);
    Rng rng; rng.key = rt_rng_key(seed, 0, i, 3); rng.ctr = 0;
    f3 p, wi, ln; col3 Le; bool ok;
    light_emit(S, rng, (float)n_vpl, true, p, wi, Le, ln, ok);
    if (!ok) return;
    int slot = 0;
    rec[0] = make_float4(p.x, p.y, p.z, __int_as_float(-1));
    rec[1] = make_float4(ln.x, ln.y, ln.z, 0); rec[2] = make_float4(0, 0, 0, 0); rec[3] = make_float4(Le.r, Le.g, Le.b, 1.0f);
    slot = 1;
    f3 o = p + (wi * RT_EPSILON), d = wi;
    col3 T = mkc(1, 1, 1);
    uint32_t cn = 0, ct = 0, nrays = 0;
    for (int depth = 0; depth <= RT_MAX_DEPTH; depth++) {
        HitRec h; h.t = RT_FLT_MAX; h.u = h.v = 0; h.rank = 0xFFFFFFFFu;
        nrays++;
        if (S.root_ref != RT_REF_EMPTY) rt_walk_reference<false, false>(S, o, d, 0.0f, h, spill2 + i, gstride, cn, ct);
        if (!(h.t < RT_FLT_MAX)) break;
        Shading sd = shading_data(S, o, d, h.t, h.u, h.v, h.rank);
        float lum = lumc(T);
        float rr = lum < 0.9f ? lum : 0.9f;
        if (rt_rng_next(rng) < rr) T = T / rr; else break;
        col3 c = T * Le;
        float4* e = rec + slot * 4;
        e[0] = make_float4(sd.x.x, sd.x.y, sd.x.z, __int_as_float((int)sd.mat));
        e[1] = make_float4(sd.sN.x, sd.sN.y, sd.sN.z, sd.tu);
        e[2] = make_float4(sd.wo.x, sd.wo.y, sd.wo.z, sd.tv);
        e[3] = make_float4(c.r, c.g, c.b, 1.0f);
        slot++;
        col3 f; float pdf;
        f3 w2 = bsdf_sample(S, sd, rng, f, pdf);
        T = ((T * f) * fabsf(dot3(w2, sd.sN))) / pdf;
        o = sd.x + (w2 * RT_EPSILON); d = w2;
    }
    atomicAdd(&ctl->tot_closest, (unsigned long long)nrays);
}

// computeVPLContribution (Renderer.h:870-907): one lane per shading point, VPLs in slot order; cosine tests run
// before the shadow ray (same sum, fewer rays than upstream which traces first).
// One item = one (shading point, VPL) pair, VPL-major so that the lanes of a wave read consecutive requests and add to
// different paths.  The pairs go through the same persistent any-hit driver as the NEE shadow rays (lane refill, voted
// steps); a visible pair adds its term to the path's radiance with float atomics, so the order of the sum over VPLs is
// not fixed (upstream sums in VPL order, Renderer.h:870-907; the terms are the same).
struct GatherJob {
    const DScene& S;
    const RBuf& R;
    uint32_t n, v0;
    unsigned long long nrays;
    static constexpr uint32_t refill_idle = RT_REFILL_IDLE;
    __device__ __forceinline__ bool terms(uint32_t k, Shading& sd, f3& vx, f3& vn, f3& wi, float& G, float4& r0, float4& r3, col3& T, uint32_t& path, uint32_t& v) const {
        v = v0 + k / n;
        const uint32_t i = k - (k / n) * n;
        const float4* r = R.vpl + (size_t)v * 4;
        r3 = r[3];
        if (r3.w == 0.0f) return false;
        r0 = r[0];
        const float4 r1 = r[1];
        const float4* e = R.gather + (size_t)i * 4;
        const float4 a = e[0], b = e[1];
        sd.x = mk3(a.x, a.y, a.z); sd.sN = mk3(b.x, b.y, b.z);
        vx = mk3(r0.x, r0.y, r0.z); vn = mk3(r1.x, r1.y, r1.z);
        wi = vx - sd.x;
        float dist = len3(wi);
        wi = norm3(wi);
        float cx_ = dot3(sd.sN, wi), cxi = dot3(vn, -wi);
        if (cx_ <= 0.0f || cxi <= 0.0f) return false;
        G = cx_ * cxi / (dist * dist);
        if (G <= 0.0f) return false;
        const float4 c = e[2], dd = e[3];
        sd.wo = mk3(c.x, c.y, c.z); sd.tu = b.w; sd.tv = c.w; sd.mat = __float_as_uint(dd.w);
        T = mkc(dd.x, dd.y, dd.z);
        path = __float_as_uint(a.w);
        return true;
    }
    __device__ __forceinline__ bool load(uint32_t k, f3& p1, f3& p2) {
        Shading sd; f3 vx, vn, wi; float G; float4 r0, r3; col3 T; uint32_t path, v;
        if (!terms(k, sd, vx, vn, wi, G, r0, r3, T, path, v)) return false;
        p1 = sd.x; p2 = vx;
        nrays++;
        return true;
    }
    __device__ __forceinline__ void retire(uint32_t k, const Walk&, bool visible) {
        if (!visible) return;
        Shading sd; f3 vx, vn, wi; float G; float4 r0, r3; col3 T; uint32_t path, v;
        if (!terms(k, sd, vx, vn, wi, G, r0, r3, T, path, v)) return;
        sd.frame = frame_from(sd.sN);
        sd.albedo = tex_sample(S, S.mats[sd.mat].tex, sd.tu, sd.tv);
        col3 fx = bsdf_eval(S, sd, wi);
        col3 Le = mkc(r3.x, r3.y, r3.z), term;
        int vmat = __float_as_int(r0.w);
        if (vmat < 0 || (S.mats[vmat].flags & RT_MF_LIGHT)) term = (fx * G) * Le;
        else {
            const float4* r = R.vpl + (size_t)v * 4;
            const float4 r1 = r[1], r2 = r[2];
            Shading vs;
            vs.x = vx; vs.sN = vn; vs.wo = mk3(r2.x, r2.y, r2.z); vs.tu = r1.w; vs.tv = r2.w; vs.mat = (uint32_t)vmat;
            vs.frame = frame_from(vn);
            vs.albedo = tex_sample(S, S.mats[vs.mat].tex, vs.tu, vs.tv);
            col3 fxi = bsdf_eval(S, vs, -wi);
            term = ((fx * G) * fxi) * Le;
        }
        col3 add = T * term;
        float* L = reinterpret_cast<float*>(R.rad + path);
        atomicAdd(L + 0, add.r); atomicAdd(L + 1, add.g); atomicAdd(L + 2, add.b);
    }
};
// VPL slots [v0, v0 + nv) against every gather request of the bounce
__global__ __launch_bounds__(RT_BLOCK, RT_TRAV_WAVES) void k_gather(DScene S, RBuf R, Ctl* ctl, uint32_t v0, uint32_t nv, uint32_t* spillw, uint32_t* spill2, DCounters* counters) {
    __shared__ __align__(16) uint32_t lds_stack[RT_STACK_WORDS];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x, gstride = gridDim.x * RT_BLOCK;
    RT_TRAV_STACK_INIT(st, lds_stack, spillw, gtid, gstride)
    const uint32_t n = ctl->n_gather;
    if (n == 0) return;
    const uint32_t items = n * nv;   // the host keeps paths * nv below 2^31
    uint32_t cn = 0, ct = 0, cx = 0;
    GatherJob job{S, R, n, v0, 0ULL};
    XcdSource src; src.xf.init(items); src.w = ctl->work_x[2]; src.n = items;
    rt_persistent_walk<true, false>(S, job, src, 0, st, spill2 + gtid, gstride, cn, ct, cx, counters);
    unsigned long long total = job.nrays;
    for (int o = 32; o; o >>= 1) total += __shfl_xor(total, o);
    if ((threadIdx.x & 63) == 0 && total) atomicAdd(&ctl->tot_any, total);
}

// ------------------------------------------------------------------------------------------------ debug probes
__global__ void k_debug_bsdf(DScene S, uint32_t mat, const float* wo, const float* n, const float* uv, const uint32_t* kc, const float* wi,
                             size_t count, float* out, int do_sample) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Shading sd;
    sd.x = mk3(0, 0, 0); sd.wo = ld3(wo + i * 3); sd.sN = ld3(n + i * 3);
    sd.tu = uv[i * 2]; sd.tv = uv[i * 2 + 1]; sd.mat = mat;
    if ((S.mats[mat].flags & RT_MF_TWOSIDED) && dot3(sd.wo, sd.sN) < 0) sd.sN = -sd.sN;
    sd.frame = frame_from(sd.sN);
    sd.albedo = tex_sample(S, S.mats[mat].tex, sd.tu, sd.tv);
    if (do_sample) {
        Rng rng; rng.key = kc[i * 2]; rng.ctr = kc[i * 2 + 1];
        col3 f; float pdf;
        f3 w = bsdf_sample(S, sd, rng, f, pdf);
        float* o = out + i * 7;
        o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = f.r; o[4] = f.g; o[5] = f.b; o[6] = pdf;
    } else {
        f3 w = ld3(wi + i * 3);
        col3 f = bsdf_eval(S, sd, w);
        float* o = out + i * 4;
        o[0] = f.r; o[1] = f.g; o[2] = f.b; o[3] = bsdf_pdf(S, sd, w);
    }
}

// EnvironmentMap probes: which = 0 sample() after its two draws (wi, pdf), 1 SampleContinuous (d0, d1, pdf, 0), 2 PDF(wi)
__global__ void k_debug_env(DScene S, int which, const float* in, size_t count, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float* a = in + i * 3;
    float* o = out + i * 4;
    if (which == 0) { float pdf; f3 w = env_sample_dir(S, a[0], a[1], pdf); o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = pdf; }
    else if (which == 1) { float pdf, d0, d1; env_sample_continuous(S, a[0], a[1], pdf, d0, d1); o[0] = d0; o[1] = d1; o[2] = pdf; o[3] = 0; }
    else { o[0] = env_pdf(S, mk3(a[0], a[1], a[2])); o[1] = o[2] = o[3] = 0; }
}

// ================================================================================================= host side
#ifndef RT_LANES
#define RT_LANES 2
#endif
enum { RT_KID_GENERATE = 0, RT_KID_EXTEND, RT_KID_SHADE, RT_KID_CONNECT, RT_KID_GATHER, RT_KID_ACCUMULATE, RT_KID_CTL, RT_KID_COUNT };
struct rt_renderer {
    rt_scene* scene = nullptr;
    int W = 0, H = 0;
    hipStream_t stream = nullptr;
    RBuf R{};
    Ctl* ctl = nullptr;
    DCounters* counters = nullptr;
    float* film_own = nullptr;
    float* film = nullptr;
    uint8_t* tonemap_buf = nullptr;   // rt_film_tonemap's device-side image, made on first use
    float* aov_film = nullptr;        // rt_render_aov's film, made on first use
    int spp = 0;
    DFilter filter{};   // kind 0 (BoxFilter) until rt_renderer_set_filter
    // rt_render requests not yet enqueued (see rt_render): consecutive sample ranges of one mode and seed
    struct Pending { bool active = false; int mode = 0, spp = 0; uint64_t seed = 0, first = 0; } pend;
    uint32_t batch_max = 1;
    uint32_t* spillw = nullptr;
    uint32_t* spill2 = nullptr;
    int trav_blocks = 0, flat_blocks = 0;
    int n_vpl = 0;
    uint32_t vpl_cap = 0;             // light samples the VPL record / walk-stack buffers hold (rt_vpl_trace)
    uint32_t* vpl_stack = nullptr;
    // Second lane: a full second set of wavefront state on its own stream.  rt_render alternates path-tracing batches
    // between the two lanes, so the tail of one batch's persistent kernels (a launch lasts as long as its slowest ray:
    // 0.6-2 ms of nearly idle GPU per k_extend launch on the bench scene) overlaps the other batch's work.  The active
    // lane's state lives in the fields above; use_lane() swaps.
    struct Lane { hipStream_t stream = nullptr; RBuf R{}; Ctl* ctl = nullptr; uint32_t* spillw = nullptr; uint32_t* spill2 = nullptr; };
    Lane other;
    bool other_ready = false;
    int active_lane = 0;
    hipEvent_t acc_event = nullptr;   // film accumulation of the previous batch (keeps the fp32 sum order fixed)
    bool acc_pending = false;
    bool count_enabled = false;
    int lanes = RT_LANES;   // 2: batches alternate between the two streams; 1: everything on one stream (rt_renderer_set_streams)
    std::vector<void*> allocs;
    // per-kernel HIP-event timing on the renderer's own stream (rt_profile_enable)
    bool profile = false;
    struct ProfRec { int kid; hipEvent_t a, b; };
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[RT_KID_COUNT] = { 0 };
    uint32_t prof_launches[RT_KID_COUNT] = { 0 };
};

static void prof_collect(rt_renderer* r);
static void use_lane(rt_renderer* r, int lane) {
    if (r->active_lane == lane) return;
    std::swap(r->stream, r->other.stream); std::swap(r->R, r->other.R); std::swap(r->ctl, r->other.ctl);
    std::swap(r->spillw, r->other.spillw); std::swap(r->spill2, r->other.spill2);
    r->active_lane = lane;
}
static hipError_t sync_all(rt_renderer* r) {
    hipError_t e = hipStreamSynchronize(r->stream);
    if (e == hipSuccess && r->other_ready) e = hipStreamSynchronize(r->other.stream);
    if (e == hipSuccess) r->acc_pending = false;
    return e;
}

template <class T>
static int dev_alloc(rt_renderer* r, T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, sizeof(T) * std::max<size_t>(count, 1));
    if (e != hipSuccess) { rt_set_error(std::string("rt_renderer_create: hipMalloc: ") + hipGetErrorString(e)); return RT_ERR_NOMEM; }
    r->allocs.push_back(q);
    *p = (T*)q;
    return RT_OK;
}

static void flush_scene_renderers(rt_scene* s);
static int flush_pending(rt_renderer* r);
// the scene is going away first: finish what its renderers still owe, then cut them loose
static void orphan_scene_renderers(rt_scene* s) {
    flush_scene_renderers(s);
    for (void* q : s->renderers) { rt_renderer* r = static_cast<rt_renderer*>(q); (void)sync_all(r); r->scene = nullptr; }
    s->renderers.clear();
}
extern "C" void rt_renderer_destroy(rt_renderer* r) {
    if (!r) return;
    r->pend.active = false;   // requested but never looked at
    if (r->scene) {
        auto& v = r->scene->renderers;
        v.erase(std::remove(v.begin(), v.end(), static_cast<void*>(r)), v.end());
    }
    if (r->stream) { (void)sync_all(r); prof_collect(r); (void)hipStreamDestroy(r->stream); }
    if (r->other.stream) (void)hipStreamDestroy(r->other.stream);
    if (r->acc_event) (void)hipEventDestroy(r->acc_event);
    for (hipEvent_t e : r->event_pool) (void)hipEventDestroy(e);
    for (void* p : r->allocs) (void)hipFree(p);
    delete r;
}

static int alloc_state(rt_renderer* r, RBuf& R, Ctl** ctl) {
    int rc;
    if ((rc = dev_alloc(r, &R.st, (size_t)R.cap * 4)) || (rc = dev_alloc(r, &R.rad, R.cap)) || (rc = dev_alloc(r, &R.q_active[0], R.cap)) ||
        (rc = dev_alloc(r, &R.q_active[1], R.cap)) || (rc = dev_alloc(r, &R.q_class, (size_t)R.cap * RT_NCLASS)) || (rc = dev_alloc(r, &R.cls, R.cap)) ||
        (rc = dev_alloc(r, &R.shadow, (size_t)R.cap * 3)) || (rc = dev_alloc(r, ctl, 1)))
        return rc;
    return RT_OK;
}
// the second lane is only made when a render has two batches to overlap
static int ensure_other_lane(rt_renderer* r) {
    if (r->other_ready) return RT_OK;
    rt_renderer::Lane& o = r->other;
    const RBuf& A = r->active_lane == 0 ? r->R : r->other.R;
    o.R = A;
    o.R.gather = nullptr; o.R.vpl = nullptr; o.R.vpl_slots = 0;
    int rc;
    if (hipStreamCreate(&o.stream) != hipSuccess) { rt_set_error("rt_render: hipStreamCreate failed"); return RT_ERR_DEVICE; }
    if ((rc = alloc_state(r, o.R, &o.ctl))) return rc;
    size_t threads = (size_t)r->trav_blocks * RT_BLOCK;
    const HostScene& h = r->scene->host;
    if (RT_SPILL_WORDS(h.walk_stack) && (rc = dev_alloc(r, &o.spillw, threads * RT_SPILL_WORDS(h.walk_stack)))) return rc;
    if ((rc = dev_alloc(r, &o.spill2, threads * std::max<uint32_t>(h.max_stack2, 2)))) return rc;
    RT_HIP(hipMemsetAsync(o.ctl, 0, sizeof(Ctl), o.stream));
    r->other_ready = true;
    return RT_OK;
}

// ImageFilter tables, host side: the same expressions and promotions as Imaging.h:155-214 (see DFilter).
static float host_gauss1(int d) {
    float Nalpha = -1.f, radius = 1.f;
    float secondPart = (float)std::exp((double)Nalpha * std::pow((double)radius, 2));
    return (float)(std::exp((double)Nalpha * std::pow((double)d, 2)) - (double)secondPart);
}
static float host_mn1(const DFilter& F, float y) {
    float x = std::fabs(y);
    if (x <= 0.0f) return (float)((((double)F.a1 * std::pow((double)x, 3)) + ((double)F.b1 * std::pow((double)x, 2)) + (double)F.c1) / 6);
    if (x <= 1.0f) return (float)((((double)F.a2 * std::pow((double)x, 3)) + ((double)F.b2 * std::pow((double)x, 2)) + (double)(F.c2 * x) + (double)F.d2) / 6);
    return 0.0f;   // >= 2: 0 upstream; 1 < x < 2: no return statement upstream, 0 here
}
extern "C" int rt_renderer_set_filter(rt_renderer* r, int kind) {
    if (!r) { rt_set_error("rt_renderer_set_filter: NULL renderer"); return RT_ERR_INVALID; }
    if (kind != RT_FILTER_BOX && kind != RT_FILTER_GAUSSIAN && kind != RT_FILTER_MITCHELL) {
        rt_set_error("rt_renderer_set_filter: unknown filter"); return RT_ERR_INVALID;
    }
    { int frc_ = flush_pending(r); if (frc_) return frc_; }   // samples already requested keep the filter they were requested under
    DFilter F{};
    F.kind = kind;
    for (int d = -2; d <= 1; d++) F.g[d + 2] = host_gauss1(d);
    const float B = 1.f / 3.f, C = 1.f / 3.f;
    F.a1 = 12 - (9 * B) - (6 * C); F.b1 = -18 + (12 * B) + (6 * C); F.c1 = 6 - (2 * B);
    F.a2 = (-B) - (6 * C); F.b2 = (6 * B) + (30 * C); F.c2 = -(12 * B) - (48 * C); F.d2 = (8 * B) - (24 * C);
    const int size = kind == RT_FILTER_GAUSSIAN ? 1 : kind == RT_FILTER_MITCHELL ? 2 : 0, span = 2 * size + 1;
    F.wtot = 0.0f;
    for (int i = -size; i <= size; i++)
        for (int j = -size; j <= size; j++) {
            const float dx = (float)j - 0.5f, dy = (float)i - 0.5f;   // px - x for x = pixel + 0.5 (Renderer.h:527)
            float w = 1.0f;
            if (kind == RT_FILTER_GAUSSIAN) w = host_gauss1((int)dx) * host_gauss1((int)dy);
            else if (kind == RT_FILTER_MITCHELL) w = host_mn1(F, dx) * host_mn1(F, dy);
            F.w[(i + size) * span + (j + size)] = w;
            F.wtot += w;
        }
    r->filter = F;
    return RT_OK;
}

extern "C" int rt_renderer_set_streams(rt_renderer* r, int n) {
    if (!r || (n != 1 && n != 2)) { rt_set_error("rt_renderer_set_streams: 1 or 2 streams"); return RT_ERR_INVALID; }
    { int frc_ = flush_pending(r); if (frc_) return frc_; }
    RT_HIP(sync_all(r));
    r->lanes = n;
    return RT_OK;
}

extern "C" int rt_renderer_create(rt_scene* scene, rt_renderer** out) {
    if (!scene || !out) { rt_set_error("rt_renderer_create: NULL argument"); return RT_ERR_INVALID; }
    if (!scene->has_camera) { rt_set_error("rt_renderer_create: scene has no camera (rt_scene_set_camera)"); return RT_ERR_INVALID; }
    *out = nullptr;
    rt_renderer* r = new rt_renderer();
    r->scene = scene;
    r->W = scene->d.cam.iw; r->H = scene->d.cam.ih;
    RBuf& R = r->R;
    R.w = (uint32_t)r->W; R.h = (uint32_t)r->H;
    R.tiles_x = (R.w + 7) / 8;
    uint32_t tiles_y = (R.h + 7) / 8;
    R.np_pad = R.tiles_x * tiles_y * 64;
    R.q0 = 0; R.nq = R.np_pad;
    // batch: as many samples per pixel as fit ~64 M paths per lane (about 10.6 GB of wavefront state each; the card has 288 GB).
    // 32 spp per batch at 1080p measured 1.5 % faster than 16 (2492 -> 2529 Mrays/s at 64 spp per call)
#ifndef RT_BATCH_PATHS
#define RT_BATCH_PATHS (64u << 20)
#endif
#ifndef RT_SPLIT_FRAME
#define RT_SPLIT_FRAME 1   // a one-sample render() is cut into bands of the frame that alternate between the lanes (render_now)
#endif
#ifndef RT_SPLIT_FRAME_PARTS
#define RT_SPLIT_FRAME_PARTS 2
#endif
    r->batch_max = std::max<uint32_t>(1, std::min<uint32_t>(64, RT_BATCH_PATHS / R.np_pad));
    R.cap = std::max<uint32_t>(R.np_pad * r->batch_max, (RT_MLT_PATHS + 63u) / 64u * 64u);  // room for at least one PSSMLT chain
    auto fail = [&](int rc) { rt_renderer_destroy(r); return rc; };
    int rc;
    if (hipStreamCreate(&r->stream) != hipSuccess) { rt_set_error("rt_renderer_create: hipStreamCreate failed"); return fail(RT_ERR_DEVICE); }
    if ((rc = alloc_state(r, R, &r->ctl)) || (rc = dev_alloc(r, &r->counters, 1)) || (rc = dev_alloc(r, &r->film_own, (size_t)r->W * r->H * 3)))
        return fail(rc);
    if (hipEventCreateWithFlags(&r->acc_event, hipEventDisableTiming) != hipSuccess) { rt_set_error("rt_renderer_create: hipEventCreate failed"); return fail(RT_ERR_DEVICE); }
    R.gather = nullptr; R.vpl = nullptr; R.vpl_slots = 0;
    r->film = r->film_own;
    r->trav_blocks = scene->persistent_blocks;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, scene->device) != hipSuccess) return fail(RT_ERR_DEVICE);
    r->flat_blocks = prop.multiProcessorCount * 8;
    size_t threads = (size_t)r->trav_blocks * RT_BLOCK;
    if (RT_SPILL_WORDS(scene->host.walk_stack) && (rc = dev_alloc(r, &r->spillw, threads * RT_SPILL_WORDS(scene->host.walk_stack)))) return fail(rc);
    if ((rc = dev_alloc(r, &r->spill2, threads * std::max<uint32_t>(scene->host.max_stack2, 2)))) return fail(rc);
    if (hipMemsetAsync(r->ctl, 0, sizeof(Ctl), r->stream) != hipSuccess || hipMemsetAsync(r->counters, 0, sizeof(DCounters), r->stream) != hipSuccess ||
        hipMemsetAsync(r->film_own, 0, sizeof(float) * (size_t)r->W * r->H * 3, r->stream) != hipSuccess)
        return fail(RT_ERR_DEVICE);
    scene->renderers.push_back(r);
    scene->before_change = flush_scene_renderers;
    scene->before_destroy = orphan_scene_renderers;
    *out = r;
    return RT_OK;
}

static hipEvent_t prof_event(rt_renderer* r) {
    if (!r->event_pool.empty()) { hipEvent_t e = r->event_pool.back(); r->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
static void prof_collect(rt_renderer* r) {  // caller has synchronised the stream
    for (auto& p : r->prof) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { r->prof_ms[p.kid] += ms; r->prof_launches[p.kid]++; }
        r->event_pool.push_back(p.a); r->event_pool.push_back(p.b);
    }
    r->prof.clear();
}
#define LAUNCHK(kid, kernel, blocks, threads, ...)                                                   \
    do {                                                                                             \
        if (r->profile) {                                                                            \
            rt_renderer::ProfRec pr_{ kid, prof_event(r), prof_event(r) };                           \
            (void)hipEventRecord(pr_.a, r->stream);                                                  \
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, r->stream, __VA_ARGS__);      \
            (void)hipEventRecord(pr_.b, r->stream);                                                  \
            r->prof.push_back(pr_);                                                                  \
        } else hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, r->stream, __VA_ARGS__);   \
    } while (0)
#define LAUNCH(kernel, blocks, threads, ...) LAUNCHK(RT_KID_CTL, kernel, blocks, threads, __VA_ARGS__)

static int render_path_like(rt_renderer* r, int mode, uint32_t batch, uint64_t seed, uint64_t first_sample) {
    const DScene& S = r->scene->d;
    RBuf& R = r->R;
    const bool radiosity = mode == RT_MODE_RADIOSITY && r->n_vpl > 0;
    const uint32_t n = R.nq * batch;
    // pixels among the slots [q0, q0 + nq): whole tile rows of 8 scanlines (render_now's split of a one-sample frame)
    const uint32_t row0 = R.q0 / (R.tiles_x * 64u) * 8u, row1 = std::min<uint32_t>(R.h, (R.q0 + R.nq) / (R.tiles_x * 64u) * 8u);
    const uint32_t live = (row1 > row0 ? row1 - row0 : 0u) * R.w * batch;
    LAUNCHK(RT_KID_GENERATE, k_generate, r->flat_blocks, 256, S, R, n, seed, first_sample);
    const int bounces = RT_MAX_DEPTH + 2;  // depth 0..5: the `depth > MAX_DEPTH` test sits after NEE (Renderer.h:233)
    for (int b = 0; b < bounces; b++) {
        int cur = b & 1;
        LAUNCH(k_ctl_reset, 1, 1, r->ctl, cur, b == 0 ? 1 : 0, n, live);
        if (r->count_enabled) LAUNCHK(RT_KID_EXTEND, (k_extend<false, true>), r->trav_blocks, RT_BLOCK, S, R, r->ctl, cur, b == 0 ? 1 : 0, r->spillw, r->spill2, r->counters);
        else if (r->lanes == 1) LAUNCHK(RT_KID_EXTEND, (k_extend<false, false, 1>), r->trav_blocks, RT_BLOCK, S, R, r->ctl, cur, b == 0 ? 1 : 0, r->spillw, r->spill2, r->counters);
        else LAUNCHK(RT_KID_EXTEND, (k_extend<false, false>), r->trav_blocks, RT_BLOCK, S, R, r->ctl, cur, b == 0 ? 1 : 0, r->spillw, r->spill2, r->counters);
        LAUNCH(k_bin, r->flat_blocks / 2, RT_BIN_BLOCK, R, r->ctl, cur, b == 0 ? 1 : 0);
        if (radiosity && S.env_cond) LAUNCHK(RT_KID_SHADE, (k_shade<true, true>), r->flat_blocks * 256 / RT_SHADE_BLOCK, RT_SHADE_BLOCK, S, R, r->ctl, cur, 0);
        else if (radiosity) LAUNCHK(RT_KID_SHADE, (k_shade<true, false>), r->flat_blocks * 256 / RT_SHADE_BLOCK, RT_SHADE_BLOCK, S, R, r->ctl, cur, 0);
        else if (S.env_cond) LAUNCHK(RT_KID_SHADE, (k_shade<false, true>), r->flat_blocks * 256 / RT_SHADE_BLOCK, RT_SHADE_BLOCK, S, R, r->ctl, cur, 0);
        else LAUNCHK(RT_KID_SHADE, (k_shade<false, false>), r->flat_blocks * 256 / RT_SHADE_BLOCK, RT_SHADE_BLOCK, S, R, r->ctl, cur, 0);
        if (r->count_enabled) LAUNCHK(RT_KID_CONNECT, k_connect<true>, r->trav_blocks, RT_BLOCK, S, R, r->ctl, r->spillw, r->spill2, r->counters);
        else LAUNCHK(RT_KID_CONNECT, k_connect<false>, r->trav_blocks, RT_BLOCK, S, R, r->ctl, r->spillw, r->spill2, r->counters);
        if (radiosity) {
            const uint32_t per = std::max<uint32_t>(1u, (1u << 31) / std::max<uint32_t>(n, 1u));   // item indices are 32-bit: requests (<= n paths) x VPLs per launch < 2^31
            for (uint32_t v0 = 0; v0 < R.vpl_slots; v0 += per) {
                if (v0) RT_HIP(hipMemsetAsync(&r->ctl->work_x[2], 0, sizeof(r->ctl->work_x[2]), r->stream));
                LAUNCHK(RT_KID_GATHER, k_gather, r->trav_blocks, RT_BLOCK, S, R, r->ctl, v0, std::min<uint32_t>(per, R.vpl_slots - v0), r->spillw, r->spill2, r->counters);
            }
        }
    }
    LAUNCH(k_ctl_final, 1, 1, r->ctl);
    if (r->acc_pending) RT_HIP(hipStreamWaitEvent(r->stream, r->acc_event, 0));   // film += in batch order, whichever lane ran the batch
    if (r->filter.kind) LAUNCHK(RT_KID_ACCUMULATE, k_accumulate_filtered, r->flat_blocks, 256, R, r->film, batch, r->filter);
    else LAUNCHK(RT_KID_ACCUMULATE, k_accumulate, r->flat_blocks, 256, R, r->film, batch);
    RT_HIP(hipEventRecord(r->acc_event, r->stream));
    r->acc_pending = true;
    RT_HIP(hipGetLastError());
    return RT_OK;
}

static int render_aov(rt_renderer* r, int mode, uint32_t batch, uint64_t seed, uint64_t first_sample) {
    const DScene& S = r->scene->d;
    RBuf& R = r->R;
    const uint32_t n = R.np_pad * batch;
    LAUNCH(k_ctl_reset, 1, 1, r->ctl, 0, 1, n, (uint32_t)r->W * r->H * batch);
    LAUNCHK(RT_KID_EXTEND, k_aov, r->trav_blocks, RT_BLOCK, S, R, r->ctl, mode, seed, first_sample, n, r->spillw, r->spill2);
    LAUNCHK(RT_KID_CONNECT, k_connect<false>, r->trav_blocks, RT_BLOCK, S, R, r->ctl, r->spillw, r->spill2, r->counters);
    LAUNCH(k_ctl_final, 1, 1, r->ctl);
    LAUNCHK(RT_KID_ACCUMULATE, k_accumulate, r->flat_blocks, 256, R, r->film, batch);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

static int render_light(rt_renderer* r, uint32_t batch, uint64_t seed, uint64_t first_sample) {
    const DScene& S = r->scene->d;
    RBuf& R = r->R;
    const uint32_t npix = (uint32_t)r->W * r->H;
    const uint32_t n = npix * batch;  // W*H light paths per spp (Renderer.h:331-337)
    LAUNCH(k_ctl_reset, 1, 1, r->ctl, 0, 1, 0u, 0u);
    LAUNCHK(RT_KID_GENERATE, k_lt_generate, r->flat_blocks / 2, RT_LT_BLOCK, S, R, r->ctl, n, npix, seed, first_sample);
    if (r->filter.kind) LAUNCHK(RT_KID_CONNECT, k_connect_camera<true>, r->trav_blocks, RT_BLOCK, S, R, r->ctl, r->film, r->filter, r->spillw, r->spill2, r->counters);
    else LAUNCHK(RT_KID_CONNECT, k_connect_camera<false>, r->trav_blocks, RT_BLOCK, S, R, r->ctl, r->film, r->filter, r->spillw, r->spill2, r->counters);
    for (int b = 0; b < RT_MAX_DEPTH + 2; b++) {
        int cur = b & 1;
        LAUNCH(k_ctl_reset, 1, 1, r->ctl, cur, b == 0 ? 1 : 0, n, n);
        LAUNCHK(RT_KID_EXTEND, (k_extend<true, false>), r->trav_blocks, RT_BLOCK, S, R, r->ctl, cur, b == 0 ? 1 : 0, r->spillw, r->spill2, r->counters);
        LAUNCH(k_bin, r->flat_blocks / 2, RT_BIN_BLOCK, R, r->ctl, cur, b == 0 ? 1 : 0);
        LAUNCHK(RT_KID_SHADE, k_lt_shade, r->flat_blocks / 2, RT_LT_BLOCK, S, R, r->ctl, cur);
        if (r->filter.kind) LAUNCHK(RT_KID_CONNECT, k_connect_camera<true>, r->trav_blocks, RT_BLOCK, S, R, r->ctl, r->film, r->filter, r->spillw, r->spill2, r->counters);
        else LAUNCHK(RT_KID_CONNECT, k_connect_camera<false>, r->trav_blocks, RT_BLOCK, S, R, r->ctl, r->film, r->filter, r->spillw, r->spill2, r->counters);
    }
    LAUNCH(k_ctl_final, 1, 1, r->ctl);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

static int render_mlt(rt_renderer* r, uint32_t chains, uint64_t seed, uint64_t first_sample) {
    const DScene& S = r->scene->d;
    RBuf& R = r->R;
    const uint32_t n = chains * RT_MLT_PATHS;
    LAUNCHK(RT_KID_GENERATE, k_mlt_generate, r->flat_blocks, 256, S, R, n, chains, seed, first_sample);
    for (int b = 0; b < RT_MAX_DEPTH + 2; b++) {
        int cur = b & 1;
        LAUNCH(k_ctl_reset, 1, 1, r->ctl, cur, b == 0 ? 1 : 0, n, n);
        LAUNCHK(RT_KID_EXTEND, (k_extend<false, false>), r->trav_blocks, RT_BLOCK, S, R, r->ctl, cur, b == 0 ? 1 : 0, r->spillw, r->spill2, r->counters);
        LAUNCH(k_bin, r->flat_blocks / 2, RT_BIN_BLOCK, R, r->ctl, cur, b == 0 ? 1 : 0);
        if (S.env_cond) LAUNCHK(RT_KID_SHADE, (k_shade<false, true>), r->flat_blocks * 256 / RT_SHADE_BLOCK, RT_SHADE_BLOCK, S, R, r->ctl, cur, 0);
        else LAUNCHK(RT_KID_SHADE, (k_shade<false, false>), r->flat_blocks * 256 / RT_SHADE_BLOCK, RT_SHADE_BLOCK, S, R, r->ctl, cur, 0);
        LAUNCHK(RT_KID_CONNECT, k_connect<false>, r->trav_blocks, RT_BLOCK, S, R, r->ctl, r->spillw, r->spill2, r->counters);
    }
    LAUNCH(k_ctl_final, 1, 1, r->ctl);
    LAUNCHK(RT_KID_ACCUMULATE, k_mlt_chain, chains, 64, S, R, r->film, r->filter, chains, seed, first_sample);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

static uint32_t batch_limit(const rt_renderer* r, int mode) {
    if (mode == RT_MODE_LIGHT) return std::max<uint32_t>(1, r->R.cap / ((uint32_t)r->W * r->H));
    if (mode == RT_MODE_MLT) return std::max<uint32_t>(1, r->R.cap / RT_MLT_PATHS);  // one chain per spp, all chains of a batch side by side
    return r->batch_max;
}
// enqueue `spp` samples now (the film's SPP count is the caller's business)
static int render_now(rt_renderer* r, int mode, int spp, uint64_t seed, uint64_t first_sample) {
    RT_HIP(hipSetDevice(r->scene->device));
    r->count_enabled = r->scene->count_enabled;  // node/triangle fetch counting (atomics) follows rt_counters_enable
    const uint32_t bmax = batch_limit(r, mode);
    int done = 0;
    const uint32_t tile_rows = r->R.np_pad / (r->R.tiles_x * 64u);
    if (mode == RT_MODE_PATH && spp == 1 && r->lanes > 1 && RT_SPLIT_FRAME && !r->filter.kind && tile_rows >= 2 && r->active_lane == 0) {
        // One sample of the frame (upstream's render() per displayed frame, Main.cpp:79-137): the upper and the lower tile rows
        // go to the two lanes, so that the end of one half's launches overlaps the other half's work as batches do above.
        // Every pixel gets the same film += L either way; the filtered films gather across the cut and stay on one lane.
        int rc = ensure_other_lane(r);
        if (rc) return rc;
        RT_HIP(hipEventRecord(r->acc_event, r->stream));
        RT_HIP(hipStreamWaitEvent(r->other.stream, r->acc_event, 0));
        const uint32_t parts = std::min<uint32_t>(RT_SPLIT_FRAME_PARTS, tile_rows), row_slots = r->R.tiles_x * 64u;
        for (uint32_t p = 0; p < parts; p++) {
            const uint32_t a = (uint32_t)((uint64_t)tile_rows * p / parts) * row_slots, b = (uint32_t)((uint64_t)tile_rows * (p + 1) / parts) * row_slots;
            use_lane(r, (int)(p & 1u));
            r->R.q0 = a; r->R.nq = b - a;
            rc = render_path_like(r, mode, 1, seed, first_sample);
            r->R.q0 = 0; r->R.nq = r->R.np_pad;
            if (rc) { use_lane(r, 0); return rc; }
        }
        use_lane(r, 1);
        hipError_t e1 = hipEventRecord(r->acc_event, r->stream);   // lane 1
        use_lane(r, 0);
        RT_HIP(e1);
        RT_HIP(hipStreamWaitEvent(r->stream, r->acc_event, 0));
        r->acc_pending = true;
        return RT_OK;
    }
    if ((mode == RT_MODE_PATH || mode == RT_MODE_LIGHT) && spp >= 2 && r->lanes > 1) {
        // at least two batches, alternating between the two lanes (streams).  Path tracing chains its film accumulation by
        // an event (fixed fp32 sum order); the light tracer's splats are atomics and need no order (PSSMLT on two lanes measured no faster).
        int rc = ensure_other_lane(r);
        if (rc) return rc;
        uint32_t nb = std::max<uint32_t>(2u, ((uint32_t)spp + bmax - 1) / bmax);
        nb += nb & 1u;   // the same number of batches on either lane (48 spp: 4 x 12 measured 2 % faster than 3 x 16)
        // lane 1 starts behind whatever lane 0 already holds (a film clear, an earlier render): its splats are atomics on the
        // same film and would otherwise be free to land before a memset queued ahead of this render
        RT_HIP(hipEventRecord(r->acc_event, r->stream));
        RT_HIP(hipStreamWaitEvent(r->other.stream, r->acc_event, 0));
        for (uint32_t i = 0; i < nb; i++) {
            uint32_t b = (uint32_t)spp / nb + (i < (uint32_t)spp % nb ? 1u : 0u);
            use_lane(r, (int)(i & 1u));
            rc = mode == RT_MODE_LIGHT ? render_light(r, b, seed, first_sample + done) : render_path_like(r, mode, b, seed, first_sample + done);
            if (rc) { use_lane(r, 0); return rc; }
            done += (int)b;
        }
        use_lane(r, 1);
        hipError_t e1 = hipEventRecord(r->acc_event, r->stream);
        use_lane(r, 0);
        RT_HIP(e1);
        RT_HIP(hipStreamWaitEvent(r->stream, r->acc_event, 0));   // whatever is queued on lane 0 next sees the complete film
        r->acc_pending = true;
    }
    while (done < spp) {
        uint32_t b = std::min<uint32_t>(bmax, (uint32_t)(spp - done));
        int rc;
        if (mode == RT_MODE_LIGHT) rc = render_light(r, b, seed, first_sample + done);
        else if (mode == RT_MODE_MLT) rc = render_mlt(r, b, seed, first_sample + done);
        else if (mode == RT_MODE_AOV_ALBEDO || mode == RT_MODE_AOV_NORMAL || mode == RT_MODE_DIRECT) rc = render_aov(r, mode, b, seed, first_sample + done);
        else rc = render_path_like(r, mode, b, seed, first_sample + done);
        if (rc) return rc;
        done += (int)b;
    }
    return RT_OK;
}
// Requests held back by rt_render go to the device.  Every entry point that reads or changes what they depend on (film,
// counters, filter, VPLs, and through rt_scene::before_change the camera) calls this first.
static int flush_pending(rt_renderer* r) {
    if (!r->pend.active) return RT_OK;
    const rt_renderer::Pending p = r->pend;
    r->pend.active = false;
    return render_now(r, p.mode, p.spp, p.seed, p.first);
}
static void flush_scene_renderers(rt_scene* s) {
    for (void* q : s->renderers) (void)flush_pending(static_cast<rt_renderer*>(q));
}
// Upstream's main loop adds one sample per render() call (Main.cpp:79-137).  One sample per pixel is a 2 M-path launch
// sequence whose cost is mostly launch tails (16 ms/spp on the bench scene against 5 ms/spp in 16-spp batches), so calls
// that continue each other -- same mode and seed, first_sample where the last one ended -- are merged and go to the
// device together once two full batches have gathered, or when anything looks at the result.  Sample indices are
// explicit, so the film is the one the separate calls would have produced.
extern "C" int rt_render(rt_renderer* r, int mode, int spp, uint64_t seed, uint64_t first_sample) {
    if (!r || spp < 0) { rt_set_error("rt_render: bad argument"); return RT_ERR_INVALID; }
    if (!r->scene) { rt_set_error("rt_render: the renderer's scene has been destroyed"); return RT_ERR_INVALID; }
    if (mode == RT_MODE_PATH_ST) mode = RT_MODE_PATH;
    if (mode != RT_MODE_PATH && mode != RT_MODE_LIGHT && mode != RT_MODE_MLT && mode != RT_MODE_RADIOSITY && mode != RT_MODE_AOV_ALBEDO && mode != RT_MODE_AOV_NORMAL &&
        mode != RT_MODE_DIRECT) { rt_set_error("rt_render: unknown mode"); return RT_ERR_INVALID; }
    if (spp == 0) return RT_OK;
    const bool merges = mode == RT_MODE_PATH || mode == RT_MODE_LIGHT || mode == RT_MODE_MLT || mode == RT_MODE_RADIOSITY;
    const uint32_t full = std::min<uint32_t>(64u, ((mode == RT_MODE_PATH || mode == RT_MODE_LIGHT) ? 2u : 1u) * batch_limit(r, mode));
    rt_renderer::Pending& p = r->pend;
    if (p.active && !(merges && p.mode == mode && p.seed == seed && p.first + (uint64_t)p.spp == first_sample)) {
        int rc = flush_pending(r);
        if (rc) return rc;
    }
    r->spp += spp;  // film->incrementSPP() once per render() upstream (Renderer.h:399)
    if (!merges || (!p.active && (uint32_t)spp >= full)) return render_now(r, mode, spp, seed, first_sample);
    if (!p.active) { p.active = true; p.mode = mode; p.spp = 0; p.seed = seed; p.first = first_sample; }
    p.spp += spp;
    if ((uint32_t)p.spp >= full) return flush_pending(r);
    return RT_OK;
}

// albedoFilm / normalFilm (Renderer.h:100-103): the films getTileFilms (Renderer.h:577-603) fills for a denoiser next to the
// beauty film.  Renders `spp` samples of an AOV / direct pass into a film of its own and hands it back un-normalised; the
// renderer's film and SPP are not touched.
extern "C" int rt_render_aov(rt_renderer* r, int mode, int spp, uint64_t seed, uint64_t first_sample, float* rgb) {
    if (!r || !rgb || spp < 1) { rt_set_error("rt_render_aov: bad argument"); return RT_ERR_INVALID; }
    if (mode != RT_MODE_AOV_ALBEDO && mode != RT_MODE_AOV_NORMAL && mode != RT_MODE_DIRECT) { rt_set_error("rt_render_aov: mode must be an AOV or the direct pass"); return RT_ERR_INVALID; }
    if (!r->scene) { rt_set_error("rt_render_aov: the renderer's scene has been destroyed"); return RT_ERR_INVALID; }
    { int frc_ = flush_pending(r); if (frc_) return frc_; }
    RT_HIP(sync_all(r));
    const size_t n = (size_t)r->W * r->H * 3;
    if (!r->aov_film) { int rc = dev_alloc(r, &r->aov_film, n); if (rc) return rc; }
    RT_HIP(hipMemsetAsync(r->aov_film, 0, n * sizeof(float), r->stream));
    float* keep = r->film;
    r->film = r->aov_film;
    int rc = render_now(r, mode, spp, seed, first_sample);
    r->film = keep;
    if (rc) return rc;
    RT_HIP(sync_all(r));
    RT_HIP(hipMemcpy(rgb, r->aov_film, n * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_renderer_sync(rt_renderer* r) {
    if (!r) { rt_set_error("rt_renderer_sync: NULL"); return RT_ERR_INVALID; }
    { int frc_ = flush_pending(r); if (frc_) return frc_; }
    RT_HIP(sync_all(r));
    prof_collect(r);
    return RT_OK;
}
// Per-kernel timing with HIP events recorded on the renderer's stream around every launch.
// kernel ids: 0 generate, 1 extend, 2 shade, 3 connect, 4 gather, 5 accumulate, 6 control
extern "C" int rt_profile_enable(rt_renderer* r, int enable) {
    if (!r) { rt_set_error("rt_profile_enable: NULL"); return RT_ERR_INVALID; }
    { int frc_ = flush_pending(r); if (frc_) return frc_; }
    RT_HIP(sync_all(r));
    prof_collect(r);
    r->profile = enable != 0;
    return RT_OK;
}
extern "C" int rt_profile_read(rt_renderer* r, double* ms7, uint32_t* launches7, int reset) {
    if (!r || !ms7 || !launches7) { rt_set_error("rt_profile_read: NULL argument"); return RT_ERR_INVALID; }
    { int frc_ = flush_pending(r); if (frc_) return frc_; }
    RT_HIP(sync_all(r));
    prof_collect(r);
    for (int i = 0; i < RT_KID_COUNT; i++) { ms7[i] = r->prof_ms[i]; launches7[i] = r->prof_launches[i]; }
    if (reset) for (int i = 0; i < RT_KID_COUNT; i++) { r->prof_ms[i] = 0; r->prof_launches[i] = 0; }
    return RT_OK;
}
extern "C" int rt_film_clear(rt_renderer* r) {  // Film::clear (Imaging.h:269)
    if (!r) { rt_set_error("rt_film_clear: NULL"); return RT_ERR_INVALID; }
    r->pend.active = false;   // samples requested but not yet enqueued would be wiped by this clear anyway
    RT_HIP(hipMemsetAsync(r->film, 0, sizeof(float) * (size_t)r->W * r->H * 3, r->stream));
    r->spp = 0;
    return RT_OK;
}
extern "C" int rt_film_spp(const rt_renderer* r) { return r ? r->spp : 0; }
extern "C" int rt_film_set_spp(rt_renderer* r, int spp) {
    if (!r || spp < 0) { rt_set_error("rt_film_set_spp: bad argument"); return RT_ERR_INVALID; }
    { int frc_ = flush_pending(r); if (frc_) return frc_; }
    r->spp = spp;
    return RT_OK;
}
extern "C" int rt_film_read(rt_renderer* r, float* rgb, int* spp) {
    if (!r || !rgb) { rt_set_error("rt_film_read: NULL argument"); return RT_ERR_INVALID; }
    { int frc_ = flush_pending(r); if (frc_) return frc_; }
    RT_HIP(sync_all(r));
    RT_HIP(hipMemcpy(rgb, r->film, sizeof(float) * (size_t)r->W * r->H * 3, hipMemcpyDeviceToHost));
    if (spp) *spp = r->spp;
    return RT_OK;
}
extern "C" int rt_film_bind(rt_renderer* r, float* d_film) {
    if (!r) { rt_set_error("rt_film_bind: NULL"); return RT_ERR_INVALID; }
    { int frc_ = flush_pending(r); if (frc_) return frc_; }
    RT_HIP(sync_all(r));
    r->film = d_film ? d_film : r->film_own;
    return RT_OK;
}
extern "C" int rt_render_counters(rt_renderer* r, rt_counters* o, int reset) {
    if (!r || !o) { rt_set_error("rt_render_counters: NULL argument"); return RT_ERR_INVALID; }
    { int frc_ = flush_pending(r); if (frc_) return frc_; }
    RT_HIP(sync_all(r));
    Ctl c; DCounters d;
    RT_HIP(hipMemcpy(&c, r->ctl, sizeof(c), hipMemcpyDeviceToHost));
    if (r->other_ready) {
        Ctl c2;
        RT_HIP(hipMemcpy(&c2, r->other.ctl, sizeof(c2), hipMemcpyDeviceToHost));
        c.tot_closest += c2.tot_closest; c.tot_any += c2.tot_any;
    }
    RT_HIP(hipMemcpy(&d, r->counters, sizeof(d), hipMemcpyDeviceToHost));
    o->closest_rays = c.tot_closest; o->any_rays = c.tot_any; o->nodes_visited = d.nodes; o->tris_tested = d.tris; o->exact_path_rays = d.exact;
    o->any_nodes_visited = d.any_nodes; o->any_tris_tested = d.any_tris;
    o->lane_slots = d.slots; o->lane_busy = d.busy; o->any_lane_slots = d.any_slots; o->any_lane_busy = d.any_busy; o->max_steps = d.max_steps; o->any_max_steps = d.any_max_steps;
    o->gate_refused = d.refused; o->any_gate_refused = d.any_refused;
    if (reset) {
        RT_HIP(hipMemset(r->ctl, 0, sizeof(Ctl)));
        if (r->other_ready) RT_HIP(hipMemset(r->other.ctl, 0, sizeof(Ctl)));
        RT_HIP(hipMemset(r->counters, 0, sizeof(DCounters)));
    }
    return RT_OK;
}
extern "C" int rt_vpl_trace(rt_renderer* r, int n_vpls, uint64_t seed) {
    if (!r || n_vpls < 0) { rt_set_error("rt_vpl_trace: bad argument"); return RT_ERR_INVALID; }
    if (n_vpls > (1 << 20)) { rt_set_error("rt_vpl_trace: at most 2^20 light samples"); return RT_ERR_INVALID; }
    if (!r->scene) { rt_set_error("rt_vpl_trace: the renderer's scene has been destroyed"); return RT_ERR_INVALID; }
    { int frc_ = flush_pending(r); if (frc_) return frc_; }
    RT_HIP(sync_all(r));
    r->n_vpl = 0; r->R.vpl_slots = 0;
    if (n_vpls == 0) return RT_OK;
    int rc;
    float4* gather = nullptr;
    int blocks = (n_vpls + 63) / 64;
    // the record and walk-stack buffers stay with the renderer and are reused; they only grow (the old ones are freed)
    if ((uint32_t)n_vpls > r->vpl_cap) {
        for (void* old : { (void*)r->R.vpl, (void*)r->vpl_stack })
            if (old) { (void)hipFree(old); r->allocs.erase(std::remove(r->allocs.begin(), r->allocs.end(), old), r->allocs.end()); }
        r->R.vpl = nullptr; r->vpl_stack = nullptr; r->vpl_cap = 0;
        if ((rc = dev_alloc(r, &r->R.vpl, (size_t)blocks * 64 * 6 * 4))) return rc;
        if ((rc = dev_alloc(r, &r->vpl_stack, (size_t)blocks * 64 * std::max<uint32_t>(r->scene->host.max_stack2, 2)))) return rc;
        r->vpl_cap = (uint32_t)blocks * 64u;
    }
    if (!r->R.gather) { if ((rc = dev_alloc(r, &gather, (size_t)r->R.cap * 4))) return rc; r->R.gather = gather; }
    uint32_t* sp = r->vpl_stack;
    LAUNCH(k_vpl_trace, blocks, 64, r->scene->d, r->R, n_vpls, seed, sp, r->ctl);
    RT_HIP(hipGetLastError());
    { int frc_ = flush_pending(r); if (frc_) return frc_; }
    RT_HIP(sync_all(r));
    r->n_vpl = n_vpls; r->R.vpl_slots = (uint32_t)n_vpls * 6;
    return RT_OK;
}
extern "C" int rt_vpl_count(const rt_renderer* r) {
    if (!r || !r->R.vpl || r->R.vpl_slots == 0) return 0;
    std::vector<float4> h((size_t)r->R.vpl_slots * 4);
    if (hipMemcpy(h.data(), r->R.vpl, h.size() * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    int n = 0;
    for (uint32_t v = 0; v < r->R.vpl_slots; v++) if (h[(size_t)v * 4 + 3].w != 0.0f) n++;
    return n;
}

// Film::save (Imaging.h:278): Radiance RGBE of film / SPP (flat scanlines)
extern "C" int rt_film_save_hdr(rt_renderer* r, const char* filename) {
    if (!r || !filename) { rt_set_error("rt_film_save_hdr: NULL argument"); return RT_ERR_INVALID; }
    std::vector<float> img((size_t)r->W * r->H * 3);
    int spp = 0, rc = rt_film_read(r, img.data(), &spp);
    if (rc) return rc;
    FILE* f = fopen(filename, "wb");
    if (!f) { rt_set_error(std::string("rt_film_save_hdr: cannot open ") + filename); return RT_ERR_INVALID; }
    fprintf(f, "#?RADIANCE\n# Written by raytracer_amd\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=          1.0000000000000\n\n-Y %d +X %d\n", r->H, r->W);
    std::vector<unsigned char> row((size_t)r->W * 4);
    for (int y = 0; y < r->H; y++) {
        for (int x = 0; x < r->W; x++) {
            const float* p = &img[((size_t)y * r->W + x) * 3];
            float c[3] = { p[0] / (float)spp, p[1] / (float)spp, p[2] / (float)spp };
            float m = std::max(c[0], std::max(c[1], c[2]));
            unsigned char* o = &row[(size_t)x * 4];
            if (!(m > 1e-32f)) { o[0] = o[1] = o[2] = o[3] = 0; continue; }
            int e; float norm = frexpf(m, &e) * 256.0f / m;
            o[0] = (unsigned char)(c[0] * norm); o[1] = (unsigned char)(c[1] * norm); o[2] = (unsigned char)(c[2] * norm); o[3] = (unsigned char)(e + 128);
        }
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
    return RT_OK;
}
// Film::tonemap (Imaging.h:251)
// Film::tonemap (Imaging.h:251) for every pixel, on the device: upstream runs it for the whole canvas after every
// render() (Renderer.h:536-541), so a caller presenting each frame needs it to cost a 6 MB copy, not a 25 MB one plus
// six million host powf calls.
__global__ __launch_bounds__(256) void k_tonemap(const float* film, uint8_t* out, size_t n, int spp) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = film[i] * 1.0f / (float)spp;                       // film[...] * exposure / (float)SPP, exposure = 1
        float m = powf(fmaxf(v, 0.0f), 1.0f / 2.2f) * 255;           // std::max(NaN, 0) is NaN upstream too: the cast below is then unspecified, 0 here
        out[i] = (uint8_t)(m < 255.0f ? (m > 0.0f ? m : 0.0f) : 255.0f);
    }
}
extern "C" int rt_film_tonemap(rt_renderer* r, uint8_t* rgb8) {
    if (!r || !rgb8) { rt_set_error("rt_film_tonemap: NULL argument"); return RT_ERR_INVALID; }
    { int frc_ = flush_pending(r); if (frc_) return frc_; }
    RT_HIP(sync_all(r));
    const size_t n = (size_t)r->W * r->H * 3;
    if (!r->tonemap_buf) { int rc = dev_alloc(r, &r->tonemap_buf, n); if (rc) return rc; }
    LAUNCH(k_tonemap, r->flat_blocks, 256, r->film, r->tonemap_buf, n, r->spp);
    RT_HIP(hipGetLastError());
    RT_HIP(hipStreamSynchronize(r->stream));
    RT_HIP(hipMemcpy(rgb8, r->tonemap_buf, n, hipMemcpyDeviceToHost));
    return RT_OK;
}

static int debug_bsdf(rt_scene* s, uint32_t mat, const float* wo, const float* n, const float* uv, const uint32_t* kc, const float* wi, size_t count,
                      float* out, int do_sample) {
    if (!s || !wo || !n || !uv || !out || (do_sample ? !kc : !wi)) { rt_set_error("rt_debug_bsdf: NULL argument"); return RT_ERR_INVALID; }
    if (mat >= s->host.mats.size()) { rt_set_error("rt_debug_bsdf: material out of range"); return RT_ERR_INVALID; }
    if (count == 0) return RT_OK;
    size_t ow = do_sample ? 7 : 4;
    float *d_wo, *d_n, *d_uv, *d_x, *d_out;
    RT_HIP(hipMalloc((void**)&d_wo, count * 12)); RT_HIP(hipMalloc((void**)&d_n, count * 12)); RT_HIP(hipMalloc((void**)&d_uv, count * 8));
    RT_HIP(hipMalloc((void**)&d_x, count * 12)); RT_HIP(hipMalloc((void**)&d_out, count * ow * 4));
    RT_HIP(hipMemcpy(d_wo, wo, count * 12, hipMemcpyHostToDevice)); RT_HIP(hipMemcpy(d_n, n, count * 12, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(d_uv, uv, count * 8, hipMemcpyHostToDevice));
    if (do_sample) RT_HIP(hipMemcpy(d_x, kc, count * 8, hipMemcpyHostToDevice)); else RT_HIP(hipMemcpy(d_x, wi, count * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_bsdf, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, 0, s->d, mat, d_wo, d_n, d_uv, (const uint32_t*)d_x, d_x, count, d_out, do_sample);
    RT_HIP(hipGetLastError());
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(out, d_out, count * ow * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_wo); (void)hipFree(d_n); (void)hipFree(d_uv); (void)hipFree(d_x); (void)hipFree(d_out);
    return RT_OK;
}
extern "C" int rt_debug_env(rt_scene* s, int which, const float* in3, size_t count, float* out4) {
    if (!s || !in3 || !out4 || which < 0 || which > 2) { rt_set_error("rt_debug_env: bad argument"); return RT_ERR_INVALID; }
    if (!s->d.env_cond) { rt_set_error("rt_debug_env: the scene has no environment map"); return RT_ERR_INVALID; }
    if (count == 0) return RT_OK;
    float *d_in = nullptr, *d_out = nullptr;
    RT_HIP(hipMalloc((void**)&d_in, count * 12));
    if (hipMalloc((void**)&d_out, count * 16) != hipSuccess) { (void)hipFree(d_in); rt_set_error("rt_debug_env: out of device memory"); return RT_ERR_NOMEM; }
    hipError_t e = hipMemcpy(d_in, in3, count * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_env, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, 0, s->d, which, (const float*)d_in, count, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out4, d_out, count * 16, hipMemcpyDeviceToHost);
    (void)hipFree(d_in); (void)hipFree(d_out);
    if (e != hipSuccess) return rt_hip_fail(e, "rt_debug_env");
    return RT_OK;
}
extern "C" int rt_debug_bsdf_sample(rt_scene* s, uint32_t mat, const float* wo, const float* n, const float* uv, const uint32_t* kc, size_t count, float* out) {
    return debug_bsdf(s, mat, wo, n, uv, kc, nullptr, count, out, 1);
}
extern "C" int rt_debug_bsdf_eval(rt_scene* s, uint32_t mat, const float* wo, const float* n, const float* uv, const float* wi, size_t count, float* out) {
    return debug_bsdf(s, mat, wo, n, uv, nullptr, wi, count, out, 0);
}
