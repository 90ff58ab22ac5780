// rt_device.h — device-side building blocks for gfx950: float3 helpers with the reference's
// operation order, the counter RNG, ray/box/triangle tests and the two BVH walks.
//
// This translation unit set is compiled with -ffp-contract=off: the intersector must round every
// product and sum separately, exactly like the reference's scalar x86 code (SURVEY.md §7 "hard parts").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_common.h"

#define RT_BLOCK 256
#ifndef RT_TRAV_WAVES
#define RT_TRAV_WAVES 6    // __launch_bounds__ waves/SIMD target of the traversal kernels
#endif
#ifndef RT_STACK_LDS
#define RT_STACK_LDS 24          // stack entries per lane kept in LDS (24 KB per 256-thread block)
#endif
#define RT_FLT_MAX 3.402823466e+38f
#define RT_INF __builtin_inff()

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 v; v.x = x; v.y = y; v.z = z; return v; }
__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return ((a.x * b.x) + (a.y * b.y) + (a.z * b.z)); }       // Core.h:176
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {                                                              // Core.h:170
    return mk3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
__device__ __forceinline__ float len3(f3 a) { return sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); }          // Core.h:157
__device__ __forceinline__ float lensq3(f3 a) { return ((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); }
__device__ __forceinline__ f3 norm3(f3 a) {                                                                     // Core.h:161
    float l = 1.0f / sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z));
    return mk3(a.x * l, a.y * l, a.z * l);
}
// std::max / std::min and the Windows max()/min() macros, NaN behaviour included
__device__ __forceinline__ float smaxf(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float sminf(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float smaxf_w(float a, float b) { return (a > b) ? a : b; }  // max() macro of <Windows.h>

struct col3 { float r, g, b; };
__device__ __forceinline__ col3 mkc(float r, float g, float b) { col3 c; c.r = r; c.g = g; c.b = b; return c; }
__device__ __forceinline__ col3 operator+(col3 a, col3 b) { return mkc(a.r + b.r, a.g + b.g, a.b + b.b); }
__device__ __forceinline__ col3 operator*(col3 a, col3 b) { return mkc(a.r * b.r, a.g * b.g, a.b * b.b); }
__device__ __forceinline__ col3 operator*(col3 a, float s) { return mkc(a.r * s, a.g * s, a.b * s); }
__device__ __forceinline__ col3 operator/(col3 a, float s) { return mkc(a.r / s, a.g / s, a.b / s); }
__device__ __forceinline__ float lumc(col3 c) { return ((0.2126f * c.r) + (0.7152f * c.g) + (0.0722f * c.b)); } // Core.h:89

// ---- counter RNG: replaces MTRandom (Sampling.h:23).  Spec shared with the CPU oracle: DESIGN.md §RNG.
__device__ __host__ __forceinline__ uint32_t rt_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __host__ __forceinline__ uint32_t rt_rng_key(uint64_t seed, uint64_t sample, uint32_t pixel, uint32_t stream) {
    uint32_t k = rt_mix32((uint32_t)seed ^ 0xA511E9B3U);
    k = rt_mix32(k ^ (uint32_t)(seed >> 32));
    k = rt_mix32(k + stream * 0xC2B2AE35U);
    k = rt_mix32(k + (uint32_t)sample * 0x9E3779B9U);
    k = rt_mix32(k ^ (uint32_t)(sample >> 32));
    k = rt_mix32(k + pixel * 0x85EBCA6BU);
    return k;
}
struct Rng { uint32_t key, ctr; };
__device__ __host__ __forceinline__ float rt_rng_next(Rng& r) {
    uint32_t x = rt_mix32(r.key + r.ctr * 0x9E3779B9U);
    r.ctr++;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// ---- traversal stack: the first RT_STACK_LDS entries of each lane live in LDS (lane-interleaved, so a
// push/pop of the whole wave touches 64 consecutive banks); deeper entries spill to a per-launch HBM array.
#ifndef RT_STACK_KEYS
#define RT_STACK_KEYS 0   // measured: the extra LDS plane costs more occupancy than the pop-time cull saves
#endif
// An entry is a child reference plus (RT_STACK_KEYS) the padded entry distance of its box, so a subtree that became
// irrelevant while it waited on the stack is dropped at pop time without fetching its node / triangles.  References
// and keys live in two lane-interleaved 32-bit planes (LDS: [2][RT_STACK_LDS][RT_BLOCK]).
#define RT_STACK_WORDS ((RT_STACK_KEYS ? 2 : 1) * RT_STACK_LDS * RT_BLOCK)
typedef __attribute__((address_space(3))) uint32_t rt_lds_u32;  // explicit LDS address space: keeps pops ds_read, not flat_load
struct TravStack {
    rt_lds_u32* lds;   // &lds_stack[threadIdx.x]
    uint32_t* glob;    // &spill[global thread id], may be null when the scene's bound fits LDS
    uint32_t gstride;  // total threads of the launch
    uint32_t gkeys;    // offset (in words) of the key plane inside the spill array
    __device__ __forceinline__ void put(uint32_t sp, uint32_t v, float key) {
        if (sp < RT_STACK_LDS) {
            lds[sp * RT_BLOCK] = v;
            if (RT_STACK_KEYS) lds[(RT_STACK_LDS + sp) * RT_BLOCK] = __float_as_uint(key);
        } else {
            size_t o = (size_t)(sp - RT_STACK_LDS) * gstride;
            glob[o] = v;
            if (RT_STACK_KEYS) glob[o + gkeys] = __float_as_uint(key);
        }
    }
    __device__ __forceinline__ uint32_t get(uint32_t sp, float& key) const {
        // the LDS read is unconditional (clamped index) so that it stays a ds_read: with one load per branch the
        // compiler sinks both into a single flat_load, which is slower than LDS for the common case
        const uint32_t i = sp < RT_STACK_LDS ? sp : RT_STACK_LDS - 1u;
        uint32_t v = lds[i * RT_BLOCK];
        key = RT_STACK_KEYS ? __uint_as_float(lds[(RT_STACK_LDS + i) * RT_BLOCK]) : -RT_INF;
        if (sp >= RT_STACK_LDS) {
            size_t o = (size_t)(sp - RT_STACK_LDS) * gstride;
            v = glob[o];
            if (RT_STACK_KEYS) key = __uint_as_float(glob[o + gkeys]);
        }
        return v;
    }
};
#define RT_TRAV_STACK_INIT(st, lds_stack, spill4, gtid, gstride, max_stack4)                    \
    TravStack st;                                                                             \
    st.lds = (rt_lds_u32*)(lds_stack + threadIdx.x); st.glob = spill4 ? spill4 + gtid : nullptr; st.gstride = gstride; \
    st.gkeys = (max_stack4 > RT_STACK_LDS ? (max_stack4 - RT_STACK_LDS) : 0u) * gstride;

struct HitRec { float t, u, v; uint32_t rank; };

// Triangle::rayIntersect (Geometry.h:115-128), same expression tree; inv_area is the hoisted 1/Dot(e1 x e2, n).
// a, b, c, e are the four 16-B pieces of a DTri record.
__device__ __forceinline__ bool rt_tri_eval(float4 a, float4 b, float4 c, float4 e, f3 o, f3 dir, float& t, float& u, float& v) {
    f3 n = mk3(c.y, c.z, c.w);
    float denom = dot3(n, dir);
    if (denom == 0) return false;
    t = (e.x - dot3(n, o)) / denom;
    if (t < 0) return false;
    f3 v0 = mk3(a.x, a.y, a.z), v1 = mk3(a.w, b.x, b.y), v2 = mk3(b.z, b.w, c.x);
    f3 p = o + (dir * t);
    f3 e1 = v2 - v1, e2 = v0 - v2;
    u = dot3(cross3(e1, p - v1), n) * e.y;
    if (u < 0 || u > 1.0f) return false;
    v = dot3(cross3(e2, p - v2), n) * e.y;
    if (v < 0 || (u + v) > 1.0f) return false;
    return true;
}
// The same test with the early returns folded into one predicate (same operations, same NaN behaviour: a
// comparison with NaN rejects nothing, as upstream): no dependent second round of loads, no divergent branches.
__device__ __forceinline__ bool rt_tri_eval_all(float4 a, float4 b, float4 c, float4 e, f3 o, f3 dir, float& t, float& u, float& v) {
    f3 n = mk3(c.y, c.z, c.w);
    float denom = dot3(n, dir);
    t = (e.x - dot3(n, o)) / denom;
    f3 v0 = mk3(a.x, a.y, a.z), v1 = mk3(a.w, b.x, b.y), v2 = mk3(b.z, b.w, c.x);
    f3 p = o + (dir * t);
    f3 e1 = v2 - v1, e2 = v0 - v2;
    u = dot3(cross3(e1, p - v1), n) * e.y;
    v = dot3(cross3(e2, p - v2), n) * e.y;
    bool reject = (denom == 0) | (t < 0) | (u < 0) | (u > 1.0f) | (v < 0) | ((u + v) > 1.0f);
    return !reject;
}
__device__ __forceinline__ bool rt_tri_test(const DTri* __restrict__ tp, f3 o, f3 dir, float& t, float& u, float& v) {
    const float4* q = reinterpret_cast<const float4*>(tp);
    return rt_tri_eval(q[0], q[1], q[2], q[3], o, dir, t, u, v);
}

// AABB::rayAABB (Geometry.h:207-229) verbatim, ternary Min/Max + std::max/min: used by the reference-order walk.
__device__ __forceinline__ bool rt_box_exact(f3 lo, f3 hi, f3 o, f3 inv, float& tret) {
    f3 tmin = (lo - o) * inv, tmax = (hi - o) * inv;
    f3 en = mk3(tmin.x < tmax.x ? tmin.x : tmax.x, tmin.y < tmax.y ? tmin.y : tmax.y, tmin.z < tmax.z ? tmin.z : tmax.z);
    f3 ex = mk3(tmin.x > tmax.x ? tmin.x : tmax.x, tmin.y > tmax.y ? tmin.y : tmax.y, tmin.z > tmax.z ? tmin.z : tmax.z);
    float tentry = smaxf(en.x, smaxf(en.y, en.z));
    float texit = sminf(ex.x, sminf(ex.y, ex.z));
    tret = sminf(tentry, texit);
    return (tentry <= texit);
}

// A ray may take the 4-wide walk only if 1/dir is finite and tame on every axis: then no slab product is
// NaN, v_min/v_max agree with the reference's ternaries, and box containment implies test monotonicity.
__device__ __forceinline__ bool rt_ray_is_tame(f3 inv) {
    return (fabsf(inv.x) < 1e18f) && (fabsf(inv.y) < 1e18f) && (fabsf(inv.z) < 1e18f);
}
// Beyond that there are rays no renderer shoots but a caller can pass: NaN / inf / 1e38 in the origin or the direction.  Their
// slab products overflow or are NaN in ways the padded culling was not derived for (it assumes finite arithmetic), so they
// take the reference-order walk, which is upstream's own sequence of operations.  False for NaN (every comparison fails).
__device__ __forceinline__ bool rt_ray_is_moderate(f3 o, f3 dir) {
    return (fabsf(o.x) < 1e18f) && (fabsf(o.y) < 1e18f) && (fabsf(o.z) < 1e18f) &&
           (fabsf(dir.x) < 1e18f) && (fabsf(dir.y) < 1e18f) && (fabsf(dir.z) < 1e18f);
}

// ---- reference-order walk over the binary tree: BVHNode::traverse / traverseVisible (Geometry.h:441-490)
// as an explicit-stack DFS (left before right, no ordering, no distance culling).
template <bool ANY, bool COUNT>
__device__ bool rt_walk_reference(const DScene& S, f3 o, f3 dir, float maxT, HitRec& best, uint32_t* spill, uint32_t gstride,
                                  uint32_t& c_nodes, uint32_t& c_tris) {
    f3 inv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);  // Ray::init (Geometry.h:23)
    uint32_t sp = 0;
    spill[0] = 0; sp = 1;
    while (sp) {
        uint32_t ni = spill[(size_t)(--sp) * gstride];
        const float4* q = reinterpret_cast<const float4*>(S.nodes2 + ni);
        float4 a = q[0], b = q[1], c = q[2];
        if (COUNT) c_nodes++;
        float tb;
        bool pass = rt_box_exact(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), o, inv, tb);
        if (ANY) { if (!pass || tb > maxT) continue; }
        else if (!pass) continue;
        int left = __float_as_int(a.w), right = __float_as_int(b.w);
        if (left < 0 && right < 0) {
            int first = __float_as_int(c.x), count = __float_as_int(c.y);
            for (int k = 0; k < count; k++) {
                float t, u, v;
                if (COUNT) c_tris++;
                if (rt_tri_test(S.tris + first + k, o, dir, t, u, v)) {
                    if (ANY) { if (t < maxT) return false; }
                    else if (t < best.t) { best.t = t; best.u = u; best.v = v; best.rank = (uint32_t)(first + k); }
                }
            }
            continue;
        }
        if (right >= 0) spill[(size_t)(sp++) * gstride] = (uint32_t)right;
        if (left >= 0) spill[(size_t)(sp++) * gstride] = (uint32_t)left;
    }
    return true;
}

// ---- the walk of mode-0 rays that may not take the 4-wide tree (a zero / denormal direction component: 1/dir is
// infinite or huge, slab products can be NaN and box containment no longer implies test monotonicity).  Every node of
// the reference tree on the way down is tested with the reference's own expression (rt_box_exact), so the candidate
// set is the reference's by construction; what this walk adds over rt_walk_reference is near-first order and a
// conservative cull that only looks at the axes where 1/dir is tame (a box is skipped when, grown by the same slack
// as in the wide walk, it lies entirely behind the origin or entirely beyond the best hit on those axes).  Without
// it one such ray pierces thousands of boxes of a 1 M-triangle scene unculled and, alone in its wave, sets the
// duration of the whole launch: 0.6-2 ms per k_extend launch before, measured.
template <bool ANY, bool COUNT>
__device__ bool rt_walk_exact(const DScene& S, f3 o, f3 dir, float maxT, HitRec& best, uint32_t* spill, uint32_t gstride,
                              uint32_t& c_nodes, uint32_t& c_tris) {
    const f3 inv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);  // Ray::init (Geometry.h:23)
    const bool tx = fabsf(inv.x) < 1e18f, ty = fabsf(inv.y) < 1e18f, tz = fabsf(inv.z) < 1e18f;
    const float scale = fmaxf(S.abs_max, fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))));
    const float delta = scale * 6.103515625e-05f;
    const f3 pad = mk3(delta * fabsf(inv.x), delta * fabsf(inv.y), delta * fabsf(inv.z));
    auto test = [&](uint32_t ni, float& key) -> bool {
        const float4* q = reinterpret_cast<const float4*>(S.nodes2 + ni);
        float4 a = q[0], b = q[1];
        f3 lo = mk3(a.x, a.y, a.z), hi = mk3(b.x, b.y, b.z);
        if (COUNT) c_nodes++;
        float tb;
        bool pass = rt_box_exact(lo, hi, o, inv, tb);
        if (ANY) pass = pass && !(tb > maxT);
        if (!pass) return false;
        float lo_ = -RT_INF, hi_ = RT_INF;
        if (tx) { float t0 = (lo.x - o.x) * inv.x, t1 = (hi.x - o.x) * inv.x; lo_ = fmaxf(lo_, fminf(t0, t1) - pad.x); hi_ = fminf(hi_, fmaxf(t0, t1) + pad.x); }
        if (ty) { float t0 = (lo.y - o.y) * inv.y, t1 = (hi.y - o.y) * inv.y; lo_ = fmaxf(lo_, fminf(t0, t1) - pad.y); hi_ = fminf(hi_, fmaxf(t0, t1) + pad.y); }
        if (tz) { float t0 = (lo.z - o.z) * inv.z, t1 = (hi.z - o.z) * inv.z; lo_ = fmaxf(lo_, fminf(t0, t1) - pad.z); hi_ = fminf(hi_, fmaxf(t0, t1) + pad.z); }
        if (hi_ < 0.0f) return false;
        if (!ANY && lo_ > best.t) return false;
        key = lo_;
        return true;
    };
    float key;
    if (!test(0u, key)) return true;
    uint32_t sp = 0, cur = 0;
    for (;;) {
        const float4* q = reinterpret_cast<const float4*>(S.nodes2 + cur);
        const int left = __float_as_int(q[0].w), right = __float_as_int(q[1].w);
        if (left < 0 && right < 0) {
            const float4 c = q[2];
            const int first = __float_as_int(c.x), count = __float_as_int(c.y);
            for (int k = 0; k < count; k++) {
                float t, u, v;
                if (COUNT) c_tris++;
                if (rt_tri_test(S.tris + first + k, o, dir, t, u, v)) {
                    if (ANY) { if (t < maxT) return false; }
                    else if (t < best.t || (t == best.t && (uint32_t)(first + k) < best.rank)) { best.t = t; best.u = u; best.v = v; best.rank = (uint32_t)(first + k); }
                }
            }
        } else {
            float kl = 0, kr = 0;
            const bool pl = left >= 0 && test((uint32_t)left, kl), pr = right >= 0 && test((uint32_t)right, kr);
            if (pl && pr) {
                const bool swap = kr < kl;
                spill[(size_t)(sp++) * gstride] = (uint32_t)(swap ? left : right);
                cur = (uint32_t)(swap ? right : left);
                continue;
            }
            if (pl) { cur = (uint32_t)left; continue; }
            if (pr) { cur = (uint32_t)right; continue; }
        }
        if (!sp) break;
        cur = spill[(size_t)(--sp) * gstride];
    }
    return true;
}

#define RT_CSWAP(ka, ra, kb, rb)                                  \
    {                                                             \
        bool s_ = kb < ka;                                        \
        float tk_ = s_ ? kb : ka; kb = s_ ? ka : kb; ka = tk_;    \
        uint32_t tr_ = s_ ? rb : ra; rb = s_ ? ra : rb; ra = tr_; \
    }

// ---- 4-wide ordered walk, one step at a time.  The per-lane state lives in a Walk so that a wave can keep
// stepping the rays it has while lanes whose ray finished are handed new ones (rt_persistent_walk below).
//
// Closest hit: children sorted near-to-far, subtrees skipped when their box, padded by `pad` (a spatial slack of
// ~1000 ulp of the largest coordinate, expressed in t per axis), cannot hold a hit in [0, best.t].  Any hit: the
// reference's box rule (tentry <= texit && !(tentry > maxT)) plus a padded "entirely behind the origin" skip, any
// order.  Skipping only uses the padded interval; whether a child is *entered* is always the reference's own
// test on the reference's own box.
// Slack of the conservative culls, as a fraction of the largest coordinate in play (scene extent, ray origin): a box is grown
// by delta = RT_PAD_SCALE * M before the walk may skip it.  What the slack has to cover: an accepted
// hit's point p = o + dir * t lies inside its triangle -- hence inside every box around it -- up to the rounding of the
// plane equation, of p and of the barycentric test, together < 2^-19 M whatever the triangle's shape; expressing the box
// planes in t adds 8 * 2^-24 M |1/dir|.  2^-17 M (128 ulp) is three times that bound.  Round 1 used 2^-14: for rays nearly
// parallel to an axis the pad in t is delta * |1/dir|, and the eight-fold larger pad kept whole subtrees behind the origin
// and beyond the best hit alive (+4.9 % Mrays/s on the bench scene from this constant and the per-axis any-hit pad below).
#ifndef RT_PAD_SCALE
#define RT_PAD_SCALE 7.62939453125e-06f   // 2^-17
#endif
// Any-hit walks test "entirely behind the origin" with the per-axis pad as well (3 adds + 2 mins per child more than round 1's
// scalar delta * max |1/dir|, which let axis-parallel shadow rays walk everything behind them: k_connect -6 %).
struct Walk {
    f3 o, dir, inv;
    f3 pad;        // the slack in t per axis: delta * |1/dir|
    float maxT;        // any hit
    HitRec best;       // closest hit so far; any hit: best.rank = the blocker that was found
    uint32_t cur, sp;
    uint32_t pend;     // rt_vote_step: the leaf set aside while the lane walks on (RT_REF_EMPTY: none)
    uint32_t mx, my, mz;   // all ones where 1/dir is negative on that axis, else 0: picks a node's hi planes as the near planes
    bool occluded;
    bool wide;         // this item went through the 4-wide walk: its answer still has to pass rt_settle
};
__device__ __forceinline__ void rt_walk_init(const DScene& S, Walk& w, f3 o, f3 dir, f3 inv, float maxT) {
    float scale = fmaxf(S.abs_max, fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))));
    float delta = scale * RT_PAD_SCALE;
    w.o = o; w.dir = dir; w.inv = inv;
    w.pad = mk3(delta * fabsf(inv.x), delta * fabsf(inv.y), delta * fabsf(inv.z));
    w.maxT = maxT;
    w.mx = inv.x < 0.0f ? 0xFFFFFFFFu : 0u; w.my = inv.y < 0.0f ? 0xFFFFFFFFu : 0u; w.mz = inv.z < 0.0f ? 0xFFFFFFFFu : 0u;
    w.cur = S.root_ref; w.sp = 0; w.pend = RT_REF_EMPTY; w.occluded = false; w.wide = true;
}
// Box tests of one 4-wide node: pushes the passing children but the nearest, which becomes w.cur (returns true).
//
// The tests are CONSERVATIVE ONLY.  Upstream's answer is min (t, DFS rank) over the triangles that pass rayIntersect AND whose
// ancestor boxes all pass rayAABB; for a tame ray the box test is monotone under containment (section 3 of DESIGN.md), so the
// second condition is "the triangle's gate box passes" (DScene::gate).  The walk may therefore use any boxes and any
// arithmetic that never skip a true hit nearer than the best so far -- here the child boxes on a 256-step grid of the node's
// box, rounded outwards, grown by the slack `pad`, evaluated with one fma per plane -- as long as the hit it ends with is then
// checked against its gate box with upstream's own expression (rt_settle): if that check fails, which takes a hit within
// rounding of its gate box's surface, the ray is walked again by rt_walk_exact.
//   plane in t:  (org + q * s - o) * inv -+ pad  =  q * (s * inv) + ((org - o) * inv -+ pad)  =  fma(q, A, B)
// Rounding: org - o, B and the final fma round once each, <= 3 * 2^-24 * 3 M |1/dir|, inside the slack's budget (rt_walk_init).
// (Writing the slab arithmetic on child pairs as v_pk_* measured slower in round 1; the build passes -fno-slp-vectorize.)
#ifndef RT_POP_FAST
#define RT_POP_FAST 1
#endif
__device__ __forceinline__ uint32_t rt_bfi(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }   // v_bfi_b32
__device__ __forceinline__ float rt_ub0(uint32_t v) { return (float)(v & 255u); }            // v_cvt_f32_ubyte0..3
__device__ __forceinline__ float rt_ub1(uint32_t v) { return (float)((v >> 8) & 255u); }
__device__ __forceinline__ float rt_ub2(uint32_t v) { return (float)((v >> 16) & 255u); }
__device__ __forceinline__ float rt_ub3(uint32_t v) { return (float)(v >> 24); }
template <bool ANY>
__device__ __forceinline__ bool rt_node_eval(const DScene& S, Walk& w, TravStack& st) {
    const char* nb = reinterpret_cast<const char*>(S.nodesq);
    const uint32_t base = w.cur << 6;   // < 2^31: the host caps the wide tree at 2^25 nodes
    const float4 h0 = *reinterpret_cast<const float4*>(nb + base);
    const uint4 h1 = *reinterpret_cast<const uint4*>(nb + (base + 16u));
    const uint4 h2 = *reinterpret_cast<const uint4*>(nb + (base + 32u));
    const uint4 ch = *reinterpret_cast<const uint4*>(nb + (base + 48u));
    const float ax = h0.w * w.inv.x, ay = __uint_as_float(h1.x) * w.inv.y, az = __uint_as_float(h1.y) * w.inv.z;
    const float dx = h0.x - w.o.x, dy = h0.y - w.o.y, dz = h0.z - w.o.z;
    const float blx = __builtin_fmaf(dx, w.inv.x, -w.pad.x), bhx = __builtin_fmaf(dx, w.inv.x, w.pad.x);
    const float bly = __builtin_fmaf(dy, w.inv.y, -w.pad.y), bhy = __builtin_fmaf(dy, w.inv.y, w.pad.y);
    const float blz = __builtin_fmaf(dz, w.inv.z, -w.pad.z), bhz = __builtin_fmaf(dz, w.inv.z, w.pad.z);
    const uint32_t nx = rt_bfi(w.mx, h2.y, h1.z), fx = rt_bfi(w.mx, h1.z, h2.y);
    const uint32_t ny = rt_bfi(w.my, h2.z, h1.w), fy = rt_bfi(w.my, h1.w, h2.z);
    const uint32_t nz = rt_bfi(w.mz, h2.w, h2.x), fz = rt_bfi(w.mz, h2.x, h2.w);
    const float limit = ANY ? w.maxT : w.best.t;
    float k0, k1, k2, k3;
#define RT_SLOT(UB, REF, KEY)                                                                               \
    {                                                                                                      \
        float lo_ = fmaxf(__builtin_fmaf(UB(nx), ax, blx), fmaxf(__builtin_fmaf(UB(ny), ay, bly), __builtin_fmaf(UB(nz), az, blz))); \
        float hi_ = fminf(__builtin_fmaf(UB(fx), ax, bhx), fminf(__builtin_fmaf(UB(fy), ay, bhy), __builtin_fmaf(UB(fz), az, bhz))); \
        bool pass = (fmaxf(lo_, 0.0f) <= fminf(hi_, limit)) && (REF != RT_REF_EMPTY);                      \
        KEY = pass ? lo_ : RT_INF;                                                                         \
    }
    RT_SLOT(rt_ub0, ch.x, k0)
    RT_SLOT(rt_ub1, ch.y, k1)
    RT_SLOT(rt_ub2, ch.z, k2)
    RT_SLOT(rt_ub3, ch.w, k3)
#undef RT_SLOT
    uint32_t r0 = ch.x, r1 = ch.y, r2 = ch.z, r3 = ch.w;
    if (!ANY) {
#ifndef RT_SORT_SWAPS
#define RT_SORT_SWAPS 5
#endif
        RT_CSWAP(k0, r0, k1, r1) RT_CSWAP(k2, r2, k3, r3) RT_CSWAP(k0, r0, k2, r2)   // the nearest child is in slot 0 now
        if (RT_SORT_SWAPS >= 4) RT_CSWAP(k1, r1, k3, r3)
        if (RT_SORT_SWAPS >= 5) RT_CSWAP(k1, r1, k2, r2)
    } else {  // no order needed: compact the passing children to the front
        if (!(k2 < RT_INF)) { k2 = k3; r2 = r3; k3 = RT_INF; }
        if (!(k1 < RT_INF)) { k1 = k2; r1 = r2; k2 = k3; r2 = r3; k3 = RT_INF; }
        if (!(k0 < RT_INF)) { k0 = k1; r0 = r1; k1 = k2; r1 = r2; k2 = k3; r2 = r3; k3 = RT_INF; }
    }
    if (!RT_STACK_KEYS && __ballot(w.sp + 3u > (uint32_t)RT_STACK_LDS) == 0) {
        // every lane has room in LDS for three more: store unconditionally, advance by the predicate
        st.lds[w.sp * RT_BLOCK] = r3; w.sp += (k3 < RT_INF) ? 1u : 0u;
        st.lds[w.sp * RT_BLOCK] = r2; w.sp += (k2 < RT_INF) ? 1u : 0u;
        st.lds[w.sp * RT_BLOCK] = r1; w.sp += (k1 < RT_INF) ? 1u : 0u;
    } else {
        if (k3 < RT_INF) st.put(w.sp++, r3, k3);
        if (k2 < RT_INF) st.put(w.sp++, r2, k2);
        if (k1 < RT_INF) st.put(w.sp++, r1, k1);
    }
    if (k0 < RT_INF) { w.cur = r0; return true; }
    return false;
}
// pop, dropping entries whose (padded) entry distance now lies beyond the best hit; false = stack empty
template <bool ANY>
__device__ __forceinline__ bool rt_pop(Walk& w, TravStack& st) {
    if (!RT_STACK_KEYS && RT_POP_FAST) {
        if (w.sp == 0) return false;
        --w.sp;
        float key;
        w.cur = st.get(w.sp, key);
        return true;
    }
    while (w.sp) {
        float key;
        uint32_t c = st.get(--w.sp, key);
        if (ANY || !(key > w.best.t)) { w.cur = c; return true; }
    }
    return false;
}
// One node or one leaf, then a pop.  Returns true when the walk is over (stack empty, or ANY found a blocker).
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool rt_wide_step(const DScene& S, Walk& w, TravStack& st, uint32_t& c_nodes, uint32_t& c_tris) {
    const uint32_t cur = w.cur;
    if (!(cur & RT_REF_LEAF)) {
        if (COUNT) c_nodes++;
        if (rt_node_eval<ANY>(S, w, st)) return false;
    } else {
        uint32_t first = cur & RT_LEAF_FIRST_MASK, cnt = ((cur >> 27) & 15u) + 1u;
        for (uint32_t k = 0; k < cnt; k++) {
            float t, u, v;
            if (COUNT) c_tris++;
            if (rt_tri_test(S.tris + first + k, w.o, w.dir, t, u, v)) {
                if (ANY) { if (t < w.maxT) { w.occluded = true; w.best.rank = first + k; return true; } }
                else if (t < w.best.t || (t == w.best.t && (first + k) < w.best.rank)) {
                    w.best.t = t; w.best.u = u; w.best.v = v; w.best.rank = first + k;
                }
            }
        }
    }
    return !rt_pop<ANY>(w, st);
}

// ---- the wave-voted step of rt_persistent_walk.  A wave whose lanes are part in nodes and part in leaves pays
// for both code paths in rt_wide_step, each with a fraction of the lanes (PMC: ~26 % of VALU lanes live).  Here
// the wave runs ONE of the two per step -- box tests for the lanes standing on a node, or one triangle for the
// lanes standing in a leaf -- whichever serves more lanes per instruction; the others wait a step.
// All 64 lanes call it (busy = this lane holds a walk).  Returns true when this lane's walk is over.
#ifndef RT_VOTE_NUM
#define RT_VOTE_NUM 1    // nodes go when (lanes on a node) * RT_VOTE_NUM >= (lanes with a leaf) * RT_VOTE_DEN.  With postponed leaves 1:1 is
#define RT_VOTE_DEN 1    // best (3:2 -1.1 %, 1:2 -1.6 %; "triangles only once 16 / 24 / 32 lanes hold a leaf" -5 % / -4 % / -10 %)
#endif
// A lane that reaches a leaf does not stop walking: the leaf is set aside in w.pend and the lane pops on, so that it has work
// for either kind of step -- box tests for its current node, one triangle of its postponed leaf (Aila & Laine's speculative
// traversal).  What it costs: the nodes walked meanwhile were not culled by the hit the postponed leaf may hold.  A second
// leaf met while one is postponed waits in w.cur until the first is done.
#ifndef RT_POSTPONE
#define RT_POSTPONE 1
#endif
__device__ __forceinline__ bool rt_ref_is_node(uint32_t r) { return !(r & RT_REF_LEAF); }
__device__ __forceinline__ bool rt_ref_is_leaf(uint32_t r) { return (r & RT_REF_LEAF) && r != RT_REF_EMPTY; }
template <bool ANY>
__device__ __forceinline__ void rt_pop_or_none(Walk& w, TravStack& st) { if (!rt_pop<ANY>(w, st)) w.cur = RT_REF_EMPTY; }
// cur is a node, a leaf that has to wait, or none; a leaf moves to pend as soon as pend is free
template <bool ANY>
__device__ __forceinline__ void rt_settle_refs(Walk& w, TravStack& st) {
    if (rt_ref_is_leaf(w.cur) && w.pend == RT_REF_EMPTY) { w.pend = w.cur; rt_pop_or_none<ANY>(w, st); }
}
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool rt_vote_step(const DScene& S, Walk& w, bool busy, TravStack& st, uint32_t& c_nodes, uint32_t& c_tris) {
#if RT_POSTPONE
    const bool innode = busy && rt_ref_is_node(w.cur);
    const bool inleaf = busy && w.pend != RT_REF_EMPTY;
    const uint32_t nl = (uint32_t)__popcll(__ballot(inleaf)), nn = (uint32_t)__popcll(__ballot(innode));
    if (nn * (uint32_t)RT_VOTE_NUM >= nl * (uint32_t)RT_VOTE_DEN) {
#ifdef RT_DBG_WAVESTEPS
        if (COUNT) c_nodes++;
#endif
        if (!innode) return false;
#ifndef RT_DBG_WAVESTEPS
        if (COUNT) c_nodes++;
#endif
        if (!rt_node_eval<ANY>(S, w, st)) rt_pop_or_none<ANY>(w, st);
        rt_settle_refs<ANY>(w, st);
        return w.cur == RT_REF_EMPTY && w.pend == RT_REF_EMPTY;
    }
#ifdef RT_DBG_WAVESTEPS
    if (COUNT) c_tris++;
#endif
    if (!inleaf) return false;
    const uint32_t leaf = w.pend;
    const uint32_t tri = leaf & RT_LEAF_FIRST_MASK, left = (leaf >> 27) & 15u;  // triangles left in the leaf after this one
    const float4* q = reinterpret_cast<const float4*>(S.tris + tri);
    float4 a = q[0], b = q[1], c = q[2], e = q[3];
    float t, u, v;
#ifndef RT_DBG_WAVESTEPS
    if (COUNT) c_tris++;
#endif
    if (rt_tri_eval_all(a, b, c, e, w.o, w.dir, t, u, v)) {
        if (ANY) { if (t < w.maxT) { w.occluded = true; w.best.rank = tri; return true; } }
        else if (t < w.best.t || (t == w.best.t && tri < w.best.rank)) { w.best.t = t; w.best.u = u; w.best.v = v; w.best.rank = tri; }
    }
    if (left) { w.pend = RT_REF_LEAF | ((left - 1u) << 27) | (tri + 1u); return false; }
    w.pend = RT_REF_EMPTY;
    rt_settle_refs<ANY>(w, st);
    return w.cur == RT_REF_EMPTY && w.pend == RT_REF_EMPTY;
#else
    const uint32_t cur = w.cur;
    const bool inleaf = busy && (cur & RT_REF_LEAF) != 0;
    const bool innode = busy && !(cur & RT_REF_LEAF);
    const uint32_t nl = (uint32_t)__popcll(__ballot(inleaf)), nn = (uint32_t)__popcll(__ballot(innode));
    if (nn * (uint32_t)RT_VOTE_NUM >= nl * (uint32_t)RT_VOTE_DEN) {
        if (!innode) return false;
        if (COUNT) c_nodes++;
        if (rt_node_eval<ANY>(S, w, st)) return false;
        return !rt_pop<ANY>(w, st);
    }
    if (!inleaf) return false;
    const uint32_t tri = cur & RT_LEAF_FIRST_MASK, left = (cur >> 27) & 15u;  // triangles left in the leaf after this one
    const float4* q = reinterpret_cast<const float4*>(S.tris + tri);
    float4 a = q[0], b = q[1], c = q[2], e = q[3];
    float t, u, v;
    if (COUNT) c_tris++;
    if (rt_tri_eval_all(a, b, c, e, w.o, w.dir, t, u, v)) {
        if (ANY) { if (t < w.maxT) { w.occluded = true; w.best.rank = tri; return true; } }
        else if (t < w.best.t || (t == w.best.t && tri < w.best.rank)) { w.best.t = t; w.best.u = u; w.best.v = v; w.best.rank = tri; }
    }
    if (left) { w.cur = RT_REF_LEAF | ((left - 1u) << 27) | (tri + 1u); return false; }
    return !rt_pop<ANY>(w, st);
#endif
}

// Scene::traverse (Scene.h:118), set-up half: returns true when the result is already known (w.best), false when
// the lane now holds a 4-wide walk to step.  best.t == FLT_MAX <=> miss.
template <bool COUNT>
__device__ __forceinline__ bool rt_closest_begin(const DScene& S, f3 o, f3 dir, int mode, Walk& w, uint32_t* spill2, uint32_t gstride,
                                                 uint32_t& cn, uint32_t& ct, uint32_t& cexact) {
    w.best.t = RT_FLT_MAX; w.best.u = 0; w.best.v = 0; w.best.rank = 0xFFFFFFFFu;
    w.wide = false;
    if (S.root_ref == RT_REF_EMPTY) return true;
    // dir == (0,0,0) (what pathTrace shoots after a BSDF returned pdf 0 / wi 0, e.g. Materials.h:319-324): Dot(n, dir)
    // is 0 for every triangle, so Triangle::rayIntersect rejects all of them (Geometry.h:118) -- a miss, whatever
    // boxes the reference would have walked through on the way.
    if (mode == 0 && dir.x == 0.0f && dir.y == 0.0f && dir.z == 0.0f) return true;
    f3 inv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    const bool moderate = rt_ray_is_moderate(o, dir);
    if (mode == 0 && moderate && rt_ray_is_tame(inv)) { rt_walk_init(S, w, o, dir, inv, 0.0f); return false; }
    if (COUNT) cexact++;
    if (mode == 0 && moderate) rt_walk_exact<false, COUNT>(S, o, dir, 0.0f, w.best, spill2, gstride, cn, ct);
    else rt_walk_reference<false, COUNT>(S, o, dir, 0.0f, w.best, spill2, gstride, cn, ct);
    return true;
}
// Scene::visible (Scene.h:169-178), set-up half: true when `visible` is already known.
template <bool COUNT>
__device__ __forceinline__ bool rt_visible_begin(const DScene& S, f3 p1, f3 p2, int mode, Walk& w, bool& visible, uint32_t* spill2,
                                                 uint32_t gstride, uint32_t& cn, uint32_t& ct, uint32_t& cexact) {
    f3 dir = p2 - p1;
    float maxT = len3(dir) - (2.0f * RT_EPSILON);
    dir = norm3(dir);
    f3 o = p1 + (dir * RT_EPSILON);
    visible = true;
    w.wide = false;
    if (S.root_ref == RT_REF_EMPTY) return true;
    f3 inv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    const bool moderate = rt_ray_is_moderate(o, dir);
    if (mode == 0 && moderate && rt_ray_is_tame(inv)) { rt_walk_init(S, w, o, dir, inv, maxT); return false; }
    if (COUNT) cexact++;
    HitRec dummy;
    visible = (mode == 0 && moderate) ? rt_walk_exact<true, COUNT>(S, o, dir, maxT, dummy, spill2, gstride, cn, ct)
                        : rt_walk_reference<true, COUNT>(S, o, dir, maxT, dummy, spill2, gstride, cn, ct);
    return true;
}

// ---- the check that makes the conservative wide walk exact (see rt_node_eval).  Called once per item, on its final answer,
// before it is retired: the reported triangle's gate box must pass upstream's own box test for this ray (any hit: with the
// `tentry > maxT` rule of traverseVisible, Geometry.h:477).  An earlier, since replaced candidate needs no check: it only ever
// culled what lies beyond it.  If the check fails the answer is recomputed by rt_walk_exact, which enters a subtree only
// through upstream's test on upstream's boxes.
template <bool ANY, bool COUNT>
__device__ __forceinline__ void rt_settle(const DScene& S, Walk& w, bool& visible, uint32_t* spill2, uint32_t gstride,
                                          uint32_t& cn, uint32_t& ct, uint32_t& crefused) {
    if (!w.wide) return;
    w.wide = false;
    if (ANY ? visible : !(w.best.t < RT_FLT_MAX)) return;   // nothing was accepted
    const uint32_t rank = w.best.rank < S.n_tris ? w.best.rank : 0u;   // never read outside the array, whatever the walk left here
    const float4* g = reinterpret_cast<const float4*>(S.gate + rank);
    const float4 a = g[0], b = g[1];
    float tb;
    bool ok = rt_box_exact(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), w.o, w.inv, tb) && rank == w.best.rank;
    if (ANY) ok = ok && !(tb > w.maxT);
    if (ok) return;
    if (COUNT) crefused++;
    if (ANY) { HitRec dummy; visible = rt_walk_exact<true, COUNT>(S, w.o, w.dir, w.maxT, dummy, spill2, gstride, cn, ct); }
    else {
        w.best.t = RT_FLT_MAX; w.best.u = 0; w.best.v = 0; w.best.rank = 0xFFFFFFFFu;
        rt_walk_exact<false, COUNT>(S, w.o, w.dir, 0.0f, w.best, spill2, gstride, cn, ct);
    }
}

// ---- persistent traversal with lane refill (Aila & Laine's "persistent while-while" idea at wave64 width).
// With one ray per lane per 64-ray chunk a wave runs until its slowest ray is done; PMC counters put the average
// number of live lanes per VALU instruction at ~11 of 64 that way.  Here a wave keeps a private pool of item
// indices (one work atomic per grant of 64-256 items); whenever at least RT_REFILL_IDLE lanes are idle the finished
// lanes retire their result together and every idle lane is handed the next item of the pool.
//   Job:    bool load(uint32_t item, f3& a, f3& b)   -- closest: origin, direction; any: p1, p2.  false = nothing to trace
//           void retire(uint32_t item, const Walk& w, bool visible)
//   Source: bool next(uint32_t& begin, uint32_t& end) -- wave-uniform grant of items [begin, end), false when drained
// Which lane traces which item only affects speed: every item is loaded, traced and retired by exactly one lane.
// Steps between two looks at the idle-lane count: the look (a ballot, a count, the refill branch) is a fixed cost per step that
// buys nothing while fewer than RT_REFILL_IDLE lanes are idle.  1 -> 2 / 3 / 4 / 6 / 8: +1.3 / 1.9 / 2.3 / 2.5 / 1.9 % Mrays/s.
#ifndef RT_STEPS_PER_CHECK
#define RT_STEPS_PER_CHECK 4
#endif
#ifndef RT_REFILL_IDLE
#define RT_REFILL_IDLE 32
#endif
#ifndef RT_REFILL_IDLE_ANY
#define RT_REFILL_IDLE_ANY RT_REFILL_IDLE
#endif

// Tail: once the source is drained a wave is left with its slowest walks -- a few rays that take 1000+ steps where the mean
// is 50 -- and the launch lasts until they end.  From then on idle lanes take work from busy ones: every RT_STEAL_EVERY steps
// a lane that holds a walk hands the top entry of its stack (the subtree it would enter next) to a free lane, which gets a
// copy of the ray and the best hit so far and walks that subtree on its own.  The parts of one item are folded together when
// the wave is done: closest hit = the minimum by (t, rank), the order the single walk applies; any hit = OR.  A part can only
// miss hits that its copy of best.t already beats, so the result is the single walk's, bit for bit.
#ifndef RT_TAIL_STEAL
#define RT_TAIL_STEAL 1
#endif
#ifndef RT_STEAL_EVERY
#define RT_STEAL_EVERY 4
#endif
__device__ __forceinline__ float rt_shfl_f(float v, uint32_t src) { return __shfl(v, (int)src); }
__device__ __forceinline__ f3 rt_shfl_f3(f3 v, uint32_t src) { return mk3(__shfl(v.x, (int)src), __shfl(v.y, (int)src), __shfl(v.z, (int)src)); }

template <bool ANY, bool COUNT, class Job, class Source>
__device__ __forceinline__ void rt_persistent_walk(const DScene& S, Job& job, Source& src, int mode, TravStack& st, uint32_t* spill2,
                                                   uint32_t gstride, uint32_t& cn, uint32_t& ct, uint32_t& cx, DCounters* counters) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t c_slots = 0, c_busy = 0, c_mine = 0, c_max = 0, c_refused = 0;
    Walk w;
    w.wide = false;
    uint32_t item = 0;
    bool busy = false, done = false, visible = true;
    uint32_t pb = 0, pe = 0;   // the wave's pool of granted items (wave-uniform)
    bool drained = false;      // wave-uniform
    for (;;) {
        unsigned long long idle = __ballot(!busy);
        uint32_t nidle = (uint32_t)__popcll(idle);
        if (drained ? (nidle == 64u) : (nidle >= (uint32_t)(ANY ? RT_REFILL_IDLE_ANY : RT_REFILL_IDLE))) {
            if (done) { rt_settle<ANY, COUNT>(S, w, visible, spill2, gstride, cn, ct, c_refused); job.retire(item, w, visible); done = false; }
            if (drained) break;
            if (pb >= pe) {
                if (!src.next(pb, pe)) { drained = true; if (RT_TAIL_STEAL) break; continue; }
            }
            uint32_t take = pe - pb;
            take = take < nidle ? take : nidle;
            uint32_t my = (uint32_t)__popcll(idle & ((1ULL << lane) - 1ULL));
            if (!busy && my < take) {
                item = pb + my;
                f3 a, b;
                if (job.load(item, a, b)) {
                    bool fin = ANY ? rt_visible_begin<COUNT>(S, a, b, mode, w, visible, spill2, gstride, cn, ct, cx)
                                   : rt_closest_begin<COUNT>(S, a, b, mode, w, spill2, gstride, cn, ct, cx);
                    busy = !fin; done = fin;
                    if (COUNT) c_mine = 0;
                }
            }
            pb += take;
            continue;  // re-count: short grants and dead items leave lanes idle
        }
#pragma unroll
        for (int rep_ = 0; rep_ < RT_STEPS_PER_CHECK; rep_++) {
            if (COUNT) { c_slots++; c_busy += busy ? 1u : 0u; c_mine += busy ? 1u : 0u; c_max = c_mine > c_max ? c_mine : c_max; }
            if (rt_vote_step<ANY, COUNT>(S, w, busy, st, cn, ct)) {
                busy = false; done = true;
                if (ANY) visible = !w.occluded;
            }
        }
    }
    if (RT_TAIL_STEAL) {
        // every lane is busy (walking) or free here: finished lanes retired above
        unsigned long long split = 0;   // lanes whose result is one part of an item (wave-uniform)
        for (uint32_t round = 0;; round++) {
            const unsigned long long bm = __ballot(busy);
            if (!bm) break;
            if ((round % (uint32_t)RT_STEAL_EVERY) == 0u) {
                const unsigned long long fm = __ballot(!busy && !done);      // free: no walk, no result to keep
                const unsigned long long dm = __ballot(busy && w.sp >= 1u);  // donors: a subtree waiting on the stack
                uint32_t n = (uint32_t)__popcll(fm), nd = (uint32_t)__popcll(dm);
                n = n < nd ? n : nd;
                if (n) {
                    const uint32_t lt_f = (uint32_t)__popcll(fm & ((1ULL << lane) - 1ULL)), lt_d = (uint32_t)__popcll(dm & ((1ULL << lane) - 1ULL));
                    const bool thief = ((fm >> lane) & 1ULL) && lt_f < n, donor = ((dm >> lane) & 1ULL) && lt_d < n;
                    uint32_t from = lane;   // a thief's donor: the lt_f-th set bit of dm
                    unsigned long long bits = dm;
                    for (uint32_t j = 0; j < n; j++) {
                        const uint32_t l = (uint32_t)__ffsll((long long)bits) - 1u;
                        if (thief && lt_f == j) from = l;
                        bits &= bits - 1ULL;
                    }
                    uint32_t give = 0;
                    if (donor) { float key; give = st.get(--w.sp, key); }
                    // the ray and what is known about it so far travel by shuffle; non-thieves read themselves
                    const f3 o_ = rt_shfl_f3(w.o, from), d_ = rt_shfl_f3(w.dir, from), i_ = rt_shfl_f3(w.inv, from), p_ = rt_shfl_f3(w.pad, from);
                    const float mt_ = rt_shfl_f(w.maxT, from);
                    const float bt_ = rt_shfl_f(w.best.t, from), bu_ = rt_shfl_f(w.best.u, from), bv_ = rt_shfl_f(w.best.v, from);
                    const uint32_t br_ = (uint32_t)__shfl((int)w.best.rank, (int)from);
                    const uint32_t sx_ = (uint32_t)__shfl((int)w.mx, (int)from), sy_ = (uint32_t)__shfl((int)w.my, (int)from), sz_ = (uint32_t)__shfl((int)w.mz, (int)from);
                    const uint32_t it_ = (uint32_t)__shfl((int)item, (int)from), cur_ = (uint32_t)__shfl((int)give, (int)from);
                    if (thief) {
                        w.o = o_; w.dir = d_; w.inv = i_; w.pad = p_; w.maxT = mt_;
                        w.best.t = bt_; w.best.u = bu_; w.best.v = bv_; w.best.rank = br_;
                        w.mx = sx_; w.my = sy_; w.mz = sz_; w.cur = cur_; w.sp = 0; w.pend = RT_REF_EMPTY; w.occluded = false; w.wide = true;
                        if (RT_POSTPONE && rt_ref_is_leaf(w.cur)) { w.pend = w.cur; w.cur = RT_REF_EMPTY; }
                        item = it_; busy = true; visible = true;
                        if (COUNT) c_mine = 0;
                    }
                    split |= __ballot(thief || donor);
                }
            }
            if (COUNT) { c_slots++; c_busy += busy ? 1u : 0u; c_mine += busy ? 1u : 0u; c_max = c_mine > c_max ? c_mine : c_max; }
            if (rt_vote_step<ANY, COUNT>(S, w, busy, st, cn, ct)) {
                busy = false; done = true;
                if (ANY) visible = !w.occluded;
            }
        }
        // results that were never part of a split retire together
        if (done && !((split >> lane) & 1ULL)) { rt_settle<ANY, COUNT>(S, w, visible, spill2, gstride, cn, ct, c_refused); job.retire(item, w, visible); done = false; }
        unsigned long long hm = __ballot(done);
        while (hm) {
            const uint32_t leader = (uint32_t)__ffsll((long long)hm) - 1u;
            const uint32_t it = (uint32_t)__shfl((int)item, (int)leader);
            const unsigned long long same = __ballot(done && item == it);
            if (ANY) {
                const unsigned long long occ = __ballot(done && item == it && !visible);   // parts that found a blocker
                if (occ) {
                    const uint32_t r2 = (uint32_t)__shfl((int)w.best.rank, (int)((uint32_t)__ffsll((long long)occ) - 1u));
                    if (lane == leader) { visible = false; w.best.rank = r2; }   // rt_settle checks that blocker's gate box
                }
            } else {
                unsigned long long rest = same & ~(1ULL << leader);
                while (rest) {
                    const uint32_t l = (uint32_t)__ffsll((long long)rest) - 1u;
                    const float t2 = rt_shfl_f(w.best.t, l), u2 = rt_shfl_f(w.best.u, l), v2 = rt_shfl_f(w.best.v, l);
                    const uint32_t r2 = (uint32_t)__shfl((int)w.best.rank, (int)l);
                    if (lane == leader && (t2 < w.best.t || (t2 == w.best.t && r2 < w.best.rank))) { w.best.t = t2; w.best.u = u2; w.best.v = v2; w.best.rank = r2; }
                    rest &= rest - 1ULL;
                }
            }
            if (lane == leader) { w.wide = true; rt_settle<ANY, COUNT>(S, w, visible, spill2, gstride, cn, ct, c_refused); job.retire(item, w, visible); }
            hm &= ~same;
        }
    }
    if (COUNT) {
        atomicAdd(ANY ? &counters->any_slots : &counters->slots, (unsigned long long)c_slots);
        atomicAdd(ANY ? &counters->any_busy : &counters->busy, (unsigned long long)c_busy);
        atomicMax(ANY ? &counters->any_max_steps : &counters->max_steps, (unsigned long long)c_max);
        if (c_refused) atomicAdd(ANY ? &counters->any_refused : &counters->refused, (unsigned long long)c_refused);
    }
}

// Scene::traverse (Scene.h:118): closest hit, one ray per lane start to finish.
template <bool COUNT>
__device__ __forceinline__ void rt_closest(const DScene& S, f3 o, f3 dir, int mode, HitRec& best, TravStack& st,
                                           uint32_t* spill2, uint32_t gstride, uint32_t& cn, uint32_t& ct, uint32_t& cexact) {
    Walk w;
    if (!rt_closest_begin<COUNT>(S, o, dir, mode, w, spill2, gstride, cn, ct, cexact))
        while (!rt_wide_step<false, COUNT>(S, w, st, cn, ct)) {}
    bool unused = true; uint32_t refused = 0;
    rt_settle<false, COUNT>(S, w, unused, spill2, gstride, cn, ct, refused);
    best = w.best;
}

// Scene::visible (Scene.h:169-178)
template <bool COUNT>
__device__ __forceinline__ bool rt_visible(const DScene& S, f3 p1, f3 p2, int mode, TravStack& st, uint32_t* spill2, uint32_t gstride,
                                           uint32_t& cn, uint32_t& ct, uint32_t& cexact) {
    Walk w;
    bool visible;
    if (rt_visible_begin<COUNT>(S, p1, p2, mode, w, visible, spill2, gstride, cn, ct, cexact)) return visible;
    while (!rt_wide_step<true, COUNT>(S, w, st, cn, ct)) {}
    visible = !w.occluded;
    uint32_t refused = 0;
    rt_settle<true, COUNT>(S, w, visible, spill2, gstride, cn, ct, refused);
    return visible;
}
