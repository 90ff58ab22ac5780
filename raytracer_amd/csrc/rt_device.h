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
#define RT_TRAV_WAVES 5    // __launch_bounds__ waves/SIMD target of the traversal kernels (96 VGPRs: the 8-wide node step spills at 80; 4 waves: -5 %)
#endif
#ifndef RT_NODE_MIX
#define RT_NODE_MIX 1      // the node step's plane bytes through v_perm_b32 + v_fma_mix_f32 (0: v_bfi + v_cvt_f32_ubyte + v_fma_f32, the same bits)
#endif
#ifndef RT_STACK_LDS
#define RT_STACK_LDS 12          // 8-byte stack entries per lane kept in LDS (24 KB per 256-thread block)
#endif
#define RT_FLT_MAX 3.402823466e+38f
#define RT_PAD_SCALE_EXACT 6.103515625e-05f   // 2^-14: cull slack of rt_walk_exact (the wide walk's is RT_PAD_SCALE, 2^-17)
#define RT_INF __builtin_inff()

// the wave's lane mask of a predicate, straight from its compare (HIP's __ballot goes through an int: v_cndmask + v_cmp per call)
__device__ __forceinline__ unsigned long long rt_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
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

// ---- traversal stack: the first RT_STACK_LDS entries of each lane live in LDS (lane-interleaved 8-byte entries, so a
// push/pop of the whole wave is one conflict-free ds_write_b64 / ds_read_b64); deeper entries spill to a per-launch HBM array.
// An entry is a group of the 8-wide walk (rt_node_eval): {child_base, hit bits << 24 | imask} -- the children of one node
// that still have to be entered -- or {tri_base, triangle bits} -- the triangles of one node that still have to be tested.
typedef uint32_t rt_u2 __attribute__((ext_vector_type(2)));
typedef _Float16 rt_h2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) rt_u2 rt_lds_u2;  // explicit LDS address space: keeps pops ds_read, not flat_load
#define RT_STACK_WORDS (2 * RT_STACK_LDS * RT_BLOCK)
#define RT_SPILL_WORDS(walk_stack) ((walk_stack) > (uint32_t)RT_STACK_LDS ? ((walk_stack) - (uint32_t)RT_STACK_LDS) * 2u : 0u)   // per lane
struct TravStack {
    rt_lds_u2* lds;    // &lds_stack[threadIdx.x]
    rt_u2* glob;       // &spill[global thread id], may be null when the scene's bound fits LDS
    uint32_t gstride;  // total threads of the launch
    __device__ __forceinline__ void put(uint32_t sp, uint32_t b, uint32_t m) {
        rt_u2 v; v.x = b; v.y = m;
        if (sp < RT_STACK_LDS) lds[sp * RT_BLOCK] = v;
        else glob[(size_t)(sp - RT_STACK_LDS) * gstride] = v;
    }
    __device__ __forceinline__ void get(uint32_t sp, uint32_t& b, uint32_t& m) const {
        // the LDS read is unconditional (clamped index) so that it stays a ds_read: with one load per branch the
        // compiler sinks both into a single flat_load, which is slower than LDS for the common case
        const uint32_t i = sp < RT_STACK_LDS ? sp : RT_STACK_LDS - 1u;
        rt_u2 v = lds[i * RT_BLOCK];
        if (sp >= RT_STACK_LDS) v = glob[(size_t)(sp - RT_STACK_LDS) * gstride];
        b = v.x; m = v.y;
    }
};
#define RT_TRAV_STACK_INIT(st, lds_stack, spillw, gtid, gstride)                                    \
    TravStack st;                                                                                 \
    st.lds = (rt_lds_u2*)lds_stack + threadIdx.x; st.glob = spillw ? (rt_u2*)spillw + gtid : nullptr; st.gstride = gstride;

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
// Two more conditions a ray must meet to take the wide walk, both relative to the scene's size (M = the largest coordinate
// in play, scene or origin):
//  * Upstream accepts a plane hit whose point p = o + dir * t overflows, or whose barycentric products do: u or v come out
//    NaN and a comparison with NaN rejects nothing (Geometry.h:122-126).  Such a "hit" lies nowhere near its triangle, so the
//    premise of the conservative culls -- an accepted hit's point is inside every box around its triangle -- does not hold.
//    For a plane that is not orthogonal to the ray by cancellation, t <= 2 M max|1/dir|, |p| <= 2 M max|dir| max|1/dir| and
//    the cross products are below 8 M^2 max|dir| max|1/dir|; kept under 1e36 nothing overflows.  Every ray a renderer shoots
//    in a scene of ordinary size passes (|dir| ~ 1, M ~ 10: up to |1/dir| of 1e18); found by steep rays (a direction component
//    of 1e-17 beside components of 1e12) on a scene scaled by 1e12.
//  * With RT_NODE_MIX a node step multiplies 1/dir by the node's grid step * 2^24 (rt_node_eval: the plane bytes are read as
//    f16 denormals).  A grid step is at most 8/255 of M (k8_emit, walk_tree.hip), so with M max|1/dir| < 1e31 the scaled
//    factor stays finite (< 5.3e36).
// A ray that fails either takes the exact walk.  False if a product is inf or NaN.
__device__ __forceinline__ bool rt_ray_fits_scene(const DScene& S, f3 o, f3 dir, f3 inv) {
    const float M = fmaxf(S.abs_max, fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))));
    const float D = fmaxf(fabsf(dir.x), fmaxf(fabsf(dir.y), fabsf(dir.z))), I = fmaxf(fabsf(inv.x), fmaxf(fabsf(inv.y), fabsf(inv.z)));
    const float IM = I * M;
    return ((D * IM) * M < 1e36f) && (!RT_NODE_MIX || IM < 1e31f);
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
// conservative cull that only looks at the axes where 1/dir is tame (a box is skipped when, grown by RT_PAD_SCALE_EXACT
// of the largest coordinate in play -- 2^-14, eight times the wide walk's slack: this walk runs for a handful of rays per
// frame and keeps round 1's wider margin -- it lies entirely behind the origin or entirely beyond the best hit on those axes).  Without
// it one such ray pierces thousands of boxes of a 1 M-triangle scene unculled and, alone in its wave, sets the
// duration of the whole launch: 0.6-2 ms per k_extend launch before, measured.
template <bool ANY, bool COUNT>
__device__ bool rt_walk_exact(const DScene& S, f3 o, f3 dir, float maxT, HitRec& best, uint32_t* spill, uint32_t gstride,
                              uint32_t& c_nodes, uint32_t& c_tris) {
    const f3 inv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);  // Ray::init (Geometry.h:23)
    const bool tx = fabsf(inv.x) < 1e18f, ty = fabsf(inv.y) < 1e18f, tz = fabsf(inv.z) < 1e18f;
    const float scale = fmaxf(S.abs_max, fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))));
    const float delta = scale * RT_PAD_SCALE_EXACT;
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

// ---- the 8-wide walk, one step at a time.  The per-lane state lives in a Walk so that a wave can keep stepping the
// rays it has while lanes whose ray finished are handed new ones (rt_persistent_walk below).
//
// The tests are CONSERVATIVE ONLY.  Upstream's answer is min (t, DFS rank) over the triangles that pass rayIntersect AND whose
// ancestor boxes all pass rayAABB; for a tame ray the box test is monotone under containment (section 3 of DESIGN.md), so the
// second condition is "the triangle's gate box passes" (DScene::gate).  The walk may therefore use any tree, any boxes and any
// arithmetic that never skip a true hit nearer than the best so far -- here the 8-wide nodes of walk_tree.hip, their child
// boxes on a 256-step grid of the node's box, rounded outwards, grown by the slack `pad`, evaluated with one fma per plane --
// as long as the hit it ends with is then checked against its gate box with upstream's own expression (rt_settle): if that
// check fails, which takes a hit within rounding of its gate box's surface, the ray is walked again by rt_walk_exact.
//   plane in t:  (org + q * s - o) * inv -+ pad  =  q * (s * inv) + ((org - o) * inv -+ pad)  =  fma(q, A, B)
// Rounding: org - o, B and the final fma round once each, <= 3 * 2^-24 * 3 M |1/dir|, inside the slack's budget (below);
// s is a power of two, so A = ldexp(inv, e) is exact.
//
// Slack of the conservative culls, as a fraction of the largest coordinate in play (scene extent, ray origin): a box is grown
// by delta = RT_PAD_SCALE * M before the walk may skip it.  What the slack has to cover: an accepted
// hit's point p = o + dir * t lies inside its triangle -- hence inside every box around it -- up to the rounding of the
// plane equation, of p and of the barycentric test, together < 2^-19 M whatever the triangle's shape; expressing the box
// planes in t adds 8 * 2^-24 M |1/dir|.  2^-17 M (128 ulp) is three times that bound.  Round 1 used 2^-14: for rays nearly
// parallel to an axis the pad in t is delta * |1/dir|, and the eight-fold larger pad kept whole subtrees behind the origin
// and beyond the best hit alive (+4.9 % Mrays/s on the bench scene).  rt_settle only catches a hit the walk should not have
// reported (its gate box fails); a true hit culled by too small a pad is never seen at run time: lowering RT_PAD_SCALE needs
// the strained-ray, aimed-ray and scaled-scene tests of tests/test_intersect_gpu.py to pass.
#ifndef RT_PAD_SCALE
#define RT_PAD_SCALE 7.62939453125e-06f   // 2^-17
#endif
static_assert(RT_PAD_SCALE >= 7.62939453125e-06f, "the cull slack was derived and tested for 2^-17 of the largest coordinate; see the comment above");
// What a lane holds: `cur`, a group of inner children still to enter {cb = the node's child_base, cm = hit bits << 24 | imask}
// (or, while `pend` is taken, a group of triangles that waits its turn; cm == 0: nothing, and then the stack is empty too),
// and `pend`, the triangles being tested {pb = the node's tri_base, pm = one bit per triangle}.  A lane that meets triangles
// does not stop walking: they are set aside in pend and the lane goes on with the next node, so that it has work for either
// kind of wave step (Aila & Laine's speculative traversal); what that costs is that the nodes walked meanwhile were not
// culled by the hit the postponed triangles may hold.
struct Walk {
    f3 o, dir, inv;
    f3 pad;        // the slack in t per axis: delta * |1/dir|
    float maxT;        // any hit
    HitRec best;       // closest hit so far; any hit: best.rank = the blocker that was found
    uint32_t cb, cm;   // cur
    uint32_t pb, pm;   // pend
    uint32_t sp, sb;   // the stack holds entries [sb, sp): pushes and pops at sp; the tail's work stealing takes from sb, the oldest entry (the largest subtree the walk has waiting)
    uint32_t q4;       // the ray's octant (bit a set where dir is non-negative on axis a) in each of the four bytes
    uint32_t mx, my, mz;   // which of a node's lo / hi planes are the near ones on that axis, by the sign of 1/dir: a v_perm_b32 byte selector (RT_NODE_MIX) or an all-ones / zero mask
    bool occluded;
    bool wide;         // this item went through the wide walk: its answer still has to pass rt_settle
};
__device__ __forceinline__ bool rt_cur_is_nodes(const Walk& w) { return (w.cm >> 24) != 0u; }
__device__ __forceinline__ bool rt_cur_is_tris(const Walk& w) { return w.cm != 0u && (w.cm >> 24) == 0u; }
__device__ __forceinline__ bool rt_walk_done(const Walk& w) { return w.cm == 0u && w.pm == 0u; }
__device__ __forceinline__ void rt_walk_signs(Walk& w) {
#if RT_NODE_MIX
    // byte selectors of v_perm_b32 {S0 = the node's hi planes, S1 = its lo planes}: bytes 0 and 2 of the near planes' word into
    // the low bytes of the two 16-bit halves, zeros above them (0x0c); bytes 4-7 name S0, i.e. hi is near where 1/dir < 0
    w.mx = w.inv.x < 0.0f ? 0x0c060c04u : 0x0c020c00u; w.my = w.inv.y < 0.0f ? 0x0c060c04u : 0x0c020c00u; w.mz = w.inv.z < 0.0f ? 0x0c060c04u : 0x0c020c00u;
#else
    w.mx = w.inv.x < 0.0f ? 0xFFFFFFFFu : 0u; w.my = w.inv.y < 0.0f ? 0xFFFFFFFFu : 0u; w.mz = w.inv.z < 0.0f ? 0xFFFFFFFFu : 0u;
#endif
    w.q4 = ((w.inv.x < 0.0f ? 0u : 1u) | (w.inv.y < 0.0f ? 0u : 2u) | (w.inv.z < 0.0f ? 0u : 4u)) * 0x01010101u;
}
__device__ __forceinline__ void rt_walk_init(const DScene& S, Walk& w, f3 o, f3 dir, f3 inv, float maxT) {
    float scale = fmaxf(S.abs_max, fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))));
    float delta = scale * RT_PAD_SCALE;
    w.o = o; w.dir = dir; w.inv = inv;
    w.pad = mk3(delta * fabsf(inv.x), delta * fabsf(inv.y), delta * fabsf(inv.z));
    w.maxT = maxT;
    rt_walk_signs(w);
    w.cb = 0u; w.cm = 1u << 24;   // "one child left to enter below child_base 0": node 0, whatever the octant
    w.pb = 0u; w.pm = 0u; w.sp = 0u; w.sb = 0u; w.occluded = false; w.wide = true;
}
// (m & a) | (~m & b) as the one instruction it is: from the C expression the compiler keeps ~m in a register and emits v_and +
// v_and_or, 24 instructions per node step where 12 do
__device__ __forceinline__ uint32_t rt_bfi(uint32_t m, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float rt_ub0(uint32_t v) { return (float)(v & 255u); }            // v_cvt_f32_ubyte0..3
__device__ __forceinline__ float rt_ub1(uint32_t v) { return (float)((v >> 8) & 255u); }
__device__ __forceinline__ float rt_ub2(uint32_t v) { return (float)((v >> 16) & 255u); }
__device__ __forceinline__ float rt_ub3(uint32_t v) { return (float)(v >> 24); }
template <bool ANY>
__device__ __forceinline__ void rt_pop(Walk& w, TravStack& st) {
    if (w.sp == w.sb) { w.cm = 0u; w.sp = 0u; w.sb = 0u; return; }
    --w.sp;
    st.get(w.sp, w.cb, w.cm);
}
// cur after a pop: triangles move on to pend as soon as pend is free
template <bool ANY>
__device__ __forceinline__ void rt_settle_refs(Walk& w, TravStack& st) {
    if (rt_cur_is_tris(w) && w.pm == 0u) { w.pb = w.cb; w.pm = w.cm; rt_pop<ANY>(w, st); }
}
// One node: enters the next child of cur (the highest hit bit: children sit in octant order, see DNode8), tests its eight
// slots and files what passed -- inner children become cur, triangles go to pend (or wait), what was left of the old cur
// goes onto the stack.  At most two pushes and no sort; a node is five 16-byte loads.
template <bool ANY>
__device__ __forceinline__ void rt_node_eval(const DScene& S, Walk& w, TravStack& st) {
    const uint32_t k = 31u - (uint32_t)__clz((int)w.cm);
    const uint32_t rem = w.cm & ~(1u << k);
    const uint32_t slot = (k ^ w.q4) & 7u;
    const uint32_t node = w.cb + (uint32_t)__popc(rem & ((1u << slot) - 1u));   // the mask only reaches imask: hit bits sit at 24 and above
    const char* nb = reinterpret_cast<const char*>(S.nodes8);
    const uint32_t base = node * 80u;   // < 2^32: the tree is capped at 2^25 nodes
    const float4 r0 = *reinterpret_cast<const float4*>(nb + base);
    const uint4 r1 = *reinterpret_cast<const uint4*>(nb + (base + 16u));   // child_base, tri_base, meta[2]
    const uint4 r2 = *reinterpret_cast<const uint4*>(nb + (base + 32u));   // lox[2] loy[2]
    const uint4 r3 = *reinterpret_cast<const uint4*>(nb + (base + 48u));   // loz[2] hix[2]
    const uint4 r4 = *reinterpret_cast<const uint4*>(nb + (base + 64u));   // hiy[2] hiz[2]
    const uint32_t hw = __float_as_uint(r0.w);
#if RT_NODE_MIX
    // the plane bytes are read as f16 denormals (0x00qq = q * 2^-24, exactly), so the step carries the 2^24: A = ldexp(inv, e + 24)
    const float ax = ldexpf(w.inv.x, ((int)(hw << 24) >> 24) + 24), ay = ldexpf(w.inv.y, ((int)(hw << 16) >> 24) + 24), az = ldexpf(w.inv.z, ((int)(hw << 8) >> 24) + 24);
#else
    const float ax = ldexpf(w.inv.x, (int)(hw << 24) >> 24), ay = ldexpf(w.inv.y, (int)(hw << 16) >> 24), az = ldexpf(w.inv.z, (int)(hw << 8) >> 24);
#endif
    const float dx = r0.x - w.o.x, dy = r0.y - w.o.y, dz = r0.z - w.o.z;
    const float blx = __builtin_fmaf(dx, w.inv.x, -w.pad.x), bhx = __builtin_fmaf(dx, w.inv.x, w.pad.x);
    const float bly = __builtin_fmaf(dy, w.inv.y, -w.pad.y), bhy = __builtin_fmaf(dy, w.inv.y, w.pad.y);
    const float blz = __builtin_fmaf(dz, w.inv.z, -w.pad.z), bhz = __builtin_fmaf(dz, w.inv.z, w.pad.z);
    const float limit = ANY ? w.maxT : w.best.t;
    uint32_t hit = 0u;
#if RT_NODE_MIX
    // Four slots per pass.  Near / far plane bytes by the ray's sign AND their conversion in one go: v_perm_b32 puts bytes 0 and 2
    // (then 1 and 3) of the near word into the low bytes of the two halves of a register, where they read as f16 denormals, and
    // v_fma_mix_f32 takes a half as its first operand -- q * 2^-24 * (A * 2^24) + B is the same product, rounded once, as
    // v_cvt_f32_ubyte + v_fma_f32 gave: 24 v_perm + 48 v_fma_mix per node where 12 v_bfi + 48 v_cvt + 48 v_fma stood, bit-identical.
    const uint32_t sx1 = w.mx + 0x00010001u, sy1 = w.my + 0x00010001u, sz1 = w.mz + 0x00010001u;
#define RT_HX(v) ((float)__builtin_bit_cast(rt_h2, v).x)
#define RT_HY(v) ((float)__builtin_bit_cast(rt_h2, v).y)
#define RT_SLOT(H, P, SH)                                                                                                     \
    {                                                                                                                         \
        const float lo_ = fmaxf(__builtin_fmaf(H(nx##P), ax, blx), fmaxf(__builtin_fmaf(H(ny##P), ay, bly), __builtin_fmaf(H(nz##P), az, blz))); \
        const float hi_ = fminf(__builtin_fmaf(H(fx##P), ax, bhx), fminf(__builtin_fmaf(H(fy##P), ay, bhy), __builtin_fmaf(H(fz##P), az, bhz))); \
        const bool pass = fmaxf(lo_, 0.0f) <= fminf(hi_, limit);                                                               \
        const uint32_t b_ = ((bits >> SH) & 255u) << ((idx >> SH) & 255u);                                                    \
        hit |= pass ? b_ : 0u;                                                                                                \
    }
#define RT_HALF(LOX, LOY, LOZ, HIX, HIY, HIZ, META)                                                                           \
    {                                                                                                                         \
        const uint32_t nxa = __builtin_amdgcn_perm(HIX, LOX, w.mx), nxb = __builtin_amdgcn_perm(HIX, LOX, sx1);               \
        const uint32_t fxa = __builtin_amdgcn_perm(LOX, HIX, w.mx), fxb = __builtin_amdgcn_perm(LOX, HIX, sx1);               \
        const uint32_t nya = __builtin_amdgcn_perm(HIY, LOY, w.my), nyb = __builtin_amdgcn_perm(HIY, LOY, sy1);               \
        const uint32_t fya = __builtin_amdgcn_perm(LOY, HIY, w.my), fyb = __builtin_amdgcn_perm(LOY, HIY, sy1);               \
        const uint32_t nza = __builtin_amdgcn_perm(HIZ, LOZ, w.mz), nzb = __builtin_amdgcn_perm(HIZ, LOZ, sz1);               \
        const uint32_t fza = __builtin_amdgcn_perm(LOZ, HIZ, w.mz), fzb = __builtin_amdgcn_perm(LOZ, HIZ, sz1);               \
        const uint32_t inner = ((META & (META << 1)) & 0x10101010u) >> 4;   /* 1 in the bytes of inner children (index 24-31) */ \
        const uint32_t idx = (META ^ ((inner * 7u) & w.q4)) & 0x1F1F1F1Fu;  /* their index XOR the ray's octant */              \
        const uint32_t bits = (META >> 5) & 0x07070707u;                                                                      \
        RT_SLOT(RT_HX, a, 0) RT_SLOT(RT_HX, b, 8) RT_SLOT(RT_HY, a, 16) RT_SLOT(RT_HY, b, 24)                                 \
    }
    RT_HALF(r2.x, r2.z, r3.x, r3.z, r4.x, r4.z, r1.z)
    RT_HALF(r2.y, r2.w, r3.y, r3.w, r4.y, r4.w, r1.w)
#undef RT_HALF
#undef RT_SLOT
#undef RT_HX
#undef RT_HY
#else
    // four slots per pass: the near / far plane bytes by the ray's sign, and per slot the bits it stands for and where they go
#define RT_SLOT(UB, SH)                                                                                                       \
    {                                                                                                                         \
        const float lo_ = fmaxf(__builtin_fmaf(UB(nx), ax, blx), fmaxf(__builtin_fmaf(UB(ny), ay, bly), __builtin_fmaf(UB(nz), az, blz))); \
        const float hi_ = fminf(__builtin_fmaf(UB(fx), ax, bhx), fminf(__builtin_fmaf(UB(fy), ay, bhy), __builtin_fmaf(UB(fz), az, bhz))); \
        const bool pass = fmaxf(lo_, 0.0f) <= fminf(hi_, limit);                                                               \
        const uint32_t b_ = ((bits >> SH) & 255u) << ((idx >> SH) & 255u);                                                    \
        hit |= pass ? b_ : 0u;                                                                                                \
    }
#define RT_HALF(LOX, LOY, LOZ, HIX, HIY, HIZ, META)                                                                           \
    {                                                                                                                         \
        const uint32_t nx = rt_bfi(w.mx, HIX, LOX), fx = rt_bfi(w.mx, LOX, HIX);                                             \
        const uint32_t ny = rt_bfi(w.my, HIY, LOY), fy = rt_bfi(w.my, LOY, HIY);                                             \
        const uint32_t nz = rt_bfi(w.mz, HIZ, LOZ), fz = rt_bfi(w.mz, LOZ, HIZ);                                             \
        const uint32_t inner = ((META & (META << 1)) & 0x10101010u) >> 4;   /* 1 in the bytes of inner children (index 24-31) */ \
        const uint32_t idx = (META ^ ((inner * 7u) & w.q4)) & 0x1F1F1F1Fu;  /* their index XOR the ray's octant */              \
        const uint32_t bits = (META >> 5) & 0x07070707u;                                                                      \
        RT_SLOT(rt_ub0, 0) RT_SLOT(rt_ub1, 8) RT_SLOT(rt_ub2, 16) RT_SLOT(rt_ub3, 24)                                         \
    }
    RT_HALF(r2.x, r2.z, r3.x, r3.z, r4.x, r4.z, r1.z)
    RT_HALF(r2.y, r2.w, r3.y, r3.w, r4.y, r4.w, r1.w)
#undef RT_HALF
#undef RT_SLOT
#endif
    const uint32_t gm = hit & 0xFF000000u, tm = hit & 0x00FFFFFFu;
    const bool pend_free = w.pm == 0u;
    w.pb = pend_free ? r1.y : w.pb; w.pm = pend_free ? tm : w.pm;
    const bool x1 = gm != 0u;                    // inner children to enter: they become cur
    const bool x2 = tm != 0u && !pend_free;      // triangles that have to wait for pend
    const bool x3 = rem >= (1u << 24);           // the old cur has children left
    const bool push_rem = x3 && (x1 || x2), push_tri = x2 && x1;
    if (rt_ballot(w.sp + 2u > (uint32_t)RT_STACK_LDS) == 0) {
        // every lane has room in LDS for two more: store unconditionally, advance by the predicate
        rt_u2 v; v.x = w.cb; v.y = rem;
        st.lds[w.sp * RT_BLOCK] = v; w.sp += push_rem ? 1u : 0u;
        v.x = r1.y; v.y = tm;
        st.lds[w.sp * RT_BLOCK] = v; w.sp += push_tri ? 1u : 0u;
    } else {
        if (push_rem) st.put(w.sp++, w.cb, rem);
        if (push_tri) st.put(w.sp++, r1.y, tm);
    }
    const uint32_t ob = w.cb;
    w.cb = x1 ? r1.x : (x2 ? r1.y : ob);
    w.cm = x1 ? (gm | (hw >> 24)) : (x2 ? tm : (x3 ? rem : 0u));
    if (!(x1 || x2 || x3)) { rt_pop<ANY>(w, st); rt_settle_refs<ANY>(w, st); }
}
// One triangle of pend.  Returns true when an any-hit walk found its blocker.
template <bool ANY>
__device__ __forceinline__ bool rt_tri_eval_step(const DScene& S, Walk& w, TravStack& st) {
    const uint32_t j = (uint32_t)__ffs((int)w.pm) - 1u;
    w.pm &= w.pm - 1u;
    const float4* q = reinterpret_cast<const float4*>(S.trisw + (w.pb + j));
    const float4 a = q[0], b = q[1], c = q[2], e = q[3];
    float t, u, v;
    if (rt_tri_eval_all(a, b, c, e, w.o, w.dir, t, u, v)) {
        const uint32_t rank = __float_as_uint(e.z);   // trisw keeps the DFS rank where tris keeps orig_id
        if (ANY) { if (t < w.maxT) { w.occluded = true; w.best.rank = rank; w.cm = 0u; w.pm = 0u; return true; } }   // cm, pm: see rt_vote_step
        else if (t < w.best.t || (t == w.best.t && rank < w.best.rank)) { w.best.t = t; w.best.u = u; w.best.v = v; w.best.rank = rank; }
    }
    if (w.pm == 0u && rt_cur_is_tris(w)) { w.pb = w.cb; w.pm = w.cm; rt_pop<ANY>(w, st); }
    return false;
}
// One node or one triangle.  Returns true when the walk is over (nothing left, or ANY found a blocker).
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool rt_wide_step(const DScene& S, Walk& w, TravStack& st, uint32_t& c_nodes, uint32_t& c_tris) {
    if (rt_cur_is_nodes(w)) {
        if (COUNT) c_nodes++;
        rt_node_eval<ANY>(S, w, st);
    } else if (w.pm != 0u) {
        if (COUNT) c_tris++;
        if (rt_tri_eval_step<ANY>(S, w, st)) return true;
    }
    return rt_walk_done(w);
}

// ---- the wave-voted step of rt_persistent_walk.  A wave whose lanes are part on nodes and part on triangles pays
// for both code paths in rt_wide_step, each with a fraction of the lanes (PMC: ~26 % of VALU lanes live).  Here
// the wave runs ONE of the two per step -- box tests for the lanes that have a node to enter, or one triangle for the
// lanes that hold postponed triangles -- whichever serves more lanes per instruction; the others wait a step.
// All 64 lanes call it (busy = this lane holds a walk).  Returns true when this lane's walk is over.
// nodes go when (lanes with a node) * RT_VOTE_NUM >= (lanes with triangles) * RT_VOTE_DEN.  A node step costs about twice a
// triangle step (265 + 82 against 135 + 39 instructions), so lanes served per unit of issue time is what 1:2 maximises.  Bench scene /
// living room, Mrays/s of the frame: 3:2 4660 / -, 1:1 4860 / 4560, 2:3 4918 / 4679, 1:2 4922 / 4666, 2:5 4877 / -, 1:3 4851 / -.
#ifndef RT_VOTE_NUM
#define RT_VOTE_NUM 1
#define RT_VOTE_DEN 2
#endif
#ifndef RT_VOTE_NUM_ANY   // any-hit walks on their own: 1:1 / 1:2 / 1:3 give k_connect 12.95 / 12.75 / 13.2 ms per 20 spp; stealing every step / every fourth: no change
#define RT_VOTE_NUM_ANY RT_VOTE_NUM
#define RT_VOTE_DEN_ANY RT_VOTE_DEN
#endif
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool rt_vote_step(const DScene& S, Walk& w, bool busy, TravStack& st, uint32_t& c_nodes, uint32_t& c_tris) {
    // a lane without a walk holds cm == 0 and pm == 0 (rt_persistent_walk clears them at the start, a walk leaves them so
    // when it ends, and the two early exits -- an any-hit blocker, a part cut short in the tail -- clear them), so the two
    // ballots come straight from compares on the state: `busy &&` cost an s_and, a v_cndmask and a second v_cmp per ballot
    (void)busy;
    const bool innode = rt_cur_is_nodes(w);
    const bool inleaf = w.pm != 0u;
    const int nl = __builtin_popcountll(rt_ballot(inleaf)), nn = __builtin_popcountll(rt_ballot(innode));
    if (nn * (ANY ? RT_VOTE_NUM_ANY : RT_VOTE_NUM) >= nl * (ANY ? RT_VOTE_DEN_ANY : RT_VOTE_DEN)) {
#ifdef RT_DBG_WAVESTEPS
        if (COUNT) c_nodes++;
#endif
        if (!innode) return false;
#ifndef RT_DBG_WAVESTEPS
        if (COUNT) c_nodes++;
#endif
        rt_node_eval<ANY>(S, w, st);
        return rt_walk_done(w);
    }
#ifdef RT_DBG_WAVESTEPS
    if (COUNT) c_tris++;
#endif
    if (!inleaf) return false;
#ifndef RT_DBG_WAVESTEPS
    if (COUNT) c_tris++;
#endif
    if (rt_tri_eval_step<ANY>(S, w, st)) return true;
    return rt_walk_done(w);
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
    if (mode == 0 && moderate && rt_ray_is_tame(inv) && rt_ray_fits_scene(S, o, dir, inv)) { rt_walk_init(S, w, o, dir, inv, 0.0f); return false; }
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
    if (mode == 0 && moderate && rt_ray_is_tame(inv) && rt_ray_fits_scene(S, o, dir, inv)) { rt_walk_init(S, w, o, dir, inv, maxT); return false; }
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
//   Job:    static constexpr uint32_t refill_idle    -- idle lanes at which the wave retires and refills
//           bool load(uint32_t item, f3& a, f3& b)   -- closest: origin, direction; any: p1, p2.  false = nothing to trace
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
// A job names its own threshold (Job::refill_idle): the cheaper its load and retire, the earlier a refill pays.
#ifndef RT_REFILL_IDLE_ANY
#define RT_REFILL_IDLE_ANY 24   // k_connect (NEE shadow rays): 32 / 24 / 16 / 12 idle lanes give 12.75 / 12.6 / 12.5 / 12.6 ms per 20 spp (closest-hit walks: 32 / 24 / 16: 25.1 / 25.15 / 25.5);
#endif                          // k_gather and k_connect_camera, whose retire shades, lose at 24 (instant radiosity 4680 -> 4555 Mrays/s) and stay at RT_REFILL_IDLE

// Tail: once the source is drained a wave is left with its slowest walks -- a few rays that take 1000+ steps where the mean
// is 25 -- and the launch lasts until they end.  From then on idle lanes take work from busy ones: every RT_STEAL_EVERY steps
// a lane that holds a walk hands the oldest entry of its stack (siblings high up in the tree: the largest piece of work it
// has waiting) to a free lane, which gets a copy of the ray and the best hit so far and walks that subtree on its own.  The
// parts of one item are folded together as they end (closest hit = the minimum by (t, rank), the order the single walk
// applies; any hit = OR), a lane that has ended a part is free to steal again, and the lane that ends the last part retires
// the item.  A part can only miss hits that its copy of best.t already beats, so the result is the single walk's, bit for bit.
#ifndef RT_TAIL_STEAL
#define RT_TAIL_STEAL 1
#endif
#ifndef RT_STEAL_EVERY
#define RT_STEAL_EVERY 2
#endif
__device__ __forceinline__ float rt_shfl_f(float v, uint32_t src) { return __shfl(v, (int)src); }
__device__ __forceinline__ f3 rt_shfl_f3(f3 v, uint32_t src) { return mk3(__shfl(v.x, (int)src), __shfl(v.y, (int)src), __shfl(v.z, (int)src)); }

template <bool ANY, bool COUNT, class Job, class Source>
__device__ __forceinline__ void rt_persistent_walk(const DScene& S, Job& job, Source& src, int mode, TravStack& st, uint32_t* spill2,
                                                   uint32_t gstride, uint32_t& cn, uint32_t& ct, uint32_t& cx, DCounters* counters) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t c_slots = 0, c_busy = 0, c_mine = 0, c_max = 0, c_refused = 0;
    Walk w;
    w.wide = false; w.cm = 0u; w.pm = 0u;
    uint32_t item = 0;
    bool busy = false, done = false, visible = true;
    uint32_t pb = 0, pe = 0;   // the wave's pool of granted items (wave-uniform)
    bool drained = false;      // wave-uniform
#ifdef RT_DBG_PHASE
    // Debug build: where a wave's time goes -- shader cycles inside the retire / refill branch (settle + retire, the work atomic,
    // load + begin) against all cycles of the main loop, reported through the counters of a counting pass (tools/dbg_phase.py):
    // lane_busy / lane_slots = the refill share, nodes / tris = its first two parts.  Bench frame: 0.39 of a k_extend wave's time
    // (0.21 settle + retire, 0.07 work atomic, 0.12 load + begin), 0.40 of a k_connect wave's -- all of it memory latency, which
    // the other four waves of the SIMD cover: issuing the new items' first loads before the retire (bit-exact, built and
    // measured) changed nothing, nor does refilling at 24 or 16 idle lanes (lane occupancy 0.71 -> 0.76 -> 0.81, same time).
    unsigned long long ph_all = clock64(), ph_ref = 0, ph_t0 = 0, ph_t1 = 0, ph_ret = 0, ph_nxt = 0;
#endif
    for (;;) {
        unsigned long long idle = rt_ballot(!busy);
        uint32_t nidle = (uint32_t)__popcll(idle);
        if (drained ? (nidle == 64u) : (nidle >= (uint32_t)Job::refill_idle)) {
#ifdef RT_DBG_PHASE
            ph_t0 = clock64();
#endif
            if (done) { rt_settle<ANY, COUNT>(S, w, visible, spill2, gstride, cn, ct, c_refused); job.retire(item, w, visible); done = false; }
#ifdef RT_DBG_PHASE
            { const unsigned long long t1_ = clock64(); ph_ret += t1_ - ph_t0; ph_t1 = t1_; }
#endif
            if (drained) break;
            if (pb >= pe) {
                if (!src.next(pb, pe)) { drained = true; if (RT_TAIL_STEAL) break; continue; }
            }
#ifdef RT_DBG_PHASE
            ph_nxt += clock64() - ph_t1;
#endif
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
#ifdef RT_DBG_PHASE
            ph_ref += clock64() - ph_t0;
#endif
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
#ifdef RT_DBG_PHASE
    ph_all = clock64() - ph_all;
    if (COUNT && (threadIdx.x & 63) == 0) {
        c_slots = 0; c_busy = 0; cn = 0; ct = 0;
        atomicAdd(ANY ? &counters->any_slots : &counters->slots, ph_all >> 6); atomicAdd(ANY ? &counters->any_busy : &counters->busy, ph_ref >> 6);
        atomicAdd(ANY ? &counters->any_nodes : &counters->nodes, ph_ret >> 6); atomicAdd(ANY ? &counters->any_tris : &counters->tris, ph_nxt >> 6);
    }
#endif
    if (RT_TAIL_STEAL) {
        // Every lane is busy (walking its own item) or free here: finished lanes retired above.  From now on an item can be in
        // several lanes at once.  It keeps one record in LDS, in the slot of the lane that held it when the tail began (`own`):
        // the best answer of the parts that have ended, as one 64-bit key so that an LDS atomic min folds a part in -- closest
        // hit: (t, DFS rank), the order the single walk applies; any hit: the blocker's rank, or all ones for "visible" -- and
        // the number of parts still walking.  The lane that ends the last part retires the item; every lane that ends a part
        // is free at once and steals again.  Only this wave touches its 64 slots, and a wave's LDS operations execute in
        // program order, so no barrier is needed.
        __shared__ unsigned long long tail_key[RT_BLOCK];
        __shared__ uint32_t tail_parts[RT_BLOCK], tail_item[RT_BLOCK];
        const uint32_t wbase = threadIdx.x & ~63u;
        uint32_t own = lane;
        tail_key[threadIdx.x] = ~0ULL; tail_parts[threadIdx.x] = busy ? 1u : 0u; tail_item[threadIdx.x] = item;
        for (uint32_t round = 0;; round++) {
            if (!rt_ballot(busy)) break;
            bool ended = false;
            if ((round % (uint32_t)RT_STEAL_EVERY) == 0u) {
                // any hit: a blocker found by one part ends the others
                if (ANY && busy && *(volatile unsigned long long*)&tail_key[wbase + own] != ~0ULL) { ended = true; w.cm = 0u; w.pm = 0u; }
                const unsigned long long fm = rt_ballot(!busy);                          // free: no walk
                const unsigned long long dm = rt_ballot(busy && !ended && w.sp > w.sb);  // donors: a subtree waiting on the stack
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
                    uint32_t give_b = 0, give_m = 0;
                    if (donor) {   // the oldest entry: siblings high up in the tree
                        st.get(w.sb, give_b, give_m);
                        if (++w.sb == w.sp) { w.sp = 0u; w.sb = 0u; }
                    }
                    // the ray and what is known about it so far travel by shuffle; non-thieves read themselves
                    const f3 o_ = rt_shfl_f3(w.o, from), d_ = rt_shfl_f3(w.dir, from), i_ = rt_shfl_f3(w.inv, from), p_ = rt_shfl_f3(w.pad, from);
                    const float mt_ = rt_shfl_f(w.maxT, from);
                    const float bt_ = rt_shfl_f(w.best.t, from), bu_ = rt_shfl_f(w.best.u, from), bv_ = rt_shfl_f(w.best.v, from);
                    const uint32_t br_ = (uint32_t)__shfl((int)w.best.rank, (int)from);
                    const uint32_t own_ = (uint32_t)__shfl((int)own, (int)from);
                    const uint32_t cb_ = (uint32_t)__shfl((int)give_b, (int)from), cm_ = (uint32_t)__shfl((int)give_m, (int)from);
                    if (thief) {
                        w.o = o_; w.dir = d_; w.inv = i_; w.pad = p_; w.maxT = mt_;
                        w.best.t = bt_; w.best.u = bu_; w.best.v = bv_; w.best.rank = br_;   // the donor's best so far: culls, and folds in harmlessly
                        rt_walk_signs(w);
                        w.cb = cb_; w.cm = cm_; w.pb = 0u; w.pm = 0u; w.sp = 0u; w.sb = 0u; w.occluded = false; w.wide = true;
                        if (rt_cur_is_tris(w)) { w.pb = w.cb; w.pm = w.cm; w.cm = 0u; }
                        own = own_; busy = true; visible = true;
                        atomicAdd(&tail_parts[wbase + own], 1u);
                        if (COUNT) c_mine = 0;
                    }
                }
            }
            if (COUNT) { c_slots++; c_busy += busy ? 1u : 0u; c_mine += busy ? 1u : 0u; c_max = c_mine > c_max ? c_mine : c_max; }
            if (rt_vote_step<ANY, COUNT>(S, w, busy && !ended, st, cn, ct)) {
                ended = true;
                if (ANY) visible = !w.occluded;
            } else if (ended) visible = true;   // cut short by another part's blocker: nothing of its own to report
            if (ended) {
                busy = false;
                const uint32_t slot = wbase + own;
                const unsigned long long key = ANY ? (visible ? ~0ULL : (unsigned long long)w.best.rank)
                                                   : (((unsigned long long)__float_as_uint(w.best.t + 0.0f) << 32) | (unsigned long long)w.best.rank);
                atomicMin(&tail_key[slot], key);
                if (atomicSub(&tail_parts[slot], 1u) == 1u) {   // the item's last part: this lane holds its ray and retires it
                    const unsigned long long k = *(volatile unsigned long long*)&tail_key[slot];
                    const uint32_t rank = (uint32_t)k;
                    if (ANY) { visible = k == ~0ULL; w.best.rank = rank; }
                    else if (rank != w.best.rank) {
                        // another part won: the same test on the same record gives its t, u, v again, bit for bit
                        w.best.t = RT_FLT_MAX; w.best.u = 0; w.best.v = 0; w.best.rank = rank;
                        if (rank != 0xFFFFFFFFu) { float t, u, v; (void)rt_tri_test(S.tris + rank, w.o, w.dir, t, u, v); w.best.t = t; w.best.u = u; w.best.v = v; }
                    }
                    w.wide = true;
                    rt_settle<ANY, COUNT>(S, w, visible, spill2, gstride, cn, ct, c_refused);
                    job.retire(*(volatile uint32_t*)&tail_item[slot], w, visible);
                }
            }
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
