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
struct TravStack {
    uint32_t* lds;     // &lds_stack[threadIdx.x]
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
        if (sp < RT_STACK_LDS) {
            key = RT_STACK_KEYS ? __uint_as_float(lds[(RT_STACK_LDS + sp) * RT_BLOCK]) : -RT_INF;
            return lds[sp * RT_BLOCK];
        }
        size_t o = (size_t)(sp - RT_STACK_LDS) * gstride;
        key = RT_STACK_KEYS ? __uint_as_float(glob[o + gkeys]) : -RT_INF;
        return glob[o];
    }
};
#define RT_TRAV_STACK_INIT(st, lds_stack, spill4, gtid, gstride, max_stack4)                    \
    TravStack st;                                                                             \
    st.lds = lds_stack + threadIdx.x; st.glob = spill4 ? spill4 + gtid : nullptr; st.gstride = gstride; \
    st.gkeys = (max_stack4 > RT_STACK_LDS ? (max_stack4 - RT_STACK_LDS) : 0u) * gstride;

struct HitRec { float t, u, v; uint32_t rank; };

// Triangle::rayIntersect (Geometry.h:115-128), same expression tree; inv_area is the hoisted 1/Dot(e1 x e2, n).
__device__ __forceinline__ bool rt_tri_test(const DTri* __restrict__ tp, f3 o, f3 dir, float& t, float& u, float& v) {
    const float4* q = reinterpret_cast<const float4*>(tp);
    float4 a = q[0], b = q[1], c = q[2], e = q[3];
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

#define RT_CSWAP(ka, ra, kb, rb)                                  \
    {                                                             \
        bool s_ = kb < ka;                                        \
        float tk_ = s_ ? kb : ka; kb = s_ ? ka : kb; ka = tk_;    \
        uint32_t tr_ = s_ ? rb : ra; rb = s_ ? ra : rb; ra = tr_; \
    }

// ---- 4-wide ordered walk.  Closest hit: children sorted near-to-far, subtrees skipped when their box,
// padded by `padt` (a spatial slack of ~1000 ulp of the largest coordinate, expressed in t), cannot hold a hit
// in [0, best.t].  Skipping only uses the padded interval; whether a child is *entered* is always the
// reference's own test on the reference's own box.  Any hit: the reference's box rule
// (tentry <= texit && !(tentry > maxT)) plus the same padded "entirely behind the origin" skip, any order.
// Shape: while-while -- every lane descends inner nodes until it holds a leaf (or is done), then the wave
// tests triangles together, so the expensive leaf code runs with as many lanes as possible.
template <bool ANY, bool COUNT>
__device__ bool rt_walk_wide(const DScene& S, f3 o, f3 dir, f3 inv, float maxT, HitRec& best, TravStack& st,
                             uint32_t& c_nodes, uint32_t& c_tris) {
    float scale = fmaxf(S.abs_max, fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))));
    float delta = scale * 6.103515625e-05f;  // 2^-14
    float padt = delta * fmaxf(fabsf(inv.x), fmaxf(fabsf(inv.y), fabsf(inv.z)));
    uint32_t sp = 0;
    uint32_t cur = S.root_ref;  // RT_REF_EMPTY doubles as "done": it has the leaf bit set, so it leaves the inner loop
    for (;;) {
        while (!(cur & RT_REF_LEAF)) {
            const float4* q = reinterpret_cast<const float4*>(S.nodes4 + cur);
            float4 lox = q[0], loy = q[1], loz = q[2], hix = q[3], hiy = q[4], hiz = q[5];
            uint4 ch = *reinterpret_cast<const uint4*>(q + 6);
            if (COUNT) c_nodes++;
            float k0, k1, k2, k3;
#define RT_SLOT(LX, LY, LZ, HX, HY, HZ, CH, KEY)                                                         \
    {                                                                                                      \
        float ax = (LX - o.x) * inv.x, bx = (HX - o.x) * inv.x;                                            \
        float ay = (LY - o.y) * inv.y, by = (HY - o.y) * inv.y;                                            \
        float az = (LZ - o.z) * inv.z, bz = (HZ - o.z) * inv.z;                                            \
        float tentry = fmaxf(fminf(ax, bx), fmaxf(fminf(ay, by), fminf(az, bz)));                          \
        float texit = fminf(fmaxf(ax, bx), fminf(fmaxf(ay, by), fmaxf(az, bz)));                           \
        bool pass = (tentry <= texit) && (CH != RT_REF_EMPTY) && !((texit + padt) < 0.0f);                 \
        if (ANY) pass = pass && !(tentry > maxT);                                                          \
        else pass = pass && !((tentry - padt) > best.t);                                                   \
        KEY = pass ? tentry : RT_INF;                                                                      \
    }
            RT_SLOT(lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, ch.x, k0)
            RT_SLOT(lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, ch.y, k1)
            RT_SLOT(lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, ch.z, k2)
            RT_SLOT(lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, ch.w, k3)
#undef RT_SLOT
            uint32_t r0 = ch.x, r1 = ch.y, r2 = ch.z, r3 = ch.w;
            if (!ANY) { RT_CSWAP(k0, r0, k1, r1) RT_CSWAP(k2, r2, k3, r3) RT_CSWAP(k0, r0, k2, r2) RT_CSWAP(k1, r1, k3, r3) RT_CSWAP(k1, r1, k2, r2) }
            if (ANY) {  // no order needed: compact the passing children to the front
                if (!(k2 < RT_INF)) { k2 = k3; r2 = r3; k3 = RT_INF; }
                if (!(k1 < RT_INF)) { k1 = k2; r1 = r2; k2 = k3; r2 = r3; k3 = RT_INF; }
                if (!(k0 < RT_INF)) { k0 = k1; r0 = r1; k1 = k2; r1 = r2; k2 = k3; r2 = r3; k3 = RT_INF; }
            }
            if (k3 < RT_INF) st.put(sp++, r3, k3 - padt);
            if (k2 < RT_INF) st.put(sp++, r2, k2 - padt);
            if (k1 < RT_INF) st.put(sp++, r1, k1 - padt);
            if (k0 < RT_INF) cur = r0;
            else {
                cur = RT_REF_EMPTY;
                while (sp) { float key; uint32_t c = st.get(--sp, key); if (ANY || !(key > best.t)) { cur = c; break; } }
            }
        }
        if (cur == RT_REF_EMPTY) break;
        uint32_t first = cur & RT_LEAF_FIRST_MASK, cnt = ((cur >> 27) & 15u) + 1u;
        for (uint32_t k = 0; k < cnt; k++) {
            float t, u, v;
            if (COUNT) c_tris++;
            if (rt_tri_test(S.tris + first + k, o, dir, t, u, v)) {
                if (ANY) { if (t < maxT) return false; }
                else if (t < best.t || (t == best.t && (first + k) < best.rank)) {
                    best.t = t; best.u = u; best.v = v; best.rank = first + k;
                }
            }
        }
        cur = RT_REF_EMPTY;
        while (sp) { float key; uint32_t c = st.get(--sp, key); if (ANY || !(key > best.t)) { cur = c; break; } }
    }
    return true;
}

// ---- 4-wide ordered walk.  Closest hit: children sorted near-to-far, subtrees skipped when their box,
// padded by `pad` (a spatial slack expressed in t per axis), cannot hold a hit in [0, best.t].  Skipping only
// uses the padded interval; whether a child is *entered* is always the reference's own test on the
// reference's own box.  Any hit: the reference's box rule (tentry <= texit && !(tentry > maxT)), any order.
template <bool ANY, bool COUNT>
__device__ bool rt_walk_wide_v1(const DScene& S, f3 o, f3 dir, f3 inv, float maxT, HitRec& best, TravStack& st,
                             uint32_t& c_nodes, uint32_t& c_tris) {
    float scale = fmaxf(S.abs_max, fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))));
    float delta = scale * 6.103515625e-05f;  // 2^-14: ~1000 ulp of the largest coordinate in play
    f3 pad = mk3(delta * fabsf(inv.x), delta * fabsf(inv.y), delta * fabsf(inv.z));
    uint32_t sp = 0;
    uint32_t cur = S.root_ref;
    while (true) {
        if (!(cur & RT_REF_LEAF)) {
            const float4* q = reinterpret_cast<const float4*>(S.nodes4 + cur);
            float4 lox = q[0], loy = q[1], loz = q[2], hix = q[3], hiy = q[4], hiz = q[5];
            uint4 ch = *reinterpret_cast<const uint4*>(q + 6);
            if (COUNT) c_nodes++;
            float k0, k1, k2, k3;
#define RT_SLOT(i, LX, LY, LZ, HX, HY, HZ, CH, KEY)                                                      \
    {                                                                                                      \
        float ax = (LX - o.x) * inv.x, bx = (HX - o.x) * inv.x;                                            \
        float ay = (LY - o.y) * inv.y, by = (HY - o.y) * inv.y;                                            \
        float az = (LZ - o.z) * inv.z, bz = (HZ - o.z) * inv.z;                                            \
        float enx = fminf(ax, bx), eny = fminf(ay, by), enz = fminf(az, bz);                               \
        float exx = fmaxf(ax, bx), exy = fmaxf(ay, by), exz = fmaxf(az, bz);                               \
        float tentry = fmaxf(enx, fmaxf(eny, enz));                                                        \
        float texit = fminf(exx, fminf(exy, exz));                                                         \
        bool pass = (tentry <= texit) && (CH != RT_REF_EMPTY);                                             \
        if (ANY) pass = pass && !(tentry > maxT);                                                          \
        else {                                                                                             \
            float lo_ = fmaxf(enx - pad.x, fmaxf(eny - pad.y, enz - pad.z));                               \
            float hi_ = fminf(exx + pad.x, fminf(exy + pad.y, exz + pad.z));                               \
            pass = pass && !(lo_ > best.t) && !(hi_ < 0.0f);                                               \
        }                                                                                                  \
        KEY = pass ? (ANY ? tentry : fmaxf(enx - pad.x, fmaxf(eny - pad.y, enz - pad.z))) : RT_INF;        \
    }
            RT_SLOT(0, lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, ch.x, k0)
            RT_SLOT(1, lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, ch.y, k1)
            RT_SLOT(2, lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, ch.z, k2)
            RT_SLOT(3, lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, ch.w, k3)
#undef RT_SLOT
            uint32_t r0 = ch.x, r1 = ch.y, r2 = ch.z, r3 = ch.w;
            if (!ANY) {
                RT_CSWAP(k0, r0, k1, r1) RT_CSWAP(k2, r2, k3, r3) RT_CSWAP(k0, r0, k2, r2) RT_CSWAP(k1, r1, k3, r3) RT_CSWAP(k1, r1, k2, r2)
                if (k3 < RT_INF) st.put(sp++, r3, k3);
                if (k2 < RT_INF) st.put(sp++, r2, k2);
                if (k1 < RT_INF) st.put(sp++, r1, k1);
                if (k0 < RT_INF) { cur = r0; continue; }
            } else {
                if (k3 < RT_INF) st.put(sp++, r3, k3);
                if (k2 < RT_INF) st.put(sp++, r2, k2);
                if (k1 < RT_INF) st.put(sp++, r1, k1);
                if (k0 < RT_INF) { cur = r0; continue; }
            }
        } else {
            uint32_t first = cur & RT_LEAF_FIRST_MASK, cnt = ((cur >> 27) & 15u) + 1u;
            for (uint32_t k = 0; k < cnt; k++) {
                float t, u, v;
                if (COUNT) c_tris++;
                if (rt_tri_test(S.tris + first + k, o, dir, t, u, v)) {
                    if (ANY) { if (t < maxT) return false; }
                    else if (t < best.t || (t == best.t && (first + k) < best.rank)) {
                        best.t = t; best.u = u; best.v = v; best.rank = first + k;
                    }
                }
            }
        }
        // pop, dropping entries whose (padded) entry distance now lies beyond the best hit
        bool got = false;
        while (sp) { float key; uint32_t c = st.get(--sp, key); if (ANY || !(key > best.t)) { cur = c; got = true; break; } }
        if (!got) break;
    }
    return true;
}

// Scene::traverse (Scene.h:118): closest hit.  best.t == FLT_MAX <=> miss.
template <bool COUNT>
__device__ __forceinline__ void rt_closest(const DScene& S, f3 o, f3 dir, int mode, HitRec& best, TravStack& st,
                                           uint32_t* spill2, uint32_t gstride, uint32_t& cn, uint32_t& ct, uint32_t& cexact) {
    best.t = RT_FLT_MAX; best.u = 0; best.v = 0; best.rank = 0xFFFFFFFFu;
    if (S.root_ref == RT_REF_EMPTY) return;
    // dir == (0,0,0) (what pathTrace shoots after a BSDF returned pdf 0 / wi 0, e.g. Materials.h:319-324): Dot(n, dir)
    // is 0 for every triangle, so Triangle::rayIntersect rejects all of them (Geometry.h:118) -- a miss, whatever
    // boxes the reference would have walked through on the way.
    if (mode == 0 && dir.x == 0.0f && dir.y == 0.0f && dir.z == 0.0f) return;
    f3 inv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    if (mode == 0 && rt_ray_is_tame(inv)) {
        rt_walk_wide_v1<false, COUNT>(S, o, dir, inv, 0.0f, best, st, cn, ct);
    }
    else { if (COUNT) cexact++; rt_walk_reference<false, COUNT>(S, o, dir, 0.0f, best, spill2, gstride, cn, ct); }
}

// Scene::visible (Scene.h:169-178)
template <bool COUNT>
__device__ __forceinline__ bool rt_visible(const DScene& S, f3 p1, f3 p2, int mode, TravStack& st, uint32_t* spill2, uint32_t gstride,
                                           uint32_t& cn, uint32_t& ct, uint32_t& cexact) {
    f3 dir = p2 - p1;
    float maxT = len3(dir) - (2.0f * RT_EPSILON);
    dir = norm3(dir);
    f3 o = p1 + (dir * RT_EPSILON);
    if (S.root_ref == RT_REF_EMPTY) return true;
    f3 inv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    HitRec dummy;
    if (mode == 0 && rt_ray_is_tame(inv)) return rt_walk_wide<true, COUNT>(S, o, dir, inv, maxT, dummy, st, cn, ct);
    if (COUNT) cexact++;
    return rt_walk_reference<true, COUNT>(S, o, dir, maxT, dummy, spill2, gstride, cn, ct);
}
