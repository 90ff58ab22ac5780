/* TEST INFRASTRUCTURE — CPU oracle (see rt_oracle.h).  Plain C restatement of
 * the reference algorithms; "ref:" comments give the file:line followed
 * (paths relative to /root/reference/RTBase/).
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off (no fast-math) so every float
 * expression rounds exactly as the reference's scalar SSE2 build does.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---------------------------------------------------------------- math (ref: Core.h:16-195) */
typedef struct { float x, y, z; } V3;
typedef struct { float r, g, b; } Col;

static inline V3 v3(float x, float y, float z) { V3 v = { x, y, z }; return v; }
static inline V3 vadd(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }          /* Core.h:128 */
static inline V3 vsub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }          /* Core.h:132 */
static inline V3 vmulf(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }            /* Core.h:136 */
static inline V3 vdivf(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }            /* Core.h:140 */
static inline V3 vmul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }          /* Core.h:144 */
static inline V3 vneg(V3 a) { return v3(-a.x, -a.y, -a.z); }
static inline float vdot(V3 a, V3 b) { return ((a.x * b.x) + (a.y * b.y) + (a.z * b.z)); } /* Core.h:176 */
static inline V3 vcross(V3 a, V3 b) {                                                      /* Core.h:170 */
    return v3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
static inline float vlen2(V3 a) { return ((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); }
static inline float vlen(V3 a) { return sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); } /* Core.h:157 */
static inline V3 vnorm(V3 a) {                                                             /* Core.h:161 */
    float l = 1.0f / sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z));
    return v3(a.x * l, a.y * l, a.z * l);
}
static inline V3 vmax(V3 a, V3 b) { return v3(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z); } /* Core.h:187 */
static inline V3 vmin(V3 a, V3 b) { return v3(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z); } /* Core.h:192 */
static inline float vcoord(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
/* std::max / std::min semantics (NaN-sensitive): max(a,b) = (a<b)?b:a ; min(a,b) = (b<a)?b:a */
static inline float smax(float a, float b) { return (a < b) ? b : a; }
static inline float smin(float a, float b) { return (b < a) ? b : a; }

static inline Col col(float r, float g, float b) { Col c = { r, g, b }; return c; }
static inline Col cadd(Col a, Col b) { return col(a.r + b.r, a.g + b.g, a.b + b.b); }
static inline Col cmul(Col a, Col b) { return col(a.r * b.r, a.g * b.g, a.b * b.b); }
static inline Col cmulf(Col a, float s) { return col(a.r * s, a.g * s, a.b * s); }
static inline Col cdivf(Col a, float s) { return col(a.r / s, a.g / s, a.b / s); }
static inline float clum(Col c) { return ((0.2126f * c.r) + (0.7152f * c.g) + (0.0722f * c.b)); } /* Core.h:89 */

/* Matrix: row-major float[16] (ref: Core.h:205-505) */
static void mat_identity(float* m) { memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.0f; }
static V3 mat_mul_vec(const float* m, V3 v) { /* Core.h:295 */
    return v3((v.x * m[0] + v.y * m[1] + v.z * m[2]), (v.x * m[4] + v.y * m[5] + v.z * m[6]), (v.x * m[8] + v.y * m[9] + v.z * m[10]));
}
static V3 mat_mul_point(const float* m, V3 v) { /* Core.h:302 */
    return v3((v.x * m[0] + v.y * m[1] + v.z * m[2]) + m[3], (v.x * m[4] + v.y * m[5] + v.z * m[6]) + m[7],
              (v.x * m[8] + v.y * m[9] + v.z * m[10]) + m[11]);
}
static V3 mat_mul_point_div(const float* m, V3 v) { /* Core.h:310 */
    V3 v1 = v3((v.x * m[0] + v.y * m[1] + v.z * m[2]) + m[3], (v.x * m[4] + v.y * m[5] + v.z * m[6]) + m[7],
               (v.x * m[8] + v.y * m[9] + v.z * m[10]) + m[11]);
    float w = (m[12] * v.x) + (m[13] * v.y) + (m[14] * v.z) + m[15];
    w = 1.0f / w;
    return vmulf(v1, w);
}
/* Core.h:326-438: cofactor expansion (MESA). Written as a generic 3x3-minor
 * evaluation would round differently, so the 16 cofactors are spelled with the
 * same term order as the reference: each is a sum of six signed triple products. */
static void mat_invert(const float* m, float* out) {
    float inv[16];
#define T3(a, b, c) (m[a] * m[b] * m[c])
    inv[0] = T3(5, 10, 15) - T3(5, 11, 14) - T3(9, 6, 15) + T3(9, 7, 14) + T3(13, 6, 11) - T3(13, 7, 10);
    inv[4] = -m[4] * m[10] * m[15] + T3(4, 11, 14) + T3(8, 6, 15) - T3(8, 7, 14) - T3(12, 6, 11) + T3(12, 7, 10);
    inv[8] = T3(4, 9, 15) - T3(4, 11, 13) - T3(8, 5, 15) + T3(8, 7, 13) + T3(12, 5, 11) - T3(12, 7, 9);
    inv[12] = -m[4] * m[9] * m[14] + T3(4, 10, 13) + T3(8, 5, 14) - T3(8, 6, 13) - T3(12, 5, 10) + T3(12, 6, 9);
    inv[1] = -m[1] * m[10] * m[15] + T3(1, 11, 14) + T3(9, 2, 15) - T3(9, 3, 14) - T3(13, 2, 11) + T3(13, 3, 10);
    inv[5] = T3(0, 10, 15) - T3(0, 11, 14) - T3(8, 2, 15) + T3(8, 3, 14) + T3(12, 2, 11) - T3(12, 3, 10);
    inv[9] = -m[0] * m[9] * m[15] + T3(0, 11, 13) + T3(8, 1, 15) - T3(8, 3, 13) - T3(12, 1, 11) + T3(12, 3, 9);
    inv[13] = T3(0, 9, 14) - T3(0, 10, 13) - T3(8, 1, 14) + T3(8, 2, 13) + T3(12, 1, 10) - T3(12, 2, 9);
    inv[2] = T3(1, 6, 15) - T3(1, 7, 14) - T3(5, 2, 15) + T3(5, 3, 14) + T3(13, 2, 7) - T3(13, 3, 6);
    inv[6] = -m[0] * m[6] * m[15] + T3(0, 7, 14) + T3(4, 2, 15) - T3(4, 3, 14) - T3(12, 2, 7) + T3(12, 3, 6);
    inv[10] = T3(0, 5, 15) - T3(0, 7, 13) - T3(4, 1, 15) + T3(4, 3, 13) + T3(12, 1, 7) - T3(12, 3, 5);
    inv[14] = -m[0] * m[5] * m[14] + T3(0, 6, 13) + T3(4, 1, 14) - T3(4, 2, 13) - T3(12, 1, 6) + T3(12, 2, 5);
    inv[3] = -m[1] * m[6] * m[11] + T3(1, 7, 10) + T3(5, 2, 11) - T3(5, 3, 10) - T3(9, 2, 7) + T3(9, 3, 6);
    inv[7] = T3(0, 6, 11) - T3(0, 7, 10) - T3(4, 2, 11) + T3(4, 3, 10) + T3(8, 2, 7) - T3(8, 3, 6);
    inv[11] = -m[0] * m[5] * m[11] + T3(0, 7, 9) + T3(4, 1, 11) - T3(4, 3, 9) - T3(8, 1, 7) + T3(8, 3, 5);
    inv[15] = T3(0, 5, 10) - T3(0, 6, 9) - T3(4, 1, 10) + T3(4, 2, 9) + T3(8, 1, 6) - T3(8, 2, 5);
#undef T3
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0) { /* Core.h:426-431: `identity()` resets *this*, not inv; inv is still scaled by 1 */
        det = 1.0f;
    }
    det = 1.0f / det;
    for (int i = 0; i < 16; i++) out[i] = inv[i] * det;
}
static void mat_lookat(V3 from, V3 to, V3 up, float* m) { /* Core.h:439 */
    mat_identity(m);
    V3 dir = vnorm(vsub(from, to));
    V3 left = vnorm(vcross(up, dir));
    V3 newUp = vcross(dir, left);
    m[0] = left.x; m[1] = left.y; m[2] = left.z;
    m[4] = newUp.x; m[5] = newUp.y; m[6] = newUp.z;
    m[8] = dir.x; m[9] = dir.y; m[10] = dir.z;
    m[3] = -vdot(from, left); m[7] = -vdot(from, newUp); m[11] = -vdot(from, dir);
    m[15] = 1;
}
static void mat_perspective(float n, float f, float aspect, float fov, float* m) { /* Core.h:460 */
    memset(m, 0, 64);
    float t = 1.0f / (tanf(fov * 0.5f * 3.141592654f / 180.0f));
    m[0] = t / aspect; m[5] = t; m[10] = -f / (f - n); m[11] = -(f * n) / (f - n); m[14] = -1.0f;
}

typedef struct { V3 u, v, w; } Frame;
static Frame frame_from_vector(V3 n) { /* Core.h:513 */
    Frame f;
    f.w = vnorm(n);
    if (fabsf(f.w.x) > fabsf(f.w.y)) {
        float l = 1.0f / sqrtf(f.w.x * f.w.x + f.w.z * f.w.z);
        f.u = v3(f.w.z * l, 0.0f, -f.w.x * l);
    } else {
        float l = 1.0f / sqrtf(f.w.y * f.w.y + f.w.z * f.w.z);
        f.u = v3(0, f.w.z * l, -f.w.y * l);
    }
    f.v = vcross(f.w, f.u);
    return f;
}
static inline V3 frame_to_local(const Frame* f, V3 a) { return v3(vdot(a, f->u), vdot(a, f->v), vdot(a, f->w)); } /* Core.h:534 */
static inline V3 frame_to_world(const Frame* f, V3 a) { /* Core.h:538 */
    return vadd(vadd(vmulf(f->u, a.x), vmulf(f->v, a.y)), vmulf(f->w, a.z));
}

/* ---------------------------------------------------------------- RNG (shared spec with the GPU) */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
typedef struct { uint32_t key, ctr; } Rng;
static inline Rng rng_make(uint64_t seed, uint64_t sample, uint32_t pixel, uint32_t stream) {
    uint32_t k = mix32((uint32_t)seed ^ 0xA511E9B3U);
    k = mix32(k ^ (uint32_t)(seed >> 32));
    k = mix32(k + stream * 0xC2B2AE35U);
    k = mix32(k + (uint32_t)sample * 0x9E3779B9U);
    k = mix32(k ^ (uint32_t)(sample >> 32));
    k = mix32(k + pixel * 0x85EBCA6BU);
    Rng r = { k, 0 };
    return r;
}
static inline float rng_next(Rng* r) { /* replaces MTRandom::next (Sampling.h:32): uniform in [0,1) */
    uint32_t x = mix32(r->key + r->ctr * 0x9E3779B9U);
    r->ctr++;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

/* ---------------------------------------------------------------- geometry (ref: Geometry.h) */
#define EPSILON 0.001f /* Geometry.h:49 */
#define BINS_COUNT 8   /* Geometry.h:271 */
#define MAX_DEPTH 4    /* Renderer.h:18 */

typedef struct {
    V3 p[3], nrm[3];
    float u[3], v[3];
    V3 e1, e2, n;
    float area, d;
    uint32_t mat;
} Tri;

static void tri_init(Tri* t, const float* f, uint32_t mat) { /* Geometry.h:66-82 */
    for (int k = 0; k < 3; k++) {
        t->p[k] = v3(f[k * 8 + 0], f[k * 8 + 1], f[k * 8 + 2]);
        t->nrm[k] = v3(f[k * 8 + 3], f[k * 8 + 4], f[k * 8 + 5]);
        t->u[k] = f[k * 8 + 6];
        t->v[k] = f[k * 8 + 7];
    }
    t->mat = mat;
    t->e1 = vsub(t->p[2], t->p[1]);
    t->e2 = vsub(t->p[0], t->p[2]);
    t->n = vnorm(vcross(t->e1, t->e2));
    t->area = vlen(vcross(t->e1, t->e2)) * 0.5f;
    t->d = vdot(t->n, t->p[0]);
}
static inline V3 tri_centre(const Tri* t) { /* Geometry.h:83 */
    return vdivf(vadd(vadd(t->p[0], t->p[1]), t->p[2]), 3.0f);
}
/* Geometry.h:115-128 — the production intersector. u,v weight vertices 0 and 1. */
static inline int tri_intersect(const Tri* tr, V3 o, V3 dir, float* t, float* u, float* v) {
    float denom = vdot(tr->n, dir);
    if (denom == 0) return 0;
    *t = (tr->d - vdot(tr->n, o)) / denom;
    if (*t < 0) return 0;
    V3 p = vadd(o, vmulf(dir, *t));
    float invArea = 1.0f / vdot(vcross(tr->e1, tr->e2), tr->n);
    *u = vdot(vcross(tr->e1, vsub(p, tr->p[1])), tr->n) * invArea;
    if (*u < 0 || *u > 1.0f) return 0;
    *v = vdot(vcross(tr->e2, vsub(p, tr->p[2])), tr->n) * invArea;
    if (*v < 0 || (*u + *v) > 1.0f) return 0;
    return 1;
}
static inline V3 tri_gnormal(const Tri* t) { /* Geometry.h:175 */
    return vmulf(t->n, (vdot(t->nrm[0], t->n) > 0 ? 1.0f : -1.0f));
}
static inline V3 tri_sample(const Tri* t, float r1, float r2, float* pdf) { /* Geometry.h:162 */
    float alpha = 1 - sqrtf(r1);
    float beta = r2 * sqrtf(r1);
    float gamma = 1 - (alpha + beta);
    *pdf = 1 / t->area;
    return vadd(vadd(vmulf(t->p[0], alpha), vmulf(t->p[1], beta)), vmulf(t->p[2], gamma));
}

typedef struct { V3 bmax, bmin; } Box; /* Geometry.h:181 */
static inline void box_reset(Box* b) { b->bmax = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX); b->bmin = v3(FLT_MAX, FLT_MAX, FLT_MAX); }
static inline void box_extend(Box* b, V3 p) { b->bmax = vmax(b->bmax, p); b->bmin = vmin(b->bmin, p); } /* Geometry.h:195 */
static inline void box_extend_box(Box* b, const Box* a) { box_extend(b, a->bmax); box_extend(b, a->bmin); } /* Geometry.h:201 */
static inline float box_area(const Box* b) { /* Geometry.h:231 */
    V3 s = vsub(b->bmax, b->bmin);
    return ((s.x * s.y) + (s.y * s.z) + (s.x * s.z)) * 2.0f;
}
/* Geometry.h:207-229: slab test with ternary Min/Max and std::max/min, no clamp to [0, best]. */
static inline int box_ray(const Box* b, V3 o, V3 inv, float* t) {
    V3 tmin = vmul(vsub(b->bmin, o), inv);
    V3 tmax = vmul(vsub(b->bmax, o), inv);
    V3 Tentry = vmin(tmin, tmax);
    V3 Texit = vmax(tmin, tmax);
    float tentry = smax(Tentry.x, smax(Tentry.y, Tentry.z));
    float texit = smin(Texit.x, smin(Texit.y, Texit.z));
    *t = smin(tentry, texit);
    return (tentry <= texit);
}

typedef struct {
    Box bounds;
    int l, r;   /* child node indices or -1 */
    int num;    /* triangles under this node */
    int first;  /* leaves: offset into leaf_tris; */
    int nleaf;  /* leaves: number of triangles kept (0 for inner nodes, ref `triangles.clear()`) */
} Node;

/* ---------------------------------------------------------------- scene */
typedef struct { int width, height; Col* texels; float* alpha; } Texture;

typedef struct {
    V3 x, wo, sN, gN;
    float tu, tv;
    Frame frame;
    int mat; /* -1 == NULL bsdf */
    float t;
} Shading;

typedef struct { Shading sd; Col Le; } VPL; /* Renderer.h:21 */

typedef struct { int kind; /* 0 area, 1 background colour, 2 env map */ uint32_t tri; } LightRef;

struct orc_scene {
    Tri* tris; uint32_t ntris;
    Node* nodes; int nnodes, cap_nodes;
    uint32_t* leaf_tris; uint32_t nleaf_tris;
    int max_depth, nleaves, max_leaf;
    orc_material* mats; uint32_t nmats;
    Texture* tex; uint32_t ntex;
    orc_background bg;
    LightRef* lights; uint32_t nlights;
    Box bounds; V3 scene_centre; float scene_radius; /* SceneLoader.h:286-287 */
    /* camera (Scene.h:14-45) */
    float P[16], invP[16], cam[16], cam2view[16];
    float width, height;
    V3 origin, view_dir; float Afilm;
    VPL* vpls; int nvpls, cap_vpls;
    int filter; /* 0 BoxFilter (Renderer.h:61), 1 GaussianFilter, 2 MitchellNetravaliFilter */
    /* EnvironmentMap::build (Lights.h:207-253): per-row-normalised pdfs, row pdfs, total; NULL without an environment map */
    float* env_cond; float* env_marg; float env_func_int; int env_nu, env_nv;
};

/* ---------------------------------------------------------------- BVH build (ref: Geometry.h:310-438) */
static int node_new(orc_scene* s) {
    if (s->nnodes == s->cap_nodes) {
        s->cap_nodes = s->cap_nodes ? s->cap_nodes * 2 : 1024;
        s->nodes = (Node*)realloc(s->nodes, sizeof(Node) * (size_t)s->cap_nodes);
    }
    Node* n = &s->nodes[s->nnodes];
    box_reset(&n->bounds);
    n->l = n->r = -1; n->num = 0; n->first = 0; n->nleaf = 0;
    return s->nnodes++;
}

/* Geometry.h:319-383 */
static float find_best_split(const orc_scene* s, const Box* bounds, const uint32_t* tris, int num, int* axis, float* splitPos) {
    float bestCost = FLT_MAX;
    float parentArea = box_area(bounds);
    for (int a = 0; a < 3; a++) {
        float boundsMin = FLT_MAX, boundsMax = -FLT_MAX;
        for (int i = 0; i < num; i++) {
            float c = vcoord(tri_centre(&s->tris[tris[i]]), a);
            boundsMin = smin(boundsMin, c);
            boundsMax = smax(boundsMax, c);
        }
        if (boundsMin == boundsMax) continue;
        Box binb[BINS_COUNT]; int binc[BINS_COUNT];
        for (int i = 0; i < BINS_COUNT; i++) { box_reset(&binb[i]); binc[i] = 0; }
        float scale = BINS_COUNT / (boundsMax - boundsMin);
        for (int i = 0; i < num; i++) {
            const Tri* t = &s->tris[tris[i]];
            float c = vcoord(tri_centre(t), a);
            int bi = (int)((c - boundsMin) * scale);
            if (bi > BINS_COUNT - 1) bi = BINS_COUNT - 1; /* std::min(BINS_COUNT-1, idx) */
            binc[bi]++;
            box_extend(&binb[bi], t->p[0]); box_extend(&binb[bi], t->p[1]); box_extend(&binb[bi], t->p[2]);
        }
        float leftArea[BINS_COUNT - 1], rightArea[BINS_COUNT - 1];
        int leftCount[BINS_COUNT - 1], rightCount[BINS_COUNT - 1];
        Box leftBox, rightBox; box_reset(&leftBox); box_reset(&rightBox);
        int leftSum = 0, rightSum = 0;
        for (int i = 0; i < BINS_COUNT - 1; i++) {
            leftSum += binc[i];
            leftCount[i] = leftSum;
            if (binc[i] > 0) { box_extend_box(&leftBox, &binb[i]); leftArea[i] = box_area(&leftBox); }
            else leftArea[i] = 0; /* quirk: area is zeroed after an empty bin, Geometry.h:360 */
            rightSum += binc[(BINS_COUNT - 1) - i];
            rightCount[(BINS_COUNT - 2) - i] = rightSum;
            if (binc[(BINS_COUNT - 1) - i] > 0) {
                box_extend_box(&rightBox, &binb[(BINS_COUNT - 1) - i]);
                rightArea[(BINS_COUNT - 2) - i] = box_area(&rightBox);
            } else rightArea[(BINS_COUNT - 2) - i] = 0;
        }
        float BinScale = (boundsMax - boundsMin) / BINS_COUNT;
        for (int i = 0; i < BINS_COUNT - 1; i++) {
            float cost = 1.0f + ((leftArea[i] / parentArea) * leftCount[i] * 1.0f) + ((rightArea[i] / parentArea) * rightCount[i] * 1.0f);
            if (cost < bestCost) { bestCost = cost; *axis = a; *splitPos = boundsMin + BinScale * (i + 1); }
        }
    }
    return bestCost;
}

/* Geometry.h:391-438.  `tris` is consumed (freed) here. */
static void build_rec(orc_scene* s, int ni, uint32_t* tris, int num, int depth) {
    if (depth > s->max_depth) s->max_depth = depth;
    s->nodes[ni].num = num;
    if (num > 1) {
        Box b; box_reset(&b);
        for (int i = 0; i < num; i++) {
            const Tri* t = &s->tris[tris[i]];
            box_extend(&b, t->p[0]); box_extend(&b, t->p[1]); box_extend(&b, t->p[2]);
        }
        s->nodes[ni].bounds = b;
        int axis = -1; float splitPos = 0.0f;
        find_best_split(s, &b, tris, num, &axis, &splitPos);
        /* axis == -1 (every centroid identical on all axes): the reference reads
         * coords[-1] (UB, Geometry.h:418).  The oracle defines it as "no split". */
        if (axis >= 0) {
            uint32_t* left = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)num);
            uint32_t* right = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)num);
            int nl = 0, nr = 0;
            for (int i = 0; i < num; i++) {
                if (vcoord(tri_centre(&s->tris[tris[i]]), axis) < splitPos) left[nl++] = tris[i];
                else right[nr++] = tris[i];
            }
            if (nl > 0 && nr > 0) {
                free(tris);
                int l = node_new(s); s->nodes[ni].l = l;
                build_rec(s, l, left, nl, depth + 1); /* frees left; left subtree emits its leaves first */
                int r = node_new(s); s->nodes[ni].r = r;
                build_rec(s, r, right, nr, depth + 1);
                return;
            }
            free(left); free(right);
        }
    }
    /* leaf: num <= 1 (box left inverted, Geometry.h:401) or a failed split (Geometry.h:426) */
    s->nodes[ni].first = (int)s->nleaf_tris;
    s->nodes[ni].nleaf = num;
    for (int i = 0; i < num; i++) s->leaf_tris[s->nleaf_tris++] = tris[i];
    s->nleaves++;
    if (num > s->max_leaf) s->max_leaf = num;
    free(tris);
}

/* ---------------------------------------------------------------- traversal (ref: Geometry.h:441-490) */
typedef struct { uint64_t nodes, tris; } Cnt;

static void traverse_rec(const orc_scene* s, int ni, V3 o, V3 dir, V3 inv, orc_hit* best, Cnt* c) {
    const Node* n = &s->nodes[ni];
    float tb;
    c->nodes++;
    if (!box_ray(&n->bounds, o, inv, &tb)) return;
    if (n->l < 0 && n->r < 0) {
        for (int i = 0; i < n->nleaf; i++) {
            uint32_t id = s->leaf_tris[n->first + i];
            float t, u, v;
            c->tris++;
            if (tri_intersect(&s->tris[id], o, dir, &t, &u, &v) && t < best->t) {
                best->t = t; best->alpha = u; best->beta = v; best->gamma = 1.0f - (u + v); best->ID = id;
            }
        }
        return;
    }
    if (n->l >= 0) traverse_rec(s, n->l, o, dir, inv, best, c);
    if (n->r >= 0) traverse_rec(s, n->r, o, dir, inv, best, c);
}
static int visible_rec(const orc_scene* s, int ni, V3 o, V3 dir, V3 inv, float maxT, Cnt* c) {
    const Node* n = &s->nodes[ni];
    float t, u, v;
    c->nodes++;
    if (!box_ray(&n->bounds, o, inv, &t) || t > maxT) return 1;
    if (n->l < 0 && n->r < 0) {
        for (int i = 0; i < n->nleaf; i++) {
            c->tris++;
            if (tri_intersect(&s->tris[s->leaf_tris[n->first + i]], o, dir, &t, &u, &v) && t < maxT) return 0;
        }
        return 1;
    }
    if (n->l >= 0 && !visible_rec(s, n->l, o, dir, inv, maxT, c)) return 0;
    if (n->r >= 0 && !visible_rec(s, n->r, o, dir, inv, maxT, c)) return 0;
    return 1;
}
/* Scene::traverse (Scene.h:118) with Ray::init (Geometry.h:19) */
static orc_hit scene_traverse(const orc_scene* s, V3 o, V3 dir, Cnt* c) {
    orc_hit h; h.t = FLT_MAX; h.ID = 0xFFFFFFFFu; h.alpha = h.beta = h.gamma = 0;
    V3 inv = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    if (s->nnodes > 0) traverse_rec(s, 0, o, dir, inv, &h, c);
    return h;
}
/* Scene::visible (Scene.h:169) */
static int scene_visible(const orc_scene* s, V3 p1, V3 p2, Cnt* c) {
    V3 dir = vsub(p2, p1);
    float maxT = vlen(dir) - (2.0f * EPSILON);
    dir = vnorm(dir);
    V3 o = vadd(p1, vmulf(dir, EPSILON));
    V3 inv = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    if (s->nnodes == 0) return 1;
    return visible_rec(s, 0, o, dir, inv, maxT, c);
}

/* ---------------------------------------------------------------- textures (ref: Imaging.h:71-93) */
static Col tex_sample(const Texture* tx, float tu, float tv) {
    float u = smax(0.0f, fabsf(tu)) * tx->width;
    float v = smax(0.0f, fabsf(tv)) * tx->height;
    int x = (int)floorf(u), y = (int)floorf(v);
    float fu = u - x, fv = v - y;
    float w0 = (1.0f - fu) * (1.0f - fv), w1 = fu * (1.0f - fv), w2 = (1.0f - fu) * fv, w3 = fu * fv;
    x = x % tx->width; y = y % tx->height;
    Col s0 = tx->texels[y * tx->width + x];
    Col s1 = tx->texels[y * tx->width + ((x + 1) % tx->width)];
    Col s2 = tx->texels[((y + 1) % tx->height) * tx->width + x];
    Col s3 = tx->texels[((y + 1) % tx->height) * tx->width + ((x + 1) % tx->width)];
    return cadd(cadd(cadd(cmulf(s0, w0), cmulf(s1, w1)), cmulf(s2, w2)), cmulf(s3, w3));
}

/* ---------------------------------------------------------------- sampling (ref: Sampling.h:143-197) */
static V3 cosine_sample_hemisphere(float r1, float r2) { /* Sampling.h:163 */
    float theta = acosf(sqrtf(r1));
    float phi = 2 * M_PI * r2;
    return v3(1.f * sinf(theta) * cosf(phi), 1.f * sinf(theta) * sinf(phi), 1.f * cosf(theta));
}
static float cosine_hemisphere_pdf(V3 wi) { return wi.z / M_PI; } /* Sampling.h:175 */
static V3 uniform_sample_sphere(float r1, float r2) { /* Sampling.h:180 */
    float theta = acosf(1 - (2 * r1));
    float phi = 2 * M_PI * r2;
    return v3(1.f * sinf(theta) * cosf(phi), 1.f * sinf(theta) * sinf(phi), 1.f * cosf(theta));
}
static float uniform_sphere_pdf(void) { return 1.0f / 4 * M_PI; } /* Sampling.h:192: == pi/4 (sic) */
static V3 uniform_sample_hemisphere(float r1, float r2) { /* Sampling.h:146 */
    float theta = acosf(r1);
    float phi = 2 * M_PI * r2;
    return v3(1.f * sinf(theta) * cosf(phi), 1.f * sinf(theta) * sinf(phi), 1.f * cosf(theta));
}

/* ---------------------------------------------------------------- BSDFs (ref: Materials.h) */
enum { M_DIFFUSE = 0, M_MIRROR, M_CONDUCTOR, M_GLASS, M_DIELECTRIC, M_ORENNAYAR, M_PLASTIC };

static inline float clampf(float x, float lo, float hi) { return smax(lo, smin(x, hi)); } /* Sampling.h:16 */

static float fresnel_dielectric(float cosTheta, float iorInt, float iorExt) { /* Materials.h:39-66 */
    cosTheta = clampf(cosTheta, -1.0f, 1.0f);
    int entering = cosTheta > 0.f;
    if (entering) { float tmp = iorInt; iorInt = iorExt; iorExt = tmp; cosTheta = fabsf(cosTheta); }
    float ri = iorExt / iorInt;
    float cosI = cosTheta;
    float sinI = sqrtf(smax(0.0f, 1 - powf(cosI, 2)));
    float sinT = ri * sinI;
    if (sinT >= 1.f) return 1.f;
    float cosT = sqrtf(smax(0.0f, 1 - powf(sinT, 2)));
    float par = (cosI - (ri * cosT)) / (cosI + (ri * cosT));
    float per = ((cosI * ri) - cosT) / ((cosI * ri) + cosT);
    return 0.5f * ((par * par) + (per * per));
}
static Col fresnel_conductor(float cosTheta, Col ior, Col k) { /* Materials.h:67-90 */
    cosTheta = fabsf(cosTheta);
    float c2 = powf(cosTheta, 2);
    float s2 = 1 - c2;
    Col cos2 = col(c2, c2, c2), sin2 = col(s2, s2, s2);
    Col nk2 = cadd(cmul(ior, ior), cmul(k, k));
    Col twoNCos = cmulf(cmulf(ior, cosTheta), 2);
    Col num = cadd(col(nk2.r * c2 - twoNCos.r, nk2.g * c2 - twoNCos.g, nk2.b * c2 - twoNCos.b), sin2);
    Col den = cadd(cadd(cmulf(nk2, c2), twoNCos), sin2);
    Col par = col(num.r / den.r, num.g / den.g, num.b / den.b);
    Col num2 = cadd(col(nk2.r - twoNCos.r, nk2.g - twoNCos.g, nk2.b - twoNCos.b), cos2);
    Col den2 = cadd(cadd(nk2, twoNCos), cos2);
    Col per = col(num2.r / den2.r, num2.g / den2.g, num2.b / den2.b);
    /* SQ(x) is the macro (x * x) with unparenthesised Colour expressions: fine for plain identifiers */
    return cmulf(cadd(cmul(par, par), cmul(per, per)), 0.5f);
}
static float lambda_ggx(V3 wi, float alpha) { /* Materials.h:93-101 */
    float sin2 = smax(0.0f, 1 - wi.z * wi.z);
    float tanT = sqrtf(sin2) / wi.z;
    float a = fabsf(tanT);
    float a2t2 = (alpha * a) * (alpha * a);
    return (-1 + sqrtf(1 + a2t2)) * 0.5f;
}
static float g_ggx(V3 wi, V3 wo, float alpha) { return 1.0f / (1.0f + lambda_ggx(wi, alpha) + lambda_ggx(wo, alpha)); } /* :102 */
static float d_ggx(V3 h, float alpha) { /* Materials.h:109-117 */
    float c2 = h.z * h.z;
    float a2 = powf(alpha, 2);
    float den = (c2 * (a2 - 1)) + 1;
    return a2 / (den * den * M_PI);
}
static void set_ior(float cosTheta, float* iorI, float* iorT, float intIOR, float extIOR) { /* Materials.h:119 */
    int entering = cosTheta > 0;
    *iorI = entering ? extIOR : intIOR;
    *iorT = entering ? intIOR : extIOR;
}
static inline float mat_alpha(const orc_material* m) { return 1.62142f * sqrtf(m->roughness); } /* Materials.h:307,543,855 */
static inline Col mat_emission(const orc_material* m) { return col(m->emission[0], m->emission[1], m->emission[2]); }
static inline int mat_is_light(const orc_material* m) { return clum(mat_emission(m)) > 0; } /* Materials.h:184 */
static inline int mat_is_specular(const orc_material* m) { return m->type == M_MIRROR || m->type == M_GLASS; }
static inline int mat_two_sided(const orc_material* m) { return (m->flags & 1u) ? 1 : (m->type != M_DIELECTRIC); } /* :738, :1044 */
static inline Col mat_albedo(const orc_scene* s, const orc_material* m, float tu, float tv) {
    return tex_sample(&s->tex[m->albedo_tex], tu, tv);
}
static inline float plastic_exponent(float alpha) { /* Materials.h:857: SQ(std::max(alpha,0.001f)) expands to max*max */
    float a = smax(alpha, 0.001f);
    return (2.0f / (a * a)) - 2.0f;
}

/* BSDF::evaluate */
static Col bsdf_eval(const orc_scene* s, const Shading* sd, V3 wi) {
    const orc_material* m = &s->mats[sd->mat];
    switch (m->type) {
    case M_DIFFUSE:   /* Materials.h:224 */
    case M_ORENNAYAR: /* Materials.h:777 (early return: Lambert) */
        return cdivf(mat_albedo(s, m, sd->tu, sd->tv), (float)M_PI);
    case M_MIRROR: case M_GLASS: /* Materials.h:268, :503 */
        return col(0, 0, 0);
    case M_CONDUCTOR: { /* Materials.h:360-391 */
        float alpha = mat_alpha(m);
        V3 wo = frame_to_local(&sd->frame, sd->wo), wl = frame_to_local(&sd->frame, wi);
        if (wo.z <= 0.0f || wl.z <= 0.0f) return col(0, 0, 0);
        V3 wm = vnorm(vadd(wl, wo));
        if (wm.z <= 0.0f) return col(0, 0, 0);
        if (alpha < 0.001f) return col(0, 0, 0);
        float D = d_ggx(wm, alpha), G = g_ggx(wl, wo, alpha);
        Col F = fresnel_conductor(vdot(wo, wm), col(m->eta[0], m->eta[1], m->eta[2]), col(m->k[0], m->k[1], m->k[2]));
        Col brdf = cdivf(cmulf(cmulf(F, D), G), (4.0f * wo.z * wl.z));
        return cmul(mat_albedo(s, m, sd->tu, sd->tv), brdf);
    }
    case M_DIELECTRIC: { /* Materials.h:638-686 */
        float alpha = mat_alpha(m);
        V3 wo = frame_to_local(&sd->frame, sd->wo), wl = frame_to_local(&sd->frame, wi);
        if (wo.z <= 0.0f) return col(0, 0, 0);
        float iorT, iorI; set_ior(wo.z, &iorI, &iorT, m->int_ior, m->ext_ior);
        float ri = iorI / iorT;
        V3 wm;
        int reflect = wl.z * wo.z > 0;
        if (reflect) wm = vnorm(vadd(wl, wo));
        else { wm = vnorm(vadd(vmulf(wl, ri), wo)); if (wm.z < 0) wm = vneg(wm); }
        if (wm.z <= 0.0f) return col(0, 0, 0);
        float F = fresnel_dielectric(vdot(wo, wm), iorI, iorT);
        float D = d_ggx(wm, alpha), G = g_ggx(wl, wo, alpha);
        float result;
        if (reflect) result = F * D * G / (4.0f * wo.z * wl.z);
        else {
            float den = ri * vdot(wl, wm) + vdot(wo, wm);
            result = (1 - F) * D * G * vdot(wl, wm) * vdot(wo, wm) / (powf(den, 2) * wl.z * wo.z);
        }
        return cmulf(mat_albedo(s, m, sd->tu, sd->tv), result);
    }
    case M_PLASTIC: { /* Materials.h:931-961 */
        V3 wo = frame_to_local(&sd->frame, sd->wo), wl = frame_to_local(&sd->frame, wi);
        if (wl.z <= 0.0f || wo.z <= 0.0f) return col(0, 0, 0);
        float cosTheta = wo.z, iorT, iorI; set_ior(cosTheta, &iorI, &iorT, m->int_ior, m->ext_ior);
        float ks = fresnel_dielectric(cosTheta, iorI, iorT), kd = 1.0f - ks;
        float e = plastic_exponent(mat_alpha(m));
        Col diffuse = cdivf(cmulf(mat_albedo(s, m, sd->tu, sd->tv), kd), (float)M_PI);
        V3 wr = v3(-wo.x, -wo.y, wo.z);
        float cosAlpha = smax(0.0f, vdot(wr, wl));
        float gloss = ks * ((e + 2) / (2 * M_PI)) * powf(cosAlpha, e);
        return cadd(diffuse, col(gloss, gloss, gloss));
    }
    }
    return col(0, 0, 0);
}

/* BSDF::PDF (Materials.h:228, :274, :392, :509, :687, :822, :962) */
static float bsdf_pdf(const orc_scene* s, const Shading* sd, V3 wi) {
    const orc_material* m = &s->mats[sd->mat];
    V3 wo = frame_to_local(&sd->frame, sd->wo), wl = frame_to_local(&sd->frame, wi);
    switch (m->type) {
    case M_DIFFUSE: case M_ORENNAYAR: return cosine_hemisphere_pdf(wl);
    case M_MIRROR: case M_GLASS: return 0;
    case M_CONDUCTOR: {
        float alpha = mat_alpha(m);
        if (wo.z <= 0.0f || wl.z <= 0.0f) return 0.0f;
        V3 wm = vnorm(vadd(wl, wo));
        if (wm.z <= 0.0f) return 0.0f;
        if (alpha < 0.001f) return 0.0f;
        return d_ggx(wm, alpha) * wm.z / (4 * vdot(wo, wm));
    }
    case M_DIELECTRIC: {
        float alpha = mat_alpha(m);
        if (wo.z <= 0.0f) return 0.0f;
        float iorT, iorI; set_ior(wo.z, &iorI, &iorT, m->int_ior, m->ext_ior);
        float ri = iorI / iorT;
        V3 wm; int reflect = wl.z * wo.z > 0;
        if (reflect) wm = vnorm(vadd(wl, wo));
        else { wm = vnorm(vadd(vmulf(wl, ri), wo)); if (wm.z < 0) wm = vneg(wm); }
        if (wm.z <= 0.0f) return 0.0f;
        float F = fresnel_dielectric(vdot(wo, wm), iorI, iorT);
        float D = d_ggx(wm, alpha);
        float pr = D * wm.z / (4 * vdot(wo, wm)), pt = 0.0f;
        if (!reflect) { float den = ri * vdot(wl, wm) + vdot(wo, wm); pt = (D * vdot(wl, wm)) / powf(den, 2); }
        return F * pr + (1 - F) * pt;
    }
    case M_PLASTIC: {
        if (wl.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
        float iorT, iorI; set_ior(wo.z, &iorI, &iorT, m->int_ior, m->ext_ior);
        float ks = fresnel_dielectric(wo.z, iorI, iorT), kd = 1.0f - ks;
        float e = plastic_exponent(mat_alpha(m));
        float pd = wl.z / M_PI;
        V3 wr = v3(-wo.x, -wo.y, wo.z);
        float cosAlpha = smax(0.0f, vdot(wr, wl));
        float pg = ((e + 1) * powf(cosAlpha, e) / (2 * M_PI));
        return (kd * pd) + (ks * pg);
    }
    }
    return 0;
}

/* BSDF::sample.  Random numbers are drawn in statement order; where the
 * reference draws two inside one call expression (unspecified order) the first
 * draw is r1. */
static V3 bsdf_sample(const orc_scene* s, const Shading* sd, Rng* rng, Col* f, float* pdf) {
    const orc_material* m = &s->mats[sd->mat];
    switch (m->type) {
    case M_DIFFUSE: { /* Materials.h:208-222 */
        float r1 = rng_next(rng), r2 = rng_next(rng);
        V3 wi = cosine_sample_hemisphere(r1, r2);
        *pdf = wi.z / M_PI;
        *f = cdivf(mat_albedo(s, m, sd->tu, sd->tv), (float)M_PI);
        return frame_to_world(&sd->frame, wi);
    }
    case M_ORENNAYAR: { /* Materials.h:759-773 */
        float r1 = rng_next(rng), r2 = rng_next(rng);
        V3 wl = cosine_sample_hemisphere(r1, r2);
        *pdf = wl.z / M_PI;
        V3 wi = frame_to_world(&sd->frame, wl);
        *f = bsdf_eval(s, sd, wi);
        return wi;
    }
    case M_MIRROR: { /* Materials.h:257-267 */
        V3 wo = frame_to_local(&sd->frame, sd->wo);
        V3 wi = v3(-wo.x, -wo.y, wo.z);
        *pdf = 1.0f;
        *f = cdivf(mat_albedo(s, m, sd->tu, sd->tv), wi.z);
        return frame_to_world(&sd->frame, wi);
    }
    case M_CONDUCTOR: { /* Materials.h:309-359 */
        float alpha = mat_alpha(m);
        Col eta = col(m->eta[0], m->eta[1], m->eta[2]), kk = col(m->k[0], m->k[1], m->k[2]);
        V3 wo = frame_to_local(&sd->frame, sd->wo);
        if (wo.z <= 0.0f) { *pdf = 0.0f; *f = col(0, 0, 0); return v3(0, 0, 0); }
        if (alpha < 0.001f) {
            V3 wi = v3(-wo.x, -wo.y, wo.z);
            *pdf = 1.0f;
            Col F = fresnel_conductor(wi.z, eta, kk);
            *f = cmul(mat_albedo(s, m, sd->tu, sd->tv), F);
            return frame_to_world(&sd->frame, wi);
        }
        float r1 = rng_next(rng), r2 = rng_next(rng);
        float a2 = powf(alpha, 2);
        float cosTheta = sqrtf((1 - r1) / (r1 * ((a2 - 1) + 1))); /* sic: not a GGX inversion */
        float sinTheta = sqrtf(smax(0.0f, 1 - powf(cosTheta, 2)));
        float phi = 2 * M_PI * r2;
        V3 wm = v3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
        V3 wl = vsub(vmulf(vmulf(wm, vdot(wo, wm)), 2), wo);
        float D = d_ggx(wm, alpha), G = g_ggx(wl, wo, alpha);
        Col F = fresnel_conductor(vdot(wl, wm), eta, kk);
        Col brdf = cdivf(cmulf(cmulf(F, D), G), (4.0f * wo.z * wl.z));
        *pdf = D * wm.z / (4 * vdot(wo, wm));
        *f = cdivf(cmulf(cmul(mat_albedo(s, m, sd->tu, sd->tv), brdf), fabsf(wl.z)), *pdf);
        return frame_to_world(&sd->frame, wl);
    }
    case M_GLASS: { /* Materials.h:447-502 */
        V3 wo = frame_to_local(&sd->frame, sd->wo);
        float cosTheta = wo.z, iorT, iorI;
        set_ior(cosTheta, &iorI, &iorT, m->int_ior, m->ext_ior);
        float ri = iorI / iorT;
        float F = fresnel_dielectric(cosTheta, iorI, iorT);
        V3 wi = v3(0, 0, 0);
        int reflect = rng_next(rng) < F;
        if (reflect) {
            wi = v3(-wo.x, -wo.y, wo.z);
            *f = mat_albedo(s, m, sd->tu, sd->tv);
            *pdf = F;
        } else {
            float sinI = sqrtf(smax(0.0f, 1.0f - powf(cosTheta, 2)));
            float sinT = ri * sinI;
            if (sinT >= 1.0f) {
                wi = v3(-wo.x, -wo.y, wo.z);
                *f = mat_albedo(s, m, sd->tu, sd->tv);
                *pdf = 1.0f;
            } else {
                float cosT = sqrtf(smax(0.0f, 1 - powf(sinT, 2)));
                wi = v3(-wo.x * ri, -wo.y * ri, 0);
                wi.z = cosTheta > 0 ? -cosT : cosT;
                wi = vnorm(wi);
                *pdf = (1.0f - F);
                *f = cmulf(mat_albedo(s, m, sd->tu, sd->tv), (*pdf * ri * ri));
            }
        }
        return frame_to_world(&sd->frame, vnorm(wi));
    }
    case M_DIELECTRIC: { /* Materials.h:545-637 */
        float alpha = mat_alpha(m);
        V3 wo = frame_to_local(&sd->frame, sd->wo);
        float cosTheta = wo.z;
        if (cosTheta <= 0.0f) { *pdf = 0.0f; *f = col(0, 0, 0); return v3(0, 0, 0); }
        float iorT, iorI; set_ior(cosTheta, &iorI, &iorT, m->int_ior, m->ext_ior);
        float ri = iorI / iorT;
        float r1 = rng_next(rng), r2 = rng_next(rng);
        float a2 = powf(alpha, 2);
        float cosM = sqrtf((1 - r1) / (r1 * ((a2 - 1) + 1)));
        float sinM = sqrtf(smax(0.0f, 1 - powf(cosM, 2)));
        float phi = 2 * M_PI * r2;
        V3 wm = v3(sinM * cosf(phi), sinM * sinf(phi), cosM);
        V3 wl = v3(0, 0, 0); /* Materials.h:577: default-constructed Vec3 => Fresnel at cos = 0 */
        float pr, pt;
        float F = fresnel_dielectric(vdot(wl, wm), iorI, iorT);
        float D = d_ggx(wm, alpha);
        int reflect = 0;
        if (rng_next(rng) < F) {
            reflect = 1;
            wl = vadd(vneg(wo), vmulf(vmulf(wm, vdot(wo, wm)), 2));
            pr = D * cosM / (4 * vdot(wo, wm)); pt = 0;
        } else {
            float cosI = vdot(wo, wm);
            float sinI = sqrtf(smax(0.0f, 1 - powf(cosI, 2)));
            float sinT = ri * sinI;
            if (sinT >= 1.0f) {
                reflect = 1;
                wl = vadd(vneg(wo), vmulf(vmulf(wm, vdot(wo, wm)), 2));
                pr = D * cosM / (4 * vdot(wo, wm)); pt = 0; F = 1.0f;
            } else {
                float cosT = sqrtf(smax(0.0f, 1 - powf(sinT, 2)));
                wl = vadd(vmulf(vneg(wo), ri), vmulf(wm, (ri * cosI - cosT)));
                if (wo.z > 0) wl = vneg(wl);
                pr = 0.0f;
                float den = ri * vdot(wl, wm) + vdot(wo, wm);
                pt = (D * vdot(wl, wm)) / powf(den, 2);
            }
        }
        *pdf = F * pr + (1 - F) * pt;
        float G = g_ggx(wl, wo, alpha);
        if (reflect) {
            *pdf = D * wm.z / (4 * vdot(wo, wm));
            float brdf = F * D * G / (4.0f * wo.z * wl.z);
            *f = cdivf(cmulf(cmulf(mat_albedo(s, m, sd->tu, sd->tv), brdf), fabsf(wl.z)), *pdf);
        } else {
            float den = ri * vdot(wl, wm) + vdot(wo, wm);
            float bsdf = ((1 - F) * D * G * vdot(wl, wm) * vdot(wo, wm)) / (powf(den, 2) * wl.z * wo.z);
            *f = cdivf(cmulf(cmulf(mat_albedo(s, m, sd->tu, sd->tv), bsdf), fabsf(wl.z)), *pdf);
        }
        return frame_to_world(&sd->frame, wl);
    }
    case M_PLASTIC: { /* Materials.h:861-930 */
        V3 wo = frame_to_local(&sd->frame, sd->wo);
        float cosTheta = wo.z;
        if (cosTheta <= 0.0f) { *pdf = 1.0f; *f = col(0, 0, 0); return v3(0, 0, 0); }
        float iorT, iorI; set_ior(cosTheta, &iorI, &iorT, m->int_ior, m->ext_ior);
        float ks = fresnel_dielectric(cosTheta, iorI, iorT), kd = 1.0f - ks;
        V3 wl;
        float e = plastic_exponent(mat_alpha(m));
        float s1 = rng_next(rng), s2 = rng_next(rng);
        V3 wr = v3(-wo.x, -wo.y, wo.z);
        if (rng_next(rng) > ks) {
            wl = cosine_sample_hemisphere(s1, s2);
        } else {
            float ct = powf(s1, 1.0f / (e + 1.f));
            float st = sqrtf(smax(0.0f, 1.0f - powf(ct, 2)));
            float phi = 2 * M_PI * s2;
            V3 lobe = v3(st * cosf(phi), st * sinf(phi), ct);
            Frame fr = frame_from_vector(wr);
            wl = frame_to_local(&fr, lobe); /* sic: toLocal, Materials.h:914 */
        }
        float cosAlpha = smax(0.0f, vdot(wr, wl));
        *pdf = (kd * (wl.z / M_PI)) + (ks * (e + 1) * powf(cosAlpha, e) / (2 * M_PI));
        Col diffuse = cdivf(cmulf(mat_albedo(s, m, sd->tu, sd->tv), kd), (float)M_PI);
        float gloss = ks * ((e + 2) / (2 * M_PI)) * powf(cosAlpha, e);
        *f = cadd(diffuse, col(gloss, gloss, gloss));
        return frame_to_world(&sd->frame, wl);
    }
    }
    *pdf = 0; *f = col(0, 0, 0);
    return v3(0, 0, 0);
}

/* ---------------------------------------------------------------- lights (ref: Lights.h) */
static Col bg_evaluate(const orc_scene* s, V3 wi) {
    if (s->bg.kind == 0) return col(s->bg.colour[0], s->bg.colour[1], s->bg.colour[2]); /* Lights.h:102 */
    /* EnvironmentMap::evaluate, Lights.h:162-169 */
    float u = atan2f(wi.z, wi.x);
    u = (u < 0.0f) ? u + (2.0f * M_PI) : u;
    u = u / (2.0f * M_PI);
    float v = acosf(wi.y) / M_PI;
    return tex_sample(&s->tex[s->bg.env_tex], u, v);
}
static float bg_power(const orc_scene* s) {
    if (s->bg.kind == 0) return clum(col(s->bg.colour[0], s->bg.colour[1], s->bg.colour[2])) * 4.0f * M_PI; /* Lights.h:118 */
    const Texture* e = &s->tex[s->bg.env_tex]; /* Lights.h:184-197 */
    float total = 0;
    for (int i = 0; i < e->height; i++) {
        float st = sinf(((float)i / (float)e->height) * M_PI);
        for (int n = 0; n < e->width; n++) total += (clum(e->texels[(i * e->width) + n]) * st);
    }
    total = total / (float)(e->width * e->height);
    return total * 4.0f * M_PI;
}

/* ---- EnvironmentMap's tabulated importance sampling (Lights.h:207-253, :306-325, :347-424), as written upstream.
 * build() leaves per-row-normalised PDFS in `conditional_cdf` and row pdfs in `marginal_cdf`; Sample1D walks a running sum
 * over them.  Everything the tables produce then passes through Vec2::operator[], which returns int (Sampling.h:204,
 * pinned by tests/golden/scalars_r2.npz: 0 for every coordinate in [0, 1)), so sample() / sampleDirectionFromLight() / PDF()
 * end up at theta = 0: wi = (0, 1, 0) and pdf = (table pdf) * 2 pi^2 / sin(0) = +inf.  The steps are kept so that the cases
 * that do not collapse (a coordinate of exactly 1, a walk that falls off the end of the table) come out as upstream's. */
static void env_build(orc_scene* s) {
    const Texture* e = &s->tex[s->bg.env_tex];
    int nu = e->width, nv = e->height;
    s->env_nu = nu; s->env_nv = nv;
    float* func = (float*)malloc(sizeof(float) * (size_t)nu * nv);
    s->env_cond = (float*)malloc(sizeof(float) * (size_t)nu * nv);
    s->env_marg = (float*)malloc(sizeof(float) * (size_t)nv);
    float funcInt = 0.0f;
    for (int v = 0; v < nv; ++v) {
        float theta = (v + 0.5f) / nv * M_PI;
        float sinTheta = sinf(theta);
        for (int u = 0; u < nu; ++u) {
            int idx = v * nu + u;
            float lum = clum(e->texels[idx]);
            func[idx] = lum * sinTheta;
            funcInt += func[idx];
        }
    }
    for (int v = 0; v < nv; ++v) {
        float* row = s->env_cond + (size_t)v * nu;
        float sum = 0.0f;
        for (int u = 0; u < nu; ++u) { row[u] = func[v * nu + u]; sum += row[u]; }
        if (sum > 0.0f) for (int u = 0; u < nu; ++u) row[u] /= sum;
    }
    for (int v = 0; v < nv; ++v) {
        float sum = 0.0f;
        for (int u = 0; u < nu; ++u) sum += func[v * nu + u];
        s->env_marg[v] = sum;
    }
    if (funcInt > 0.0f) for (int v = 0; v < nv; ++v) s->env_marg[v] /= funcInt;
    s->env_func_int = funcInt;
    free(func);
}
static float env_sample1d(const float* dist, int n, float u, float* pdf, int* offset) { /* Lights.h:306-325 */
    float cdf = 0.0f;
    int index = 0;
    for (int i = 0; i < n; i++) {
        cdf += dist[i];
        if (u <= cdf) { index = i; break; }
    }
    index = index < n - 1 ? index : n - 1;
    if (offset) *offset = index;
    if (pdf) *pdf = dist[index] / (float)n;   /* dist[index] / dist.size(): size_t -> float */
    float t = (u - (cdf - dist[index])) / dist[index];
    return (index + t) / (float)n;
}
static void env_sample_continuous(const orc_scene* s, float u0, float u1, float* pdf, float* d0, float* d1) { /* Lights.h:347-356 */
    float pdfs[2];
    int v;
    *d1 = env_sample1d(s->env_marg, s->env_nv, u1, &pdfs[1], &v);
    *d0 = env_sample1d(s->env_cond + (size_t)v * s->env_nu, s->env_nu, u0, &pdfs[0], NULL);
    *pdf = pdfs[0] * pdfs[1];
}
/* int Vec2::operator[] (Sampling.h:204): float -> int conversion; NaN (a zero-pdf entry picked by u == 0) is undefined upstream, 0 here */
static int vec2_index(float x) { return (x != x) ? 0 : (int)x; }
static V3 env_uv_to_direction(float u, float v) { /* Lights.h:366-374 */
    float theta = v * M_PI;
    float phi = u * 2.0f * M_PI;
    float sinTheta = sinf(theta), cosTheta = cosf(theta), sinPhi = sinf(phi), cosPhi = cosf(phi);
    return v3(sinTheta * cosPhi, cosTheta, sinTheta * sinPhi);
}
/* EnvironmentMap::sample / sampleDirectionFromLight after their two draws (Lights.h:385-399, :411-424) */
static V3 env_sample_dir(const orc_scene* s, float u0, float u1, float* pdf) {
    float d0, d1;
    env_sample_continuous(s, u0, u1, pdf, &d0, &d1);
    int iu = vec2_index(d0), iv = vec2_index(d1);            /* uv[0], uv[1] */
    V3 wi = env_uv_to_direction((float)iu, (float)iv);
    float theta = iv * M_PI;
    float sinTheta = sinf(theta);
    *pdf *= (2.0f * M_PI * M_PI) / sinTheta;
    return wi;
}
static float env_pdf(const orc_scene* s, V3 wi) { /* EnvironmentMap::PDF, PdfUV, DirectionToUV (Lights.h:359-363, :377-383, :402-409) */
    float u = atan2f(wi.z, wi.x);
    u = (u < 0.0f) ? u + (2.0f * M_PI) : u;
    u = u / (2.0f * M_PI);
    float v = acosf(wi.y) / M_PI;
    int p0 = vec2_index(u), p1 = vec2_index(v);
    int iu = p0 * s->env_nu, iv = p1 * s->env_nv;             /* static_cast<int>(p[0] * nu): int * int */
    iu = iu < 0 ? 0 : (iu > s->env_nu - 1 ? s->env_nu - 1 : iu);
    iv = iv < 0 ? 0 : (iv > s->env_nv - 1 ? s->env_nv - 1 : iv);
    float pdfUV = (s->env_cond[(size_t)iv * s->env_nu + iu] * s->env_nu * s->env_nv) / s->env_func_int;
    float theta = p1 * M_PI;
    float sinTheta = sinf(theta);
    return pdfUV * (2.0f * M_PI * M_PI) / sinTheta;
}

/* Scene::sampleLight (Scene.h:144) */
static const LightRef* sample_light(const orc_scene* s, Rng* rng, float* pmf) {
    *pmf = 1.f / s->nlights;
    int i = (int)(s->nlights * rng_next(rng));
    int last = (int)s->nlights - 1;
    return &s->lights[i < last ? i : last];
}
static V3 light_normal(const orc_scene* s, const LightRef* l, V3 wi) {
    if (l->kind == 0) return tri_gnormal(&s->tris[l->tri]); /* Lights.h:58 */
    return vneg(wi);                                        /* Lights.h:114, :180 */
}
static Col light_evaluate(const orc_scene* s, const LightRef* l, V3 wi) {
    if (l->kind == 0) { /* Lights.h:42 */
        const Tri* t = &s->tris[l->tri];
        if (vdot(wi, tri_gnormal(t)) < 0) return mat_emission(&s->mats[t->mat]);
        return col(0, 0, 0);
    }
    return bg_evaluate(s, wi);
}
/* Light::sample (Lights.h:37, :95, :385) */
static V3 light_sample(const orc_scene* s, const LightRef* l, Rng* rng, Col* emitted, float* pdf) {
    if (l->kind == 0) {
        const Tri* t = &s->tris[l->tri];
        *emitted = mat_emission(&s->mats[t->mat]);
        float r1 = rng_next(rng), r2 = rng_next(rng);
        return tri_sample(t, r1, r2, pdf);
    }
    float r1 = rng_next(rng), r2 = rng_next(rng);
    if (l->kind == 1) {
        V3 wi = uniform_sample_sphere(r1, r2);
        *pdf = uniform_sphere_pdf();
        *emitted = col(s->bg.colour[0], s->bg.colour[1], s->bg.colour[2]);
        return wi;
    }
    /* EnvironmentMap::sample (Lights.h:385-399) */
    V3 wi = env_sample_dir(s, r1, r2, pdf);
    *emitted = bg_evaluate(s, wi);
    return wi;
}
static V3 light_sample_position(const orc_scene* s, const LightRef* l, Rng* rng, float* pdf) {
    float r1 = rng_next(rng), r2 = rng_next(rng);
    if (l->kind == 0) return tri_sample(&s->tris[l->tri], r1, r2, pdf); /* Lights.h:66 */
    V3 p = uniform_sample_sphere(r1, r2);                                /* Lights.h:122, :327 */
    p = vmulf(p, s->scene_radius);
    p = vadd(p, s->scene_centre);
    if (l->kind == 1) *pdf = 4 * M_PI * s->scene_radius * s->scene_radius; /* sic: area, not 1/area */
    else *pdf = 1.0f / (4 * M_PI * (s->scene_radius * s->scene_radius));
    return p;
}
static V3 light_sample_direction(const orc_scene* s, const LightRef* l, Rng* rng, float* pdf) {
    float r1 = rng_next(rng), r2 = rng_next(rng);
    if (l->kind == 0) { /* Lights.h:70-84 */
        V3 wi = cosine_sample_hemisphere(r1, r2);
        *pdf = cosine_hemisphere_pdf(wi);
        Frame fr = frame_from_vector(tri_gnormal(&s->tris[l->tri]));
        return frame_to_world(&fr, wi);
    }
    if (l->kind == 1) { V3 wi = uniform_sample_sphere(r1, r2); *pdf = uniform_sphere_pdf(); return wi; } /* Lights.h:130 */
    return env_sample_dir(s, r1, r2, pdf); /* Lights.h:411-424 */
}

/* ---------------------------------------------------------------- camera (ref: Scene.h:14-79) */
void orc_scene_set_camera(orc_scene* s, const orc_camera* c) {
    memcpy(s->P, c->P, 64);
    mat_invert(c->P, s->invP);       /* Camera::init, Scene.h:26 */
    s->width = (float)c->width; s->height = (float)c->height;
    float Wlens = (2.0f / c->P[5]);
    float aspect = c->P[0] / c->P[5];
    float Hlens = Wlens * aspect;
    s->Afilm = Wlens * Hlens;
    memcpy(s->cam, c->V, 64);        /* Camera::updateView, Scene.h:37 */
    mat_invert(c->V, s->cam2view);
    s->origin = mat_mul_point(s->cam, v3(0, 0, 0));
    s->view_dir = mat_mul_point_div(s->invP, v3(0, 0, 1));
    s->view_dir = mat_mul_vec(s->cam, s->view_dir);
    s->view_dir = vnorm(s->view_dir);
}
static void camera_ray(const orc_scene* s, float x, float y, V3* o, V3* dir) { /* Scene.h:47-64 */
    float xc = ((2 * x) / (s->width - 1)) - 1;
    float yc = 1 - ((2 * y) / (s->height - 1));
    V3 pcam = mat_mul_point_div(s->invP, v3(xc, yc, 0));
    V3 dcam = mat_mul_point_div(s->cam, pcam);
    *dir = vsub(dcam, s->origin);
    *o = s->origin;
}
static int camera_project(const orc_scene* s, V3 p, float* x, float* y) { /* Scene.h:65-79 */
    V3 pview = mat_mul_point(s->cam2view, p);
    V3 pproj = mat_mul_point_div(s->P, pview);
    *x = (pproj.x + 1.0f) * 0.5f;
    *y = (pproj.y + 1.0f) * 0.5f;
    if (*x < 0 || *x > 1.0f || *y < 0 || *y > 1.0f) return 0;
    *x = *x * s->width;
    *y = 1.0f - *y;
    *y = *y * s->height;
    return 1;
}

/* ---------------------------------------------------------------- shading data (ref: Scene.h:185-215) */
static Shading shading_data(const orc_scene* s, const orc_hit* h, V3 o, V3 dir) {
    Shading sd; memset(&sd, 0, sizeof(sd));
    sd.mat = -1;
    if (h->t < FLT_MAX) {
        const Tri* t = &s->tris[h->ID];
        sd.x = vadd(o, vmulf(dir, h->t));
        sd.gN = tri_gnormal(t);
        V3 n = vadd(vadd(vmulf(t->nrm[0], h->alpha), vmulf(t->nrm[1], h->beta)), vmulf(t->nrm[2], h->gamma)); /* Geometry.h:154 */
        sd.sN = vnorm(n);
        sd.tu = t->u[0] * h->alpha + t->u[1] * h->beta + t->u[2] * h->gamma;
        sd.tv = t->v[0] * h->alpha + t->v[1] * h->beta + t->v[2] * h->gamma;
        sd.mat = (int)t->mat;
        sd.wo = vneg(dir);
        if (mat_two_sided(&s->mats[t->mat])) {
            if (vdot(sd.wo, sd.sN) < 0) sd.sN = vneg(sd.sN);
            if (vdot(sd.wo, sd.gN) < 0) sd.gN = vneg(sd.gN);
        }
        sd.frame = frame_from_vector(sd.sN);
        sd.t = h->t;
    } else {
        sd.wo = vneg(dir);
        sd.t = h->t;
    }
    return sd;
}

/* ---------------------------------------------------------------- integrators (ref: Renderer.h) */
typedef struct {
    const orc_scene* s;
    Cnt cnt;
    uint64_t closest, any;
} Ctx;

static orc_hit ctx_traverse(Ctx* c, V3 o, V3 d) { c->closest++; return scene_traverse(c->s, o, d, &c->cnt); }
static int ctx_visible(Ctx* c, V3 a, V3 b) { c->any++; return scene_visible(c->s, a, b, &c->cnt); }
static inline float wmin(float a, float b) { return (a < b) ? a : b; } /* Windows min() macro */
static inline float wmax(float a, float b) { return (a > b) ? a : b; } /* Windows max() macro */

/* Renderer.h:155-201 */
static Col compute_direct(Ctx* c, const Shading* sd, Rng* rng) {
    const orc_scene* s = c->s;
    const orc_material* m = &s->mats[sd->mat];
    if (mat_is_specular(m)) return col(0, 0, 0);
    if (s->nlights == 0) return col(0, 0, 0); /* upstream indexes lights[-1] (Scene.h:147): defined here as "no direct light" */
    float pmf, pdf; Col emitted;
    const LightRef* light = sample_light(s, rng, &pmf);
    V3 p = light_sample(s, light, rng, &emitted, &pdf);
    if (light->kind == 0) {
        V3 wi = vsub(p, sd->x);
        float l = vlen2(wi);
        wi = vnorm(wi);
        float G = (wmax(vdot(wi, sd->sN), 0.0f) * wmax(-vdot(wi, light_normal(s, light, wi)), 0.0f)) / l;
        if (G > 0) {
            if (ctx_visible(c, sd->x, p))
                return cdivf(cmulf(cmul(bsdf_eval(s, sd, wi), emitted), G), (pmf * pdf));
        }
    } else {
        V3 wi = p;
        float G = wmax(vdot(wi, sd->sN), 0.0f);
        if (G > 0) {
            if (ctx_visible(c, sd->x, vadd(sd->x, vmulf(p, 10000.0f))))
                return cdivf(cmulf(cmul(bsdf_eval(s, sd, wi), emitted), G), (pmf * pdf));
        }
    }
    return col(0, 0, 0);
}

/* Renderer.h:870-907 */
static Col vpl_contribution(Ctx* c, const Shading* sd) {
    const orc_scene* s = c->s;
    Col indirect = col(0, 0, 0);
    for (int i = 0; i < s->nvpls; i++) {
        const VPL* vpl = &s->vpls[i];
        V3 wi = vsub(vpl->sd.x, sd->x);
        float distance = vlen(wi);
        wi = vnorm(wi);
        if (!ctx_visible(c, sd->x, vpl->sd.x)) continue;
        float cx = vdot(sd->sN, wi), cxi = vdot(vpl->sd.sN, vneg(wi));
        if (cx <= 0.0f || cxi <= 0.0f) continue;
        float G = cx * cxi / (distance * distance);
        if (G <= 0.0f) continue;
        Col fx = bsdf_eval(s, sd, wi);
        if (vpl->sd.mat < 0 || mat_is_light(&s->mats[vpl->sd.mat])) indirect = cadd(indirect, cmul(cmulf(fx, G), vpl->Le));
        else {
            Col fxi = bsdf_eval(s, &vpl->sd, vneg(wi));
            indirect = cadd(indirect, cmul(cmul(cmulf(fx, G), fxi), vpl->Le));
        }
    }
    return indirect;
}

/* Renderer.h:215-255 (radiosity == 0) and :910-962 (radiosity == 1), iterative
 * form of the tail recursion; the per-vertex terms are summed right-to-left as
 * the recursion returns them. */
/* debugging aid of the tests: when set, path_trace logs 20 floats per path vertex (orc_trace_path) */
static __thread float* g_trace = NULL;
static __thread int g_trace_n = 0, g_trace_cap = 0;
static void trace_vertex(const Shading* sd, Col T_in, Col direct, Col f, float pdf, V3 wi) {
    if (!g_trace || g_trace_n >= g_trace_cap) return;
    float* r = g_trace + 20 * g_trace_n++;
    r[0] = (float)sd->mat; r[1] = T_in.r; r[2] = T_in.g; r[3] = T_in.b; r[4] = direct.r; r[5] = direct.g; r[6] = direct.b;
    r[7] = f.r; r[8] = f.g; r[9] = f.b; r[10] = pdf; r[11] = wi.x; r[12] = wi.y; r[13] = wi.z;
    r[14] = sd->wo.x; r[15] = sd->wo.y; r[16] = sd->wo.z; r[17] = sd->sN.x; r[18] = sd->sN.y; r[19] = sd->sN.z;
}
static Col path_trace(Ctx* c, V3 o, V3 dir, Rng* rng, int radiosity) {
    const orc_scene* s = c->s;
    Col T = col(1, 1, 1);
    Col terms[MAX_DEPTH + 4]; int nterms = 0;
    int canHitLight = 1;
    for (int depth = 0;; depth++) {
        orc_hit h = ctx_traverse(c, o, dir);
        Shading sd = shading_data(s, &h, o, dir);
        if (!(sd.t < FLT_MAX)) { terms[nterms++] = bg_evaluate(s, dir); break; } /* Renderer.h:254: not scaled by throughput */
        const orc_material* m = &s->mats[sd.mat];
        if (mat_is_light(m)) {
            terms[nterms++] = canHitLight ? cmul(T, mat_emission(m)) : col(0, 0, 0);
            break;
        }
        Col direct = cmul(T, compute_direct(c, &sd, rng));
        if (radiosity && !mat_is_specular(m)) direct = cadd(direct, cmul(T, vpl_contribution(c, &sd)));
        terms[nterms++] = direct;
        if (depth > MAX_DEPTH) break;
        float rr = wmin(clum(T), 0.9f);
        if (rng_next(rng) < rr) T = cdivf(T, rr);
        else break;
        Col f; float pdf;
        V3 wi = bsdf_sample(s, &sd, rng, &f, &pdf);
        trace_vertex(&sd, T, direct, f, pdf, wi);
        T = cdivf(cmulf(cmul(T, f), fabsf(vdot(wi, sd.sN))), pdf);
        o = vadd(sd.x, vmulf(wi, EPSILON));
        dir = wi;
        canHitLight = mat_is_specular(m);
    }
    Col L = terms[nterms - 1];
    for (int i = nterms - 2; i >= 0; i--) L = cadd(terms[i], L);
    return L;
}

/* ImageFilter::filter / size (Imaging.h:138-214).  The C++ promotions are kept: pow(int, int) and pow(float, int) are
 * double, the products with the float constants are double, the functions return float. */
static float filter_gauss1(int d) { /* GaussianFilter::gauss(int d, int radius): the float offsets arrive truncated to int */
    float Nalpha = -1.f, radius = 1.f;                                     /* Imaging.h:155-156 (globals) */
    float secondPart = (float)exp((double)Nalpha * pow((double)radius, 2)); /* Imaging.h:158 */
    return (float)(exp((double)Nalpha * pow((double)d, 2)) - (double)secondPart);
}
static float filter_mn1(float y) { /* MitchellNetravaliFilter::MN, Imaging.h:193-205 */
    float B = 1.f / 3.f, C = 1.f / 3.f;
    float a1 = 12 - (9 * B) - (6 * C), b1 = -18 + (12 * B) + (6 * C), c1 = 6 - (2 * B);
    float a2 = (-B) - (6 * C), b2 = (6 * B) + (30 * C), c2 = -(12 * B) - (48 * C), d2 = (8 * B) - (24 * C);
    float x = fabsf(y);
    if (fabsf(x) <= 0.0f && fabs(x) < 1.0f) return (float)((((double)a1 * pow((double)x, 3)) + ((double)b1 * pow((double)x, 2)) + (double)(c1)) / 6);
    else if (fabsf(x) <= 1.0f && fabs(x) < 2.0f) return (float)((((double)a2 * pow((double)x, 3)) + ((double)b2 * pow((double)x, 2)) + (double)(c2 * x) + (double)d2) / 6);
    else if (fabsf(x) >= 2) return 0;
    return 0; /* 1 < |x| < 2: upstream falls off the end of the function (undefined value); defined here as 0 */
}
static float filter_eval(int kind, float x, float y) {
    if (kind == 1) return filter_gauss1((int)x) * filter_gauss1((int)y);
    if (kind == 2) return filter_mn1(x) * filter_mn1(y);
    return (fabsf(x) <= 1.0f && fabs(y) <= 1.0f) ? 1.0f : 0.0f; /* BoxFilter */
}
static inline int filter_size(int kind) { return kind == 1 ? 1 : kind == 2 ? 2 : 0; }
static void film_atomic_add(float* p, float v) {
    _Atomic uint32_t* a = (_Atomic uint32_t*)p;
    uint32_t old = atomic_load_explicit(a, memory_order_relaxed), nw;
    do { float fo; memcpy(&fo, &old, 4); float fn = fo + v; memcpy(&nw, &fn, 4); }
    while (!atomic_compare_exchange_weak_explicit(a, &old, nw, memory_order_relaxed, memory_order_relaxed));
}
/* Film::splat, Imaging.h:224-247.  With the size-0 BoxFilter this is film[(int)y * W + (int)x] += L * 1 / 1. */
static void film_add(float* film, int W, int H, float x, float y, Col L, int atomic, int filter) {
    float w[25]; int idx[25]; int used = 0; float total = 0;
    int size = filter_size(filter);
    for (int i = -size; i <= size; i++)
        for (int j = -size; j <= size; j++) {
            int px = (int)x + j, py = (int)y + i;
            if (px >= 0 && px < W && py >= 0 && py < H) {
                idx[used] = (py * W) + px;
                w[used] = filter ? filter_eval(filter, px - x, py - y) : 1.0f;
                total += w[used];
                used++;
            }
        }
    if (filter) atomic = 1; /* footprints of neighbouring pixels overlap across render threads */
    for (int i = 0; i < used; i++) {
        float* p = film + (size_t)idx[i] * 3;
        float v[3] = { filter ? L.r * w[i] / total : L.r, filter ? L.g * w[i] / total : L.g, filter ? L.b * w[i] / total : L.b };
        for (int k = 0; k < 3; k++) {
            if (!atomic) p[k] = p[k] + v[k];
            else film_atomic_add(&p[k], v[k]);
        }
    }
}

/* Renderer.h:669-692 */
static void connect_to_camera(Ctx* c, V3 p, Col colr, float* film) {
    const orc_scene* s = c->s;
    float x, y;
    if (!camera_project(s, p, &x, &y) || !ctx_visible(c, p, s->origin)) return;
    V3 cd = vnorm(vsub(p, s->origin));
    float ct = vdot(s->view_dir, cd);
    if (ct <= 0) return;
    float we = 1.f / (s->Afilm * pow((double)ct, 4.0));
    film_add(film, (int)s->width, (int)s->height, x, y, cmulf(colr, we), 1, s->filter);
}
/* Renderer.h:695-767 */
static void light_trace(Ctx* c, Rng* rng, float* film) {
    const orc_scene* s = c->s;
    if (s->nlights == 0) return;
    float pmf, pdfPos, pdfDir;
    const LightRef* light = sample_light(s, rng, &pmf);
    V3 p = light_sample_position(s, light, rng, &pdfPos);
    V3 wi = light_sample_direction(s, light, rng, &pdfDir);
    Col Le = cdivf(light_evaluate(s, light, vneg(wi)), pdfPos);
    connect_to_camera(c, p, Le, film);
    V3 o = p, dir = wi;
    Col T = col(1, 1, 1);
    for (int depth = 0;; depth++) {
        orc_hit h = ctx_traverse(c, o, dir);
        Shading sd = shading_data(s, &h, o, dir);
        if (!(sd.t < FLT_MAX)) return;
        const orc_material* m = &s->mats[sd.mat];
        V3 wc = vnorm(vsub(s->origin, sd.x));
        Col colr = cmul(cmul(T, bsdf_eval(s, &sd, wc)), Le);
        connect_to_camera(c, sd.x, colr, film);
        if (mat_is_light(m)) return;
        if (depth > MAX_DEPTH) return;
        float rr = wmin(clum(T), 0.9f);
        if (rng_next(rng) < rr) T = cdivf(T, rr);
        else return;
        float r1 = rng_next(rng), r2 = rng_next(rng);
        V3 nw = cosine_sample_hemisphere(r1, r2);
        float pdf = cosine_hemisphere_pdf(wc); /* sic: world-space direction to the camera, Renderer.h:757 */
        nw = frame_to_world(&sd.frame, nw);
        Col f = bsdf_eval(s, &sd, nw);
        T = cdivf(cmulf(cmul(T, f), fabsf(vdot(nw, sd.sN))), pdf);
        o = vadd(sd.x, vmulf(nw, EPSILON));
        dir = nw;
    }
}

/* Renderer.h:772-863 */
static void vpl_push(orc_scene* s, const Shading* sd, Col Le) {
    if (s->nvpls == s->cap_vpls) {
        s->cap_vpls = s->cap_vpls ? s->cap_vpls * 2 : 256;
        s->vpls = (VPL*)realloc(s->vpls, sizeof(VPL) * (size_t)s->cap_vpls);
    }
    s->vpls[s->nvpls].sd = *sd;
    s->vpls[s->nvpls].Le = Le;
    s->nvpls++;
}
int orc_vpl_trace(orc_scene* s, int N, uint64_t seed) {
    s->nvpls = 0;
    if (s->nlights == 0) return 0;
    Ctx c; memset(&c, 0, sizeof(c)); c.s = s;
    for (int i = 0; i < N; i++) {
        Rng rng = rng_make(seed, 0, (uint32_t)i, 3);
        float pmf, pdfPos, pdfDir, pdf; Col emitted;
        const LightRef* light = sample_light(s, &rng, &pmf);
        V3 p = light_sample_position(s, light, &rng, &pdfPos);
        V3 wi = light_sample_direction(s, light, &rng, &pdfDir);
        V3 ln = light_normal(s, light, p);
        light_sample(s, light, &rng, &emitted, &pdf);
        Col Le = cdivf(cmulf(emitted, fabsf(vdot(wi, ln))), (pmf * pdfPos * pdfDir * (float)N));
        Shading v0; memset(&v0, 0, sizeof(v0)); v0.x = p; v0.gN = ln; v0.sN = ln; v0.mat = -1; /* ShadingData(p, n), Materials.h:27 */
        vpl_push(s, &v0, Le);
        V3 o = vadd(p, vmulf(wi, EPSILON)), dir = wi;
        Col T = col(1, 1, 1);
        for (int depth = 0; depth <= MAX_DEPTH; depth++) { /* Renderer.h:813 */
            orc_hit h = ctx_traverse(&c, o, dir);
            Shading sd = shading_data(s, &h, o, dir);
            if (!(sd.t < FLT_MAX)) break;
            float rr = wmin(clum(T), 0.9f);
            if (rng_next(&rng) < rr) T = cdivf(T, rr);
            else break;
            vpl_push(s, &sd, cmul(T, Le));
            Col f; float bp;
            V3 w2 = bsdf_sample(s, &sd, &rng, &f, &bp);
            T = cdivf(cmulf(cmul(T, f), fabsf(vdot(w2, sd.sN))), bp);
            o = vadd(sd.x, vmulf(w2, EPSILON));
            dir = w2;
        }
    }
    return s->nvpls;
}
int orc_vpl_count(orc_scene* s) { return s->nvpls; }
void orc_set_filter(orc_scene* s, int kind) { s->filter = (kind == 1 || kind == 2) ? kind : 0; }
float orc_filter_eval(int kind, float x, float y) { return filter_eval(kind, x, y); }
void orc_vpl_get(orc_scene* s, float* out) {
    for (int i = 0; i < s->nvpls; i++) {
        const VPL* v = &s->vpls[i];
        float* o = out + (size_t)i * 10;
        o[0] = v->sd.x.x; o[1] = v->sd.x.y; o[2] = v->sd.x.z;
        o[3] = v->sd.sN.x; o[4] = v->sd.sN.y; o[5] = v->sd.sN.z;
        o[6] = v->Le.r; o[7] = v->Le.g; o[8] = v->Le.b; o[9] = (float)v->sd.mat;
    }
}

/* AOV passes, Renderer.h:257-296 */
static Col aov(Ctx* c, int mode, V3 o, V3 dir, Rng* rng) {
    const orc_scene* s = c->s;
    orc_hit h = ctx_traverse(c, o, dir);
    if (mode == 11) { /* viewNormals */
        if (h.t < FLT_MAX) { Shading sd = shading_data(s, &h, o, dir); return col(fabsf(sd.sN.x), fabsf(sd.sN.y), fabsf(sd.sN.z)); }
        return col(0, 0, 0);
    }
    Shading sd = shading_data(s, &h, o, dir);
    if (sd.t < FLT_MAX) {
        if (mat_is_light(&s->mats[sd.mat])) return mat_emission(&s->mats[sd.mat]);
        if (mode == 10) return bsdf_eval(s, &sd, v3(0, 1, 0)); /* albedo */
        return compute_direct(c, &sd, rng);                     /* direct */
    }
    return mode == 10 ? bg_evaluate(s, dir) : col(0, 0, 0);
}

/* ---------------------------------------------------------------- PSSMLT (ref: Renderer.h:964-1063, :353-370; Sampling.h:43-139)
 * What the upstream sampler does, read off its code: PSS::next (Sampling.h:60-67) hands out coordinate
 * currentIndex++ and currentIndex is only ever reset by PSS::reset (largeStep, :118-122).  A small step perturbs the
 * coordinates already handed out (smallStepPerturb, :85-94) and the next nextPSSP() indexes past them, pushing a
 * fresh uniform: every proposal's (x, y) is a fresh uniform draw whether the step was "small" or "large".
 * pathTrace(x, y, sampler) takes all its numbers from sampler->next() == PSSSampler::random (:132-134), not from
 * the primary sample space.  So proposals are independent of the chain state; the mutated coordinates are never
 * read and are not restated.  What remains is restated in upstream's order: b from M = 100 paths (:995-1006), an
 * initial state (:1022), 100 000 proposals per renderSTMLT call (:357) with the large/small splat weights
 * (:1040-1047) and the acceptance draw (:1053).
 * Shared number spec (GPU and oracle): path i of sample k draws from stream 2 with pixel = i (i < 100: the b
 * estimate, i = 100: initial state, i = 101 + j: proposal j); the chain's two draws per proposal come from stream 4. */
#define MLT_M 100
#define MLT_PROPOSALS 100000
static float mlt_acceptance(Col current, Col proposed) { /* Renderer.h:965-973 */
    float cl = clum(current), pl = clum(proposed);
    if (cl <= 0.0f) return 1.0f;
    if (pl <= 0.0f) return 0.0f;
    float r = pl / cl;
    return (r < 1.0f) ? r : 1.0f; /* (std::min)(1.0f, r) */
}
/* the two primary-sample-space coordinates of path `index`: nextPSSC() / nextPSSP() always return a fresh push (pinned by
 * tests/golden/scalars_r3.npz from the compiled Sampling.h), here the first two numbers of the path's own generator */
static Rng mlt_coords(uint64_t seed, uint64_t sample, uint32_t index, float* u, float* v) {
    Rng rng = rng_make(seed, sample, index, 2);
    *u = rng_next(&rng);
    *v = rng_next(&rng);
    return rng;
}
/* test probe: what the sampler hands the chain for the first n proposals, without tracing anything: per proposal the step
 * draw, the two coordinates and the accept draw, in the order orc_mlt_render consumes them */
void orc_mlt_trace(uint64_t seed, uint64_t sample, int n, float* out4) {
    Rng chain = rng_make(seed, sample, 0, 4);
    for (int i = 0; i < n; i++) {
        out4[i * 4 + 0] = rng_next(&chain);
        (void)mlt_coords(seed, sample, (uint32_t)(MLT_M + 1 + i), &out4[i * 4 + 1], &out4[i * 4 + 2]);
        out4[i * 4 + 3] = rng_next(&chain);
    }
}
static Col mlt_path(Ctx* c, uint64_t seed, uint64_t sample, uint32_t index, float* x, float* y) { /* genPathC / genPathP, Renderer.h:978-992 */
    const orc_scene* s = c->s;
    float pu, pv;
    Rng rng = mlt_coords(seed, sample, index, &pu, &pv);
    *x = pu * s->width;
    *y = pv * s->height;
    float u1 = rng_next(&rng), u2 = rng_next(&rng);
    V3 o, d;
    camera_ray(s, *x + u1, *y + u2, &o, &d); /* Renderer.h:205 */
    return path_trace(c, o, d, &rng, 0);
}
void orc_mlt_render(orc_scene* s, int spp, uint64_t seed, uint64_t first, float* film) {
    int W = (int)s->width, H = (int)s->height;
    Ctx c; memset(&c, 0, sizeof(c)); c.s = s;
    const float largeStepProb = 0.3f; /* Renderer.h:1010 */
    for (int k = 0; k < spp; k++) {
        uint64_t sample = first + (uint64_t)k;
        float b = 0.0f, x, y;
        for (int i = 0; i < MLT_M; i++) b += clum(mlt_path(&c, seed, sample, (uint32_t)i, &x, &y));
        b /= MLT_M;
        if (b <= 0.0f) b = 1.0f;
        float cx, cy, px, py;
        Col current = mlt_path(&c, seed, sample, MLT_M, &cx, &cy);
        Rng chain = rng_make(seed, sample, 0, 4);
        for (int i = 0; i < MLT_PROPOSALS; i++) {
            int largeStep = rng_next(&chain) < largeStepProb;
            Col proposed = mlt_path(&c, seed, sample, (uint32_t)(MLT_M + 1 + i), &px, &py);
            float a = mlt_acceptance(current, proposed);
            if (largeStep) film_add(film, W, H, px, py, cdivf(cmulf(proposed, a), (b + largeStepProb * clum(proposed))), 0, s->filter);
            else film_add(film, W, H, px, py, cdivf(cmulf(proposed, a), b), 0, s->filter);
            film_add(film, W, H, cx, cy, cdivf(cmulf(current, (1.0f - a)), b), 0, s->filter);
            if (rng_next(&chain) < a) { current = proposed; cx = px; cy = py; }
        }
    }
}

/* ---------------------------------------------------------------- render driver (ref: Renderer.h:397-428, :516-545, :327-351) */
typedef struct {
    orc_scene* s; int mode, spp; uint64_t seed, first; float* film;
    atomic_int* next_tile; int tiles_x, tiles_y;
    Ctx ctx;
} Job;
#define TILE 16 /* Renderer.h:16 */

static void render_pixel(Job* j, int x, int y) {
    orc_scene* s = j->s;
    int W = (int)s->width, H = (int)s->height;
    uint32_t pix = (uint32_t)(y * W + x);
    float px = x + 0.5f, py = y + 0.5f; /* Renderer.h:527-528 */
    for (int k = 0; k < j->spp; k++) {
        uint64_t sample = j->first + (uint64_t)k;
        if (j->mode == 1) { /* one light path per pixel per spp, Renderer.h:327-337 */
            Rng rng = rng_make(j->seed, sample, pix, 1);
            light_trace(&j->ctx, &rng, j->film);
            continue;
        }
        Rng rng = rng_make(j->seed, sample, pix, 0);
        V3 o, d; Col L;
        if (j->mode >= 10) {
            camera_ray(s, px, py, &o, &d); /* Renderer.h:590 */
            L = aov(&j->ctx, j->mode, o, d, &rng);
        } else {
            float u1 = rng_next(&rng), u2 = rng_next(&rng);
            camera_ray(s, px + u1, py + u2, &o, &d); /* Renderer.h:205 */
            L = path_trace(&j->ctx, o, d, &rng, j->mode == 4);
        }
        film_add(j->film, W, H, px, py, L, 0, j->mode >= 10 ? 0 : s->filter); /* Renderer.h:535; the AOV films keep their BoxFilter (Renderer.h:100-103) */
    }
}
/* One path-tracing sample of one pixel with a per-vertex log (20 floats each: material, throughput before the vertex, NEE term,
 * f, pdf, wi, wo, shading normal); returns the number of vertices logged and the sample's radiance in L[3]. */
int orc_trace_path(orc_scene* s, int x, int y, uint64_t seed, uint64_t sample, float* log, int cap, float* L3) {
    Ctx ctx; memset(&ctx, 0, sizeof(ctx)); ctx.s = s;
    Rng rng = rng_make(seed, sample, (uint32_t)(y * (int)s->width + x), 0);
    float px = x + 0.5f, py = y + 0.5f;
    float u1 = rng_next(&rng), u2 = rng_next(&rng);
    V3 o, d;
    camera_ray(s, px + u1, py + u2, &o, &d);
    g_trace = log; g_trace_n = 0; g_trace_cap = cap;
    Col L = path_trace(&ctx, o, d, &rng, 0);
    g_trace = NULL;
    L3[0] = L.r; L3[1] = L.g; L3[2] = L.b;
    return g_trace_n;
}
static void* render_worker(void* arg) {
    Job* j = (Job*)arg;
    int W = (int)j->s->width, H = (int)j->s->height;
    int total = j->tiles_x * j->tiles_y;
    for (;;) {
        int i = atomic_fetch_add(j->next_tile, 1);
        if (i >= total) break;
        int tx = i % j->tiles_x, ty = i / j->tiles_x;
        int x1 = (tx + 1) * TILE < W ? (tx + 1) * TILE : W, y1 = (ty + 1) * TILE < H ? (ty + 1) * TILE : H;
        for (int y = ty * TILE; y < y1; y++)
            for (int x = tx * TILE; x < x1; x++) render_pixel(j, x, y);
    }
    return NULL;
}
void orc_render(orc_scene* s, int mode, int spp, uint64_t seed, uint64_t first, float* film, int nthreads, orc_stats* st) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    int W = (int)s->width, H = (int)s->height;
    atomic_int next; atomic_init(&next, 0);
    Job* jobs = (Job*)calloc((size_t)nthreads, sizeof(Job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int i = 0; i < nthreads; i++) {
        Job* j = &jobs[i];
        j->s = s; j->mode = mode; j->spp = spp; j->seed = seed; j->first = first; j->film = film;
        j->next_tile = &next; j->tiles_x = (W + TILE - 1) / TILE; j->tiles_y = (H + TILE - 1) / TILE;
        j->ctx.s = s;
        if (i > 0) pthread_create(&th[i], NULL, render_worker, j);
    }
    render_worker(&jobs[0]);
    for (int i = 1; i < nthreads; i++) pthread_join(th[i], NULL);
    if (st) {
        memset(st, 0, sizeof(*st));
        for (int i = 0; i < nthreads; i++) {
            st->closest_rays += jobs[i].ctx.closest; st->any_rays += jobs[i].ctx.any;
            st->nodes_visited += jobs[i].ctx.cnt.nodes; st->tris_tested += jobs[i].ctx.cnt.tris;
        }
    }
    free(jobs); free(th);
}

/* ---------------------------------------------------------------- scene construction (ref: Scene.h:92-168, SceneLoader.h:283-287) */
orc_scene* orc_scene_create(const float* verts, const uint32_t* tri_mat, uint32_t ntris, const orc_material* mats, uint32_t nmats,
                            const orc_texture* tex, uint32_t ntex, const orc_background* bg) {
    orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene));
    s->ntris = ntris;
    s->tris = (Tri*)malloc(sizeof(Tri) * (size_t)(ntris ? ntris : 1));
    box_reset(&s->bounds);
    for (uint32_t i = 0; i < ntris; i++) {
        tri_init(&s->tris[i], verts + (size_t)i * 24, tri_mat ? tri_mat[i] : 0);
        box_extend(&s->bounds, s->tris[i].p[0]); box_extend(&s->bounds, s->tris[i].p[1]); box_extend(&s->bounds, s->tris[i].p[2]);
    }
    s->nmats = nmats;
    s->mats = (orc_material*)malloc(sizeof(orc_material) * (size_t)(nmats ? nmats : 1));
    if (nmats) memcpy(s->mats, mats, sizeof(orc_material) * nmats);
    s->ntex = ntex;
    s->tex = (Texture*)calloc(ntex ? ntex : 1, sizeof(Texture));
    for (uint32_t i = 0; i < ntex; i++) {
        size_t n = (size_t)tex[i].width * tex[i].height;
        s->tex[i].width = tex[i].width; s->tex[i].height = tex[i].height;
        s->tex[i].texels = (Col*)malloc(sizeof(Col) * n);
        memcpy(s->tex[i].texels, tex[i].rgb, sizeof(Col) * n);
    }
    if (bg) s->bg = *bg;
    if (bg && bg->kind == 1) env_build(s); /* EnvironmentMap's constructor (Lights.h:144-151) */
    /* Scene::init: background light first if it has power (Scene.h:164) */
    s->lights = (LightRef*)malloc(sizeof(LightRef) * ((size_t)ntris + 1));
    if (bg && bg_power(s) > 0) { s->lights[s->nlights].kind = bg->kind == 0 ? 1 : 2; s->lights[s->nlights].tri = 0; s->nlights++; }
    /* Scene::build: BVH (Scene.h:96-98) */
    s->leaf_tris = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(ntris ? ntris : 1));
    int root = node_new(s);
    uint32_t* all = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(ntris ? ntris : 1));
    for (uint32_t i = 0; i < ntris; i++) all[i] = i;
    build_rec(s, root, all, (int)ntris, 0);
    /* area lights in triangle order (Scene.h:107-116) */
    for (uint32_t i = 0; i < ntris; i++)
        if (nmats && mat_is_light(&s->mats[s->tris[i].mat])) { s->lights[s->nlights].kind = 0; s->lights[s->nlights].tri = i; s->nlights++; }
    s->scene_centre = vmulf(vadd(s->bounds.bmax, s->bounds.bmin), 0.5f);
    s->scene_radius = vlen(vsub(s->bounds.bmax, s->scene_centre));
    return s;
}
void orc_scene_destroy(orc_scene* s) {
    if (!s) return;
    for (uint32_t i = 0; i < s->ntex; i++) free(s->tex[i].texels);
    free(s->tex); free(s->tris); free(s->nodes); free(s->leaf_tris); free(s->mats); free(s->lights); free(s->vpls);
    free(s->env_cond); free(s->env_marg);
    free(s);
}
void orc_scene_info(orc_scene* s, int* st) {
    st[0] = s->max_depth; st[1] = s->nnodes; st[2] = s->nleaves; st[3] = (int)s->nlights; st[4] = s->max_leaf;
}
void orc_leaf_order(orc_scene* s, uint32_t* out) { memcpy(out, s->leaf_tris, sizeof(uint32_t) * s->nleaf_tris); }
void orc_tri_derived(orc_scene* s, uint32_t i, float* o) {
    const Tri* t = &s->tris[i];
    o[0] = t->e1.x; o[1] = t->e1.y; o[2] = t->e1.z; o[3] = t->e2.x; o[4] = t->e2.y; o[5] = t->e2.z;
    o[6] = t->n.x; o[7] = t->n.y; o[8] = t->n.z; o[9] = t->area; o[10] = t->d;
}

/* ---------------------------------------------------------------- batched entry points */
void orc_traverse(orc_scene* s, const float* rays, size_t n, orc_hit* out) {
    Cnt c = { 0, 0 };
    for (size_t i = 0; i < n; i++) {
        const float* r = rays + i * 6;
        out[i] = scene_traverse(s, v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), &c);
    }
}
void orc_traverse_brute(orc_scene* s, const float* rays, size_t n, orc_hit* out) { /* Scene.h:123-142 */
    for (size_t i = 0; i < n; i++) {
        const float* r = rays + i * 6;
        V3 o = v3(r[0], r[1], r[2]), d = v3(r[3], r[4], r[5]);
        orc_hit h; h.t = FLT_MAX; h.ID = 0xFFFFFFFFu; h.alpha = h.beta = h.gamma = 0;
        for (uint32_t k = 0; k < s->ntris; k++) {
            float t, u, v;
            if (tri_intersect(&s->tris[k], o, d, &t, &u, &v)) {
                if (t < h.t) { h.t = t; h.ID = k; h.alpha = u; h.beta = v; h.gamma = 1.0f - (u + v); }
            }
        }
        out[i] = h;
    }
}
void orc_visible(orc_scene* s, const float* segs, size_t n, uint8_t* out) {
    Cnt c = { 0, 0 };
    for (size_t i = 0; i < n; i++) {
        const float* q = segs + i * 6;
        out[i] = (uint8_t)scene_visible(s, v3(q[0], q[1], q[2]), v3(q[3], q[4], q[5]), &c);
    }
}
typedef struct { orc_scene* s; const float* in; size_t n; void* out; int any; } MtJob;
static void* mt_worker(void* a) {
    MtJob* j = (MtJob*)a;
    if (j->any) orc_visible(j->s, j->in, j->n, (uint8_t*)j->out);
    else orc_traverse(j->s, j->in, j->n, (orc_hit*)j->out);
    return NULL;
}
static void run_mt(orc_scene* s, const float* in, size_t n, void* out, int any, int nt) {
    if (nt < 1) nt = 1;
    if (nt > 256) nt = 256;
    pthread_t th[256]; MtJob jobs[256];
    size_t per = (n + (size_t)nt - 1) / (size_t)nt;
    int used = 0;
    for (int i = 0; i < nt; i++) {
        size_t b = per * (size_t)i; if (b >= n) break;
        size_t e = b + per < n ? b + per : n;
        jobs[i].s = s; jobs[i].in = in + b * 6; jobs[i].n = e - b; jobs[i].any = any;
        jobs[i].out = any ? (void*)((uint8_t*)out + b) : (void*)((orc_hit*)out + b);
        pthread_create(&th[i], NULL, mt_worker, &jobs[i]);
        used++;
    }
    for (int i = 0; i < used; i++) pthread_join(th[i], NULL);
}
void orc_traverse_mt(orc_scene* s, const float* rays, size_t n, orc_hit* out, int nt) { run_mt(s, rays, n, out, 0, nt); }
void orc_visible_mt(orc_scene* s, const float* segs, size_t n, uint8_t* out, int nt) { run_mt(s, segs, n, out, 1, nt); }

void orc_generate_rays(orc_scene* s, const float* xy, size_t n, float* rays) {
    for (size_t i = 0; i < n; i++) {
        V3 o, d; camera_ray(s, xy[i * 2], xy[i * 2 + 1], &o, &d);
        float* r = rays + i * 6;
        r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
    }
}
int orc_project(orc_scene* s, const float* p, float* xy) { return camera_project(s, v3(p[0], p[1], p[2]), &xy[0], &xy[1]); }

/* ---------------------------------------------------------------- probes */
static Shading probe_sd(orc_scene* s, uint32_t mat, const float* wo, const float* n, const float* uv) {
    Shading sd; memset(&sd, 0, sizeof(sd));
    sd.wo = v3(wo[0], wo[1], wo[2]);
    sd.sN = v3(n[0], n[1], n[2]); sd.gN = sd.sN;
    sd.tu = uv[0]; sd.tv = uv[1];
    sd.mat = (int)mat;
    if (mat_two_sided(&s->mats[mat]) && vdot(sd.wo, sd.sN) < 0) { sd.sN = vneg(sd.sN); sd.gN = sd.sN; }
    sd.frame = frame_from_vector(sd.sN);
    return sd;
}
void orc_bsdf_sample(orc_scene* s, uint32_t mat, const float* wo, const float* n, const float* uv, uint32_t key, uint32_t ctr, float* out) {
    Shading sd = probe_sd(s, mat, wo, n, uv);
    Rng rng; rng.key = key; rng.ctr = ctr;
    Col f; float pdf;
    V3 wi = bsdf_sample(s, &sd, &rng, &f, &pdf);
    out[0] = wi.x; out[1] = wi.y; out[2] = wi.z; out[3] = f.r; out[4] = f.g; out[5] = f.b; out[6] = pdf;
}
void orc_bsdf_eval(orc_scene* s, uint32_t mat, const float* wo, const float* n, const float* uv, const float* wi, float* out) {
    Shading sd = probe_sd(s, mat, wo, n, uv);
    V3 w = v3(wi[0], wi[1], wi[2]);
    Col f = bsdf_eval(s, &sd, w);
    out[0] = f.r; out[1] = f.g; out[2] = f.b; out[3] = bsdf_pdf(s, &sd, w);
}
/* Triangle::centre (Geometry.h:83-86) + area after Triangle::init: out5 = centre.xyz, centre.w (Vec3 / divides w too), area */
void orc_tri_centre(const float* tri24, float* out5) {
    Tri t; tri_init(&t, tri24, 0);
    V3 c = tri_centre(&t);
    out5[0] = c.x; out5[1] = c.y; out5[2] = c.z; out5[3] = 1.0f / 3.0f; out5[4] = t.area;
}
/* EnvironmentMap probes: which = 0 sample() after its two draws (out4 = wi, pdf), 1 SampleContinuous (out4 = d0, d1, pdf, 0),
 * 2 PDF(wi) with in3 = wi (out4[0]); returns 0 without an environment map */
int orc_env_probe(orc_scene* s, int which, const float* in, float* out4) {
    if (!s->env_cond) return 0;
    if (which == 0) { float pdf; V3 w = env_sample_dir(s, in[0], in[1], &pdf); out4[0] = w.x; out4[1] = w.y; out4[2] = w.z; out4[3] = pdf; }
    else if (which == 1) { float pdf, d0, d1; env_sample_continuous(s, in[0], in[1], &pdf, &d0, &d1); out4[0] = d0; out4[1] = d1; out4[2] = pdf; out4[3] = 0; }
    else { out4[0] = env_pdf(s, v3(in[0], in[1], in[2])); out4[1] = out4[2] = out4[3] = 0; }
    return 1;
}
/* the tables themselves: cond[nv * nu], marg[nv]; returns funcInt */
float orc_env_tables(orc_scene* s, float* cond, float* marg) {
    if (!s->env_cond) return 0.0f;
    if (cond) memcpy(cond, s->env_cond, sizeof(float) * (size_t)s->env_nu * s->env_nv);
    if (marg) memcpy(marg, s->env_marg, sizeof(float) * (size_t)s->env_nv);
    return s->env_func_int;
}
void orc_rng_floats(uint64_t seed, uint64_t sample, uint32_t pixel, uint32_t stream, int n, float* out) {
    Rng r = rng_make(seed, sample, pixel, stream);
    for (int i = 0; i < n; i++) out[i] = rng_next(&r);
}
void orc_sampling(int which, float r1, float r2, float* o) {
    V3 w; float pdf;
    if (which == 0) { w = cosine_sample_hemisphere(r1, r2); pdf = cosine_hemisphere_pdf(w); }
    else if (which == 1) { w = uniform_sample_sphere(r1, r2); pdf = uniform_sphere_pdf(); }
    else { w = uniform_sample_hemisphere(r1, r2); pdf = 1.0f / 2 * M_PI; /* Sampling.h:158 */ }
    o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = pdf;
}
void orc_frame(const float* n, const float* v, float* o) {
    Frame f = frame_from_vector(v3(n[0], n[1], n[2]));
    V3 a = v3(v[0], v[1], v[2]);
    V3 l = frame_to_local(&f, a), w = frame_to_world(&f, a);
    float t[15] = { f.u.x, f.u.y, f.u.z, f.v.x, f.v.y, f.v.z, f.w.x, f.w.y, f.w.z, l.x, l.y, l.z, w.x, w.y, w.z };
    memcpy(o, t, sizeof(t));
}
void orc_matrix(int which, const float* in, float* out) {
    if (which == 0) mat_perspective(in[0], in[1], in[2], in[3], out);
    else if (which == 1) mat_lookat(v3(in[0], in[1], in[2]), v3(in[3], in[4], in[5]), v3(in[6], in[7], in[8]), out);
    else mat_invert(in, out);
}
void orc_tri_sample(const float* tri24, float r1, float r2, float* p, float* pdf) {
    Tri t; tri_init(&t, tri24, 0);
    V3 q = tri_sample(&t, r1, r2, pdf);
    p[0] = q.x; p[1] = q.y; p[2] = q.z;
}
void orc_tri_shading(const float* tri24, float a, float b, float c, float* o) {
    Tri t; tri_init(&t, tri24, 0);
    V3 g = tri_gnormal(&t);
    V3 n = vnorm(vadd(vadd(vmulf(t.nrm[0], a), vmulf(t.nrm[1], b)), vmulf(t.nrm[2], c)));
    o[0] = g.x; o[1] = g.y; o[2] = g.z; o[3] = n.x; o[4] = n.y; o[5] = n.z;
    o[6] = t.u[0] * a + t.u[1] * b + t.u[2] * c;
    o[7] = t.v[0] * a + t.v[1] * b + t.v[2] * c;
}
int orc_tri_intersect(const float* tri24, const float* r, float* tuv) {
    Tri t; tri_init(&t, tri24, 0);
    tuv[0] = tuv[1] = tuv[2] = 0;
    return tri_intersect(&t, v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), &tuv[0], &tuv[1], &tuv[2]);
}
int orc_aabb_ray(const float* bmin, const float* bmax, const float* r, float* t) {
    Box b; b.bmin = v3(bmin[0], bmin[1], bmin[2]); b.bmax = v3(bmax[0], bmax[1], bmax[2]);
    V3 d = v3(r[3], r[4], r[5]);
    return box_ray(&b, v3(r[0], r[1], r[2]), v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), t);
}
float orc_aabb_default_area(void) { Box b; box_reset(&b); return box_area(&b); }
void orc_texture_sample(orc_scene* s, int tex, float u, float v, float* rgb) {
    Col c = tex_sample(&s->tex[tex], u, v);
    rgb[0] = c.r; rgb[1] = c.g; rgb[2] = c.b;
}
