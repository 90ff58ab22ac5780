// TEST INFRASTRUCTURE — not part of the product.
//
// Thin extern "C" harness around the reference's OWN geometry code, compiled
// from the sources where they lie (-I/root/reference/RTBase, see
// oracle/Makefile target `ref`).  It includes only Geometry.h (which pulls in
// Core.h and Sampling.h) -- exactly what the reference's own unit-test project
// includes (RTTest/RTTest.cpp:7).  No reference source is copied here, and no
// stand-in headers are used: the only extras are standard C++ headers that
// MSVC leaks transitively and g++ does not (<cfloat>, <iostream>, <string>),
// injected with `-include` on the command line.
//
// Scene.h and everything above it (Imaging.h -> stb_image_write.h needs the
// Annex-K sprintf_s, Renderer.h needs <Windows.h>/D3D11/OIDN) is NOT buildable
// in this image without stand-ins and is therefore not built; see DESIGN.md.
// The two lines of Scene::visible (Scene.h:169-178) that set up the shadow ray
// are restated in ref_visible() below so BVHNode::traverseVisible can be driven.
//
// Output: oracle/_ref/libref_geom.so (git-ignored).  Used by tests/ to pin
// oracle/rt_oracle.c and to generate tests/golden/*.npz.
#include "Geometry.h"

#include <cstdint>
#include <cstring>

namespace {

struct RefScene {
    std::vector<Triangle> tris;
    BVHNode* bvh = nullptr;
};

struct ArraySampler : public Sampler {
    const float* v; size_t n; size_t i = 0;
    ArraySampler(const float* _v, size_t _n) : v(_v), n(_n) {}
    float next() override { return v[(i++) % n]; }
};

Vertex make_vertex(const float* f) {
    Vertex v;
    v.p = Vec3(f[0], f[1], f[2]);
    v.normal = Vec3(f[3], f[4], f[5]);
    v.u = f[6];
    v.v = f[7];
    return v;
}

void leaf_order(BVHNode* n, std::vector<uint32_t>& out, int depth, int& maxd, int& nodes, int& leaves) {
    nodes++;
    if (depth > maxd) maxd = depth;
    if (n->l == nullptr && n->r == nullptr) {
        leaves++;
        for (Triangle* t : n->triangles) out.push_back((uint32_t)triangleMap[t]);
        return;
    }
    if (n->l) leaf_order(n->l, out, depth + 1, maxd, nodes, leaves);
    if (n->r) leaf_order(n->r, out, depth + 1, maxd, nodes, leaves);
}

}  // namespace

extern "C" {

struct RefHit { uint32_t ID; float t, alpha, beta, gamma; };  // == IntersectionData (Geometry.h:256)

// verts: [n][3][8] = p.xyz, normal.xyz, u, v ; mats: [n]
void* ref_build(const float* verts, const uint32_t* mats, uint32_t n) {
    RefScene* s = new RefScene();
    s->tris.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        const float* f = verts + (size_t)i * 24;
        s->tris[i].init(make_vertex(f), make_vertex(f + 8), make_vertex(f + 16), mats ? mats[i] : 0);
    }
    s->bvh = new BVHNode();
    s->bvh->build(s->tris);  // Geometry.h:310 (what Scene::build calls, Scene.h:98)
    return s;
}

void ref_free(void* h) {
    // The reference never frees its BVH (raw new, no destructor); neither do we
    // beyond the scene shell -- the process is short-lived test infrastructure.
    RefScene* s = (RefScene*)h;
    for (auto& t : s->tris) triangleMap.erase(&t);
    delete s;
}

// Scene::traverse with UseBVH == true (Scene.h:118-121)
void ref_traverse(void* h, const float* rays, size_t n, RefHit* out) {
    RefScene* s = (RefScene*)h;
    for (size_t i = 0; i < n; i++) {
        const float* r = rays + i * 6;
        Ray ray(Vec3(r[0], r[1], r[2]), Vec3(r[3], r[4], r[5]));
        IntersectionData d = s->bvh->traverse(ray, s->tris);
        if (!(d.t < FLT_MAX)) { d.ID = 0xFFFFFFFFu; d.alpha = d.beta = d.gamma = 0; }
        memcpy(&out[i], &d, sizeof(RefHit));
    }
}

// Scene::traverse with UseBVH == false (Scene.h:123-142): brute force over Triangle::rayIntersect
void ref_traverse_brute(void* h, const float* rays, size_t n, RefHit* out) {
    RefScene* s = (RefScene*)h;
    for (size_t i = 0; i < n; i++) {
        const float* r = rays + i * 6;
        Ray ray(Vec3(r[0], r[1], r[2]), Vec3(r[3], r[4], r[5]));
        RefHit d; d.t = FLT_MAX; d.ID = 0xFFFFFFFFu; d.alpha = d.beta = d.gamma = 0;
        for (size_t k = 0; k < s->tris.size(); k++) {
            float t, u, v;
            if (s->tris[k].rayIntersect(ray, t, u, v) && t < d.t) {
                d.t = t; d.ID = (uint32_t)k; d.alpha = u; d.beta = v; d.gamma = 1.0f - (u + v);
            }
        }
        out[i] = d;
    }
}

// Scene::visible (Scene.h:169-178): the ray set-up lines are restated, the
// traversal is the reference's BVHNode::traverseVisible (Geometry.h:472).
void ref_visible(void* h, const float* segs, size_t n, uint8_t* out) {
    RefScene* s = (RefScene*)h;
    for (size_t i = 0; i < n; i++) {
        const float* q = segs + i * 6;
        Vec3 p1(q[0], q[1], q[2]), p2(q[3], q[4], q[5]);
        Ray ray;
        Vec3 dir = p2 - p1;
        float maxT = dir.length() - (2.0f * EPSILON);
        dir = dir.normalize();
        ray.init(p1 + (dir * EPSILON), dir);
        out[i] = s->bvh->traverseVisible(ray, s->tris, maxT) ? 1 : 0;
    }
}

// DFS (left, right) leaf order of triangle indices + tree statistics.
void ref_leaf_order(void* h, uint32_t* out, int* stats /*depth,nodes,leaves*/) {
    RefScene* s = (RefScene*)h;
    std::vector<uint32_t> o;
    int maxd = 0, nodes = 0, leaves = 0;
    leaf_order(s->bvh, o, 0, maxd, nodes, leaves);
    if (out) memcpy(out, o.data(), o.size() * sizeof(uint32_t));
    if (stats) { stats[0] = maxd; stats[1] = nodes; stats[2] = leaves; }
}

// Derived triangle fields after Triangle::init (Geometry.h:66-82): e1,e2,n,area,d (11 floats)
void ref_tri_derived(void* h, uint32_t i, float* out) {
    RefScene* s = (RefScene*)h;
    const Triangle& t = s->tris[i];
    out[0] = t.e1.x; out[1] = t.e1.y; out[2] = t.e1.z;
    out[3] = t.e2.x; out[4] = t.e2.y; out[5] = t.e2.z;
    out[6] = t.n.x;  out[7] = t.n.y;  out[8] = t.n.z;
    out[9] = t.area; out[10] = t.d;
}

// Single-triangle tests: which = 0 rayIntersect (production), 2 rayIntersect2 (MT), 3 rayIntersect3
int ref_tri_intersect(const float* tri24, const float* ray6, int which, float* tuv) {
    Triangle t;
    t.init(make_vertex(tri24), make_vertex(tri24 + 8), make_vertex(tri24 + 16), 0);
    Ray r(Vec3(ray6[0], ray6[1], ray6[2]), Vec3(ray6[3], ray6[4], ray6[5]));
    float a = 0, b = 0, c = 0;
    bool hit = which == 0 ? t.rayIntersect(r, a, b, c) : which == 2 ? t.rayIntersect2(r, a, b, c) : t.rayIntersect3(r, a, b, c);
    tuv[0] = a; tuv[1] = b; tuv[2] = c;
    return hit ? 1 : 0;
}

int ref_plane_intersect(const float* n3, float d, const float* ray6, float* t) {
    Plane p; Vec3 n(n3[0], n3[1], n3[2]); p.init(n, d);
    Ray r(Vec3(ray6[0], ray6[1], ray6[2]), Vec3(ray6[3], ray6[4], ray6[5]));
    return p.rayIntersect(r, *t) ? 1 : 0;
}

float ref_aabb_default_area() { AABB b; return b.area(); }

int ref_aabb_ray(const float* bmin, const float* bmax, const float* ray6, float* t) {
    AABB b; b.min = Vec3(bmin[0], bmin[1], bmin[2]); b.max = Vec3(bmax[0], bmax[1], bmax[2]);
    Ray r(Vec3(ray6[0], ray6[1], ray6[2]), Vec3(ray6[3], ray6[4], ray6[5]));
    return b.rayAABB(r, *t) ? 1 : 0;
}

// Triangle::sample / gNormal / interpolateAttributes (Geometry.h:154-178)
void ref_tri_sample(const float* tri24, float r1, float r2, float* p3, float* pdf) {
    Triangle t;
    t.init(make_vertex(tri24), make_vertex(tri24 + 8), make_vertex(tri24 + 16), 0);
    float rr[2] = { r1, r2 };
    ArraySampler s(rr, 2);
    Vec3 p = t.sample(&s, *pdf);
    p3[0] = p.x; p3[1] = p.y; p3[2] = p.z;
}
void ref_tri_shading(const float* tri24, float a, float b, float c, float* out8) {
    Triangle t;
    t.init(make_vertex(tri24), make_vertex(tri24 + 8), make_vertex(tri24 + 16), 0);
    Vec3 g = t.gNormal(), n; float u, v;
    t.interpolateAttributes(a, b, c, n, u, v);
    out8[0] = g.x; out8[1] = g.y; out8[2] = g.z; out8[3] = n.x; out8[4] = n.y; out8[5] = n.z; out8[6] = u; out8[7] = v;
}

// Sampling.h:143-197
void ref_sampling(int which, float r1, float r2, float* out4) {
    Vec3 w;
    float pdf = 0;
    if (which == 0) { w = SamplingDistributions::cosineSampleHemisphere(r1, r2); pdf = SamplingDistributions::cosineHemispherePDF(w); }
    else if (which == 1) { w = SamplingDistributions::uniformSampleSphere(r1, r2); pdf = SamplingDistributions::uniformSpherePDF(w); }
    else { w = SamplingDistributions::uniformSampleHemisphere(r1, r2); pdf = SamplingDistributions::uniformHemispherePDF(w); }
    out4[0] = w.x; out4[1] = w.y; out4[2] = w.z; out4[3] = pdf;
}

// Core.h:513-541 Frame::fromVector + toLocal/toWorld
void ref_frame(const float* n3, const float* v3, float* out15) {
    Frame f; f.fromVector(Vec3(n3[0], n3[1], n3[2]));
    Vec3 v(v3[0], v3[1], v3[2]);
    Vec3 l = f.toLocal(v), w = f.toWorld(v);
    float o[15] = { f.u.x, f.u.y, f.u.z, f.v.x, f.v.y, f.v.z, f.w.x, f.w.y, f.w.z, l.x, l.y, l.z, w.x, w.y, w.z };
    memcpy(out15, o, sizeof(o));
}

// Core.h:326-471 Matrix::perspective / lookAt / invert / mulPointAndPerspectiveDivide
void ref_matrix(int which, const float* in, float* out16) {
    Matrix m;
    if (which == 0) m = Matrix::perspective(in[0], in[1], in[2], in[3]);
    else if (which == 1) m = Matrix::lookAt(Vec3(in[0], in[1], in[2]), Vec3(in[3], in[4], in[5]), Vec3(in[6], in[7], in[8]));
    else { memcpy(m.m, in, 64); m = m.invert(); }
    memcpy(out16, m.m, 64);
}
void ref_mul_point_div(const float* m16, const float* p4, float* out3) {
    Matrix m; memcpy(m.m, m16, 64);
    Vec3 r = m.mulPointAndPerspectiveDivide(Vec3(p4[0], p4[1], p4[2], p4[3]));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

// Vec2::operator[] (Sampling.h:199-210): declared to return int, so the float coordinate is truncated.  EnvironmentMap's
// tabulated sampling reads its 2-D draw through it (Lights.h), which is what the device's "net effect" restatement rests on.
int ref_vec2_index(float x, float y, int i) { Vec2 v(x, y); return v[i]; }

// The instance transform of loadInstance (SceneLoader.h:196-219) as Core.h performs it: which = 0 mulPoint(v), 1 mulVec(v),
// 2 transpose() (out16), 3 invert().transpose().mulVec(v).normalize() (a loaded normal), 4 operator*(second matrix in v16) (out16)
void ref_matrix_apply(int which, const float* m16, const float* v, float* out) {
    Matrix m; memcpy(m.m, m16, 64);
    if (which == 0) { Vec3 r = m.mulPoint(Vec3(v[0], v[1], v[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
    else if (which == 1) { Vec3 r = m.mulVec(Vec3(v[0], v[1], v[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
    else if (which == 2) { Matrix t = m.transpose(); memcpy(out, t.m, 64); }
    else if (which == 3) { Matrix t = m.invert(); t = t.transpose(); Vec3 r = t.mulVec(Vec3(v[0], v[1], v[2])); r = r.normalize(); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
    else { Matrix b; memcpy(b.m, v, 64); Matrix t = m * b; memcpy(out, t.m, 64); }
}

// Triangle::centre (Geometry.h:83-86; Vec3 operator/ also divides w) and the area test of loadInstance (t.area > 0, SceneLoader.h:229)
void ref_tri_centre(const float* tri24, float* out5) {
    Triangle t;
    t.init(make_vertex(tri24), make_vertex(tri24 + 8), make_vertex(tri24 + 16), 0);
    Vec3 c = t.centre();
    out5[0] = c.x; out5[1] = c.y; out5[2] = c.z; out5[3] = c.w; out5[4] = t.area;
}

// MTRandom (Sampling.h:23-36): first n draws for a seed
void ref_mtrandom(unsigned seed, int n, float* out) {
    MTRandom r(seed);
    for (int i = 0; i < n; i++) out[i] = r.next();
}

// PSS / PSSSampler (Sampling.h:43-139) driven by the call sequence of PSSMLTRender (Renderer.h:1013-1061): genPathC's two
// nextPSSC() (:978-979), then per proposal `next() < largeStepProb` -> largeStep() / smallStep() (:1027-1034), genPathP's two
// nextPSSP() (:987-988), `path_draws` calls of next() standing in for the numbers pathTrace takes from sampler->next(), and the
// accept draw next() (:1053).  Out, 8 floats per proposal: large step?, the two nextPSSP() values, proposedState.currentIndex
// and proposedState.pssCoords.size() before those two calls, the size after them, currentTimer, the accept draw.
// out2 (2 floats): the two nextPSSC() values of the initial state.
void ref_pss_trace(int proposals, int path_draws, float large_prob, float* out8, float* out2) {
    PSSSampler sampler;
    for (int k = 0; k < path_draws; k++) (void)sampler.next();   // getNormalization's draws come from the same next() (:998-1000)
    out2[0] = sampler.nextPSSC();
    out2[1] = sampler.nextPSSC();
    for (int k = 0; k < path_draws; k++) (void)sampler.next();
    for (int i = 0; i < proposals; i++) {
        float* o = out8 + (size_t)i * 8;
        const bool large = sampler.next() < large_prob;
        if (large) sampler.largeStep(); else sampler.smallStep();
        o[0] = large ? 1.0f : 0.0f;
        o[3] = (float)sampler.proposedState.currentIndex;
        o[4] = (float)sampler.proposedState.pssCoords.size();
        o[1] = sampler.nextPSSP();
        o[2] = sampler.nextPSSP();
        o[5] = (float)sampler.proposedState.pssCoords.size();
        o[6] = (float)sampler.currentTimer;
        for (int k = 0; k < path_draws; k++) (void)sampler.next();
        o[7] = sampler.next();
    }
}

}  // extern "C"
