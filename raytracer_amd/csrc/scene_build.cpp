// scene_build.cpp — host preparation of an rt_scene (compiled with -ffp-contract=off so
// every float below rounds exactly as the reference's scalar code does).
//
//  1. Triangle::init precompute (Geometry.h:66-82) into 64-B device records.
//  2. The reference's top-down 8-bin SAH tree (BVHNode::RecursiveBuild / findBestSplitPlane,
//     Geometry.h:319-438), built iteratively over one index array.  Same cost expression, same
//     strict '<' tie rule, same stable partition, same "split always attempted" rule, so the
//     tree -- and therefore the first-visited-wins order of equal-t hits -- is the reference's.
//  3. The reference-topology node array (DNode2).  What the walk reads -- the 8-wide tree, the walk-ordered
//     triangle records and the per-triangle gate boxes -- is derived from it on the device (walk_tree.hip).
#include "scene_build.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <chrono>
#include <thread>
#include <atomic>
#include <cmath>

namespace {

struct F3 { float x, y, z; };
inline F3 f3(float x, float y, float z) { return F3{ x, y, z }; }
inline F3 sub(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline F3 add(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline float dot(F3 a, F3 b) { return ((a.x * b.x) + (a.y * b.y) + (a.z * b.z)); }
inline F3 cross(F3 a, F3 b) { return f3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)); }
inline float length(F3 a) { return sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); }
inline F3 normalize(F3 a) { float l = 1.0f / sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); return f3(a.x * l, a.y * l, a.z * l); }
inline float comp(F3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

struct Aabb {
    F3 hi{ -FLT_MAX_, -FLT_MAX_, -FLT_MAX_ }, lo{ FLT_MAX_, FLT_MAX_, FLT_MAX_ };
    static constexpr float FLT_MAX_ = 3.402823466e+38f;
    void grow(F3 p) {  // AABB::extend, ternary Max/Min (Geometry.h:195, Core.h:187-195)
        hi = f3(hi.x > p.x ? hi.x : p.x, hi.y > p.y ? hi.y : p.y, hi.z > p.z ? hi.z : p.z);
        lo = f3(lo.x < p.x ? lo.x : p.x, lo.y < p.y ? lo.y : p.y, lo.z < p.z ? lo.z : p.z);
    }
    void grow(const Aabb& b) { grow(b.hi); grow(b.lo); }  // Geometry.h:201
    float area() const {                                 // Geometry.h:231
        F3 s = sub(hi, lo);
        return ((s.x * s.y) + (s.y * s.z) + (s.x * s.z)) * 2.0f;
    }
};

struct BNode {  // reference-topology node
    Aabb box;
    int left = -1, right = -1;
    uint32_t begin = 0, count = 0;  // range in the index array (all triangles below the node)
    uint32_t first = 0, nleaf = 0;  // leaves: range in DFS order
    uint32_t depth = 0;
};

struct Builder {
    const float* verts;
    uint32_t n;
    std::vector<F3> centroid;
    std::vector<uint32_t> idx, tmp, order;
    std::vector<BNode> nodes;
    uint32_t max_depth = 0, leaves = 0, max_leaf = 0;

    F3 vert(uint32_t t, int k) const { const float* f = verts + (size_t)t * 24 + k * 8; return f3(f[0], f[1], f[2]); }

    // findBestSplitPlane (Geometry.h:319-383)
    void best_split(const BNode& nd, int& axis, float& pos) const {
        const int BINS = 8;
        float best = Aabb::FLT_MAX_;
        float parent_area = nd.box.area();
        const uint32_t* t = idx.data() + nd.begin;
        for (int a = 0; a < 3; a++) {
            float cmin = Aabb::FLT_MAX_, cmax = -Aabb::FLT_MAX_;
            for (uint32_t i = 0; i < nd.count; i++) {
                float c = comp(centroid[t[i]], a);
                cmin = std::min(cmin, c);
                cmax = std::max(cmax, c);
            }
            if (cmin == cmax) continue;
            Aabb bb[BINS]; int bc[BINS] = { 0 };
            float scale = BINS / (cmax - cmin);
            for (uint32_t i = 0; i < nd.count; i++) {
                float c = comp(centroid[t[i]], a);
                int b = std::min(BINS - 1, static_cast<int>((c - cmin) * scale));
                bc[b]++;
                bb[b].grow(vert(t[i], 0)); bb[b].grow(vert(t[i], 1)); bb[b].grow(vert(t[i], 2));
            }
            float la[BINS - 1], ra[BINS - 1]; int lc[BINS - 1], rc[BINS - 1];
            Aabb lb, rb; int ls = 0, rs = 0;
            for (int i = 0; i < BINS - 1; i++) {
                ls += bc[i]; lc[i] = ls;
                if (bc[i] > 0) { lb.grow(bb[i]); la[i] = lb.area(); } else la[i] = 0;  // area zeroed after an empty bin (Geometry.h:360)
                rs += bc[BINS - 1 - i]; rc[BINS - 2 - i] = rs;
                if (bc[BINS - 1 - i] > 0) { rb.grow(bb[BINS - 1 - i]); ra[BINS - 2 - i] = rb.area(); } else ra[BINS - 2 - i] = 0;
            }
            float bin_w = (cmax - cmin) / BINS;
            for (int i = 0; i < BINS - 1; i++) {
                float cost = 1.0f + ((la[i] / parent_area) * lc[i] * 1.0f) + ((ra[i] / parent_area) * rc[i] * 1.0f);
                if (cost < best) { best = cost; axis = a; pos = cmin + bin_w * (i + 1); }
            }
        }
    }

    void build() {
        centroid.resize(n); idx.resize(n); tmp.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            F3 s = add(add(vert(i, 0), vert(i, 1)), vert(i, 2));  // Triangle::centre (Geometry.h:83)
            centroid[i] = f3(s.x / 3.0f, s.y / 3.0f, s.z / 3.0f);
            idx[i] = i;
        }
        nodes.reserve((size_t)2 * n + 1);
        nodes.emplace_back();
        nodes[0].begin = 0; nodes[0].count = n;
        std::vector<int> stack{ 0 };
        while (!stack.empty()) {  // LIFO with "right pushed first" == left-then-right DFS, so leaves are emitted in visit order
            int ni = stack.back(); stack.pop_back();
            BNode nd = nodes[ni];
            max_depth = std::max(max_depth, nd.depth);
            bool split = false;
            if (nd.count > 1) {  // num <= 1 returns before the bounds are grown (Geometry.h:401)
                for (uint32_t i = 0; i < nd.count; i++) {
                    uint32_t t = idx[nd.begin + i];
                    nd.box.grow(vert(t, 0)); nd.box.grow(vert(t, 1)); nd.box.grow(vert(t, 2));
                }
                int axis = -1; float pos = 0.0f;
                best_split(nd, axis, pos);
                // axis == -1: every centroid coincides; upstream indexes coords[-1] (undefined). Defined here as "no split".
                if (axis >= 0) {
                    uint32_t nl = 0, nr = 0;
                    for (uint32_t i = 0; i < nd.count; i++) {  // stable partition (Geometry.h:417-424)
                        uint32_t t = idx[nd.begin + i];
                        if (comp(centroid[t], axis) < pos) idx[nd.begin + nl++] = t; else tmp[nr++] = t;
                    }
                    // idx[begin..begin+nl) was compacted in place: element i is only ever written at a position <= i
                    memcpy(idx.data() + nd.begin + nl, tmp.data(), sizeof(uint32_t) * nr);
                    if (nl > 0 && nr > 0) {
                        split = true;
                        int l = (int)nodes.size(); nodes.emplace_back();
                        int r = (int)nodes.size(); nodes.emplace_back();
                        nodes[l].begin = nd.begin; nodes[l].count = nl; nodes[l].depth = nd.depth + 1;
                        nodes[r].begin = nd.begin + nl; nodes[r].count = nr; nodes[r].depth = nd.depth + 1;
                        nd.left = l; nd.right = r;
                        stack.push_back(r); stack.push_back(l);
                    }
                }
            }
            if (!split) {
                nd.first = (uint32_t)order.size(); nd.nleaf = nd.count;
                for (uint32_t i = 0; i < nd.count; i++) order.push_back(idx[nd.begin + i]);
                leaves++; max_leaf = std::max(max_leaf, nd.count);
            }
            nodes[ni] = nd;
        }
    }
};

inline void put3(float* d, F3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
inline float lum(const float* c) { return ((0.2126f * c[0]) + (0.7152f * c[1]) + (0.0722f * c[2])); }  // Colour::Lum (Core.h:89)

}  // namespace

bool host_scene_build(const rt_scene_desc& d, HostScene& hs, std::string& err, DeviceSceneBuilder device_builder, DeviceGeometry* geom) {
    auto t0 = std::chrono::steady_clock::now();
    if (d.n_tris > RT_MAX_TRIS) { err = "too many triangles (max 2^27-1)"; return false; }
    if (d.n_tris && (!d.verts || !d.tri_material)) { err = "verts/tri_material is NULL"; return false; }
    if (d.n_materials && !d.materials) { err = "materials is NULL"; return false; }
    if (d.n_textures && !d.textures) { err = "textures is NULL"; return false; }
    for (uint32_t i = 0; i < d.n_tris; i++)
        if (d.tri_material[i] >= d.n_materials) { err = "tri_material index out of range"; return false; }
    // a NaN / inf position makes upstream's SAH binning index its 32 bins out of range (Geometry.h:319-383: (int)((centre - min) * scale))
    for (size_t i = 0; i < (size_t)d.n_tris * 3; i++) {
        const float* p = d.verts + i * 8;
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) { err = "non-finite vertex position"; return false; }
    }
    for (uint32_t i = 0; i < d.n_materials; i++) {
        if (d.materials[i].type > RT_MAT_PLASTIC) { err = "unknown material type"; return false; }
        if (d.materials[i].albedo_tex < 0 || (uint32_t)d.materials[i].albedo_tex >= d.n_textures) { err = "albedo_tex out of range"; return false; }
    }
    if (d.background.kind > 1) { err = "unknown background kind"; return false; }
    if (d.background.kind == 1 && (d.background.env_tex < 0 || (uint32_t)d.background.env_tex >= d.n_textures)) { err = "env_tex out of range"; return false; }

    // textures
    for (uint32_t i = 0; i < d.n_textures; i++) {
        const rt_texture& t = d.textures[i];
        if (t.width <= 0 || t.height <= 0 || !t.rgb) { err = "bad texture"; return false; }
        DTexture dt; dt.offset = (uint32_t)hs.texels.size(); dt.width = t.width; dt.height = t.height; dt.pad = 0;
        hs.tex.push_back(dt);
        hs.texels.insert(hs.texels.end(), t.rgb, t.rgb + (size_t)t.width * t.height * 3);
    }
    // materials
    for (uint32_t i = 0; i < d.n_materials; i++) {
        const rt_material& m = d.materials[i];
        DMaterial o; memset(&o, 0, sizeof(o));
        o.type = m.type; o.tex = m.albedo_tex;
        bool layered = (m.flags & RT_MAT_FLAG_LAYERED) != 0;
        bool two_sided = layered || m.type != RT_MAT_DIELECTRIC;          // isTwoSided (Materials.h:738, :1044)
        bool specular = m.type == RT_MAT_MIRROR || m.type == RT_MAT_GLASS; // isPureSpecular
        o.flags = (two_sided ? RT_MF_TWOSIDED : 0) | (specular ? RT_MF_SPECULAR : 0) | (lum(m.emission) > 0 ? RT_MF_LIGHT : 0);
        o.alpha = 1.62142f * sqrtf(m.roughness);
        memcpy(o.emission, m.emission, 12);
        o.int_ior = m.int_ior; o.ext_ior = m.ext_ior;
        memcpy(o.eta, m.eta, 12); memcpy(o.k, m.k, 12);
        float a = std::max(o.alpha, 0.001f);
        o.phong_e = (2.0f / (a * a)) - 2.0f;
        hs.mats.push_back(o);
    }
    hs.bg_kind = d.background.kind; hs.bg_tex = d.background.env_tex;
    memcpy(hs.bg_colour, d.background.colour, 12);
    if (d.background.kind == 1) {   // EnvironmentMap::build (Lights.h:207-253), same float operations in the same order
        const rt_texture& e = d.textures[d.background.env_tex];
        const int nu = e.width, nv = e.height;
        std::vector<float> func((size_t)nu * nv);
        float funcInt = 0.0f;
        for (int v = 0; v < nv; ++v) {
            float theta = (v + 0.5f) / nv * M_PI;
            float sinTheta = sinf(theta);
            for (int u = 0; u < nu; ++u) {
                size_t idx = (size_t)v * nu + u;
                func[idx] = lum(e.rgb + idx * 3) * sinTheta;
                funcInt += func[idx];
            }
        }
        hs.env_cond.resize((size_t)nu * nv); hs.env_marg.resize((size_t)nv);
        for (int v = 0; v < nv; ++v) {
            float* row = hs.env_cond.data() + (size_t)v * nu;
            float sum = 0.0f;
            for (int u = 0; u < nu; ++u) { row[u] = func[(size_t)v * nu + u]; sum += row[u]; }
            if (sum > 0.0f) for (int u = 0; u < nu; ++u) row[u] /= sum;
        }
        for (int v = 0; v < nv; ++v) {
            float sum = 0.0f;
            for (int u = 0; u < nu; ++u) sum += func[(size_t)v * nu + u];
            hs.env_marg[v] = sum;
        }
        if (funcInt > 0.0f) for (int v = 0; v < nv; ++v) hs.env_marg[v] /= funcInt;
        hs.env_func_int = funcInt; hs.env_nu = nu; hs.env_nv = nv;
    }

    std::vector<uint32_t> rank_of(d.n_tris);
    int on_device = 2;
    if (device_builder && geom) {
        on_device = device_builder(d, hs, *geom, err);
        if (on_device == 0) return false;
    }
    if (on_device == 1) {
        for (uint32_t r = 0; r < d.n_tris; r++) rank_of[hs.leaf_order[r]] = r;
    } else {
    // reference tree
    Builder b; b.verts = d.verts; b.n = d.n_tris;
    b.build();
    if (getenv("RT_BUILD_TIMING")) fprintf(stderr, "bvh2 build %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    hs.leaf_order = b.order;
    hs.depth2 = b.max_depth; hs.leaves2 = b.leaves; hs.max_leaf2 = b.max_leaf;
    hs.max_stack2 = b.max_depth + 2;

    // triangle records in DFS order + scene bounds (Scene::init, Scene.h:152-158) on several threads (the partial bounds are
    // folded in order below), the reference-topology nodes beside them
    Aabb bounds;
    hs.tris.resize(d.n_tris); hs.trishade.resize(d.n_tris);
    float amax = 0;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const uint32_t nthr = (uint32_t)std::min<uint64_t>(std::min(16u, hw), std::max<uint64_t>(1, d.n_tris / 65536));
    std::vector<Aabb> part_bounds(nthr);
    std::vector<float> part_amax(nthr, 0.0f);
    auto records = [&](uint32_t ti) {
        const uint32_t r0 = (uint32_t)((uint64_t)d.n_tris * ti / nthr), r1 = (uint32_t)((uint64_t)d.n_tris * (ti + 1) / nthr);
        Aabb bounds; float amax = 0;
        for (uint32_t r = r0; r < r1; r++) {
            uint32_t t = b.order[r];
            rank_of[t] = r;
            const float* f = d.verts + (size_t)t * 24;
            F3 p0 = f3(f[0], f[1], f[2]), p1 = f3(f[8], f[9], f[10]), p2 = f3(f[16], f[17], f[18]);
            F3 e1 = sub(p2, p1), e2 = sub(p0, p2);
            F3 n = normalize(cross(e1, e2));
            DTri& o = hs.tris[r];
            put3(o.v0, p0); put3(o.v1, p1); put3(o.v2, p2); put3(o.n, n);
            o.d = dot(n, p0);
            o.inv_area = 1.0f / dot(cross(e1, e2), n);
            o.orig_id = t;
            F3 n0 = f3(f[3], f[4], f[5]);
            bool flip = !(dot(n0, n) > 0);  // gNormal = n * (Dot(v0.normal, n) > 0 ? 1 : -1)
            o.mat_gsign = d.tri_material[t] | (flip ? 0x80000000u : 0u);
            DTriShade& s = hs.trishade[r];
            put3(s.n0, n0); put3(s.n1, f3(f[11], f[12], f[13])); put3(s.n2, f3(f[19], f[20], f[21]));
            s.uv[0] = f[6]; s.uv[1] = f[7]; s.uv[2] = f[14]; s.uv[3] = f[15]; s.uv[4] = f[22]; s.uv[5] = f[23];
            s.mat_gsign = o.mat_gsign;
            bounds.grow(p0); bounds.grow(p1); bounds.grow(p2);
            for (int k = 0; k < 3; k++) for (int c = 0; c < 3; c++) amax = std::max(amax, fabsf(f[k * 8 + c]));
        }
        part_bounds[ti] = bounds; part_amax[ti] = amax;
    };
    std::vector<std::thread> workers;
    for (uint32_t ti = 1; ti < nthr; ti++) workers.emplace_back(records, ti);
    std::thread tree([&]() {
        // reference-topology device nodes
        hs.nodes2.resize(b.nodes.size());
        for (size_t i = 0; i < b.nodes.size(); i++) {
            const BNode& s = b.nodes[i];
            DNode2& o = hs.nodes2[i];
            put3(o.lo, s.box.lo); put3(o.hi, s.box.hi);
            o.left = s.left; o.right = s.right; o.first = (int32_t)s.first; o.count = (int32_t)s.nleaf; o.pad0 = (int32_t)s.begin; o.pad1 = (int32_t)s.count;
        }
        if (getenv("RT_BUILD_TIMING")) fprintf(stderr, "  nodes2 thread done at %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    });
    records(0);
    for (auto& t : workers) t.join();
    if (getenv("RT_BUILD_TIMING")) fprintf(stderr, "  records (%u threads) done at %.3f s\n", nthr, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    tree.join();
    for (uint32_t ti = 0; ti < nthr; ti++) {
        const Aabb& q = part_bounds[ti];   // folded in order with extend's own ternaries: the same bits as one pass, signed zeros included
        bounds.hi = f3(bounds.hi.x > q.hi.x ? bounds.hi.x : q.hi.x, bounds.hi.y > q.hi.y ? bounds.hi.y : q.hi.y, bounds.hi.z > q.hi.z ? bounds.hi.z : q.hi.z);
        bounds.lo = f3(bounds.lo.x < q.lo.x ? bounds.lo.x : q.lo.x, bounds.lo.y < q.lo.y ? bounds.lo.y : q.lo.y, bounds.lo.z < q.lo.z ? bounds.lo.z : q.lo.z);
        amax = std::max(amax, part_amax[ti]);
    }
    hs.abs_max = amax;
    put3(hs.bounds_lo, bounds.lo); put3(hs.bounds_hi, bounds.hi);
    hs.n_tris = d.n_tris; hs.n_nodes2 = (uint32_t)hs.nodes2.size();
    }
    {   // SceneBounds (SceneLoader.h:286-287)
        F3 lo = f3(hs.bounds_lo[0], hs.bounds_lo[1], hs.bounds_lo[2]), hi = f3(hs.bounds_hi[0], hs.bounds_hi[1], hs.bounds_hi[2]);
        F3 c = f3((hi.x + lo.x) * 0.5f, (hi.y + lo.y) * 0.5f, (hi.z + lo.z) * 0.5f);
        put3(hs.centre, c);
        hs.radius = length(sub(hi, c));
    }

    if (getenv("RT_BUILD_TIMING")) fprintf(stderr, "records + nodes2 done at %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    // light list: background first if it has power (Scene.h:164), then emissive triangles in caller order (Scene.h:107)
    {
        float power = 0;
        if (d.background.kind == 0) power = lum(d.background.colour) * 4.0f * (float)M_PI;
        else {  // EnvironmentMap::totalIntegratedPower (Lights.h:184)
            const rt_texture& e = d.textures[d.background.env_tex];
            float total = 0;
            for (int i = 0; i < e.height; i++) {
                float st = sinf(((float)i / (float)e.height) * M_PI);
                for (int k = 0; k < e.width; k++) total += (lum(e.rgb + ((size_t)i * e.width + k) * 3) * st);
            }
            total = total / (float)(e.width * e.height);
            power = total * 4.0f * M_PI;
        }
        if (power > 0) hs.lights.push_back(DLight{ d.background.kind == 0 ? 1u : 2u, 0u, 0.0f, 0u });
        for (uint32_t t = 0; t < d.n_tris; t++)
            if (hs.mats[d.tri_material[t]].flags & RT_MF_LIGHT) {
                const float* f = d.verts + (size_t)t * 24;   // Triangle::init's area (Geometry.h:80), the expression k_records' device twin uses
                F3 p0 = f3(f[0], f[1], f[2]), p1 = f3(f[8], f[9], f[10]), p2 = f3(f[16], f[17], f[18]);
                hs.lights.push_back(DLight{ 0u, rank_of[t], length(cross(sub(p2, p1), sub(p0, p2))) * 0.5f, 0u });
            }
    }
    hs.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

// ---- camera (Scene.h:26-45) with Matrix::invert (Core.h:326) ----------------------------------------------------
namespace {
void invert4(const float* m, float* out) {
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0) det = 1.0f;
    det = 1.0f / det;
    for (int i = 0; i < 16; i++) out[i] = inv[i] * det;
}
F3 mul_point(const float* m, F3 v) {
    return f3((v.x * m[0] + v.y * m[1] + v.z * m[2]) + m[3], (v.x * m[4] + v.y * m[5] + v.z * m[6]) + m[7], (v.x * m[8] + v.y * m[9] + v.z * m[10]) + m[11]);
}
F3 mul_point_div(const float* m, F3 v) {
    F3 a = mul_point(m, v);
    float w = (m[12] * v.x) + (m[13] * v.y) + (m[14] * v.z) + m[15];
    w = 1.0f / w;
    return f3(a.x * w, a.y * w, a.z * w);
}
F3 mul_vec(const float* m, F3 v) {
    return f3((v.x * m[0] + v.y * m[1] + v.z * m[2]), (v.x * m[4] + v.y * m[5] + v.z * m[6]), (v.x * m[8] + v.y * m[9] + v.z * m[10]));
}
}  // namespace

// Matrix::invert (Core.h:326) for host code above the ABI (lookAt().invert(), instance normal matrices)
extern "C" int rt_matrix_invert(const float* m16, float* out16) {
    if (!m16 || !out16) return -1;
    invert4(m16, out16);
    return 0;
}

void host_camera_build(const rt_camera& c, DCamera& o) {
    memcpy(o.P, c.P, 64);
    invert4(c.P, o.invP);
    o.width = (float)c.width; o.height = (float)c.height; o.iw = c.width; o.ih = c.height;
    float Wlens = (2.0f / c.P[5]);
    float aspect = c.P[0] / c.P[5];
    float Hlens = Wlens * aspect;
    o.Afilm = Wlens * Hlens;
    memcpy(o.cam, c.V, 64);
    invert4(c.V, o.cam2view);
    put3(o.origin, mul_point(o.cam, f3(0, 0, 0)));
    F3 vd = mul_point_div(o.invP, f3(0, 0, 1));
    vd = mul_vec(o.cam, vd);
    put3(o.view_dir, normalize(vd));
}
