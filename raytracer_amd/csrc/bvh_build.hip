// bvh_build.hip — BVHNode::build / findBestSplitPlane (Geometry.h:310-438) on the device, same tree as the host builder.
//
// Why the same tree comes out of a parallel build: every quantity upstream accumulates over a node's triangles is a
// min, a max or an integer count (node bounds, centroid range, 8 bin counts and bin bounds per axis), all exact and
// order-independent; the only order-dependent step is the stable partition `centre()[axis] < splitPos`
// (Geometry.h:417-424), reproduced by a prefix sum over the flags.  (The ternary Min/Max of upstream can return -0.0 or
// +0.0 depending on visiting order; no decision reads the sign of a zero — areas, costs, `cmin == cmax`, bin indices and the
// partition compare by value — so only the stored sign bit of a zero box plane may differ from the host build.)
// The float expressions of the split search are the host builder's, evaluated by one thread per node.
//
// Level-synchronous: all nodes of one depth are processed together, 8 kernel launches + 2 scans per level:
//   owner scan -> bounds/centroid range -> binning -> split search (per node) -> flags -> flag scan -> scatter -> children.
// Big nodes are reduced per wave before touching the per-node accumulators (one atomic per wave and value); nodes that
// share a wave with others go straight to global atomics, where contention is low by construction.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "scene_build.h"
#include <string.h>
#include <utility>

namespace {

#define BB_CHECK(call)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) { err = std::string("bvh_build: ") + #call + ": " + hipGetErrorString(e_); return false; } \
    } while (0)

constexpr int BINS = 8;
constexpr float FMAX = 3.402823466e+38f;
// accumulator layout per pending node (uint32 words): 0-2 box lo, 3-5 box hi, 6-8 centroid min, 9-11 centroid max,
// then per axis: 8 counts, 8 x (lo xyz, hi xyz)
constexpr int ACC_BOX = 0, ACC_CEN = 6, ACC_AXIS = 12, ACC_AXIS_WORDS = BINS + BINS * 6, ACC_WORDS = ACC_AXIS + 3 * ACC_AXIS_WORDS;

struct GNode { float lo[3], hi[3]; int32_t left, right; uint32_t begin, count, depth; int32_t slot; /* accumulator slot while pending, else -1 */ };
struct Split { int32_t axis; float pos; };

// order-preserving float <-> uint key, so integer atomicMin/atomicMax implement float min/max
__device__ __forceinline__ uint32_t fkey(float f) { uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float funkey(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }
constexpr uint32_t KEY_POS_MAX = 0xFF7FFFFFu;   // fkey(+FLT_MAX)
constexpr uint32_t KEY_NEG_MAX = 0x00800000u;   // fkey(-FLT_MAX)

__device__ __forceinline__ float centroid_of(const float* __restrict__ verts, uint32_t t, int a) {  // Triangle::centre (Geometry.h:83)
    const float* f = verts + (size_t)t * 24 + a;
    return ((f[0] + f[8]) + f[16]) / 3.0f;
}

__global__ void k_init(uint32_t n, uint32_t* idx, uint32_t* start_mark, int32_t* node_at_start, GNode* nodes) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { idx[i] = i; start_mark[i] = i == 0 ? 1u : 0u; node_at_start[i] = i == 0 ? 0 : -1; }
    if (i == 0) {
        GNode r;
        for (int k = 0; k < 3; k++) { r.lo[k] = FMAX; r.hi[k] = -FMAX; }
        r.left = r.right = -1; r.begin = 0; r.count = n; r.depth = 0; r.slot = n > 1 ? 0 : -1;
        nodes[0] = r;
    }
}
__global__ void k_acc_reset(uint32_t* acc, uint32_t n_pending) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_pending * ACC_WORDS) return;
    uint32_t w = (uint32_t)(i % ACC_WORDS);
    uint32_t v;
    if (w < ACC_AXIS) v = ((w % 6) < 3) ? KEY_POS_MAX : KEY_NEG_MAX;                 // lo / min start at +FLT_MAX, hi / max at -FLT_MAX
    else { uint32_t r = (w - ACC_AXIS) % ACC_AXIS_WORDS; v = r < BINS ? 0u : (((r - BINS) % 6) < 3 ? KEY_POS_MAX : KEY_NEG_MAX); }
    acc[i] = v;
}
// pending slot of the node that owns position p, or -1
__device__ __forceinline__ int32_t slot_at(uint32_t p, const uint32_t* seg, const int32_t* node_at_start, const GNode* nodes, int32_t& node) {
    node = node_at_start[seg[p] - 1u];
    return nodes[node].slot;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v) { for (int o = 32; o; o >>= 1) { uint32_t t = __shfl_xor(v, o); v = t < v ? t : v; } return v; }
__device__ __forceinline__ uint32_t wave_max(uint32_t v) { for (int o = 32; o; o >>= 1) { uint32_t t = __shfl_xor(v, o); v = t > v ? t : v; } return v; }

// node bounds over all vertices (Geometry.h:403-409) and centroid range per axis (:326-333)
__global__ void k_bounds(uint32_t n, const float* __restrict__ verts, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ seg,
                         const int32_t* __restrict__ node_at_start, const GNode* __restrict__ nodes, uint32_t* acc) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    int32_t slot = -1, node;
    uint32_t lo[3], hi[3], cmn[3], cmx[3];
    for (int a = 0; a < 3; a++) { lo[a] = cmn[a] = 0xFFFFFFFFu; hi[a] = cmx[a] = 0u; }
    if (p < n) {
        slot = slot_at(p, seg, node_at_start, nodes, node);
        if (slot >= 0) {
            uint32_t t = idx[p];
            const float* f = verts + (size_t)t * 24;
            for (int a = 0; a < 3; a++) {
                float v0 = f[a], v1 = f[8 + a], v2 = f[16 + a];
                lo[a] = fkey(fminf(v0, fminf(v1, v2))); hi[a] = fkey(fmaxf(v0, fmaxf(v1, v2)));
                cmn[a] = cmx[a] = fkey(centroid_of(verts, t, a));
            }
        }
    }
    // a wave that lies inside one node reduces first: one atomic per value per wave
    int32_t s0 = __shfl(slot, 0);
    bool uniform = __all(slot == s0);
    if (uniform) {
        if (s0 < 0) return;
        for (int a = 0; a < 3; a++) { lo[a] = wave_min(lo[a]); hi[a] = wave_max(hi[a]); cmn[a] = wave_min(cmn[a]); cmx[a] = wave_max(cmx[a]); }
        if ((threadIdx.x & 63) != 0) return;
    } else if (slot < 0) return;
    uint32_t* A = acc + (size_t)slot * ACC_WORDS;
    for (int a = 0; a < 3; a++) {
        atomicMin(A + ACC_BOX + a, lo[a]); atomicMax(A + ACC_BOX + 3 + a, hi[a]);
        atomicMin(A + ACC_CEN + a, cmn[a]); atomicMax(A + ACC_CEN + 3 + a, cmx[a]);
    }
}
// bin counts and bin bounds (Geometry.h:335-349).  A block that lies inside one node (every block of the top levels)
// bins into LDS and flushes 168 values with one global atomic each; mixed blocks use global atomics directly.
__global__ __launch_bounds__(256) void k_bin(uint32_t n, const float* __restrict__ verts, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ seg,
                      const int32_t* __restrict__ node_at_start, const GNode* __restrict__ nodes, uint32_t* acc) {
    constexpr int W = 3 * ACC_AXIS_WORDS;
    __shared__ uint32_t s_acc[W];
    __shared__ int32_t s_slot;
    __shared__ int s_mixed;
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    int32_t node, slot = -2;   // -2: position beyond the array, matches anything
    if (p < n) slot = slot_at(p, seg, node_at_start, nodes, node);
    if (threadIdx.x == 0) { s_slot = slot; s_mixed = 0; }
    for (int w = threadIdx.x; w < W; w += blockDim.x) { int r = w % ACC_AXIS_WORDS; s_acc[w] = r < BINS ? 0u : (((r - BINS) % 6) < 3 ? KEY_POS_MAX : KEY_NEG_MAX); }
    __syncthreads();
    if (slot != -2 && slot != s_slot) s_mixed = 1;
    __syncthreads();
    const bool local = !s_mixed;
    if (local && s_slot < 0) return;   // nothing pending in this block
    if (slot >= 0) {
        uint32_t* A = acc + (size_t)slot * ACC_WORDS;
        uint32_t t = idx[p];
        const float* f = verts + (size_t)t * 24;
        uint32_t lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            float v0 = f[a], v1 = f[8 + a], v2 = f[16 + a];
            lo[a] = fkey(fminf(v0, fminf(v1, v2))); hi[a] = fkey(fmaxf(v0, fmaxf(v1, v2)));
        }
        for (int a = 0; a < 3; a++) {
            float cmin = funkey(A[ACC_CEN + a]), cmax = funkey(A[ACC_CEN + 3 + a]);
            if (cmin == cmax) continue;
            float scale = BINS / (cmax - cmin);
            float c = centroid_of(verts, t, a);
            int b = static_cast<int>((c - cmin) * scale);
            b = b < BINS - 1 ? b : BINS - 1;   // std::min(BINS - 1, ...)
            uint32_t* X = local ? s_acc + a * ACC_AXIS_WORDS : A + ACC_AXIS + a * ACC_AXIS_WORDS;
            atomicAdd(X + b, 1u);
            uint32_t* B = X + BINS + b * 6;
            for (int k = 0; k < 3; k++) { atomicMin(B + k, lo[k]); atomicMax(B + 3 + k, hi[k]); }
        }
    }
    if (!local) return;
    __syncthreads();
    uint32_t* A = acc + (size_t)s_slot * ACC_WORDS + ACC_AXIS;
    for (int w = threadIdx.x; w < W; w += blockDim.x) {
        int r = w % ACC_AXIS_WORDS;
        uint32_t v = s_acc[w];
        if (r < BINS) { if (v) atomicAdd(A + w, v); }
        else if (((r - BINS) % 6) < 3) { if (v != KEY_POS_MAX) atomicMin(A + w, v); }
        else if (v != KEY_NEG_MAX) atomicMax(A + w, v);
    }
}
struct Box {
    float hi[3] = { -FMAX, -FMAX, -FMAX }, lo[3] = { FMAX, FMAX, FMAX };
    __device__ void grow(const float* p) {  // AABB::extend (Geometry.h:195), ternary Max/Min
        for (int k = 0; k < 3; k++) { hi[k] = hi[k] > p[k] ? hi[k] : p[k]; lo[k] = lo[k] < p[k] ? lo[k] : p[k]; }
    }
    __device__ void grow(const Box& b) { grow(b.hi); grow(b.lo); }
    __device__ float area() const { float sx = hi[0] - lo[0], sy = hi[1] - lo[1], sz = hi[2] - lo[2]; return ((sx * sy) + (sy * sz) + (sx * sz)) * 2.0f; }
};
// findBestSplitPlane's sweep (Geometry.h:351-383), one thread per pending node; also stores the node's box
__global__ void k_split(uint32_t n_pending, const uint32_t* __restrict__ pending, GNode* nodes, const uint32_t* __restrict__ acc, Split* splits) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pending) return;
    GNode& nd = nodes[pending[i]];
    const uint32_t* A = acc + (size_t)i * ACC_WORDS;
    Box nb;
    for (int k = 0; k < 3; k++) { nb.lo[k] = funkey(A[ACC_BOX + k]); nb.hi[k] = funkey(A[ACC_BOX + 3 + k]); nd.lo[k] = nb.lo[k]; nd.hi[k] = nb.hi[k]; }
    float best = FMAX;
    float parent_area = nb.area();
    int axis = -1; float pos = 0.0f;
    for (int a = 0; a < 3; a++) {
        float cmin = funkey(A[ACC_CEN + a]), cmax = funkey(A[ACC_CEN + 3 + a]);
        if (cmin == cmax) continue;
        const uint32_t* X = A + ACC_AXIS + a * ACC_AXIS_WORDS;
        Box bb[BINS]; int bc[BINS];
        for (int b = 0; b < BINS; b++) {
            bc[b] = (int)X[b];
            if (bc[b] > 0) for (int k = 0; k < 3; k++) { bb[b].lo[k] = funkey(X[BINS + b * 6 + k]); bb[b].hi[k] = funkey(X[BINS + b * 6 + 3 + k]); }
        }
        float la[BINS - 1], ra[BINS - 1]; int lc[BINS - 1], rc[BINS - 1];
        Box lb, rb; int ls = 0, rs = 0;
        for (int b = 0; b < BINS - 1; b++) {
            ls += bc[b]; lc[b] = ls;
            if (bc[b] > 0) { lb.grow(bb[b]); la[b] = lb.area(); } else la[b] = 0;   // area zeroed after an empty bin (Geometry.h:360)
            rs += bc[BINS - 1 - b]; rc[BINS - 2 - b] = rs;
            if (bc[BINS - 1 - b] > 0) { rb.grow(bb[BINS - 1 - b]); ra[BINS - 2 - b] = rb.area(); } else ra[BINS - 2 - b] = 0;
        }
        float bin_w = (cmax - cmin) / BINS;
        for (int b = 0; b < BINS - 1; b++) {
            float cost = 1.0f + ((la[b] / parent_area) * lc[b] * 1.0f) + ((ra[b] / parent_area) * rc[b] * 1.0f);
            if (cost < best) { best = cost; axis = a; pos = cmin + bin_w * (b + 1); }
        }
    }
    splits[i].axis = axis; splits[i].pos = pos;
}
__global__ void k_flags(uint32_t n, const float* __restrict__ verts, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ seg,
                        const int32_t* __restrict__ node_at_start, const GNode* __restrict__ nodes, const Split* __restrict__ splits, uint32_t* flags) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int32_t node;
    int32_t slot = slot_at(p, seg, node_at_start, nodes, node);
    uint32_t f = 0;
    if (slot >= 0 && splits[slot].axis >= 0) f = centroid_of(verts, idx[p], splits[slot].axis) < splits[slot].pos ? 1u : 0u;   // Geometry.h:418
    flags[p] = f;
}
// stable partition of every pending node's range (Geometry.h:417-424); `scan` is the exclusive prefix sum of flags, n + 1 entries
__global__ void k_scatter(uint32_t n, const uint32_t* __restrict__ idx, uint32_t* __restrict__ idx_out, const uint32_t* __restrict__ seg,
                          const int32_t* __restrict__ node_at_start, const GNode* __restrict__ nodes, const Split* __restrict__ splits,
                          const uint32_t* __restrict__ flags, const uint32_t* __restrict__ scan) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int32_t node;
    int32_t slot = slot_at(p, seg, node_at_start, nodes, node);
    uint32_t dst = p;
    if (slot >= 0 && splits[slot].axis >= 0) {
        const GNode& nd = nodes[node];
        uint32_t b = nd.begin, before = scan[p] - scan[b], nl = scan[b + nd.count] - scan[b];
        dst = flags[p] ? b + before : b + nl + ((p - b) - before);
    }
    idx_out[dst] = idx[p];
}
// make the children (Geometry.h:426-437) and queue those that need a split search of their own
__global__ void k_children(uint32_t n_pending, const uint32_t* __restrict__ pending, GNode* nodes, const Split* __restrict__ splits,
                           const uint32_t* __restrict__ scan, uint32_t* counters /* [0] node count, [1] next pending count */, uint32_t* next_pending,
                           uint32_t* start_mark, int32_t* node_at_start) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pending) return;
    uint32_t ni = pending[i];
    GNode nd = nodes[ni];
    nd.slot = -1;
    if (splits[i].axis >= 0) {
        uint32_t nl = scan[nd.begin + nd.count] - scan[nd.begin], nr = nd.count - nl;
        if (nl > 0 && nr > 0) {
            uint32_t l = atomicAdd(&counters[0], 2u), r = l + 1;
            GNode c;
            for (int k = 0; k < 3; k++) { c.lo[k] = FMAX; c.hi[k] = -FMAX; }
            c.left = c.right = -1; c.depth = nd.depth + 1;
            c.begin = nd.begin; c.count = nl; c.slot = -1;
            if (nl > 1) { c.slot = (int32_t)atomicAdd(&counters[1], 1u); next_pending[c.slot] = l; }
            nodes[l] = c;
            c.begin = nd.begin + nl; c.count = nr; c.slot = -1;
            if (nr > 1) { c.slot = (int32_t)atomicAdd(&counters[1], 1u); next_pending[c.slot] = r; }
            nodes[r] = c;
            nd.left = (int32_t)l; nd.right = (int32_t)r;
            node_at_start[nd.begin] = (int32_t)l;
            node_at_start[nd.begin + nl] = (int32_t)r;
            start_mark[nd.begin + nl] = nd.begin + nl + 1u;
        }
    }
    nodes[ni] = nd;
}

// ================================================================================================ after the tree
// Everything else that is per triangle or per node, on the arrays the build left in device memory.

__device__ __forceinline__ float d_dot(const float* a, const float* b) { return ((a[0] * b[0]) + (a[1] * b[1]) + (a[2] * b[2])); }
__device__ __forceinline__ void d_cross(const float* a, const float* b, float* o) {
    o[0] = (a[1] * b[2]) - (a[2] * b[1]); o[1] = (a[2] * b[0]) - (a[0] * b[2]); o[2] = (a[0] * b[1]) - (a[1] * b[0]);
}
// Triangle::init (Geometry.h:66-82) into the 64-B records, one thread per DFS rank; scene bounds (Scene::init, Scene.h:152-158)
// and the largest |coordinate| by integer atomics on order-preserving keys.  Same expressions as scene_build.cpp's host loop.
__global__ void k_records(uint32_t n, const float* __restrict__ verts, const uint32_t* __restrict__ tri_mat, const uint32_t* __restrict__ order,
                          DTri* __restrict__ tris, DTriShade* __restrict__ shade, uint32_t* bounds_keys /* lo xyz, hi xyz, amax */) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t lo[3] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu }, hi[3] = { 0u, 0u, 0u }, am = 0u;
    if (r < n) {
        const uint32_t t = order[r];
        const float* f = verts + (size_t)t * 24;
        float p0[3] = { f[0], f[1], f[2] }, p1[3] = { f[8], f[9], f[10] }, p2[3] = { f[16], f[17], f[18] };
        float e1[3] = { p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2] }, e2[3] = { p0[0] - p2[0], p0[1] - p2[1], p0[2] - p2[2] };
        float c[3]; d_cross(e1, e2, c);
        float l = 1.0f / sqrtf((c[0] * c[0]) + (c[1] * c[1]) + (c[2] * c[2]));
        float nn[3] = { c[0] * l, c[1] * l, c[2] * l };
        DTri o;
        for (int k = 0; k < 3; k++) { o.v0[k] = p0[k]; o.v1[k] = p1[k]; o.v2[k] = p2[k]; o.n[k] = nn[k]; }
        o.d = d_dot(nn, p0);
        o.inv_area = 1.0f / d_dot(c, nn);
        o.orig_id = t;
        float n0[3] = { f[3], f[4], f[5] };
        const bool flip = !(d_dot(n0, nn) > 0);   // gNormal = n * (Dot(v0.normal, n) > 0 ? 1 : -1)
        o.mat_gsign = tri_mat[t] | (flip ? 0x80000000u : 0u);
        tris[r] = o;
        DTriShade sh;
        for (int k = 0; k < 3; k++) { sh.n0[k] = f[3 + k]; sh.n1[k] = f[11 + k]; sh.n2[k] = f[19 + k]; }
        sh.uv[0] = f[6]; sh.uv[1] = f[7]; sh.uv[2] = f[14]; sh.uv[3] = f[15]; sh.uv[4] = f[22]; sh.uv[5] = f[23];
        sh.mat_gsign = o.mat_gsign;
        shade[r] = sh;
        for (int k = 0; k < 3; k++) {
            lo[k] = fkey(fminf(p0[k], fminf(p1[k], p2[k]))); hi[k] = fkey(fmaxf(p0[k], fmaxf(p1[k], p2[k])));
            am = max(am, __float_as_uint(fmaxf(fabsf(p0[k]), fmaxf(fabsf(p1[k]), fabsf(p2[k])))));   // non-negative floats order like their bits
        }
    }
    for (int k = 0; k < 3; k++) { lo[k] = wave_min(lo[k]); hi[k] = wave_max(hi[k]); }
    am = wave_max(am);
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 3; k++) { atomicMin(bounds_keys + k, lo[k]); atomicMax(bounds_keys + 3 + k, hi[k]); }
        atomicMax(bounds_keys + 6, am);
    }
}
// reference-topology nodes for the validation / fallback walk + the tree statistics rt_scene_info reports
__global__ void k_nodes2(uint32_t n_nodes, const GNode* __restrict__ nodes, DNode2* __restrict__ out, uint32_t* stats /* max depth, leaves, max leaf */) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const GNode s = nodes[i];
    DNode2 o;
    for (int k = 0; k < 3; k++) { o.lo[k] = s.lo[k]; o.hi[k] = s.hi[k]; }
    o.left = s.left; o.right = s.right;
    const bool leaf = s.left < 0;
    o.first = leaf ? (int32_t)s.begin : 0; o.count = leaf ? (int32_t)s.count : 0; o.pad0 = (int32_t)s.begin; o.pad1 = (int32_t)s.count;
    out[i] = o;
    atomicMax(stats + 0, s.depth);
    if (leaf) { atomicAdd(stats + 1, 1u); atomicMax(stats + 2, s.count); }
}

struct DevMem {
    std::vector<void*> ptrs;
    ~DevMem() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t alloc(T** p, size_t count) { hipError_t e = hipMalloc((void**)p, sizeof(T) * (count ? count : 1)); if (e == hipSuccess) ptrs.push_back(*p); return e; }
};

}  // namespace

// The binary tree level by level (see the top of this file), then -- still on the device -- the triangle records in leaf
// order and the reference-topology node array.  Only the leaf order (4 B per triangle), a few counters per level and the
// scene statistics come back to the host; what the walk reads is derived from these arrays by walk_tree.hip.
int device_scene_build(const rt_scene_desc& d, HostScene& hs, DeviceGeometry& out, std::string& err_) {
    std::string& err = err_;
    const uint32_t n = d.n_tris;
    if (n < 2) return 2;   // nothing to build in parallel; the host's special cases (empty scene, single-triangle root) apply
    auto t0 = std::chrono::steady_clock::now();
#define BB_CHECK0(call)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) { err = std::string("bvh_build: ") + #call + ": " + hipGetErrorString(e_); return 0; } \
    } while (0)
    uint32_t levels = 0;
    DevMem mem;
    float* verts; uint32_t *idx[2], *start_mark, *seg, *flags, *scan, *acc, *pending[2], *counters; int32_t* node_at_start; GNode* nodes; Split* splits;
    const size_t max_nodes = (size_t)2 * n + 1, max_pending = (size_t)n / 2 + 1;
    BB_CHECK0(mem.alloc(&verts, (size_t)n * 24)); BB_CHECK0(mem.alloc(&idx[0], n)); BB_CHECK0(mem.alloc(&idx[1], n));
    BB_CHECK0(mem.alloc(&start_mark, n)); BB_CHECK0(mem.alloc(&seg, n)); BB_CHECK0(mem.alloc(&flags, (size_t)n + 1)); BB_CHECK0(mem.alloc(&scan, (size_t)n + 1));
    BB_CHECK0(mem.alloc(&node_at_start, n)); BB_CHECK0(mem.alloc(&nodes, max_nodes)); BB_CHECK0(mem.alloc(&splits, max_pending));
    BB_CHECK0(mem.alloc(&pending[0], max_pending)); BB_CHECK0(mem.alloc(&pending[1], max_pending)); BB_CHECK0(mem.alloc(&counters, 2));
    BB_CHECK0(mem.alloc(&acc, max_pending * ACC_WORDS));
    size_t tmp_a = 0, tmp_b = 0;
    BB_CHECK0(hipcub::DeviceScan::InclusiveScan(nullptr, tmp_a, start_mark, seg, hipcub::Max(), (int)n));
    BB_CHECK0(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_b, flags, scan, (int)n + 1));
    void* tmp; size_t tmp_bytes = std::max(tmp_a, tmp_b);
    BB_CHECK0(mem.alloc((char**)&tmp, tmp_bytes));
    BB_CHECK0(hipMemcpy(verts, d.verts, sizeof(float) * (size_t)n * 24, hipMemcpyHostToDevice));
    if (getenv("RT_BUILD_TIMING")) fprintf(stderr, "device build: alloc + upload %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    const unsigned T = 256, gn = (n + T - 1) / T;
    hipLaunchKernelGGL(k_init, dim3(gn), dim3(T), 0, 0, n, idx[0], start_mark, node_at_start, nodes);
    uint32_t h_counters[2] = { 1u, 0u }, zero = 0;
    BB_CHECK0(hipMemcpy(counters, h_counters, sizeof(h_counters), hipMemcpyHostToDevice));
    BB_CHECK0(hipMemcpy(pending[0], &zero, sizeof(uint32_t), hipMemcpyHostToDevice));
    BB_CHECK0(hipMemset(flags + n, 0, sizeof(uint32_t)));
    uint32_t n_pending = n > 1 ? 1u : 0u;
    int cur = 0, pc = 0;
    for (uint32_t level = 0; n_pending > 0; level++) {
        if (level > 4096) { err = "bvh_build: tree deeper than 4096 levels"; return 0; }
        const unsigned gp = (n_pending + T - 1) / T;
        BB_CHECK0(hipcub::DeviceScan::InclusiveScan(tmp, tmp_bytes, start_mark, seg, hipcub::Max(), (int)n));
        hipLaunchKernelGGL(k_acc_reset, dim3((unsigned)(((size_t)n_pending * ACC_WORDS + T - 1) / T)), dim3(T), 0, 0, acc, n_pending);
        hipLaunchKernelGGL(k_bounds, dim3(gn), dim3(T), 0, 0, n, verts, idx[cur], seg, node_at_start, nodes, acc);
        hipLaunchKernelGGL(k_bin, dim3(gn), dim3(T), 0, 0, n, verts, idx[cur], seg, node_at_start, nodes, acc);
        hipLaunchKernelGGL(k_split, dim3(gp), dim3(T), 0, 0, n_pending, pending[pc], nodes, acc, splits);
        hipLaunchKernelGGL(k_flags, dim3(gn), dim3(T), 0, 0, n, verts, idx[cur], seg, node_at_start, nodes, splits, flags);
        BB_CHECK0(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, flags, scan, (int)n + 1));
        hipLaunchKernelGGL(k_scatter, dim3(gn), dim3(T), 0, 0, n, idx[cur], idx[cur ^ 1], seg, node_at_start, nodes, splits, flags, scan);
        BB_CHECK0(hipMemset(counters + 1, 0, sizeof(uint32_t)));
        hipLaunchKernelGGL(k_children, dim3(gp), dim3(T), 0, 0, n_pending, pending[pc], nodes, splits, scan, counters, pending[pc ^ 1], start_mark, node_at_start);
        BB_CHECK0(hipGetLastError());
        BB_CHECK0(hipMemcpy(h_counters, counters, sizeof(h_counters), hipMemcpyDeviceToHost));
        if (h_counters[0] > max_nodes || h_counters[1] > max_pending) { err = "bvh_build: node bookkeeping overflow"; return 0; }
        if (getenv("RT_BUILD_TIMING") && level < 12) fprintf(stderr, "  level %u (%u nodes) done at %.4f s\n", level, n_pending, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        n_pending = h_counters[1];
        cur ^= 1; pc ^= 1; levels = level + 1;
    }
    const uint32_t n_nodes = h_counters[0];
    const uint32_t* order = idx[cur];
    if (getenv("RT_BUILD_TIMING")) fprintf(stderr, "device build: %u levels, binary tree done at %.3f s\n", levels, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());

    // records + scene bounds, reference-topology nodes + statistics
    struct Owned { void* p = nullptr; ~Owned() { if (p) (void)hipFree(p); } void* release() { void* q = p; p = nullptr; return q; } };
    Owned o_tris, o_shade, o_nodes2;
    uint32_t *tri_mat, *keys;
    BB_CHECK0(mem.alloc(&tri_mat, n)); BB_CHECK0(mem.alloc(&keys, 16));
    BB_CHECK0(hipMalloc(&o_tris.p, sizeof(DTri) * (size_t)n)); BB_CHECK0(hipMalloc(&o_shade.p, sizeof(DTriShade) * (size_t)n));
    BB_CHECK0(hipMalloc(&o_nodes2.p, sizeof(DNode2) * (size_t)n_nodes));
    BB_CHECK0(hipMemcpy(tri_mat, d.tri_material, sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice));
    const uint32_t h_keys[16] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, /* stats */ 0u, 0u, 0u };
    BB_CHECK0(hipMemcpy(keys, h_keys, sizeof(h_keys), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_records, dim3(gn), dim3(T), 0, 0, n, verts, tri_mat, order, (DTri*)o_tris.p, (DTriShade*)o_shade.p, keys);
    hipLaunchKernelGGL(k_nodes2, dim3((n_nodes + T - 1) / T), dim3(T), 0, 0, n_nodes, nodes, (DNode2*)o_nodes2.p, keys + 7);
    BB_CHECK0(hipGetLastError());

    // what the host needs to know
    uint32_t r_keys[16];
    BB_CHECK0(hipDeviceSynchronize());
    BB_CHECK0(hipMemcpy(r_keys, keys, sizeof(r_keys), hipMemcpyDeviceToHost));
    hs.leaf_order.resize(n);
    BB_CHECK0(hipMemcpy(hs.leaf_order.data(), order, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
    auto unkey = [](uint32_t k) { uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; float f; memcpy(&f, &b, 4); return f; };
    for (int k = 0; k < 3; k++) { hs.bounds_lo[k] = unkey(r_keys[k]); hs.bounds_hi[k] = unkey(r_keys[3 + k]); }
    memcpy(&hs.abs_max, &r_keys[6], 4);
    hs.depth2 = r_keys[7]; hs.leaves2 = r_keys[8]; hs.max_leaf2 = r_keys[9];
    hs.max_stack2 = hs.depth2 + 2;
    hs.n_tris = n; hs.n_nodes2 = n_nodes;
    out.tris = (DTri*)o_tris.release(); out.trishade = (DTriShade*)o_shade.release();
    out.nodes2 = (DNode2*)o_nodes2.release();
    if (getenv("RT_BUILD_TIMING")) fprintf(stderr, "device build: records + nodes2 done, total %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return 1;
#undef BB_CHECK0
}
