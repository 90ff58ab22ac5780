// walk_tree.hip — the walk's own view of the tree, derived on the device from what the builders leave there.
//
// The builders (scene_build.cpp on the host, bvh_build.hip on the device) produce the triangle records in DFS leaf order
// (`tris`) and the reference-topology binary tree (`nodes2`), the same from both.  The walk is
// only required to be conservative -- rt_settle restores upstream's candidate set per ray from the gate box of the hit it
// reports (rt_device.h) -- so the tree it descends is built for what the traversal kernels are bound
// by, instruction issue: few, wide steps and few divergent loads per ray.  This file makes
//   gate   : per triangle, the box of the last reference node upstream really tests on the way down to it (what rt_settle tests),
//   nodes8 : the 8-wide tree, 80 B per node (DNode8, rt_common.h), level by level from nodes2: a wide node starts as one
//            reference node and cuts slots in two -- a child node into the reference node's children, a leaf slot of 2-3
//            triangles into tighter boxes -- whichever saves most by box area, until eight slots are taken (w8_gather); up to
//            RT_LEAF_MAX triangles make a leaf slot, bounded by its own triangles; children are put into the slot whose
//            octant (relative to the node's centre) they lie in, and numbered consecutively,
//   trisw  : the triangle records in the order nodes8's leaf slots name them, DFS rank inside.
// One implementation serves both builders and reads only the tree's shape, its boxes and the triangles, so device build ==
// host build holds for these arrays by construction (tests/test_bvh_build_gpu.py still compares them).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <math.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "rt_device.h"
#include "scene_dev.h"

// What opening a child node inside its parent is worth against cutting a leaf slot, per unit of box area (w8_gather).
// 2 / 1 / 0.5 / 0.25 on the bench scene: 7.3 / 6.7 / 5.9 / 5.5 triangle tests and 14.5 / 14.6 / 14.9 / 15.4 node steps per ray of
// the frame, 4318 / 4357 / 4407 / 4377 Mrays/s (largest box first, leaf slots cut last: 7.9 / 14.5, 4234).
#ifndef RT_W8_CNODE
#define RT_W8_CNODE 0.5f
#endif
namespace {

// A slot stands for a run of triangles in DFS rank order: everything below a reference node (bnode >= 0), or a part of a
// long leaf (bnode < 0).  Up to RT_LEAF_MAX triangles make a leaf slot; more make a child node.
struct W8Slot { float lo[3], hi[3]; int32_t bnode; uint32_t first, count; };
struct W8Item { int32_t bnode; uint32_t first, count; };

__device__ __forceinline__ float w8_area(const W8Slot& s) {
    const float sx = s.hi[0] - s.lo[0], sy = s.hi[1] - s.lo[1], sz = s.hi[2] - s.lo[2];
    return ((sx * sy) + (sy * sz) + (sx * sz)) * 2.0f;
}
// bounds of the vertices of triangles [first, first + count)
__device__ inline void w8_range_box(const DTri* __restrict__ tris, uint32_t first, uint32_t count, float* lo, float* hi) {
    for (int a = 0; a < 3; a++) { lo[a] = RT_INF; hi[a] = -RT_INF; }
    for (uint32_t t = 0; t < count; t++) {
        const DTri& tr = tris[first + t];
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], fminf(tr.v0[a], fminf(tr.v1[a], tr.v2[a])));
            hi[a] = fmaxf(hi[a], fmaxf(tr.v0[a], fmaxf(tr.v1[a], tr.v2[a])));
        }
    }
}
// What the walk culls with.  A child node: the reference node's box (the bounds of every vertex below it; a long leaf's
// parts: the bounds of their own triangles).  A leaf slot: the bounds of its triangles' vertices -- what a hit needs is
// that the slot's box holds the triangle; its reference box (the gate) can be far larger: a single-triangle leaf has none
// of its own upstream (Geometry.h:401), a one-triangle scene none at all.
__device__ inline W8Slot w8_slot_of_node(const DNode2* __restrict__ n2, const DTri* __restrict__ tris, int32_t node) {
    const DNode2 c = n2[node];
    W8Slot s; s.bnode = node; s.first = (uint32_t)c.pad0; s.count = (uint32_t)c.pad1;
    if (s.count > RT_LEAF_MAX) { for (int a = 0; a < 3; a++) { s.lo[a] = c.lo[a]; s.hi[a] = c.hi[a]; } }
    else w8_range_box(tris, s.first, s.count, s.lo, s.hi);
    return s;
}
__device__ inline W8Slot w8_slot_of_range(const DTri* __restrict__ tris, uint32_t first, uint32_t count) {
    W8Slot s; s.bnode = -1; s.first = first; s.count = count;
    w8_range_box(tris, first, count, s.lo, s.hi);
    return s;
}
// a slot cut in two: a reference node into its children, a long leaf (or a part of one) into halves, a leaf slot that is no
// reference node into its first triangle and the rest
__device__ inline void w8_cut(const DNode2* __restrict__ n2, const DTri* __restrict__ tris, const W8Slot& e, W8Slot& l, W8Slot& r) {
    if (e.bnode >= 0) {
        const DNode2 b = n2[e.bnode];
        if (b.left >= 0 && b.right >= 0) { l = w8_slot_of_node(n2, tris, b.left); r = w8_slot_of_node(n2, tris, b.right); return; }
    }
    const uint32_t h = e.count > RT_LEAF_MAX ? e.count / 2u : 1u;
    l = w8_slot_of_range(tris, e.first, h); r = w8_slot_of_range(tris, e.first + h, e.count - h);
}
// Greedy cuts: slots are cut in two, the cut that saves most first, until eight are taken or no cut saves anything.  Cutting
// a child node (more than RT_LEAF_MAX triangles; only with `nodes`) brings its children up into this node: worth its box
// area * RT_W8_CNODE.  Cutting a leaf slot of 2-3 triangles spares the rays that cross the slot's box but not the boxes of its
// parts their triangle tests: worth area * triangles of the slot minus the same of the parts (every leaf slot is bounded by
// its own triangles' vertices).  Slots keep the left-to-right order of the reference tree.
__device__ int w8_greedy(const DNode2* __restrict__ n2, const DTri* __restrict__ tris, W8Slot* s, int ns, bool nodes) {
    while (ns < 8) {
        int pick = -1; float best = 0.0f;
        for (int i = 0; i < ns; i++) {
            float gain;
            if (s[i].count > RT_LEAF_MAX) {
                if (!nodes) continue;
                gain = w8_area(s[i]) * (float)(RT_W8_CNODE);
                if (!(gain == gain)) gain = RT_INF;            // inf - inf: the item itself
            } else if (s[i].count >= 2u) {
                W8Slot a, b;
                w8_cut(n2, tris, s[i], a, b);
                gain = w8_area(s[i]) * (float)s[i].count - (w8_area(a) * (float)a.count + w8_area(b) * (float)b.count);
            } else continue;
            if (gain > best) { best = gain; pick = i; }
        }
        if (pick < 0) break;   // a child node with a box of no area stays a child: it is the item of its own node, and opened there
        W8Slot l, r;
        w8_cut(n2, tris, s[pick], l, r);
        for (int i = ns; i > pick + 1; i--) s[i] = s[i - 1];
        s[pick] = l; s[pick + 1] = r; ns++;
    }
    return ns;
}

// ---- the collapse plan: which reference nodes become slots of which wide node, chosen to minimise the surface-area cost of
// the walk (Ylitie, Karras & Laine 2017, section 3.1, on this walk's slots).  Bottom-up over the reference tree, per node n and
// slot budget i = 1..7:  C(n, i) = the cheapest way to stand for n's subtree with at most i slots of some wide node,
//   C(n, 1) = area(n) * triangles(n)                              when n holds at most RT_LEAF_MAX triangles: a leaf slot
//           = area(n) * RT_W8_DP_CNODE + min_k C(l, k) + C(r, 8 - k)  else: a child node, its eight slots shared out below
//   C(n, i) = min(C(n, i - 1), min_k C(l, k) + C(r, i - k))        n's two children share the i slots
// in units of one triangle test per unit of box area.  The plan word of a node keeps the argmins: for j = 2..8 slots shared out,
// how many go left (3 bits each), and for a budget of i = 2..7 how many of them are really used (3 bits each).
#ifndef RT_W8_DP
#define RT_W8_DP 1
#endif
#ifndef RT_W8_DP_CNODE
#define RT_W8_DP_CNODE 1.5f   // a node step in triangle tests.  1 / 1.5 / 2 / 3 on the bench scene: 14.8 / 14.6 / 14.4 / 14.1 node steps and 4.7 / 5.0 / 5.4 / 6.1 triangle tests per ray of the frame, 4599 / 4585 / 4574 / 4519 Mrays/s (greedy cuts alone: 14.9 / 5.9, 4419)
#endif
struct W8Cost { float c[7]; };
__device__ __forceinline__ uint32_t w8_plan_left(unsigned long long pl, uint32_t j) { return (uint32_t)(pl >> (3u * (j - 2u))) & 7u; }
__device__ __forceinline__ uint32_t w8_plan_used(unsigned long long pl, uint32_t i) { return ((uint32_t)(pl >> (21u + 3u * (i - 2u))) & 7u) + 1u; }

__global__ void k8_parents(uint32_t n, const DNode2* __restrict__ n2, int32_t* __restrict__ parent) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0u) parent[0] = -1;
    const DNode2 b = n2[i];
    if (b.left >= 0 && b.right >= 0) { parent[b.left] = (int32_t)i; parent[b.right] = (int32_t)i; }
}
// One thread per leaf of the reference tree climbs towards the root; of the two threads that reach a node the second one
// finds both children done and goes on (the first one stops), so every inner node is computed once, after its children.
__global__ void k8_plan(uint32_t n, const DNode2* __restrict__ n2, const DTri* __restrict__ tris, const int32_t* __restrict__ parent,
                        uint32_t* __restrict__ arrive, W8Cost* cost, unsigned long long* __restrict__ plan) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    {
        const DNode2 b = n2[i];
        if (b.left >= 0 && b.right >= 0) return;
        W8Slot s = w8_slot_of_range(tris, (uint32_t)b.pad0, (uint32_t)b.pad1);
        float c = w8_area(s) * (float)s.count;
        if (s.count > RT_LEAF_MAX) c = (w8_area(s) * (float)(RT_W8_DP_CNODE)) + (c * 0.5f);   // a long leaf: a child node of halves
        W8Cost o;
        for (int k = 0; k < 7; k++) o.c[k] = c;
        cost[i] = o;
    }
    int32_t node = (int32_t)i;
    for (;;) {
        const int32_t p = parent[node];
        if (p < 0) break;
        __threadfence();
        if (atomicAdd(&arrive[p], 1u) == 0u) break;
        __threadfence();
        const DNode2 b = n2[p];
        const W8Cost cl = cost[b.left], cr = cost[b.right];
        float D[9]; uint32_t K[9];
        for (uint32_t j = 2; j <= 8; j++) {
            D[j] = RT_INF; K[j] = 1u;
            for (uint32_t k = (j > 7u ? j - 7u : 1u); k <= 7u && k < j; k++) {
                const float c = cl.c[k - 1] + cr.c[j - k - 1];
                if (c < D[j]) { D[j] = c; K[j] = k; }
            }
        }
        const float sx = b.hi[0] - b.lo[0], sy = b.hi[1] - b.lo[1], sz = b.hi[2] - b.lo[2];
        const float area = ((sx * sy) + (sy * sz) + (sx * sz)) * 2.0f;
        const uint32_t cnt = (uint32_t)b.pad1;
        W8Cost o;
        uint32_t used = 1u;
        unsigned long long pl = 0ULL;
        o.c[0] = cnt <= RT_LEAF_MAX ? area * (float)cnt : (area * (float)(RT_W8_DP_CNODE)) + D[8];
        for (uint32_t j = 2; j <= 8; j++) pl |= (unsigned long long)K[j] << (3u * (j - 2u));
        for (uint32_t k = 2; k <= 7; k++) {
            if (D[k] < o.c[k - 2]) { o.c[k - 1] = D[k]; used = k; } else o.c[k - 1] = o.c[k - 2];
            pl |= (unsigned long long)(used - 1u) << (21u + 3u * (k - 2u));
        }
        cost[p] = o; plan[p] = pl;
        node = p;
    }
}

// The up to eight slots of the wide node that stands for `item`: by the plan where the item is an inner node of the reference
// tree (its eight slots are shared out between its children as the plan says, down to the nodes that take one slot), else --
// long leaves, parts of them, scenes of a few triangles -- by the greedy cuts alone.  Slots that the plan leaves free go to
// cuts of leaf slots that save something.  Both passes of a level call this and must decide alike: it reads only the tree,
// the triangles and the plan.
__device__ int w8_gather(const DNode2* __restrict__ n2, const DTri* __restrict__ tris, const unsigned long long* __restrict__ plan,
                         const W8Item& item, W8Slot* s) {
    if (RT_W8_DP && plan && item.bnode >= 0 && item.count > RT_LEAF_MAX) {
        const DNode2 b = n2[item.bnode];
        if (b.left >= 0 && b.right >= 0) {
            int32_t todo_node[8]; uint32_t todo_slots[8];
            int sp = 0, ns = 0;
            const uint32_t k8 = w8_plan_left(plan[item.bnode], 8u);
            todo_node[sp] = b.right; todo_slots[sp++] = 8u - k8;
            todo_node[sp] = b.left; todo_slots[sp++] = k8;
            while (sp > 0) {
                const int32_t node = todo_node[--sp];
                const uint32_t budget = todo_slots[sp];
                const DNode2 c = n2[node];
                uint32_t used = 1u;
                unsigned long long pl = 0ULL;
                if (budget >= 2u && c.left >= 0 && c.right >= 0) { pl = plan[node]; used = w8_plan_used(pl, budget); }
                if (used < 2u) { s[ns++] = w8_slot_of_node(n2, tris, node); continue; }
                const uint32_t k = w8_plan_left(pl, used);
                todo_node[sp] = c.right; todo_slots[sp++] = used - k;
                todo_node[sp] = c.left; todo_slots[sp++] = k;
            }
            return w8_greedy(n2, tris, s, ns, false);
        }
    }
    int ns = 1;
    s[0].bnode = item.bnode; s[0].first = item.first; s[0].count = item.count;
    for (int a = 0; a < 3; a++) { s[0].lo[a] = -RT_INF; s[0].hi[a] = RT_INF; }   // the item itself is always opened first
    if (item.count <= RT_LEAF_MAX) s[0] = w8_slot_of_range(tris, item.first, item.count);   // a scene of 1-3 triangles
    return w8_greedy(n2, tris, s, ns, true);
}

// pass 1 of a level: children that become wide nodes of the next level, and leaf triangles, per wide node
__global__ void k8_count(uint32_t n_items, const DNode2* __restrict__ n2, const DTri* __restrict__ tris, const unsigned long long* __restrict__ plan,
                         const W8Item* __restrict__ items, uint32_t* __restrict__ cnt_inner, uint32_t* __restrict__ cnt_tris) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_items) return;
    if (i == n_items) { cnt_inner[i] = 0u; cnt_tris[i] = 0u; return; }   // the scans' extra entry: its exclusive sum is the level's total
    W8Slot s[8];
    const int ns = w8_gather(n2, tris, plan, items[i], s);
    uint32_t ci = 0, ct = 0;
    for (int k = 0; k < ns; k++) {
        if (s[k].count <= RT_LEAF_MAX) ct += s[k].count; else ci++;
    }
    cnt_inner[i] = ci; cnt_tris[i] = ct;
}

// pass 2: the level's nodes, the next level's work items (in slot order, so that child_base + rank finds them) and the
// level's stretch of trisw.  The grid arithmetic is done in double: a plane that lands a rounding error (2^-52 of the
// node's extent) inside the box it should contain is covered many times over by the walk's slack (2^-17 of the scene's extent).
__global__ void k8_emit(uint32_t n_items, uint32_t level_base, uint32_t next_base, uint32_t tri_level_base, const DNode2* __restrict__ n2,
                        const DTri* __restrict__ tris, const unsigned long long* __restrict__ plan, const W8Item* __restrict__ items, const uint32_t* __restrict__ scan_inner,
                        const uint32_t* __restrict__ scan_tris, DNode8* __restrict__ out, W8Item* __restrict__ next_items, DTri* __restrict__ trisw,
                        uint32_t* __restrict__ shape /* per node: inner children | leaf slots << 8 */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    W8Slot s[8];
    const int ns = w8_gather(n2, tris, plan, items[i], s);
    float nlo[3] = { RT_INF, RT_INF, RT_INF }, nhi[3] = { -RT_INF, -RT_INF, -RT_INF };
    for (int k = 0; k < ns; k++)
        for (int a = 0; a < 3; a++) { nlo[a] = fminf(nlo[a], s[k].lo[a]); nhi[a] = fmaxf(nhi[a], s[k].hi[a]); }
    // children into the slot whose octant their centre lies in: greedy assignment by the projection of (child centre -
    // node centre) on the slot's diagonal, the largest first (first of equals: lowest child, then lowest slot)
    int slot_of[8], child_in[8];
    for (int k = 0; k < 8; k++) { slot_of[k] = -1; child_in[k] = -1; }
    {
        float c[8][3];
        for (int k = 0; k < ns; k++)
            for (int a = 0; a < 3; a++) c[k][a] = ((s[k].lo[a] * 0.5f) + (s[k].hi[a] * 0.5f)) - ((nlo[a] * 0.5f) + (nhi[a] * 0.5f));
        for (int round = 0; round < ns; round++) {
            int bk = -1, bs = -1; float best = -RT_INF;
            for (int k = 0; k < ns; k++) {
                if (slot_of[k] >= 0) continue;
                for (int sl = 0; sl < 8; sl++) {
                    if (child_in[sl] >= 0) continue;
                    float v = ((sl & 1) ? c[k][0] : -c[k][0]) + ((sl & 2) ? c[k][1] : -c[k][1]) + ((sl & 4) ? c[k][2] : -c[k][2]);
                    if (!(v == v)) v = 0.0f;   // inf - inf on a scene of +-FLT_MAX boxes: any slot will do
                    if (bk < 0 || v > best) { best = v; bk = k; bs = sl; }
                }
            }
            slot_of[bk] = bs; child_in[bs] = bk;
        }
    }
    DNode8 q;
    uint32_t wlo[3][2], whi[3][2], exps = 0u;
    for (int a = 0; a < 3; a++) {
        float org = nlo[a], top = nhi[a];
        if (!(org <= top)) { org = 0.0f; top = 0.0f; }   // a node without children (never made)
        const double ext = (double)top - (double)org;
        int e = 0;
        bool have = false;
        if (ext > 0.0 && ext < (double)RT_INF) {
            (void)frexp(ext / 255.0, &e);                  // ext / 255 = m * 2^e, m in [0.5, 1)
            if (e < -126) e = -126;                         // keep the step a normal float
            have = true;
        } else if (ext > 0.0) { e = 127; have = true; }     // an extent beyond FLT_MAX: the largest step there is
        for (;;) {
            bool fits = true;
            uint32_t l[2] = { 0u, 0u }, h[2] = { 0u, 0u };
            const double st = ldexp(1.0, e);
            for (int sl = 0; sl < 8; sl++) {
                uint32_t ql = 255u, qh = 0u;                // unused slot: an inverted box (its meta byte is 0 as well)
                const int k = child_in[sl];
                if (k >= 0 && have) {
                    const double fl = floor(((double)s[k].lo[a] - (double)org) / st), ce = ceil(((double)s[k].hi[a] - (double)org) / st);
                    if (ce > 255.0 && e < 127) { fits = false; break; }
                    ql = fl < 0.0 ? 0u : (fl > 255.0 ? 255u : (uint32_t)fl); qh = ce > 255.0 ? 255u : (ce < 0.0 ? 0u : (uint32_t)ce);
                } else if (k >= 0) { ql = 0u; qh = 0u; }
                l[sl >> 2] |= ql << (8 * (sl & 3)); h[sl >> 2] |= qh << (8 * (sl & 3));
            }
            if (fits) { wlo[a][0] = l[0]; wlo[a][1] = l[1]; whi[a][0] = h[0]; whi[a][1] = h[1]; break; }
            e++;
        }
        q.org[a] = org;
        exps |= ((uint32_t)e & 255u) << (8 * a);
    }
    // slots in ascending order: inner children take consecutive node numbers, leaf triangles consecutive places in trisw
    uint32_t imask = 0u, meta[2] = { 0u, 0u }, n_inner = 0u, n_leaf = 0u, off = 0u;
    const uint32_t child_base = next_base + scan_inner[i], tri_base = tri_level_base + scan_tris[i];
    for (int sl = 0; sl < 8; sl++) {
        const int k = child_in[sl];
        if (k < 0) continue;
        uint32_t m;
        if (s[k].count > RT_LEAF_MAX) {
            imask |= 1u << sl;
            m = 0x20u | (24u + (uint32_t)sl);
            W8Item it; it.bnode = s[k].bnode; it.first = s[k].first; it.count = s[k].count;
            next_items[scan_inner[i] + n_inner++] = it;
        } else {
            const uint32_t first = s[k].first, cnt = s[k].count;
            m = (((1u << cnt) - 1u) << 5) | off;
            for (uint32_t t = 0; t < cnt; t++) {
                DTri r = tris[first + t];
                r.orig_id = first + t;   // the DFS rank: what the walk reports and breaks ties with
                trisw[tri_base + off + t] = r;
            }
            off += cnt; n_leaf++;
        }
        meta[sl >> 2] |= m << (8 * (sl & 3));
    }
    q.exp_imask = exps | (imask << 24);
    q.child_base = child_base; q.tri_base = tri_base; q.meta[0] = meta[0]; q.meta[1] = meta[1];
    q.lox[0] = wlo[0][0]; q.lox[1] = wlo[0][1]; q.loy[0] = wlo[1][0]; q.loy[1] = wlo[1][1]; q.loz[0] = wlo[2][0]; q.loz[1] = wlo[2][1];
    q.hix[0] = whi[0][0]; q.hix[1] = whi[0][1]; q.hiy[0] = whi[1][0]; q.hiy[1] = whi[1][1]; q.hiz[0] = whi[2][0]; q.hiz[1] = whi[2][1];
    out[level_base + i] = q;
    shape[level_base + i] = n_inner | (n_leaf << 8);
}

// Entries a walk can have waiting while it is below wide node i: the rest of the node's inner children (one entry, if there
// are two or more) and the node's triangles (one entry, if the lane already holds postponed triangles), plus the worst of
// what a child needs.  Levels last to first.
__global__ void k8_stack_need(uint32_t level_base, uint32_t n_items, const DNode8* __restrict__ nodes, const uint32_t* __restrict__ shape,
                              uint32_t* __restrict__ need) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const uint32_t sh = shape[level_base + i], n_inner = sh & 255u, n_leaf = sh >> 8;
    uint32_t worst = 0;
    for (uint32_t k = 0; k < n_inner; k++) worst = max(worst, need[nodes[level_base + i].child_base + k]);
    need[level_base + i] = (n_inner >= 2u ? 1u : 0u) + (n_leaf ? 1u : 0u) + worst;
}

// gate: per triangle, the reference box whose rayAABB decides whether upstream tests that triangle at all -- the last box on
// the way down to it that BVHNode::traverse really tests (Geometry.h:441-471): its leaf's own box, or, for a leaf of one
// triangle, whose box upstream never initialises (Geometry.h:401: an inverted box passes every ray), the parent's; a scene
// of one triangle has no tested box at all.  One thread per reference node, writing the gates of its leaf children.
__global__ void k_gate(const DNode2* __restrict__ n2, uint32_t n, DGate* __restrict__ gate) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DNode2 b = n2[i];
    const bool leaf = b.left < 0 && b.right < 0;
    if (leaf && i != 0u) return;            // written by its parent
    DGate own;
    own.lo[0] = b.lo[0]; own.lo[1] = b.lo[1]; own.lo[2] = b.lo[2]; own.pad0 = 0.0f;
    own.hi[0] = b.hi[0]; own.hi[1] = b.hi[1]; own.hi[2] = b.hi[2]; own.pad1 = 0.0f;
    if (leaf) {                              // the root is the only node
        if (b.count < 2) { for (int a = 0; a < 3; a++) { own.lo[a] = -RT_FLT_MAX; own.hi[a] = RT_FLT_MAX; } }
        for (int32_t t = 0; t < b.count; t++) gate[b.first + t] = own;
        return;
    }
    const int32_t kids[2] = { b.left, b.right };
    for (int k = 0; k < 2; k++) {
        if (kids[k] < 0) continue;
        const DNode2 c = n2[kids[k]];
        if (!(c.left < 0 && c.right < 0)) continue;
        DGate g = own;
        if (c.count >= 2) { g.lo[0] = c.lo[0]; g.lo[1] = c.lo[1]; g.lo[2] = c.lo[2]; g.hi[0] = c.hi[0]; g.hi[1] = c.hi[1]; g.hi[2] = c.hi[2]; }
        for (int32_t t = 0; t < c.count; t++) gate[c.first + t] = g;
    }
}

struct Tmp {
    std::vector<void*> ptrs;
    ~Tmp() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t alloc(T** p, size_t count) { hipError_t e = hipMalloc((void**)p, sizeof(T) * (count ? count : 1)); if (e == hipSuccess) ptrs.push_back(*p); return e; }
};

}  // namespace

// Fills s->d.gate / nodes8 / trisw / n_nodes8 / walk_stack from s->d.nodes2 and s->d.tris (both already in device memory).
int rt_build_walk_tree(rt_scene* s) {
    HostScene& h = s->host;
    DScene& d = s->d;
    const uint32_t nn2 = h.n_nodes2, nt = h.n_tris;
    auto keep = [&](void** p, size_t bytes) -> int {
        if (hipMalloc(p, bytes ? bytes : 16) != hipSuccess) { rt_set_error("rt_scene_create: out of device memory (walk tree)"); return RT_ERR_NOMEM; }
        s->allocs.push_back(*p); s->device_bytes += bytes;
        return RT_OK;
    };
    void *p_gate = nullptr, *p_trisw = nullptr, *p_nodes8 = nullptr;
    int rc;
    if ((rc = keep(&p_gate, sizeof(DGate) * (size_t)nt)) || (rc = keep(&p_trisw, sizeof(DTri) * (size_t)nt))) return rc;
    d.gate = (const DGate*)p_gate; d.trisw = (const DTri*)p_trisw;
    h.n_nodes8 = 0; h.walk_stack = 0;
    if (!nn2 || !nt) {
        if ((rc = keep(&p_nodes8, sizeof(DNode8)))) return rc;
        d.nodes8 = (const DNode8*)p_nodes8; d.n_nodes8 = 0; d.walk_stack = 0;
        return RT_OK;
    }
    const unsigned T = 256;
    hipLaunchKernelGGL(k_gate, dim3((nn2 + T - 1) / T), dim3(T), 0, 0, d.nodes2, nn2, (DGate*)p_gate);
    RT_HIP(hipGetLastError());

    Tmp tmp;
    // a wide node that is not the root holds more than RT_LEAF_MAX triangles, none of them shared with a node of its level:
    // fewer than nt nodes per level, and (every node but the root has a parent with >= 2 slots) fewer than nt in all
    const size_t cap = (size_t)nt + 1;
    DNode8* wide; W8Item* items[2]; uint32_t *cnt_inner, *cnt_tris, *scan_inner, *scan_tris, *shape, *need;
    RT_HIP(tmp.alloc(&wide, cap)); RT_HIP(tmp.alloc(&items[0], cap)); RT_HIP(tmp.alloc(&items[1], cap));
    RT_HIP(tmp.alloc(&cnt_inner, cap + 1)); RT_HIP(tmp.alloc(&cnt_tris, cap + 1));
    RT_HIP(tmp.alloc(&scan_inner, cap + 1)); RT_HIP(tmp.alloc(&scan_tris, cap + 1));
    RT_HIP(tmp.alloc(&shape, cap)); RT_HIP(tmp.alloc(&need, cap));
    size_t scan_bytes = 0;
    RT_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, cnt_inner, scan_inner, (int)cap + 1));
    char* scan_tmp; RT_HIP(tmp.alloc(&scan_tmp, scan_bytes));
    // the collapse plan (temporary: the level loop below reads it, the walk never does)
    unsigned long long* plan = nullptr;
    if (RT_W8_DP) {
        int32_t* parent; uint32_t* arrive; W8Cost* cost;
        RT_HIP(tmp.alloc(&parent, nn2)); RT_HIP(tmp.alloc(&arrive, nn2)); RT_HIP(tmp.alloc(&cost, nn2)); RT_HIP(tmp.alloc(&plan, nn2));
        RT_HIP(hipMemset(arrive, 0, sizeof(uint32_t) * (size_t)nn2));
        RT_HIP(hipMemset(plan, 0, sizeof(unsigned long long) * (size_t)nn2));
        hipLaunchKernelGGL(k8_parents, dim3((nn2 + T - 1) / T), dim3(T), 0, 0, nn2, d.nodes2, parent);
        hipLaunchKernelGGL(k8_plan, dim3((nn2 + T - 1) / T), dim3(T), 0, 0, nn2, d.nodes2, d.tris, parent, arrive, cost, plan);
        RT_HIP(hipGetLastError());
    }
    const W8Item root{ 0, 0u, nt };   // level 0: the wide node that stands for the reference tree's root
    RT_HIP(hipMemcpy(items[0], &root, sizeof(root), hipMemcpyHostToDevice));
    std::vector<std::pair<uint32_t, uint32_t>> level_of;   // (first node, nodes) per level
    uint32_t level_base = 0, n_items = 1, tri_total = 0;
    int wc = 0;
    while (n_items > 0) {
        if ((size_t)level_base + n_items > cap) { rt_set_error("rt_scene_create: 8-wide tree bookkeeping overflow"); return RT_ERR_DEVICE; }
        hipLaunchKernelGGL(k8_count, dim3((n_items + 1 + T - 1) / T), dim3(T), 0, 0, n_items, d.nodes2, d.tris, plan, items[wc], cnt_inner, cnt_tris);
        RT_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, cnt_inner, scan_inner, (int)n_items + 1));
        RT_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, cnt_tris, scan_tris, (int)n_items + 1));
        uint32_t n_next = 0, n_level_tris = 0;
        RT_HIP(hipMemcpy(&n_next, scan_inner + n_items, sizeof(uint32_t), hipMemcpyDeviceToHost));
        RT_HIP(hipMemcpy(&n_level_tris, scan_tris + n_items, sizeof(uint32_t), hipMemcpyDeviceToHost));
        const uint32_t next_base = level_base + n_items;
        if ((size_t)next_base + n_next > cap || (size_t)tri_total + n_level_tris > nt) { rt_set_error("rt_scene_create: 8-wide tree bookkeeping overflow"); return RT_ERR_DEVICE; }
        hipLaunchKernelGGL(k8_emit, dim3((n_items + T - 1) / T), dim3(T), 0, 0, n_items, level_base, next_base, tri_total, d.nodes2, d.tris, plan, items[wc],
                           scan_inner, scan_tris, wide, items[wc ^ 1], (DTri*)p_trisw, shape);
        RT_HIP(hipGetLastError());
        level_of.emplace_back(level_base, n_items);
        level_base = next_base; n_items = n_next; tri_total += n_level_tris; wc ^= 1;
    }
    const uint32_t n8 = level_base;
    if (tri_total != nt) { rt_set_error("rt_scene_create: 8-wide tree does not hold every triangle exactly once"); return RT_ERR_DEVICE; }
    if (n8 > (1u << 25)) { rt_set_error("wide BVH exceeds 2^25 nodes (32-bit node offsets)"); return RT_ERR_INVALID; }
    for (size_t l = level_of.size(); l-- > 0;)
        hipLaunchKernelGGL(k8_stack_need, dim3((level_of[l].second + T - 1) / T), dim3(T), 0, 0, level_of[l].first, level_of[l].second, wide, shape, need);
    RT_HIP(hipGetLastError());
    if ((rc = keep(&p_nodes8, sizeof(DNode8) * (size_t)n8))) return rc;
    RT_HIP(hipMemcpy(p_nodes8, wide, sizeof(DNode8) * (size_t)n8, hipMemcpyDeviceToDevice));
    uint32_t r_need = 0;
    RT_HIP(hipMemcpy(&r_need, need, sizeof(uint32_t), hipMemcpyDeviceToHost));
    RT_HIP(hipDeviceSynchronize());
    d.nodes8 = (const DNode8*)p_nodes8; d.n_nodes8 = n8; d.walk_stack = r_need + 2u;
    h.n_nodes8 = n8; h.walk_stack = d.walk_stack; h.walk_levels = (uint32_t)level_of.size();
    return RT_OK;
}
