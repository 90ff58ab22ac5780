// rt_common.h — data layouts shared by the host builder and the gfx950 kernels.
//
// HBM layout (all arrays live for the lifetime of an rt_scene):
//   tris     : 64 B / triangle, 64-B aligned, stored in the DFS leaf order of the
//              reference tree, so "first visited wins ties" (Geometry.h:449) becomes
//              "lowest index wins".  One lane fetches one record as 4 x dwordx4.
//   trishade : 64 B / triangle (vertex normals, uvs, area), same order; only read at a hit.
//   nodes4   : 128 B (one L2 line) per 4-wide node: SoA child boxes + child refs.  What the builders produce; the walk reads
//   nodesq   : 64 B per 4-wide node, the same tree with the child boxes on a 256-step grid of the node's own box, rounded
//              outwards.  A lane fetches a node as 4 x dwordx4 instead of 7: the traversal kernels are bound by the number of
//              divergent load instructions per step, not by arithmetic.
//   gate     : 32 B per triangle: the reference box whose rayAABB decides whether upstream tests that triangle at all
//              (its leaf's box, or the parent's for a single-triangle leaf); read once per ray, for the hit it reports.
//   nodes2   : 48 B per node of the reference-topology binary tree (validation / fallback walk).
#pragma once
#include <stdint.h>

#define RT_EPSILON 0.001f   // Geometry.h:49
#define RT_MAX_DEPTH 4      // Renderer.h:18

// child reference encoding of the 4-wide tree
#define RT_REF_EMPTY 0xFFFFFFFFu
#define RT_REF_LEAF 0x80000000u
// Triangles per leaf slot of the wide tree.  The reference tree's leaves hold 1-2 triangles except where a split found every
// centroid on one side (Geometry.h:426): on the bench scene some twenty leaves of 8-16 large triangles (walls, floor) whose
// box nearly every ray crosses.  Longer leaves are cut into small subtrees of slots, each bounded by its own triangles:
// 16 -> 4 took the triangle tests per ray from 9.4 to 6.6 and added one node visit (23.3 -> 24.3): +5 % Mrays/s (3, 6, 8 the same).
#ifndef RT_LEAF_MAX
#define RT_LEAF_MAX 4u
#endif
#define RT_LEAF_FIRST_MASK 0x07FFFFFFu
static inline uint32_t rt_make_leaf_ref(uint32_t first, uint32_t count) { return RT_REF_LEAF | ((count - 1u) << 27) | first; }

struct DTri {        // intersection record (Triangle::rayIntersect reads n, d, e1, e2, v1, v2: Geometry.h:115-128)
    float v0[3], v1[3], v2[3];
    float n[3];
    float d;
    float inv_area;  // 1.0f / Dot(e1.cross(e2), n), hoisted out of the per-ray call (same float ops, same bits)
    uint32_t orig_id;   // index in the caller's triangle array (== triangleMap[tri], Geometry.h:454)
    uint32_t mat_gsign; // material index | (gNormal flipped ? 1u<<31 : 0)   (Geometry.h:175)
};
struct DTriShade {   // shading record (Triangle::interpolateAttributes / sample, Geometry.h:154-174)
    float n0[3], n1[3], n2[3];
    float uv[6];
    float area;
};
struct DNode4 {      // 128 B
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    uint32_t child[4];
    uint32_t pad[4];
};
struct DNodeQ {      // 64 B: child k's box is org + q * scale per axis, q = byte k of the plane words (lo rounded down, hi up)
    float org[3]; float scale_x;
    float scale_y, scale_z; uint32_t lox, loy;
    uint32_t loz, hix, hiy, hiz;
    uint32_t child[4];   // as DNode4::child; the walk tells unused slots by RT_REF_EMPTY here
};
struct DGate { float lo[3], pad0, hi[3], pad1; };   // 32 B
struct DNode2 {      // 48 B, reference-topology node (BVHNode, Geometry.h:290): leaf <=> left < 0 && right < 0
    float lo[3]; int32_t left;
    float hi[3]; int32_t right;
    int32_t first, count, pad0, pad1;
};
static_assert(sizeof(DTri) == 64 && sizeof(DTriShade) == 64 && sizeof(DNode4) == 128 && sizeof(DNode2) == 48 && sizeof(DNodeQ) == 64 && sizeof(DGate) == 32, "layout");

struct DMaterial {   // 80 B
    uint32_t type;       // RT_MAT_*
    int32_t tex;
    uint32_t flags;      // bit0 two-sided, bit1 pure specular, bit2 is light
    float alpha;         // 1.62142f * sqrtf(roughness)
    float emission[3];
    float int_ior;
    float ext_ior;
    float eta[3];
    float k[3];
    float phong_e;       // PlasticBSDF::alphaToPhongExponent (Materials.h:857)
    float pad[3];
};
#define RT_MF_TWOSIDED 1u
#define RT_MF_SPECULAR 2u
#define RT_MF_LIGHT 4u

struct DTexture { uint32_t offset; int32_t width, height; int32_t pad; };  // offset in floats into the texel pool

struct DLight { uint32_t kind; uint32_t tri; };  // kind 0 area (tri = DFS rank), 1 background colour, 2 environment map

struct DCamera {     // Camera (Scene.h:14-45)
    float P[16], invP[16], cam[16], cam2view[16];
    float width, height;
    float origin[3], view_dir[3];
    float Afilm;
    int32_t iw, ih;
};

struct DScene {
    const DTri* tris;
    const DTriShade* trishade;
    const DNode4* nodes4;
    const DNodeQ* nodesq;
    const DGate* gate;
    const DNode2* nodes2;
    const DMaterial* mats;
    const DTexture* tex;
    const float* texels;
    const DLight* lights;
    uint32_t n_tris, n_lights, n_nodes4, n_nodes2;
    uint32_t root_ref;        // RT_REF_EMPTY for an empty scene, else node 0
    uint32_t max_stack4, max_stack2;
    float abs_max;            // max |coordinate| over the scene, scales the conservative cull pad
    float bg_colour[3];
    uint32_t bg_kind; int32_t bg_tex;
    float scene_centre[3], scene_radius;   // SceneBounds (Lights.h:12, SceneLoader.h:286-287)
    const float* env_cond;    // EnvironmentMap's tables (Lights.h:207-253); null without an environment map
    const float* env_marg;
    float env_func_int; int32_t env_nu, env_nv;
    DCamera cam;
};

struct DCounters { unsigned long long closest, any, nodes, tris, exact, any_nodes, any_tris, slots, busy, any_slots, any_busy, max_steps, any_max_steps, refused, any_refused; };
