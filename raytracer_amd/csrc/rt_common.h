// rt_common.h — data layouts shared by the host builder and the gfx950 kernels.
//
// HBM layout (all arrays live for the lifetime of an rt_scene):
//   tris     : 64 B / triangle, 64-B aligned, stored in the DFS leaf order of the
//              reference tree, so "first visited wins ties" (Geometry.h:449) becomes
//              "lowest index wins".  One lane fetches one record as 4 x dwordx4.
//   trishade : 64 B / triangle (vertex normals, uvs, material word), same order; only read at a hit.
//   nodes8   : 80 B per 8-wide node, what the walk reads: derived on the device from nodes2 (walk_tree.hip) by opening the child
//              with the largest box until eight slots are taken.  Child boxes sit on a 256-step grid of the node's
//              own box (one byte per plane, rounded outwards), children are numbered consecutively (child_base + rank among
//              the inner slots) and so are the node's leaf triangles (tri_base + offset in trisw), so a node needs no child
//              references: 5 x dwordx4 per 8 children.  The traversal kernels are bound by the number of divergent load
//              instructions per ray; the 4-wide nodes of round 2 took 4 loads per 4 children.
//   trisw    : the 64-B triangle records once more, in the order the 8-wide tree's leaves name them (level, node, slot); the
//              word that is orig_id in `tris` holds the triangle's DFS rank, which the walk reports and breaks ties with.
//   gate     : 32 B per triangle: the reference box whose rayAABB decides whether upstream tests that triangle at all
//              (its leaf's box, or the parent's for a single-triangle leaf); read once per ray, for the hit it reports.
//   nodes2   : 48 B per node of the reference-topology binary tree (validation / fallback walk).
#pragma once
#include <stdint.h>

#define RT_EPSILON 0.001f   // Geometry.h:49
#define RT_MAX_DEPTH 4      // Renderer.h:18

#define RT_REF_EMPTY 0xFFFFFFFFu   // DScene::root_ref of an empty scene
// Triangles per leaf slot of the walk tree.  The reference tree's leaves hold 1-2 triangles except where a split found every
// centroid on one side (Geometry.h:426): on the bench scene some twenty leaves of 8-16 large triangles (walls, floor) whose
// box nearly every ray crosses.  Longer runs are cut into small subtrees of slots, each bounded by its own triangles.
// At most 3: an 8-wide walk node names a leaf slot's triangles by three mask bits (DNode8::meta).
#ifndef RT_LEAF_MAX
#define RT_LEAF_MAX 3u
#endif
static_assert(RT_LEAF_MAX >= 1u && RT_LEAF_MAX <= 3u, "a leaf slot holds 1-3 triangles: DNode8::meta has three bits for them");
#define RT_MAX_TRIS 0x07FFFFFFu    // 2^27 - 1 triangles per scene

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
    uint32_t mat_gsign;   // DTri::mat_gsign again: a shading point needs the material and nothing else of the intersection record, so it reads this line alone
};
// 80 B, five 16-B rows.  Slot s's box is org + q * 2^e per axis, q = byte s of the two plane words (lo rounded down, hi up).
// meta byte of slot s: 0 = unused; inner child: 0x20 | (24 + s); leaf: (unary triangle count 1 / 3 / 7) << 5 | offset, the
// slot's triangles being trisw[tri_base + offset ...].  A slot that passes the box test ORs (meta >> 5) << (meta & 31) into
// the node's hit mask -- bits 24-31 inner children, bits 0-23 triangles -- after the low three bits of an inner child's
// index were XORed with the ray's octant: children sit in the slot whose octant their centre lies in (as far as a greedy
// assignment manages), so "highest bit first" visits them roughly front to back without a sort (the compressed wide BVH of
// Ylitie, Karras & Laine 2017).  Inner child in slot s = node child_base + popcount(imask & ((1 << s) - 1)).
struct DNode8 {
    float org[3]; uint32_t exp_imask;    // int8 ex | int8 ey << 8 | int8 ez << 16 | imask << 24
    uint32_t child_base, tri_base, meta[2];
    uint32_t lox[2], loy[2];
    uint32_t loz[2], hix[2];
    uint32_t hiy[2], hiz[2];
};
struct DGate { float lo[3], pad0, hi[3], pad1; };   // 32 B
struct DNode2 {      // 48 B, reference-topology node (BVHNode, Geometry.h:290): leaf <=> left < 0 && right < 0
    float lo[3]; int32_t left;
    float hi[3]; int32_t right;
    int32_t first, count;   // leaves: their triangles (DFS ranks)
    int32_t pad0, pad1;     // every node: first DFS rank and number of the triangles below it (what walk_tree.hip builds the 8-wide tree from)
};
static_assert(sizeof(DTri) == 64 && sizeof(DTriShade) == 64 && sizeof(DNode2) == 48 && sizeof(DNode8) == 80 && sizeof(DGate) == 32, "layout");

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
// BSDF class of a material type (rtamd.h's RT_MAT_* order): the queues k_shade walks are sorted by it, so that a wave runs one BSDF's code
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int rt_class_of(uint32_t type) {
    switch (type) {
    case 1: return 1;   // RT_MAT_MIRROR
    case 3: return 2;   // RT_MAT_GLASS
    case 2: return 3;   // RT_MAT_CONDUCTOR
    case 4: return 4;   // RT_MAT_DIELECTRIC
    case 6: return 5;   // RT_MAT_PLASTIC
    default: return 0;  // diffuse, Oren-Nayar
    }
}

struct DTexture { uint32_t offset; int32_t width, height; int32_t pad; };  // offset in floats into the texel pool

struct DLight { uint32_t kind; uint32_t tri; float area; uint32_t pad; };  // kind 0 area (tri = DFS rank; area = Triangle::area, Geometry.h:80), 1 background colour, 2 environment map

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
    const DNode8* nodes8;
    const DTri* trisw;
    const DGate* gate;
    const DNode2* nodes2;
    const DMaterial* mats;
    const DTexture* tex;
    const float* texels;
    const DLight* lights;
    const uint8_t* tri_cls;   // per DFS rank: the BSDF class of the triangle's material (rt_class_of), 0x80 for an emitter -- what a closest-hit walk needs to
                              // file its hit, from a 1-B-per-triangle array that stays in L2 instead of a line of `tris` and one of the material table
    uint32_t n_tris, n_lights, n_nodes2, n_nodes8;
    uint32_t root_ref;        // RT_REF_EMPTY for an empty scene, else node 0
    uint32_t walk_stack, max_stack2;   // most 8-byte entries a walk over nodes8 can have waiting; most words of the reference-order walk
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
