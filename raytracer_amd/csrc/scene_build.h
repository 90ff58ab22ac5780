// Host-side scene preparation (no HIP calls): Triangle::init precompute, the
// reference-topology binned-SAH tree, its 4-wide collapse and the flat arrays the
// kernels read.  See scene_build.cpp.
#pragma once
#include <string>
#include <vector>

#include "../../include/rtamd.h"
#include "rt_common.h"

struct HostScene {
    std::vector<DTri> tris;            // DFS leaf order
    std::vector<DTriShade> trishade;
    std::vector<DNode4> nodes4;
    std::vector<DNode2> nodes2;
    std::vector<DMaterial> mats;
    std::vector<DTexture> tex;
    std::vector<float> texels;
    std::vector<DLight> lights;
    std::vector<uint32_t> leaf_order;  // DFS rank -> caller triangle index
    uint32_t depth2 = 0, leaves2 = 0, max_leaf2 = 0;
    uint32_t max_stack4 = 0, max_stack2 = 0;
    float abs_max = 0;
    float bounds_lo[3], bounds_hi[3];
    float centre[3], radius = 0;
    float bg_colour[3] = { 0, 0, 0 };
    uint32_t bg_kind = 0; int32_t bg_tex = -1;
    // EnvironmentMap::build (Lights.h:207-253): per-row-normalised pdfs [nv][nu], row pdfs [nv], their total; empty without an env map
    std::vector<float> env_cond, env_marg;
    float env_func_int = 0; int32_t env_nu = 0, env_nv = 0;
    double build_seconds = 0;
};

// The reference-topology binary tree as a builder hands it over: node 0 is the root, leaves have left == right == -1,
// [begin, begin + count) is the node's range in `order` (= DFS leaf order: the partitions keep the left child first).
// Nodes with count <= 1 keep the inverted default box (Geometry.h:401 returns before the bounds are grown).
struct Bvh2Node { float lo[3], hi[3]; int32_t left, right; uint32_t begin, count, depth; };
struct Bvh2 { std::vector<Bvh2Node> nodes; std::vector<uint32_t> order; };
typedef bool (*Bvh2Builder)(const float* verts, uint32_t n_tris, Bvh2& out, std::string& err);

// Returns false and fills `err` on invalid input.  `device_builder` (bvh_build.hip) replaces the host's own
// binned-SAH build of the binary tree when given; everything downstream is the same code.
bool host_scene_build(const rt_scene_desc& d, HostScene& out, std::string& err, Bvh2Builder device_builder = nullptr);
// Camera::init + Camera::updateView (Scene.h:26-45)
void host_camera_build(const rt_camera& c, DCamera& out);
