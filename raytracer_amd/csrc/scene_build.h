// Host-side scene preparation (no HIP calls): Triangle::init precompute, the
// reference-topology binned-SAH tree and the flat arrays the kernels read.  See scene_build.cpp.
#pragma once
#include <string>
#include <vector>

#include "../../include/rtamd.h"
#include "rt_common.h"

struct HostScene {
    std::vector<DTri> tris;            // DFS leaf order
    std::vector<DTriShade> trishade;
    std::vector<DNode2> nodes2;
    std::vector<DMaterial> mats;
    std::vector<DTexture> tex;
    std::vector<float> texels;
    std::vector<DLight> lights;
    std::vector<uint32_t> leaf_order;  // DFS rank -> caller triangle index
    uint32_t depth2 = 0, leaves2 = 0, max_leaf2 = 0;
    uint32_t max_stack2 = 0;
    float abs_max = 0;
    float bounds_lo[3], bounds_hi[3];
    float centre[3], radius = 0;
    float bg_colour[3] = { 0, 0, 0 };
    uint32_t bg_kind = 0; int32_t bg_tex = -1;
    // EnvironmentMap::build (Lights.h:207-253): per-row-normalised pdfs [nv][nu], row pdfs [nv], their total; empty without an env map
    std::vector<float> env_cond, env_marg;
    float env_func_int = 0; int32_t env_nu = 0, env_nv = 0;
    double build_seconds = 0;
    // sizes of the geometry arrays (equal to the vectors' sizes after a host build; a device-side build leaves the vectors empty)
    uint32_t n_tris = 0, n_nodes2 = 0;
    // the tree the walk reads (walk_tree.hip, derived on the device from nodes2 + tris whichever builder made them): 8-wide nodes, the most
    // 8-byte stack entries a walk can have waiting, and its levels
    uint32_t n_nodes8 = 0, walk_stack = 0, walk_levels = 0;
};

// What a device-side build leaves in device memory (hipMalloc'ed; the caller owns the three arrays afterwards).
struct DeviceGeometry { DTri* tris = nullptr; DTriShade* trishade = nullptr; DNode2* nodes2 = nullptr; };
// Builds the binary tree, the triangle records and the reference-topology nodes on the GPU and fills the geometry fields
// of `hs` (leaf_order, statistics, stack bound, scene bounds, sizes).  Returns 1 when it did, 2 when it declines (scenes
// the host handles: fewer than two triangles), 0 on error (`err`).
typedef int (*DeviceSceneBuilder)(const rt_scene_desc& d, HostScene& hs, DeviceGeometry& out, std::string& err);

// Returns false and fills `err` on invalid input.  With `device_builder` (bvh_build.hip) and `geom`, everything that is
// per triangle or per node happens on the device and hs.tris / trishade / nodes2 stay empty (geom holds the device
// arrays); without it the host builds them.  Both give the same tree, hence -- walk_tree.hip reads only the tree's shape,
// its boxes and the triangles -- the same walk nodes, byte for byte (tests/test_bvh_build_gpu.py).
bool host_scene_build(const rt_scene_desc& d, HostScene& out, std::string& err, DeviceSceneBuilder device_builder = nullptr,
                      DeviceGeometry* geom = nullptr);
// Camera::init + Camera::updateView (Scene.h:26-45)
void host_camera_build(const rt_camera& c, DCamera& out);
