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
    double build_seconds = 0;
};

// Returns false and fills `err` on invalid input.
bool host_scene_build(const rt_scene_desc& d, HostScene& out, std::string& err);
// Camera::init + Camera::updateView (Scene.h:26-45)
void host_camera_build(const rt_camera& c, DCamera& out);
