// scene_dev.h — the opaque rt_scene / error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/rtamd.h"
#include "rt_common.h"
#include "scene_build.h"

void rt_set_error(const std::string& msg);
int rt_hip_fail(hipError_t e, const char* what);
#define RT_HIP(call)                                             \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return rt_hip_fail(e_, #call);     \
    } while (0)

// Launch geometry of the persistent traversal kernels: enough 256-thread blocks to fill every CU at the
// occupancy the kernel's registers/LDS allow; waves then pull 64-ray chunks from a work counter.
struct LaunchCfg { int blocks = 0; int threads = 0; };

struct rt_scene {
    int device = 0;
    HostScene host;          // kept for info/leaf-order queries (device arrays are the copies the kernels read)
    DScene d{};              // device pointers + scalars, passed to kernels by value
    bool has_camera = false;
    std::vector<void*> allocs;
    uint64_t device_bytes = 0;
    // per-scene scratch for the batched intersect entry points
    DCounters* d_counters = nullptr;
    unsigned long long* d_work = nullptr;   // work-fetch counters
    uint32_t* d_spillw = nullptr;           // overflow of the LDS stack (wide walk, 8-byte entries), may be null
    uint32_t* d_spill2 = nullptr;           // stack of the reference-order walk
    hipEvent_t scratch_event = nullptr;     // recorded behind the last kernel that used the three arrays above (intersect.hip)
    bool scratch_busy = false;
    int persistent_blocks = 0;
    bool count_enabled = false;
    // renderers of this scene hold rt_render requests back to merge them (render.hip): whatever changes what those
    // requests would see (camera, counting) runs this hook first.  Null while no renderer exists.
    void (*before_change)(rt_scene*) = nullptr;
    void (*before_destroy)(rt_scene*) = nullptr;   // lets those renderers forget the scene
    std::vector<void*> renderers;
    std::mutex mu;
};

// walk_tree.hip: gate boxes, the 8-wide walk nodes and the walk-ordered triangle records, from nodes2 + tris on the device
int rt_build_walk_tree(rt_scene* s);
// shared launch helpers (intersect.hip)
int rt_scene_persistent_blocks(const rt_scene* s);
