// intersect.hip — rt_scene lifetime + the batched Scene::traverse / Scene::visible entry points.
//
// Kernels are persistent: a fixed grid (CUs x resident blocks) of 256-thread blocks; each wave pulls
// 64-ray chunks from a device work counter until the batch is drained, so a wave that finishes a cheap
// chunk immediately takes more work and ray order (hence BVH/L2 locality of neighbouring rays) is kept.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include "rt_device.h"
#include "scene_dev.h"

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
void rt_set_error(const std::string& msg) { g_err = msg; }
int rt_hip_fail(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return RT_ERR_DEVICE;
}
extern "C" const char* rt_last_error(void) { return g_err.c_str(); }
extern "C" int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int rt_device_set(int device) {
    RT_HIP(hipSetDevice(device));
    return RT_OK;
}

// ------------------------------------------------------------------------------------------------ kernels
// Work source of the batch kernels: one counter, `grant` consecutive items per atomic -- 256 for big batches (one
// contended atomic costs ~11 ns), fewer when the batch is small enough that a wave's last grant would be a large part of
// the launch (about 16 grants per wave, at least 64 items).
struct CounterSource {
    unsigned long long* work;
    uint32_t n, grant;
    __device__ __forceinline__ void init(unsigned long long* w, uint32_t count) {
        work = w; n = count;
        const uint32_t want = count / (gridDim.x * (blockDim.x >> 6) * 16u);
        grant = 256u;
        while (grant > 64u && grant > want) grant >>= 1;
    }
    __device__ __forceinline__ bool next(uint32_t& begin, uint32_t& end) {
        unsigned long long base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(work, (unsigned long long)grant);
        unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base);
        unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
        if (hi != 0 || lo >= n) return false;
        begin = lo;
        end = (n - lo) < grant ? n : lo + grant;
        return true;
    }
};

struct ClosestJob {
    const DScene& S;
    const float* __restrict__ rays;
    rt_hit* __restrict__ out;
    uint32_t done;
    static constexpr uint32_t refill_idle = RT_REFILL_IDLE;
    __device__ __forceinline__ bool load(uint32_t i, f3& o, f3& d) {
        const float* r = rays + (size_t)i * 6;
        o = mk3(r[0], r[1], r[2]); d = mk3(r[3], r[4], r[5]);
        return true;
    }
    __device__ __forceinline__ void retire(uint32_t i, const Walk& w, bool) {
        const HitRec& h = w.best;
        rt_hit res;
        if (h.t < RT_FLT_MAX) {
            res.ID = S.tris[h.rank].orig_id; res.t = h.t; res.alpha = h.u; res.beta = h.v; res.gamma = 1.0f - (h.u + h.v);  // Geometry.h:450-454
        } else { res.ID = 0xFFFFFFFFu; res.t = RT_FLT_MAX; res.alpha = res.beta = res.gamma = 0.0f; }
        out[i] = res;
        done++;
    }
};
struct AnyJob {
    const float* __restrict__ segs;
    uint8_t* __restrict__ out;
    uint32_t done;
    static constexpr uint32_t refill_idle = RT_REFILL_IDLE_ANY;
    __device__ __forceinline__ bool load(uint32_t i, f3& p1, f3& p2) {
        const float* q = segs + (size_t)i * 6;
        p1 = mk3(q[0], q[1], q[2]); p2 = mk3(q[3], q[4], q[5]);
        return true;
    }
    __device__ __forceinline__ void retire(uint32_t i, const Walk&, bool visible) { out[i] = visible ? 1 : 0; done++; }
};

// n <= RT_LAUNCH_MAX items per launch (the host cuts longer batches), so item indices fit 32 bits
#define RT_LAUNCH_MAX (1u << 30)
template <bool COUNT>
__global__ __launch_bounds__(RT_BLOCK, RT_TRAV_WAVES) void k_intersect_closest(DScene S, const float* __restrict__ rays, uint32_t n,
                                                                rt_hit* __restrict__ out, int mode, unsigned long long* work,
                                                                uint32_t* spillw, uint32_t* spill2, DCounters* counters) {
    __shared__ __align__(16) uint32_t lds_stack[RT_STACK_WORDS];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x, gstride = gridDim.x * RT_BLOCK;
    RT_TRAV_STACK_INIT(st, lds_stack, spillw, gtid, gstride)
    uint32_t cn = 0, ct = 0, cx = 0;
    ClosestJob job{S, rays, out, 0u};
    CounterSource src; src.init(work, n);
    rt_persistent_walk<false, COUNT>(S, job, src, mode, st, spill2 + gtid, gstride, cn, ct, cx, counters);
    if (COUNT) {
        atomicAdd(&counters->closest, (unsigned long long)job.done); atomicAdd(&counters->nodes, (unsigned long long)cn);
        atomicAdd(&counters->tris, (unsigned long long)ct); atomicAdd(&counters->exact, (unsigned long long)cx);
    }
}

template <bool COUNT>
__global__ __launch_bounds__(RT_BLOCK, RT_TRAV_WAVES) void k_intersect_any(DScene S, const float* __restrict__ segs, uint32_t n,
                                                            uint8_t* __restrict__ out, int mode, unsigned long long* work,
                                                            uint32_t* spillw, uint32_t* spill2, DCounters* counters) {
    __shared__ __align__(16) uint32_t lds_stack[RT_STACK_WORDS];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x, gstride = gridDim.x * RT_BLOCK;
    RT_TRAV_STACK_INIT(st, lds_stack, spillw, gtid, gstride)
    uint32_t cn = 0, ct = 0, cx = 0;
    AnyJob job{segs, out, 0u};
    CounterSource src; src.init(work, n);
    rt_persistent_walk<true, COUNT>(S, job, src, mode, st, spill2 + gtid, gstride, cn, ct, cx, counters);
    if (COUNT) {
        atomicAdd(&counters->any, (unsigned long long)job.done); atomicAdd(&counters->any_nodes, (unsigned long long)cn);
        atomicAdd(&counters->any_tris, (unsigned long long)ct); atomicAdd(&counters->exact, (unsigned long long)cx);
    }
}

// Camera::generateRay (Scene.h:47-64)
__device__ __forceinline__ f3 rt_mul_point_div(const float* m, f3 v) {  // Core.h:310
    f3 a = mk3((v.x * m[0] + v.y * m[1] + v.z * m[2]) + m[3], (v.x * m[4] + v.y * m[5] + v.z * m[6]) + m[7],
               (v.x * m[8] + v.y * m[9] + v.z * m[10]) + m[11]);
    float w = (m[12] * v.x) + (m[13] * v.y) + (m[14] * v.z) + m[15];
    w = 1.0f / w;
    return a * w;
}
__global__ void k_generate_rays(DScene S, const float* __restrict__ xy, size_t n, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DCamera& c = S.cam;
    float x = xy[i * 2], y = xy[i * 2 + 1];
    float xc = ((2 * x) / (c.width - 1)) - 1;
    float yc = 1 - ((2 * y) / (c.height - 1));
    f3 pcam = rt_mul_point_div(c.invP, mk3(xc, yc, 0));
    f3 dcam = rt_mul_point_div(c.cam, pcam);
    f3 org = ld3(c.origin);
    f3 dir = dcam - org;
    float* r = out + i * 6;
    r[0] = org.x; r[1] = org.y; r[2] = org.z; r[3] = dir.x; r[4] = dir.y; r[5] = dir.z;
}

// ------------------------------------------------------------------------------------------------ scene
template <class T>
static int upload(rt_scene* s, const std::vector<T>& v, const T** dst) {
    size_t bytes = sizeof(T) * std::max<size_t>(v.size(), 1);
    void* p = nullptr;
    RT_HIP(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    s->device_bytes += bytes;
    if (!v.empty()) RT_HIP(hipMemcpy(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    *dst = (const T*)p;
    return RT_OK;
}

int rt_scene_persistent_blocks(const rt_scene* s) { return s->persistent_blocks; }

int device_scene_build(const rt_scene_desc& d, HostScene& hs, DeviceGeometry& out, std::string& err);   // bvh_build.hip

static int na_scratch_setup(int device);   // scratch of the no-allocation entry point (below)

// DScene::tri_cls: one byte per triangle (DFS rank) with what ExtendJob::retire needs of the hit's material
__global__ void k_tri_class(uint32_t n, const DTri* __restrict__ tris, const DMaterial* __restrict__ mats, uint8_t* __restrict__ out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const DMaterial& m = mats[tris[r].mat_gsign & 0x7FFFFFFFu];
    out[r] = (uint8_t)((uint32_t)rt_class_of(m.type) | ((m.flags & RT_MF_LIGHT) ? 0x80u : 0u));
}

extern "C" int rt_scene_create(const rt_scene_desc* desc, rt_scene** out) { return rt_scene_create_ex(desc, 0u, out); }

extern "C" int rt_scene_create_ex(const rt_scene_desc* desc, uint32_t flags, rt_scene** out) {
    if (!desc || !out) { rt_set_error("rt_scene_create: NULL argument"); return RT_ERR_INVALID; }
    if (flags != 0u && flags != RT_SCENE_BUILD_DEVICE && flags != RT_SCENE_BUILD_HOST) { rt_set_error("rt_scene_create_ex: flags must be 0, RT_SCENE_BUILD_DEVICE or RT_SCENE_BUILD_HOST"); return RT_ERR_INVALID; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        rt_set_error("rt_scene_create: no HIP device visible (this library has no CPU path)");
        return RT_ERR_DEVICE;
    }
    rt_scene* s = new rt_scene();
    std::string err;
    int rc = RT_OK;
    auto fail = [&](int code) { rt_scene_destroy(s); return code; };
    if (hipGetDevice(&s->device) != hipSuccess) return fail(RT_ERR_DEVICE);
    DeviceGeometry geom;
    const bool built = host_scene_build(*desc, s->host, err, flags == RT_SCENE_BUILD_HOST ? nullptr : device_scene_build, &geom);
    HostScene& h = s->host;
    DScene& d = s->d;
    if (geom.tris) {   // a device-side build leaves its arrays in device memory: they are the scene's from here on
        for (void* p : { (void*)geom.tris, (void*)geom.trishade, (void*)geom.nodes2 }) s->allocs.push_back(p);
        s->device_bytes += (uint64_t)h.n_tris * (sizeof(DTri) + sizeof(DTriShade)) + (uint64_t)h.n_nodes2 * sizeof(DNode2);
        d.tris = geom.tris; d.trishade = geom.trishade; d.nodes2 = geom.nodes2;
    }
    if (!built) { rt_set_error("rt_scene_create: " + err); return fail(RT_ERR_INVALID); }
    if ((rc = na_scratch_setup(s->device))) return fail(rc);
    if (!geom.tris && ((rc = upload(s, h.tris, &d.tris)) || (rc = upload(s, h.trishade, &d.trishade)) || (rc = upload(s, h.nodes2, &d.nodes2))))
        return fail(rc);
    if ((rc = upload(s, h.mats, &d.mats)) || (rc = upload(s, h.tex, &d.tex)) || (rc = upload(s, h.texels, &d.texels)) || (rc = upload(s, h.lights, &d.lights)))
        return fail(rc);
    d.env_cond = nullptr; d.env_marg = nullptr; d.env_func_int = h.env_func_int; d.env_nu = h.env_nu; d.env_nv = h.env_nv;
    if (!h.env_cond.empty() && ((rc = upload(s, h.env_cond, &d.env_cond)) || (rc = upload(s, h.env_marg, &d.env_marg)))) return fail(rc);
    d.n_tris = h.n_tris; d.n_lights = (uint32_t)h.lights.size();
    d.n_nodes2 = h.n_nodes2;
    {
        void* pc = nullptr;
        if (hipMalloc(&pc, h.n_tris ? h.n_tris : 16) != hipSuccess) { rt_set_error("rt_scene_create: out of device memory (triangle classes)"); return fail(RT_ERR_NOMEM); }
        s->allocs.push_back(pc); s->device_bytes += h.n_tris; d.tri_cls = (const uint8_t*)pc;
        if (h.n_tris) {
            hipLaunchKernelGGL(k_tri_class, dim3((h.n_tris + 255) / 256), dim3(256), 0, 0, h.n_tris, d.tris, d.mats, (uint8_t*)pc);
            if (hipGetLastError() != hipSuccess) { rt_set_error("rt_scene_create: k_tri_class launch failed"); return fail(RT_ERR_DEVICE); }
        }
    }
    // the walk's own arrays (gate boxes, 8-wide nodes, walk-ordered triangle records), derived on the device from whichever
    // builder made tris + nodes2 (walk_tree.hip)
    if ((rc = rt_build_walk_tree(s))) return fail(rc);
    d.root_ref = d.n_tris ? 0u : RT_REF_EMPTY;
    d.max_stack2 = h.max_stack2;   // d.walk_stack: set by rt_build_walk_tree
    d.abs_max = h.abs_max;
    memcpy(d.bg_colour, h.bg_colour, 12); d.bg_kind = h.bg_kind; d.bg_tex = h.bg_tex;
    memcpy(d.scene_centre, h.centre, 12); d.scene_radius = h.radius;

    // persistent launch geometry + traversal scratch
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, s->device) != hipSuccess) return fail(RT_ERR_DEVICE);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_intersect_closest<false>, RT_BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (per_cu > 8) per_cu = 8;
    s->persistent_blocks = prop.multiProcessorCount * per_cu;
    size_t threads = (size_t)s->persistent_blocks * RT_BLOCK;
    void* p = nullptr;
    if (RT_SPILL_WORDS(h.walk_stack)) {
        size_t bytes = threads * RT_SPILL_WORDS(h.walk_stack) * sizeof(uint32_t);
        if (hipMalloc(&p, bytes) != hipSuccess) { rt_set_error("rt_scene_create: out of device memory (stack spill)"); return fail(RT_ERR_NOMEM); }
        s->allocs.push_back(p); s->device_bytes += bytes; s->d_spillw = (uint32_t*)p;
    }
    {
        size_t bytes = threads * std::max<uint32_t>(h.max_stack2, 2) * sizeof(uint32_t);
        if (hipMalloc(&p, bytes) != hipSuccess) { rt_set_error("rt_scene_create: out of device memory (reference-walk stack)"); return fail(RT_ERR_NOMEM); }
        s->allocs.push_back(p); s->device_bytes += bytes; s->d_spill2 = (uint32_t*)p;
    }
    if (hipMalloc(&p, sizeof(DCounters)) != hipSuccess) return fail(RT_ERR_NOMEM);
    s->allocs.push_back(p); s->d_counters = (DCounters*)p;
    if (hipMemset(p, 0, sizeof(DCounters)) != hipSuccess) return fail(RT_ERR_DEVICE);
    if (hipMalloc(&p, 256) != hipSuccess) return fail(RT_ERR_NOMEM);
    s->allocs.push_back(p); s->d_work = (unsigned long long*)p;
    *out = s;
    return RT_OK;
}

extern "C" void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    if (s->before_destroy) s->before_destroy(s);
    if (s->scratch_event) { (void)hipEventSynchronize(s->scratch_event); (void)hipEventDestroy(s->scratch_event); }
    for (void* p : s->allocs) (void)hipFree(p);
    delete s;
}

extern "C" int rt_scene_set_camera(rt_scene* s, const rt_camera* cam) {
    if (!s || !cam) { rt_set_error("rt_scene_set_camera: NULL argument"); return RT_ERR_INVALID; }
    if (cam->width < 2 || cam->height < 2) { rt_set_error("rt_scene_set_camera: width/height must be >= 2"); return RT_ERR_INVALID; }
    if (s->before_change) s->before_change(s);   // samples already requested were asked for with the old camera
    host_camera_build(*cam, s->d.cam);
    s->has_camera = true;
    return RT_OK;
}

extern "C" int rt_scene_get_info(const rt_scene* s, rt_scene_info* o) {
    if (!s || !o) { rt_set_error("rt_scene_get_info: NULL argument"); return RT_ERR_INVALID; }
    const HostScene& h = s->host;
    o->n_tris = h.n_tris; o->n_lights = (uint32_t)h.lights.size();
    o->bvh2_nodes = h.n_nodes2; o->bvh2_depth = h.depth2; o->bvh2_leaves = h.leaves2; o->bvh2_max_leaf = h.max_leaf2;
    o->wide_nodes = h.n_nodes8; o->wide_max_stack = h.walk_stack; o->wide_levels = h.walk_levels;
    o->device_bytes = s->device_bytes; o->build_seconds = (float)h.build_seconds;
    return RT_OK;
}

// The reference-topology binary tree as the builders leave it in device memory (48 B per node: lo[3] left hi[3] right first count
// subtree_first subtree_count): lets a test recompute the gate boxes and check the walk tree against the tree it stands for.
extern "C" int rt_scene_debug_ref_nodes(rt_scene* s, void* out, size_t max_nodes) {
    if (!s || !out) { rt_set_error("rt_scene_debug_ref_nodes: NULL argument"); return RT_ERR_INVALID; }
    if (max_nodes < s->host.n_nodes2) { rt_set_error("rt_scene_debug_ref_nodes: buffer too small"); return RT_ERR_INVALID; }
    if (s->host.n_nodes2) RT_HIP(hipMemcpy(out, s->d.nodes2, (size_t)s->host.n_nodes2 * sizeof(DNode2), hipMemcpyDeviceToHost));
    return RT_OK;
}
// The tree the walk reads (80 B per 8-wide node), the per-triangle gate boxes (32 B each, DFS rank order) and the walk-ordered
// triangle records (64 B each); any pointer may be NULL.
extern "C" int rt_scene_debug_walk_nodes(rt_scene* s, void* nodes_out, size_t max_nodes, void* gate_out, size_t max_tris, void* trisw_out) {
    if (!s) { rt_set_error("rt_scene_debug_walk_nodes: NULL scene"); return RT_ERR_INVALID; }
    if ((nodes_out && max_nodes < s->host.n_nodes8) || ((gate_out || trisw_out) && max_tris < s->host.n_tris)) { rt_set_error("rt_scene_debug_walk_nodes: buffer too small"); return RT_ERR_INVALID; }
    if (nodes_out && s->host.n_nodes8) RT_HIP(hipMemcpy(nodes_out, s->d.nodes8, (size_t)s->host.n_nodes8 * sizeof(DNode8), hipMemcpyDeviceToHost));
    if (gate_out && s->host.n_tris) RT_HIP(hipMemcpy(gate_out, s->d.gate, (size_t)s->host.n_tris * sizeof(DGate), hipMemcpyDeviceToHost));
    if (trisw_out && s->host.n_tris) RT_HIP(hipMemcpy(trisw_out, s->d.trisw, (size_t)s->host.n_tris * sizeof(DTri), hipMemcpyDeviceToHost));
    return RT_OK;
}
extern "C" int rt_scene_debug_tri_classes(rt_scene* s, uint8_t* out, size_t max_tris) {
    if (!s || !out) { rt_set_error("rt_scene_debug_tri_classes: NULL argument"); return RT_ERR_INVALID; }
    if (max_tris < s->host.n_tris) { rt_set_error("rt_scene_debug_tri_classes: buffer too small"); return RT_ERR_INVALID; }
    if (s->host.n_tris) RT_HIP(hipMemcpy(out, s->d.tri_cls, (size_t)s->host.n_tris, hipMemcpyDeviceToHost));
    return RT_OK;
}
extern "C" int rt_scene_leaf_order(const rt_scene* s, uint32_t* out) {
    if (!s || !out) { rt_set_error("rt_scene_leaf_order: NULL argument"); return RT_ERR_INVALID; }
    memcpy(out, s->host.leaf_order.data(), s->host.leaf_order.size() * sizeof(uint32_t));
    return RT_OK;
}

// Host-only structure query (no device needed): lets CPU-side tests check the builder against the oracle.
extern "C" int rt_host_build_info(const rt_scene_desc* desc, uint32_t* leaf_order, rt_scene_info* info) {
    if (!desc) { rt_set_error("rt_host_build_info: NULL argument"); return RT_ERR_INVALID; }
    HostScene h; std::string err;
    if (!host_scene_build(*desc, h, err)) { rt_set_error("rt_host_build_info: " + err); return RT_ERR_INVALID; }
    if (leaf_order) memcpy(leaf_order, h.leaf_order.data(), h.leaf_order.size() * sizeof(uint32_t));
    if (info) {
        memset(info, 0, sizeof(*info));
        info->n_tris = h.n_tris; info->n_lights = (uint32_t)h.lights.size();
        info->bvh2_nodes = h.n_nodes2; info->bvh2_depth = h.depth2; info->bvh2_leaves = h.leaves2; info->bvh2_max_leaf = h.max_leaf2;
        info->wide_nodes = 0; info->wide_max_stack = 0; info->wide_levels = 0; info->build_seconds = (float)h.build_seconds;
    }
    return RT_OK;
}

extern "C" int rt_counters_enable(rt_scene* s, int enable) {
    if (!s) { rt_set_error("rt_counters_enable: NULL scene"); return RT_ERR_INVALID; }
    if (s->before_change) s->before_change(s);
    s->count_enabled = enable != 0;
    return RT_OK;
}
extern "C" int rt_counters_read(rt_scene* s, rt_counters* o, int reset) {
    if (!s || !o) { rt_set_error("rt_counters_read: NULL argument"); return RT_ERR_INVALID; }
    DCounters c;
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(&c, s->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    o->closest_rays = c.closest; o->any_rays = c.any; o->nodes_visited = c.nodes; o->tris_tested = c.tris; o->exact_path_rays = c.exact;
    o->any_nodes_visited = c.any_nodes; o->any_tris_tested = c.any_tris;
    o->lane_slots = c.slots; o->lane_busy = c.busy; o->any_lane_slots = c.any_slots; o->any_lane_busy = c.any_busy; o->max_steps = c.max_steps; o->any_max_steps = c.any_max_steps;
    o->gate_refused = c.refused; o->any_gate_refused = c.any_refused;
    if (reset) RT_HIP(hipMemset(s->d_counters, 0, sizeof(DCounters)));
    return RT_OK;
}

// ------------------------------------------------------------------------------------------------ batched intersect
static int check_mode(int mode) {
    if (mode == 0 || mode == 1) return RT_OK;
    rt_set_error("intersect: mode must be 0 (wide walk) or 1 (reference-order walk)");
    return RT_ERR_INVALID;
}

// The work counter and the two stack-spill arrays are one set per scene, and the kernels that use them run after the
// host lock is gone: two launches on different streams (two caller streams, or two host threads in host_batch's chunked
// path, each with a run stream of its own) would overlap on the device, reset each other's counter and push into the same
// spill slots.  Launches that share the scratch are therefore chained on the device: each waits for the event the
// previous one recorded behind its kernel.  Both helpers are called with s->mu held.
static int scratch_acquire(rt_scene* s, hipStream_t st) {
    if (!s->scratch_event) RT_HIP(hipEventCreateWithFlags(&s->scratch_event, hipEventDisableTiming));
    if (s->scratch_busy) RT_HIP(hipStreamWaitEvent(st, s->scratch_event, 0));
    return RT_OK;
}
static int scratch_release(rt_scene* s, hipStream_t st) {
    RT_HIP(hipEventRecord(s->scratch_event, st));
    s->scratch_busy = true;
    return RT_OK;
}

extern "C" int rt_intersect_closest_dev(rt_scene* s, const rt_ray* d_rays, size_t n, rt_hit* d_out, int mode, void* stream) {
    if (!s || (n && (!d_rays || !d_out))) { rt_set_error("rt_intersect_closest_dev: NULL argument"); return RT_ERR_INVALID; }
    if (check_mode(mode)) return RT_ERR_INVALID;
    if (n == 0) return RT_OK;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(s->mu);  // the work counter and the stack spill are per scene
    { int rc = scratch_acquire(s, st); if (rc) return rc; }
    for (size_t off = 0; off < n; off += RT_LAUNCH_MAX) {
        const uint32_t m = (uint32_t)std::min<size_t>(n - off, RT_LAUNCH_MAX);
        const float* in = (const float*)(d_rays + off);
        RT_HIP(hipMemsetAsync(s->d_work, 0, sizeof(unsigned long long), st));
        int blocks = (int)std::min<size_t>(((size_t)m + RT_BLOCK - 1) / RT_BLOCK, (size_t)s->persistent_blocks);
        if (s->count_enabled)
            hipLaunchKernelGGL(k_intersect_closest<true>, dim3(blocks), dim3(RT_BLOCK), 0, st, s->d, in, m, d_out + off, mode, s->d_work,
                               s->d_spillw, s->d_spill2, s->d_counters);
        else
            hipLaunchKernelGGL(k_intersect_closest<false>, dim3(blocks), dim3(RT_BLOCK), 0, st, s->d, in, m, d_out + off, mode, s->d_work,
                               s->d_spillw, s->d_spill2, s->d_counters);
        RT_HIP(hipGetLastError());
    }
    return scratch_release(s, st);
}

extern "C" int rt_intersect_any_dev(rt_scene* s, const rt_segment* d_segs, size_t n, uint8_t* d_out, int mode, void* stream) {
    if (!s || (n && (!d_segs || !d_out))) { rt_set_error("rt_intersect_any_dev: NULL argument"); return RT_ERR_INVALID; }
    if (check_mode(mode)) return RT_ERR_INVALID;
    if (n == 0) return RT_OK;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(s->mu);
    { int rc = scratch_acquire(s, st); if (rc) return rc; }
    for (size_t off = 0; off < n; off += RT_LAUNCH_MAX) {
        const uint32_t m = (uint32_t)std::min<size_t>(n - off, RT_LAUNCH_MAX);
        const float* in = (const float*)(d_segs + off);
        RT_HIP(hipMemsetAsync(s->d_work, 0, sizeof(unsigned long long), st));
        int blocks = (int)std::min<size_t>(((size_t)m + RT_BLOCK - 1) / RT_BLOCK, (size_t)s->persistent_blocks);
        if (s->count_enabled)
            hipLaunchKernelGGL(k_intersect_any<true>, dim3(blocks), dim3(RT_BLOCK), 0, st, s->d, in, m, d_out + off, mode, s->d_work,
                               s->d_spillw, s->d_spill2, s->d_counters);
        else
            hipLaunchKernelGGL(k_intersect_any<false>, dim3(blocks), dim3(RT_BLOCK), 0, st, s->d, in, m, d_out + off, mode, s->d_work,
                               s->d_spillw, s->d_spill2, s->d_counters);
        RT_HIP(hipGetLastError());
    }
    return scratch_release(s, st);
}

// ------------------------------------------------------------------------------------------------ no-allocation entry point
// rt_intersect_closest_dev for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid
// hipMalloc).  The traversal kernel needs scratch beyond its inputs and outputs -- the work counter its waves draw ray
// indices from and the overflow of the per-lane traversal stacks -- which the other entry points take from arrays the
// scene allocated with hipMalloc.  Here it lives in one file-scope device array of fixed size (RT_NA_BYTES, ~42 MB per GPU
// the library runs on, held until the process ends), so the call itself allocates nothing; the price is a fixed, smaller
// grid (RT_NA_BLOCKS blocks), a cap on the batch (RT_NA_MAX_RAYS) and on the scene's stack bounds (RT_NA_WORDS per lane).
// `scratch` is a host-side stand-in, not a device address: host code only ever uses the per-device pointer that
// hipGetSymbolAddress returned for it (na_scratch_ptr), taken once per device at scene creation with that device current.
// Every call on a GPU shares that GPU's array: the entry point chains calls that arrive on different streams (below).
#define RT_NA_BLOCKS 512u
#define RT_NA_THREADS (RT_NA_BLOCKS * RT_BLOCK)
#define RT_NA_WORDS 80u                       // stack words per lane: (wide_max_stack - RT_STACK_LDS) + max(bvh2 stack bound, 2)
#define RT_NA_HEADER 64u                      // words in front of the stacks: the work counter, in a 256-B line of its own
#define RT_NA_MAX_RAYS ((size_t)1 << 24)
#define RT_NA_BYTES ((size_t)(RT_NA_HEADER + (size_t)RT_NA_THREADS * RT_NA_WORDS) * sizeof(uint32_t))
static __device__ __align__(256) uint32_t scratch[RT_NA_HEADER + (size_t)RT_NA_THREADS * RT_NA_WORDS];
#define RT_NA_MAX_DEVICES 64
static uint32_t* na_scratch_ptr[RT_NA_MAX_DEVICES];   // device address of `scratch` on each GPU (every GPU holds its own copy)
static hipEvent_t na_event[RT_NA_MAX_DEVICES];        // recorded behind the last launch that used that GPU's array
static hipStream_t na_last_stream[RT_NA_MAX_DEVICES];
static bool na_busy[RT_NA_MAX_DEVICES];
static std::mutex na_mu;

// once per device, with that device current (rt_scene_create_ex calls it): the array's address and the event, so that the
// entry point itself creates nothing
static int na_scratch_setup(int device) {
    if (device < 0 || device >= RT_NA_MAX_DEVICES) return RT_OK;   // such a device simply has no no-allocation path
    std::lock_guard<std::mutex> lk(na_mu);
    if (na_scratch_ptr[device]) return RT_OK;
    uint32_t* ptr = nullptr;
    RT_HIP(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
    RT_HIP(hipEventCreateWithFlags(&na_event[device], hipEventDisableTiming));
    na_scratch_ptr[device] = ptr;
    return RT_OK;
}

extern "C" int rt_intersect_closest_noalloc(rt_scene* s, const rt_ray* d_rays, size_t n, rt_hit* d_out, int mode, void* stream) {
    if (!s || (n && (!d_rays || !d_out))) { rt_set_error("rt_intersect_closest_noalloc: NULL argument"); return RT_ERR_INVALID; }
    if (check_mode(mode)) return RT_ERR_INVALID;
    if (n > RT_NA_MAX_RAYS) { rt_set_error("rt_intersect_closest_noalloc: at most 2^24 rays per call (use rt_intersect_closest_dev for larger batches)"); return RT_ERR_INVALID; }
    const uint32_t s4 = RT_SPILL_WORDS(s->host.walk_stack);
    const uint32_t s2 = std::max<uint32_t>(s->host.max_stack2, 2);
    if (s4 + s2 > RT_NA_WORDS) {
        rt_set_error("rt_intersect_closest_noalloc: the scene's traversal stack bounds (" + std::to_string(s4) + " + " + std::to_string(s2) +
                     " words per lane) exceed the static scratch (" + std::to_string(RT_NA_WORDS) + "); use rt_intersect_closest_dev");
        return RT_ERR_UNSUPPORTED;
    }
    uint32_t* ptr = (s->device >= 0 && s->device < RT_NA_MAX_DEVICES) ? na_scratch_ptr[s->device] : nullptr;
    if (!ptr) { rt_set_error("rt_intersect_closest_noalloc: no static scratch on this device"); return RT_ERR_UNSUPPORTED; }
    // the scratch pointer and the scene's arrays belong to the scene's device: a launch under another current device would
    // go there with foreign pointers
    int cur_dev = -1;
    RT_HIP(hipGetDevice(&cur_dev));
    if (cur_dev != s->device) { rt_set_error("rt_intersect_closest_noalloc: the scene's device (" + std::to_string(s->device) + ") is not current (" + std::to_string(cur_dev) + ")"); return RT_ERR_INVALID; }
    if (n == 0) return RT_OK;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(na_mu);
    // Every call on a GPU shares that GPU's array.  Calls on one stream are ordered by the stream; a call on another stream
    // than the last one's waits, on the device, for the event recorded behind that launch (no allocation either way), so
    // overlapping calls are serialised instead of racing.  A stream under graph capture is left alone: the graph orders what
    // it captured, and its owner must not replay it while another call on this GPU is in flight.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (!capturing && na_busy[s->device] && na_last_stream[s->device] != st) RT_HIP(hipStreamWaitEvent(st, na_event[s->device], 0));
    // the array keeps the last call's contents: the work counter must start at zero (stack slots are written before they are read)
    RT_HIP(hipMemsetAsync(ptr, 0, sizeof(unsigned long long), st));
    const uint32_t m = (uint32_t)n;
    const unsigned blocks = (unsigned)std::min<size_t>(((size_t)m + RT_BLOCK - 1) / RT_BLOCK, std::min<size_t>(RT_NA_BLOCKS, (size_t)s->persistent_blocks));
    uint32_t* spillw = s4 ? ptr + RT_NA_HEADER : nullptr;
    uint32_t* spill2 = ptr + RT_NA_HEADER + (size_t)RT_NA_THREADS * s4;
    if (s->count_enabled)   // counted like rt_intersect_closest_dev's rays (rt_counters_read)
        hipLaunchKernelGGL(k_intersect_closest<true>, dim3(blocks), dim3(RT_BLOCK), 0, st, s->d, (const float*)d_rays, m, d_out, mode,
                           (unsigned long long*)ptr, spillw, spill2, s->d_counters);
    else
        hipLaunchKernelGGL(k_intersect_closest<false>, dim3(blocks), dim3(RT_BLOCK), 0, st, s->d, (const float*)d_rays, m, d_out, mode,
                           (unsigned long long*)ptr, spillw, spill2, s->d_counters);
    RT_HIP(hipGetLastError());
    if (!capturing) {
        RT_HIP(hipEventRecord(na_event[s->device], st));
        na_busy[s->device] = true; na_last_stream[s->device] = st;
    }
    return RT_OK;
}

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};
}  // namespace

// Host-buffer batches.  Up to RT_HOST_CHUNK items go up, through the kernel and back in one piece; larger batches are cut into
// chunks on three streams so that chunk i+1 is on its way up and chunk i-1 on its way down while chunk i is traced (the
// caller's memory is pageable, so the copies are what bounds this path: 44 B per closest-hit ray over PCIe; pinning it
// for the call with hipHostRegister measured no faster).  Device memory stays at two chunks whatever the batch size.
#define RT_HOST_CHUNK ((size_t)1 << 21)
namespace {
struct HostPipe {
    hipStream_t up = nullptr, run = nullptr, down = nullptr;
    hipEvent_t in_ready[2] = { nullptr, nullptr }, traced[2] = { nullptr, nullptr }, out_read[2] = { nullptr, nullptr };
    DevBuf din[2], dout[2];
    ~HostPipe() {
        for (hipStream_t st : { up, run, down }) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        for (int k = 0; k < 2; k++) for (hipEvent_t e : { in_ready[k], traced[k], out_read[k] }) if (e) (void)hipEventDestroy(e);
    }
};
template <class In, class Out, class Launch>
int host_batch(const In* in, size_t n, Out* out, Launch launch) {
    if (n <= RT_HOST_CHUNK) {
        DevBuf di, d_o;
        RT_HIP(hipMalloc(&di.p, n * sizeof(In)));
        RT_HIP(hipMalloc(&d_o.p, n * sizeof(Out)));
        RT_HIP(hipMemcpy(di.p, in, n * sizeof(In), hipMemcpyHostToDevice));
        int rc = launch((const In*)di.p, n, (Out*)d_o.p, nullptr);
        if (rc) return rc;
        RT_HIP(hipDeviceSynchronize());
        RT_HIP(hipMemcpy(out, d_o.p, n * sizeof(Out), hipMemcpyDeviceToHost));
        return RT_OK;
    }
    HostPipe p;
    RT_HIP(hipStreamCreate(&p.up)); RT_HIP(hipStreamCreate(&p.run)); RT_HIP(hipStreamCreate(&p.down));
    for (int k = 0; k < 2; k++) {
        RT_HIP(hipEventCreateWithFlags(&p.in_ready[k], hipEventDisableTiming)); RT_HIP(hipEventCreateWithFlags(&p.traced[k], hipEventDisableTiming));
        RT_HIP(hipEventCreateWithFlags(&p.out_read[k], hipEventDisableTiming));
        RT_HIP(hipMalloc(&p.din[k].p, RT_HOST_CHUNK * sizeof(In))); RT_HIP(hipMalloc(&p.dout[k].p, RT_HOST_CHUNK * sizeof(Out)));
    }
    const size_t chunks = (n + RT_HOST_CHUNK - 1) / RT_HOST_CHUNK;
    for (size_t i = 0; i <= chunks; i++) {
        if (i < chunks) {
            const int k = (int)(i & 1);
            const size_t off = i * RT_HOST_CHUNK, m = std::min(RT_HOST_CHUNK, n - off);
            if (i >= 2) RT_HIP(hipStreamWaitEvent(p.up, p.traced[k], 0));        // chunk i-2 no longer reads this input buffer
            RT_HIP(hipMemcpyAsync(p.din[k].p, in + off, m * sizeof(In), hipMemcpyHostToDevice, p.up));
            RT_HIP(hipEventRecord(p.in_ready[k], p.up));
            RT_HIP(hipStreamWaitEvent(p.run, p.in_ready[k], 0));
            if (i >= 2) RT_HIP(hipStreamWaitEvent(p.run, p.out_read[k], 0));    // chunk i-2's results have left this output buffer
            int rc = launch((const In*)p.din[k].p, m, (Out*)p.dout[k].p, p.run);
            if (rc) return rc;
            RT_HIP(hipEventRecord(p.traced[k], p.run));
        }
        if (i >= 1) {   // results of the previous chunk come down while this one is traced
            const int k = (int)((i - 1) & 1);
            const size_t off = (i - 1) * RT_HOST_CHUNK, m = std::min(RT_HOST_CHUNK, n - off);
            RT_HIP(hipStreamWaitEvent(p.down, p.traced[k], 0));
            RT_HIP(hipMemcpyAsync(out + off, p.dout[k].p, m * sizeof(Out), hipMemcpyDeviceToHost, p.down));
            RT_HIP(hipEventRecord(p.out_read[k], p.down));
        }
    }
    RT_HIP(hipStreamSynchronize(p.up)); RT_HIP(hipStreamSynchronize(p.run)); RT_HIP(hipStreamSynchronize(p.down));
    return RT_OK;
}
}  // namespace

extern "C" int rt_intersect_closest(rt_scene* s, const rt_ray* rays, size_t n, rt_hit* out, int mode) {
    if (!s || (n && (!rays || !out))) { rt_set_error("rt_intersect_closest: NULL argument"); return RT_ERR_INVALID; }
    if (check_mode(mode)) return RT_ERR_INVALID;
    if (n == 0) return RT_OK;
    return host_batch(rays, n, out, [&](const rt_ray* d_in, size_t m, rt_hit* d_out, hipStream_t st) {
        return rt_intersect_closest_dev(s, d_in, m, d_out, mode, st);
    });
}

extern "C" int rt_intersect_any(rt_scene* s, const rt_segment* segs, size_t n, uint8_t* out, int mode) {
    if (!s || (n && (!segs || !out))) { rt_set_error("rt_intersect_any: NULL argument"); return RT_ERR_INVALID; }
    if (check_mode(mode)) return RT_ERR_INVALID;
    if (n == 0) return RT_OK;
    return host_batch(segs, n, out, [&](const rt_segment* d_in, size_t m, uint8_t* d_out, hipStream_t st) {
        return rt_intersect_any_dev(s, d_in, m, d_out, mode, st);
    });
}

extern "C" int rt_generate_rays(rt_scene* s, const float* xy, size_t n, rt_ray* out) {
    if (!s || (n && (!xy || !out))) { rt_set_error("rt_generate_rays: NULL argument"); return RT_ERR_INVALID; }
    if (!s->has_camera) { rt_set_error("rt_generate_rays: no camera set"); return RT_ERR_INVALID; }
    if (n == 0) return RT_OK;
    DevBuf in, o;
    RT_HIP(hipMalloc(&in.p, n * 8));
    RT_HIP(hipMalloc(&o.p, n * sizeof(rt_ray)));
    RT_HIP(hipMemcpy(in.p, xy, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_generate_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, s->d, (const float*)in.p, n, (float*)o.p);
    RT_HIP(hipGetLastError());
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(out, o.p, n * sizeof(rt_ray), hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_debug_rng(uint64_t seed, uint64_t sample, uint32_t pixel, uint32_t stream, int n, float* out) {
    if (!out || n < 0) { rt_set_error("rt_debug_rng: bad argument"); return RT_ERR_INVALID; }
    Rng r; r.key = rt_rng_key(seed, sample, pixel, stream); r.ctr = 0;
    for (int i = 0; i < n; i++) out[i] = rt_rng_next(r);
    return RT_OK;
}
