/* rtamd.h — C ABI of the MI355X-native RTBase hot path (librtamd.so).
 *
 * The reference (charlescookey/RayTracer, "RTBase") has no plugin/FFI layer:
 * its hot path sits behind four C++ seams inside one translation unit.  Each
 * entry point below replaces one of those seams; the citation names the
 * reference interface it stands in for (paths relative to RTBase/).
 *
 * Conventions: plain pointers and sizes, no C++ or torch types; every call
 * returns RT_OK (0) or a negative rt_status and leaves a message retrievable
 * with rt_last_error() (the reference itself only prints / exit(0)s).  Host
 * buffers passed in are copied; the caller keeps ownership.  A scene's
 * geometry is immutable after creation and may be traversed (rt_intersect_*)
 * from several host threads and on several streams (launches that share the
 * scene's traversal scratch are chained on the device); a renderer must be driven from one thread at a
 * time (RayTracer::render is called serially from main, Main.cpp:119), and
 * rt_scene_set_camera / rt_counters_enable / creating or destroying a
 * renderer of a scene must not overlap calls on that scene's renderers.  All device memory is owned by the
 * opaque handles and lives on the HIP device current at creation time.
 */
#ifndef RTAMD_H
#define RTAMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RT_OK = 0,
    RT_ERR_INVALID = -1,   /* bad argument / inconsistent descriptor */
    RT_ERR_DEVICE = -2,    /* HIP runtime error (message has hipGetErrorString) */
    RT_ERR_NOMEM = -3,
    RT_ERR_UNSUPPORTED = -4
} rt_status;

/* BSDF classes of Materials.h (DiffuseBSDF:199 MirrorBSDF:248 ConductorBSDF:294 GlassBSDF:434
 * DielectricBSDF:530 OrenNayarBSDF:748 PlasticBSDF:842); LayeredBSDF:1008 is flag bit 0 on its base. */
enum { RT_MAT_DIFFUSE = 0, RT_MAT_MIRROR, RT_MAT_CONDUCTOR, RT_MAT_GLASS, RT_MAT_DIELECTRIC, RT_MAT_ORENNAYAR, RT_MAT_PLASTIC };
#define RT_MAT_FLAG_LAYERED 1u

typedef struct {
    uint32_t type;
    int32_t albedo_tex;     /* index into rt_scene_desc.textures (solid colours are 1x1 textures, as upstream) */
    uint32_t flags;
    float emission[3];      /* BSDF::emission; isLight() <=> Lum(emission) > 0 (Materials.h:184) */
    float int_ior, ext_ior;
    float roughness;        /* alpha = 1.62142*sqrt(roughness) is derived inside (Materials.h:307) */
    float eta[3], k[3];     /* conductor */
    float sigma;            /* oren-nayar (evaluates as Lambert upstream, Materials.h:777) */
} rt_material;

typedef struct {
    int32_t width, height;
    const float* rgb;       /* width*height*3, row-major, as Texture::texels (Imaging.h:18) */
    const float* alpha;     /* optional, NULL if absent */
} rt_texture;

typedef struct {
    uint32_t kind;          /* 0 BackgroundColour (Lights.h:87), 1 EnvironmentMap (Lights.h:138) */
    float colour[3];
    int32_t env_tex;
} rt_background;

typedef struct {
    float P[16];            /* projection, row-major (Matrix::perspective, Core.h:460) */
    float V[16];            /* camera-to-world = lookAt().invert() (SceneLoader.h:251-252) */
    int32_t width, height;
} rt_camera;

typedef struct {
    const float* verts;             /* n_tris * 3 vertices * {p.xyz, normal.xyz, u, v} (Core.h:197 Vertex) */
    const uint32_t* tri_material;   /* n_tris */
    uint32_t n_tris;
    const rt_material* materials;
    uint32_t n_materials;
    const rt_texture* textures;
    uint32_t n_textures;
    rt_background background;
} rt_scene_desc;

typedef struct { float o[3]; float d[3]; } rt_ray;          /* Ray::o, Ray::dir (Geometry.h:9-10); invDir is derived */
typedef struct { float p1[3]; float p2[3]; } rt_segment;    /* arguments of Scene::visible */
typedef struct { uint32_t ID; float t, alpha, beta, gamma; } rt_hit;  /* IntersectionData (Geometry.h:256); miss <=> t == FLT_MAX, ID = 0xFFFFFFFF */

typedef struct {
    uint32_t n_tris, n_lights;
    uint32_t bvh2_nodes, bvh2_depth, bvh2_leaves, bvh2_max_leaf;   /* the reference-topology tree (Geometry.h:391) */
    uint32_t wide_nodes, wide_max_stack;                           /* the 8-wide tree the kernels walk: nodes (80 B each), and the most 8-byte stack entries a walk can have waiting */
    uint64_t device_bytes;
    float build_seconds;
    uint32_t wide_levels;                                          /* levels of the 8-wide tree */
} rt_scene_info;

typedef struct {
    uint64_t closest_rays, any_rays;        /* rays issued (Scene::traverse / Scene::visible equivalents) */
    uint64_t nodes_visited, tris_tested;    /* closest-hit walks: wide-node fetches (80 B each: the 8-wide quantised nodes) and triangle-record fetches (64 B each) */
    uint64_t exact_path_rays;               /* rays routed through the reference-order fallback walk */
    uint64_t any_nodes_visited, any_tris_tested;  /* the same two counts for any-hit (Scene::visible) walks */
    uint64_t lane_slots, lane_busy;         /* closest-hit walks: 64 x wave traversal steps, and how many of those lanes held a ray */
    uint64_t any_lane_slots, any_lane_busy; /* the same for any-hit walks (SIMD lane occupancy = busy / slots) */
    uint64_t max_steps, any_max_steps;      /* most wave steps any single ray stayed in flight (its share of a launch's tail) */
    uint64_t gate_refused, any_gate_refused; /* answers of the conservative wide walk that failed upstream's box test on the hit's gate box and were recomputed by the exact walk */
} rt_counters;

typedef struct rt_scene rt_scene;
typedef struct rt_renderer rt_renderer;

const char* rt_last_error(void);
/* HIP device count / selection; rt_device_set must precede handle creation on that thread. */
int rt_device_count(void);
int rt_device_set(int device);

/* --- Scene::init + Scene::build (Scene.h:150, :92): Triangle::init precompute, reference-topology
 * binned-SAH BVH (Geometry.h:310-438), light list (Scene.h:107-116), flatten + upload.
 * The binned-SAH build of the binary tree (BVHNode::build, Geometry.h:391-438), Triangle::init's records and the
 * reference-topology node array all run on the GPU, level by level; the tree, the leaf order, what the walk reads
 * (derived from them on the GPU either way) and every traversal result are the same as with the host build
 * (tests/test_bvh_build_gpu.py), which rt_scene_create_ex can still ask for. */
int rt_scene_create(const rt_scene_desc* desc, rt_scene** out);
#define RT_SCENE_BUILD_DEVICE 1u   /* the default of rt_scene_create */
#define RT_SCENE_BUILD_HOST 2u     /* everything on the host (single-threaded binary tree) */
int rt_scene_create_ex(const rt_scene_desc* desc, uint32_t flags, rt_scene** out);   /* flags: one of the two, or 0 = default */
void rt_scene_destroy(rt_scene* scene);
/* --- Camera::init + Camera::updateView (Scene.h:26, :37) */
int rt_scene_set_camera(rt_scene* scene, const rt_camera* cam);
int rt_scene_get_info(const rt_scene* scene, rt_scene_info* out);
/* DFS leaf order of the reference tree (triangle indices), for structural parity checks. */
int rt_scene_leaf_order(const rt_scene* scene, uint32_t* out_n_tris);
/* The reference-topology binary tree as it sits in device memory, for structural checks (bvh2_nodes x 48 B: lo[3] left hi[3] right
 * first count subtree_first subtree_count; leaf <=> left < 0 && right < 0; a single-triangle leaf keeps the inverted box upstream
 * never initialises, Geometry.h:401). */
int rt_scene_debug_ref_nodes(rt_scene* scene, void* out, size_t max_nodes);
/* What the traversal kernels read, derived on the device from that tree and the triangles: the 8-wide tree (wide_nodes x 80 B: org[3], int8 exponents
 * ex ey ez + imask, child_base, tri_base, meta[8], plane bytes lox[8] loy[8] loz[8] hix[8] hiy[8] hiz[8]; slot s's box is
 * org + byte s x 2^e, lo rounded down, hi up; csrc/rt_common.h DNode8), the per-triangle gate boxes (n_tris x 32 B: lo[3] pad hi[3] pad, in DFS
 * leaf order: the reference box whose rayAABB (Geometry.h:207) decides whether upstream tests that triangle) and the triangle records in the
 * order the 8-wide tree names them (n_tris x 64 B: v0 v1 v2 n d 1/area rank material).  Any pointer may be NULL. */
int rt_scene_debug_walk_nodes(rt_scene* scene, void* nodes_out, size_t max_nodes, void* gate_out, size_t max_tris, void* trisw_out);
/* One byte per triangle in DFS leaf order: the BSDF class the renderer files a hit on it under (0 diffuse / Oren-Nayar, 1 mirror, 2 glass,
 * 3 conductor, 4 dielectric, 5 plastic) | 0x80 for an emitter (csrc/rt_common.h DScene::tri_cls); no counterpart upstream, for tests. */
int rt_scene_debug_tri_classes(rt_scene* scene, uint8_t* out, size_t max_tris);
/* Host-only structure query: runs the same builder without touching a device (leaf_order/info may be NULL).
 * Not a compute path: nothing can be traversed or rendered without a GPU. */
int rt_host_build_info(const rt_scene_desc* desc, uint32_t* leaf_order, rt_scene_info* info);

/* Matrix::invert (Core.h:326, the unrolled MESA inverse in fp32): host utility, row-major 4x4, no device involved.
 * The library uses the same routine for the camera's inverse matrices; loaders use it for V = lookAt().invert()
 * (SceneLoader.h:252) and the instance normal matrix (SceneLoader.h:199). */
int rt_matrix_invert(const float* m16, float* out16);

/* --- IntersectionData Scene::traverse(const Ray&) (Scene.h:118), batched.
 * `mode`: 0 = 8-wide conservative walk + gate-box check; rays outside its domain take an exact walk over upstream's own boxes (a zero /
 * denormal direction component, NaN / inf / >= 1e18 anywhere, or -- relative to the scene's largest coordinate M -- M max|1/dir| >= 1e31 or
 * M^2 max|dir| max|1/dir| >= 1e36, where upstream's own hit-point arithmetic can overflow): the answers are upstream's either way, only the
 * speed differs.  1 = reference-order walk for every ray (validation).
 * Host-buffer form copies in/out; the _dev form takes device pointers and a hipStream_t. */
int rt_intersect_closest(rt_scene* scene, const rt_ray* rays, size_t n, rt_hit* out, int mode);
int rt_intersect_closest_dev(rt_scene* scene, const rt_ray* d_rays, size_t n, rt_hit* d_out, int mode, void* stream);
/* No counterpart in the reference: rt_intersect_closest_dev for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same arguments and results, bit for bit; the
 * traversal scratch (work counter + stack overflow) comes from one static device array of ~42 MB per GPU instead of the
 * scene's own hipMalloc'ed arrays, and the call itself performs no hipMalloc / hipFree.  Limits: n <= 2^24 rays per call
 * (RT_ERR_INVALID beyond), and a scene whose stack bounds -- 2 x (wide_max_stack - 12) + the binary tree's bound -- exceed 80
 * words per lane is refused with RT_ERR_UNSUPPORTED.  Every call on a GPU shares that GPU's array: calls on one stream are
 * ordered by the stream, a call on another stream than the previous one's is made to wait for it on the device (an event
 * wait, no allocation); a stream under graph capture is left alone -- the graph orders what it captured, and must not be
 * replayed while another call on that GPU is in flight.  The scene's device must be current (RT_ERR_INVALID otherwise). */
int rt_intersect_closest_noalloc(rt_scene* scene, const rt_ray* d_rays, size_t n, rt_hit* d_out, int mode, void* stream);
/* --- bool Scene::visible(const Vec3&, const Vec3&) (Scene.h:169), batched; out[i] = 1 if visible. */
int rt_intersect_any(rt_scene* scene, const rt_segment* segs, size_t n, uint8_t* out, int mode);
int rt_intersect_any_dev(rt_scene* scene, const rt_segment* d_segs, size_t n, uint8_t* d_out, int mode, void* stream);
/* --- Ray Camera::generateRay(float x, float y) (Scene.h:47), batched: xy[n][2] -> rays */
int rt_generate_rays(rt_scene* scene, const float* xy, size_t n, rt_ray* out);
/* Traversal counters accumulated since the last reset (only counted when enabled: costs atomics). */
int rt_counters_enable(rt_scene* scene, int enable);
int rt_counters_read(rt_scene* scene, rt_counters* out, int reset);

/* --- RayTracer (Renderer.h:29): init / render / clear / getSPP / film access.
 * Render modes follow `int type` of RayTracer::render (Renderer.h:372-395). */
enum {
    RT_MODE_PATH = 0,        /* renderMT -> pathTrace (Renderer.h:397, :215) */
    RT_MODE_LIGHT = 1,       /* renderSTLight -> lightTrace (Renderer.h:327, :695), atomic splats */
    RT_MODE_MLT = 2,         /* renderSTMLT -> PSSMLTRender (Renderer.h:353, :1013) */
    RT_MODE_RADIOSITY = 4,   /* renderMTRadiosity -> pathTraceRadiosity (Renderer.h:431, :910) */
    RT_MODE_PATH_ST = 5,     /* renderST: same estimator as RT_MODE_PATH */
    RT_MODE_AOV_ALBEDO = 10, /* albedo() (Renderer.h:272) */
    RT_MODE_AOV_NORMAL = 11, /* viewNormals() (Renderer.h:287) */
    RT_MODE_DIRECT = 12      /* direct() (Renderer.h:257) */
};
int rt_renderer_create(rt_scene* scene, rt_renderer** out);              /* RayTracer::init (Renderer.h:56) */
void rt_renderer_destroy(rt_renderer* r);
/* Adds `spp` samples per pixel (sample indices first_sample .. first_sample+spp-1 of RNG stream `seed`)
 * to the film and increments SPP by spp: `spp` consecutive RayTracer::render() calls. */
int rt_render(rt_renderer* r, int mode, int spp, uint64_t seed, uint64_t first_sample);
/* The film's ImageFilter for later rt_render calls: what `film->init(w, h, new BoxFilter())` (Renderer.h:61)
 * chooses upstream.  Film::splat (Imaging.h:224-247) spreads every sample over (2 size + 1)^2 pixels with
 * weights filter(px - x, py - y) / sum.  The AOV and direct passes keep their BoxFilter (Renderer.h:100-103). */
enum {
    RT_FILTER_BOX = 0,       /* BoxFilter, size 0 (Imaging.h:138): the default */
    RT_FILTER_GAUSSIAN = 1,  /* GaussianFilter, size 1 (Imaging.h:159) */
    RT_FILTER_MITCHELL = 2   /* MitchellNetravaliFilter, size 2 (Imaging.h:179) */
};
int rt_renderer_set_filter(rt_renderer* r, int filter);
/* albedoFilm / normalFilm (Renderer.h:100-103) as getTileFilms fills them for a denoiser (Renderer.h:577-603): `spp` samples of
 * RT_MODE_AOV_ALBEDO / RT_MODE_AOV_NORMAL (or RT_MODE_DIRECT) into a film of their own, returned un-normalised in
 * rgb[width*height*3]; the renderer's film and SPP are untouched. */
int rt_render_aov(rt_renderer* r, int mode, int spp, uint64_t seed, uint64_t first_sample, float* rgb);
/* How many HIP streams rt_render spreads its batches over: 2 (the default: the end of one batch's persistent kernels overlaps
 * the other batch's work) or 1 (every kernel runs alone: the unshared per-kernel durations bench.py reports beside the shared
 * ones).  Same film either way.  No counterpart upstream. */
int rt_renderer_set_streams(rt_renderer* r, int n);
int rt_film_clear(rt_renderer* r);                                        /* RayTracer::clear (Renderer.h:107) */
int rt_film_spp(const rt_renderer* r);                                    /* RayTracer::getSPP (Renderer.h:652) */
/* Un-normalised accumulation film (Film::film, Imaging.h:219) + SPP; rgb = width*height*3 floats. */
int rt_film_read(rt_renderer* r, float* rgb, int* spp);
/* Render into caller-owned device memory (width*height*3 floats), e.g. a torch tensor that is then
 * all-reduced over RCCL; NULL re-binds the internal film. */
int rt_film_bind(rt_renderer* r, float* d_film);
int rt_film_set_spp(rt_renderer* r, int spp);                             /* after an external film all-reduce */
/* RayTracer::traceVPLs (Renderer.h:772); explicit because upstream never calls it (Renderer.h:68). */
int rt_vpl_trace(rt_renderer* r, int n_vpls, uint64_t seed);
int rt_vpl_count(const rt_renderer* r);
/* Rays issued / node + triangle fetches by rt_render since the last reset. */
int rt_render_counters(rt_renderer* r, rt_counters* out, int reset);
int rt_renderer_sync(rt_renderer* r);
/* Per-kernel timing: HIP events recorded on the renderer's own stream around every launch of rt_render.
 * ms7/launches7 index: 0 generate, 1 extend (closest hit), 2 shade, 3 connect (any hit), 4 VPL gather,
 * 5 film accumulate, 6 control.  No counterpart upstream (Main.cpp:118 times render() with a wall clock). */
int rt_profile_enable(rt_renderer* r, int enable);
int rt_profile_read(rt_renderer* r, double* ms7, uint32_t* launches7, int reset);
/* Film::save (Imaging.h:278): Radiance .hdr of film/SPP; Film::tonemap (Imaging.h:251) to 8-bit RGB. */
int rt_film_save_hdr(rt_renderer* r, const char* filename);
int rt_film_tonemap(rt_renderer* r, uint8_t* rgb8);

/* --- debug probes used by the parity tests: evaluate BSDF::sample / evaluate+PDF on the device for
 * arrays of inputs (wo, shading normal, uv, rng key/counter). */
int rt_debug_bsdf_sample(rt_scene* scene, uint32_t material, const float* wo, const float* n, const float* uv,
                         const uint32_t* rng_key_ctr, size_t count, float* out7);
int rt_debug_bsdf_eval(rt_scene* scene, uint32_t material, const float* wo, const float* n, const float* uv,
                       const float* wi, size_t count, float* out4);
int rt_debug_rng(uint64_t seed, uint64_t sample, uint32_t pixel, uint32_t stream, int n, float* out);
/* EnvironmentMap's tabulated importance sampling (Lights.h:207-424) on the device, for scenes with an environment map:
 * in3[count][3] -> out4[count][4].  which = 0: sample() / sampleDirectionFromLight() after their two draws (in = u0, u1, -;
 * out = wi.xyz, pdf); 1: SampleContinuous (out = d0, d1, pdf, 0); 2: PDF(wi) (in = wi; out[0]). */
int rt_debug_env(rt_scene* scene, int which, const float* in3, size_t count, float* out4);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_H */
