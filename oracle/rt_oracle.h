/* TEST INFRASTRUCTURE — CPU oracle for the RTBase hot path.  NOT the product.
 *
 * A plain-C restatement of the reference's algorithms (charlescookey/RayTracer,
 * /root/reference/RTBase headers).  Every function in rt_oracle.c cites the
 * reference file:line it follows.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (raytracer_amd/csrc) never links, includes or calls it.
 *
 * Parity pin status:
 *   - geometry / BVH / traversal / sampling / frame / matrix: PINNED against the
 *     reference's own Geometry.h+Core.h+Sampling.h compiled in oracle/_ref
 *     (tests/test_oracle_vs_ref.py) and against tests/golden npz files generated
 *     from that build, plus the three RTTest.cpp vectors.
 *   - camera / shading data / BSDFs / lights / integrators / film: restated by
 *     reading; the reference headers for these do not compile in this image
 *     without stand-ins (Annex-K sprintf_s, <Windows.h>, OIDN) => PARITY UNPINNED
 *     against a reference binary; checked by analytic/statistical tests only.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Descriptor layouts are deliberately byte-identical to include/rtamd.h so the
 * same ctypes structures can feed both libraries. */
typedef struct {
    uint32_t type;      /* 0 diffuse 1 mirror 2 conductor 3 glass 4 dielectric 5 orennayar 6 plastic */
    int32_t albedo_tex;
    uint32_t flags;     /* bit0: wrapped in LayeredBSDF (Materials.h:1008) */
    float emission[3];
    float int_ior, ext_ior;
    float roughness;
    float eta[3];
    float k[3];
    float sigma;
} orc_material;

typedef struct {
    int32_t width, height;
    const float* rgb;   /* width*height*3 */
    const float* alpha; /* optional */
} orc_texture;

typedef struct {
    uint32_t kind;      /* 0 BackgroundColour, 1 EnvironmentMap */
    float colour[3];
    int32_t env_tex;
} orc_background;

typedef struct {
    float P[16];        /* projection (Matrix::perspective, Core.h:460) */
    float V[16];        /* camera-to-world, i.e. lookAt().invert() (SceneLoader.h:251-252) */
    int32_t width, height;
} orc_camera;

typedef struct { uint32_t ID; float t, alpha, beta, gamma; } orc_hit; /* Geometry.h:256 */

typedef struct {
    uint64_t closest_rays, any_rays; /* Scene::traverse / Scene::visible calls */
    uint64_t nodes_visited, tris_tested; /* reference-tree visits (rayAABB calls / rayIntersect calls) */
} orc_stats;

typedef struct orc_scene orc_scene;

orc_scene* orc_scene_create(const float* verts, const uint32_t* tri_mat, uint32_t ntris,
                            const orc_material* mats, uint32_t nmats,
                            const orc_texture* tex, uint32_t ntex,
                            const orc_background* bg);
void orc_scene_destroy(orc_scene*);
void orc_scene_set_camera(orc_scene*, const orc_camera*);
/* stats: [0] max depth, [1] nodes, [2] leaves, [3] lights, [4] max leaf size */
void orc_scene_info(orc_scene*, int* stats5);
void orc_leaf_order(orc_scene*, uint32_t* out);
void orc_tri_derived(orc_scene*, uint32_t i, float* out11);

void orc_traverse(orc_scene*, const float* rays6, size_t n, orc_hit* out);
void orc_traverse_brute(orc_scene*, const float* rays6, size_t n, orc_hit* out);
void orc_visible(orc_scene*, const float* segs6, size_t n, uint8_t* out);
void orc_traverse_mt(orc_scene*, const float* rays6, size_t n, orc_hit* out, int nthreads);
void orc_visible_mt(orc_scene*, const float* segs6, size_t n, uint8_t* out, int nthreads);

void orc_generate_rays(orc_scene*, const float* xy, size_t n, float* rays6);
int  orc_project(orc_scene*, const float* p3, float* xy);

/* modes follow RayTracer::render's `type` (Renderer.h:372-395):
 * 0 path tracing, 1 light tracing, 2 PSSMLT, 4 instant radiosity, 5 ST path tracing
 * (== 0 here: the RNG is counter based), 10 albedo AOV, 11 normal AOV, 12 direct only */
void orc_mlt_render(orc_scene* s, int spp, uint64_t seed, uint64_t first_sample, float* film); /* PSSMLT, Renderer.h:964-1063 */
void orc_mlt_trace(uint64_t seed, uint64_t sample, int n, float* out4); /* probe: step draw, x, y in [0,1), accept draw per proposal */
void orc_render(orc_scene*, int mode, int spp, uint64_t seed, uint64_t first_sample,
                float* film_rgb, int nthreads, orc_stats* stats);
int  orc_vpl_trace(orc_scene*, int n_vpls, uint64_t seed);
int  orc_vpl_count(orc_scene*);
void orc_set_filter(orc_scene*, int kind); /* film filter of later renders: 0 Box, 1 Gaussian, 2 MitchellNetravali (Imaging.h:138-214) */
float orc_filter_eval(int kind, float x, float y); /* ImageFilter::filter */
void orc_vpl_get(orc_scene*, float* out /* n*[x3, n3, Le3, mat] = 10 floats */);

/* single-function probes for unit tests */
void orc_bsdf_sample(orc_scene*, uint32_t mat, const float* wo3, const float* n3, const float* uv2,
                     uint32_t rng_key, uint32_t rng_ctr, float* out7 /* wi3, f3, pdf */);
void orc_bsdf_eval(orc_scene*, uint32_t mat, const float* wo3, const float* n3, const float* uv2,
                   const float* wi3, float* out4 /* f3, pdf */);
void orc_rng_floats(uint64_t seed, uint64_t sample, uint32_t pixel, uint32_t stream, int n, float* out);
void orc_sampling(int which, float r1, float r2, float* out4);
void orc_frame(const float* n3, const float* v3, float* out15);
void orc_matrix(int which, const float* in, float* out16);
void orc_tri_sample(const float* tri24, float r1, float r2, float* p3, float* pdf);
void orc_tri_shading(const float* tri24, float a, float b, float c, float* out8);
int  orc_tri_intersect(const float* tri24, const float* ray6, float* tuv);
int  orc_aabb_ray(const float* bmin, const float* bmax, const float* ray6, float* t);
float orc_aabb_default_area(void);
void orc_texture_sample(orc_scene*, int tex, float u, float v, float* rgb);

#ifdef __cplusplus
}
#endif
#endif
