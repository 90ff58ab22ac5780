// rtamd.hpp — C++ host mirror of the RTBase interfaces the hot path sits behind, over the C ABI (include/rtamd.h).
//
// Same names, argument meaning and call order as upstream so a driver written against RTBase reads the same:
//   Camera::init / updateView          (Scene.h:26, :37)
//   Scene::init / build / traverse / visible   (Scene.h:150, :92, :118, :169)
//   RayTracer::init / render / clear / getSPP / saveHDR / savePNG, `int type`   (Renderer.h:56, :374, :107, :652, :657, :662, :372)
//   loadScene(name)                    (SceneLoader.h:236)
// Differences that come with the boundary: traversal also has batched forms (one launch per batch instead of one call
// per ray), errors throw rtamd::Error carrying rt_last_error() (upstream prints / exit(0)s), and nothing here computes:
// every intersection and every sample runs in librtamd.so on the GPU, and a missing device is an exception, not a fallback.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtamd.h"

namespace rtamd {

struct Error : std::runtime_error {
    int status;
    Error(const std::string& where, int st) : std::runtime_error(where + ": " + rt_last_error()), status(st) {}
};
inline void check(int st, const char* where) { if (st != RT_OK) throw Error(where, st); }

struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() {}
    Vec3(float X, float Y, float Z) : x(X), y(Y), z(Z) {}
    Vec3 operator+(const Vec3& b) const { return Vec3(x + b.x, y + b.y, z + b.z); }
    Vec3 operator-(const Vec3& b) const { return Vec3(x - b.x, y - b.y, z - b.z); }
    Vec3 operator*(float s) const { return Vec3(x * s, y * s, z * s); }
    float dot(const Vec3& b) const { return x * b.x + y * b.y + z * b.z; }
    Vec3 cross(const Vec3& b) const { return Vec3(y * b.z - z * b.y, z * b.x - x * b.z, x * b.y - y * b.x); }
    Vec3 normalize() const { float l = 1.0f / sqrtf(dot(*this)); return *this * l; }
};
struct Colour { float r = 0, g = 0, b = 0; Colour() {} Colour(float R, float G, float B) : r(R), g(G), b(B) {} };
struct Vertex { Vec3 p, normal; float u = 0, v = 0; };                      // Core.h:197
struct Ray { Vec3 o, dir; Ray() {} Ray(Vec3 O, Vec3 D) : o(O), dir(D) {} };   // Geometry.h:6 (invDir is derived on the device)
typedef rt_hit IntersectionData;                                            // Geometry.h:256: ID, t, alpha, beta, gamma

// Row-major 4x4 with the two constructors loadScene uses (Core.h:439, :460) and a general inverse.
struct Matrix {
    float m[16];
    Matrix() { memset(m, 0, sizeof(m)); m[0] = m[5] = m[10] = m[15] = 1.0f; }
    float& a(int r, int c) { return m[r * 4 + c]; }
    static Matrix lookAt(const Vec3& from, const Vec3& to, const Vec3& up) {
        Matrix M;
        Vec3 dir = (from - to).normalize(), left = up.cross(dir).normalize(), newUp = dir.cross(left);
        M.a(0, 0) = left.x; M.a(0, 1) = left.y; M.a(0, 2) = left.z; M.a(0, 3) = -from.dot(left);
        M.a(1, 0) = newUp.x; M.a(1, 1) = newUp.y; M.a(1, 2) = newUp.z; M.a(1, 3) = -from.dot(newUp);
        M.a(2, 0) = dir.x; M.a(2, 1) = dir.y; M.a(2, 2) = dir.z; M.a(2, 3) = -from.dot(dir);
        return M;
    }
    static Matrix perspective(float n, float f, float aspect, float fov) {  // fov in degrees
        Matrix M; memset(M.m, 0, sizeof(M.m));
        float t = 1.0f / tanf(fov * 0.5f * 3.141592654f / 180.0f);
        M.a(0, 0) = t / aspect; M.a(1, 1) = t; M.a(2, 2) = -f / (f - n); M.a(2, 3) = -(f * n) / (f - n); M.a(3, 2) = -1.0f;
        return M;
    }
    Matrix invert() const { Matrix R; if (rt_matrix_invert(m, R.m) != 0) throw std::runtime_error("Matrix::invert"); return R; }  // Core.h:326, the library's fp32 MESA inverse
};

class Camera {  // Scene.h:14.  The derived members (inverse projection, origin, viewDirection, Afilm) are computed by the library.
public:
    Matrix projectionMatrix, camera;
    float width = 0, height = 0;
    void init(Matrix ProjectionMatrix, int screenwidth, int screenheight) { projectionMatrix = ProjectionMatrix; width = (float)screenwidth; height = (float)screenheight; }
    void updateView(Matrix V) { camera = V; }
    rt_camera abi() const {
        rt_camera c; memcpy(c.P, projectionMatrix.m, sizeof(c.P)); memcpy(c.V, camera.m, sizeof(c.V));
        c.width = (int32_t)width; c.height = (int32_t)height;
        return c;
    }
};

struct Texture { int width = 1, height = 1; std::vector<float> rgb; };  // Imaging.h:11 (fp32 texels)
inline Texture solidTexture(Colour c) { Texture t; t.rgb = { c.r, c.g, c.b }; return t; }
struct Triangle { Vertex vertices[3]; unsigned int materialIndex = 0; };  // what Triangle::init takes (Geometry.h:66)

class Scene {  // Scene.h:82
public:
    std::vector<Triangle> triangles;
    std::vector<rt_material> materials;   // BSDF table: type tag + parameters (Materials.h), texture by index
    std::vector<Texture> textures;
    rt_background background;
    Camera camera;
    rt_scene* handle = nullptr;

    Scene() { memset(&background, 0, sizeof(background)); background.env_tex = -1; }
    ~Scene() { if (handle) rt_scene_destroy(handle); }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;

    void init(std::vector<Triangle> meshTriangles, std::vector<rt_material> meshMaterials, std::vector<Texture> meshTextures, rt_background _background) {
        triangles = std::move(meshTriangles); materials = std::move(meshMaterials); textures = std::move(meshTextures); background = _background;
    }
    // Scene::build (Scene.h:92): BVH + light list, here also the upload; the camera must be set (loadScene does that first too).
    void build() {
        std::vector<float> verts(triangles.size() * 24);
        std::vector<uint32_t> tm(triangles.size());
        for (size_t i = 0; i < triangles.size(); i++) {
            for (int k = 0; k < 3; k++) {
                const Vertex& v = triangles[i].vertices[k];
                float* d = &verts[i * 24 + k * 8];
                d[0] = v.p.x; d[1] = v.p.y; d[2] = v.p.z; d[3] = v.normal.x; d[4] = v.normal.y; d[5] = v.normal.z; d[6] = v.u; d[7] = v.v;
            }
            tm[i] = triangles[i].materialIndex;
        }
        buildFlat(verts.data(), tm.data(), (uint32_t)triangles.size());
    }
    void buildFlat(const float* verts, const uint32_t* tri_material, uint32_t n_tris) {
        std::vector<rt_texture> tex(textures.size());
        for (size_t i = 0; i < textures.size(); i++) { tex[i].width = textures[i].width; tex[i].height = textures[i].height; tex[i].rgb = textures[i].rgb.data(); tex[i].alpha = nullptr; }
        rt_scene_desc d; memset(&d, 0, sizeof(d));
        d.verts = verts; d.tri_material = tri_material; d.n_tris = n_tris;
        d.materials = materials.data(); d.n_materials = (uint32_t)materials.size();
        d.textures = tex.data(); d.n_textures = (uint32_t)tex.size();
        d.background = background;
        if (handle) { rt_scene_destroy(handle); handle = nullptr; }
        check(rt_scene_create(&d, &handle), "Scene::build");
        rt_camera c = camera.abi();
        check(rt_scene_set_camera(handle, &c), "Scene::build (camera)");
    }
    rt_scene_info info() const { rt_scene_info i; check(rt_scene_get_info(handle, &i), "Scene::info"); return i; }
    // Upstream's renderer reads Scene::camera directly; here a changed view has to reach the device (RTCamera does this).
    void pushCamera() { if (handle) { rt_camera c = camera.abi(); check(rt_scene_set_camera(handle, &c), "Scene::pushCamera"); } }
    // |bounds.max - bounds.min| of Scene::init's box over all vertices (Scene.h:152-158), for RTCamera::movespeed
    float boundsDiagonal() const {
        if (triangles.empty()) return 0.0f;
        Vec3 lo(3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f), hi(-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f);
        for (const Triangle& t : triangles) for (int k = 0; k < 3; k++) {
            const Vec3& p = t.vertices[k].p;
            lo = Vec3(lo.x < p.x ? lo.x : p.x, lo.y < p.y ? lo.y : p.y, lo.z < p.z ? lo.z : p.z);
            hi = Vec3(hi.x > p.x ? hi.x : p.x, hi.y > p.y ? hi.y : p.y, hi.z > p.z ? hi.z : p.z);
        }
        Vec3 d = hi - lo;
        return sqrtf(d.dot(d));
    }

    // IntersectionData Scene::traverse(const Ray&) (Scene.h:118); miss <=> t == FLT_MAX
    std::vector<IntersectionData> traverse(const std::vector<Ray>& rays) const {
        static_assert(sizeof(Ray) == sizeof(rt_ray), "Ray layout");
        std::vector<IntersectionData> out(rays.size());
        check(rt_intersect_closest(handle, reinterpret_cast<const rt_ray*>(rays.data()), rays.size(), out.data(), 0), "Scene::traverse");
        return out;
    }
    IntersectionData traverse(const Ray& ray) const { return traverse(std::vector<Ray>{ ray })[0]; }
    // bool Scene::visible(const Vec3& p1, const Vec3& p2) (Scene.h:169)
    std::vector<uint8_t> visible(const std::vector<rt_segment>& segs) const {
        std::vector<uint8_t> out(segs.size());
        check(rt_intersect_any(handle, segs.data(), segs.size(), out.data(), 0), "Scene::visible");
        return out;
    }
    bool visible(const Vec3& p1, const Vec3& p2) const {
        rt_segment s = { { p1.x, p1.y, p1.z }, { p2.x, p2.y, p2.z } };
        return visible(std::vector<rt_segment>{ s })[0] != 0;
    }
};

class RayTracer {  // Renderer.h:29
public:
    Scene* scene = nullptr;
    rt_renderer* handle = nullptr;
    int type = 0;           // Renderer.h:372: 0 path (MT), 1 light tracing, 2 PSSMLT, 4 instant radiosity, 5 path (ST); 3 (+denoise) is out of scope
    uint64_t seed = 1;      // upstream seeds MTRandom per thread (Renderer.h:66); here one seed names the whole counter-based stream
    int samplesPerCall = 1; // render() adds this many spp (upstream: one per call)
    ~RayTracer() { if (handle) rt_renderer_destroy(handle); }
    void init(Scene* _scene, void* /*canvas*/ = nullptr) {
        scene = _scene;
        if (handle) { rt_renderer_destroy(handle); handle = nullptr; }
        check(rt_renderer_create(scene->handle, &handle), "RayTracer::init");
    }
    void clear() { check(rt_film_clear(handle), "RayTracer::clear"); }
    // film->init(w, h, new BoxFilter()) (Renderer.h:61) with another ImageFilter: RT_FILTER_BOX / _GAUSSIAN / _MITCHELL (Imaging.h:138-214)
    void setFilter(int filter) { check(rt_renderer_set_filter(handle, filter), "RayTracer::setFilter"); }
    // albedoFilm / normalFilm as getTileFilms fills them (Renderer.h:577-603): one AOV sample per pixel each, films of their own
    void getAOVFilms(std::vector<float>& albedo, std::vector<float>& normal) {
        albedo.assign((size_t)scene->camera.width * (size_t)scene->camera.height * 3, 0.0f); normal.assign(albedo.size(), 0.0f);
        check(rt_render_aov(handle, RT_MODE_AOV_ALBEDO, 1, seed, 0, albedo.data()), "RayTracer::getAOVFilms");
        check(rt_render_aov(handle, RT_MODE_AOV_NORMAL, 1, seed, 0, normal.data()), "RayTracer::getAOVFilms");
    }
    void traceVPLs(int N) { check(rt_vpl_trace(handle, N, seed), "RayTracer::traceVPLs"); }   // Renderer.h:772
    void render() {
        if (type == 3) throw std::runtime_error("RayTracer::render: type 3 (OIDN denoise) is outside the hot path");
        int first = rt_film_spp(handle);
        check(rt_render(handle, type, samplesPerCall, seed, (uint64_t)first), "RayTracer::render");
    }
    int getSPP() const { return rt_film_spp(handle); }
    void sync() { check(rt_renderer_sync(handle), "RayTracer::sync"); }
    std::vector<float> film(int* spp = nullptr) {   // un-normalised sum, as Film::film (Imaging.h:219)
        std::vector<float> f((size_t)scene->camera.width * (size_t)scene->camera.height * 3);
        check(rt_film_read(handle, f.data(), spp), "RayTracer::film");
        return f;
    }
    void saveHDR(std::string filename) { check(rt_film_save_hdr(handle, filename.c_str()), "RayTracer::saveHDR"); }
    // Film::tonemap per pixel (Imaging.h:251) + an uncompressed PNG (stored deflate blocks; upstream uses stb_image_write)
    void savePNG(std::string filename) {
        int W = (int)scene->camera.width, H = (int)scene->camera.height;
        std::vector<uint8_t> rgb((size_t)W * H * 3);
        check(rt_film_tonemap(handle, rgb.data()), "RayTracer::savePNG");
        std::vector<uint8_t> raw; raw.reserve((size_t)(W * 3 + 1) * H);
        for (int y = 0; y < H; y++) { raw.push_back(0); raw.insert(raw.end(), rgb.begin() + (size_t)y * W * 3, rgb.begin() + (size_t)(y + 1) * W * 3); }
        std::vector<uint8_t> z = { 0x78, 0x01 };
        uint32_t a1 = 1, a2 = 0;
        for (size_t i = 0; i < raw.size(); i++) { a1 = (a1 + raw[i]) % 65521u; a2 = (a2 + a1) % 65521u; }
        for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
            size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
            z.push_back(off + n >= raw.size() ? 1 : 0);
            z.push_back((uint8_t)(n & 255)); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)(~n & 255)); z.push_back((uint8_t)((~n >> 8) & 255));
            z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
            if (raw.empty()) break;
        }
        uint32_t ad = (a2 << 16) | a1;
        for (int k = 3; k >= 0; k--) z.push_back((uint8_t)(ad >> (8 * k)));
        FILE* f = fopen(filename.c_str(), "wb");
        if (!f) throw std::runtime_error("RayTracer::savePNG: cannot open " + filename);
        static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
        fwrite(sig, 1, 8, f);
        auto chunk = [&](const char* tag, const std::vector<uint8_t>& data) {
            std::vector<uint8_t> b(tag, tag + 4); b.insert(b.end(), data.begin(), data.end());
            uint32_t crc = 0xFFFFFFFFu;
            for (uint8_t v : b) { crc ^= v; for (int k = 0; k < 8; k++) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u))); }
            crc = ~crc;
            uint32_t n = (uint32_t)data.size();
            uint8_t len[4] = { (uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n };
            uint8_t c4[4] = { (uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc };
            fwrite(len, 1, 4, f); fwrite(b.data(), 1, b.size(), f); fwrite(c4, 1, 4, f);
        };
        std::vector<uint8_t> ihdr = { (uint8_t)(W >> 24), (uint8_t)(W >> 16), (uint8_t)(W >> 8), (uint8_t)W, (uint8_t)(H >> 24), (uint8_t)(H >> 16), (uint8_t)(H >> 8), (uint8_t)H, 8, 2, 0, 0, 0 };
        chunk("IHDR", ihdr); chunk("IDAT", z); chunk("IEND", {});
        fclose(f);
    }
};

// ---- scenes (loadScene itself is in gem.hpp).  Besides upstream's scene.json + .gem directories there are the flat RTSC
// file raytracer_amd/abi.py:save_rtsc writes and a built-in "cornell-box"; set-up order as SceneLoader.h:236: camera first,
// then init, then build.
inline rt_material diffuseMaterial(int tex, Colour emission = Colour()) {
    rt_material m; memset(&m, 0, sizeof(m));
    m.type = RT_MAT_DIFFUSE; m.albedo_tex = tex; m.emission[0] = emission.r; m.emission[1] = emission.g; m.emission[2] = emission.b;
    m.int_ior = 1.5f; m.ext_ior = 1.0f; m.sigma = 1.0f;
    return m;
}
inline void addQuad(std::vector<Triangle>& out, Vec3 o, Vec3 eu, Vec3 ev, Vec3 n, unsigned int mat) {
    Vec3 p[4] = { o, o + eu, o + eu + ev, o + ev };
    float uv[4][2] = { { 0, 0 }, { 1, 0 }, { 1, 1 }, { 0, 1 } };
    const int idx[2][3] = { { 0, 1, 2 }, { 0, 2, 3 } };
    for (int t = 0; t < 2; t++) {
        Triangle tr; tr.materialIndex = mat;
        for (int k = 0; k < 3; k++) { Vertex& v = tr.vertices[k]; v.p = p[idx[t][k]]; v.normal = n; v.u = uv[idx[t][k]][0]; v.v = uv[idx[t][k]][1]; }
        out.push_back(tr);
    }
}
inline void addBox(std::vector<Triangle>& out, Vec3 lo, Vec3 hi, unsigned int mat) {
    Vec3 d = hi - lo;
    addQuad(out, Vec3(lo.x, lo.y, hi.z), Vec3(d.x, 0, 0), Vec3(0, d.y, 0), Vec3(0, 0, 1), mat);
    addQuad(out, Vec3(hi.x, lo.y, lo.z), Vec3(-d.x, 0, 0), Vec3(0, d.y, 0), Vec3(0, 0, -1), mat);
    addQuad(out, Vec3(hi.x, lo.y, hi.z), Vec3(0, 0, -d.z), Vec3(0, d.y, 0), Vec3(1, 0, 0), mat);
    addQuad(out, Vec3(lo.x, lo.y, lo.z), Vec3(0, 0, d.z), Vec3(0, d.y, 0), Vec3(-1, 0, 0), mat);
    addQuad(out, Vec3(lo.x, hi.y, hi.z), Vec3(d.x, 0, 0), Vec3(0, 0, -d.z), Vec3(0, 1, 0), mat);
    addQuad(out, Vec3(lo.x, lo.y, lo.z), Vec3(d.x, 0, 0), Vec3(0, 0, d.z), Vec3(0, -1, 0), mat);
}
// RTCamera (SceneLoader.h:8-88): the fly camera Main.cpp drives with W/S/A/D/E/Q (Main.cpp:87-116).  Same steps and the same
// Rodrigues rotation about `up`; updateCamera() rebuilds V = lookAt(from, to, up).invert(), hands it to Camera::updateView
// and -- what upstream gets for free from sharing the Camera object -- pushes it to the device scene.
class RTCamera {
public:
    Vec3 from, to, up;
    Camera* camera = nullptr;
    Scene* scene = nullptr;     // not upstream: whose device copy to refresh
    float movespeed = 1.0f;
    float rotspeed = 5.0f;
    void forward() { Vec3 dir = (to - from).normalize() * movespeed; from = from + dir; to = from + dir; updateCamera(); }
    void back() { Vec3 dir = (to - from).normalize() * movespeed; from = from - dir; to = from + dir; updateCamera(); }
    void left() { turn(rotspeed * ((float)M_PI / 180.0f)); }
    void right() { turn(-rotspeed * ((float)M_PI / 180.0f)); }
    void flyUp() { Vec3 dir = up * movespeed; from = from + dir; to = to + dir; updateCamera(); }
    void flyDown() { Vec3 dir = up * movespeed; from = from - dir; to = to - dir; updateCamera(); }
    void updateCamera() {
        Matrix V = Matrix::lookAt(from, to, up).invert();
        if (camera) camera->updateView(V);
        if (scene) scene->pushCamera();
    }
private:
    void turn(float rad) {   // SceneLoader.h:38-69; upstream forms rotspeed * (M_PI / 180.0f) in double and narrows it here
        Vec3 dir = (to - from).normalize();
        float cosTheta = cosf(rad), sinTheta = sinf(rad);
        Vec3 k = up;
        Vec3 rotated = (dir * cosTheta) + (k.cross(dir) * sinTheta) + (k * (k.dot(dir) * (1 - cosTheta)));
        to = from + rotated;
        updateCamera();
    }
};
inline RTCamera viewcamera;   // SceneLoader.h:90

inline void setCamera(Scene* scene, Vec3 from, Vec3 to, Vec3 up, float fov, int width, int height, bool flipX = false) {
    Matrix P = Matrix::perspective(0.001f, 10000.0f, (float)width / (float)height, fov);   // SceneLoader.h:244
    if (flipX) P.a(0, 0) = -P.a(0, 0);
    Matrix V = Matrix::lookAt(from, to, up).invert();                                        // SceneLoader.h:251-252
    scene->camera.init(P, width, height);
    scene->camera.updateView(V);
    viewcamera.from = from; viewcamera.to = to; viewcamera.up = up;   // SceneLoader.h:261-264
    viewcamera.camera = &scene->camera; viewcamera.scene = scene;
}
inline Scene* cornellBox(int width, int height) {
    Scene* scene = new Scene();
    setCamera(scene, Vec3(0, 1.0f, 4.4f), Vec3(0, 1.0f, 0), Vec3(0, 1, 0), 40.0f, width, height);
    std::vector<Texture> tex = { solidTexture(Colour(0.73f, 0.73f, 0.73f)), solidTexture(Colour(0.65f, 0.05f, 0.05f)),
                                 solidTexture(Colour(0.12f, 0.45f, 0.15f)), solidTexture(Colour(1, 1, 1)) };
    std::vector<rt_material> mats = { diffuseMaterial(0), diffuseMaterial(1), diffuseMaterial(2), diffuseMaterial(3, Colour(17.0f, 12.0f, 4.0f)) };
    std::vector<Triangle> tris;
    const float W = 1.0f, H = 2.0f;
    addQuad(tris, Vec3(-W, 0, W), Vec3(2 * W, 0, 0), Vec3(0, 0, -2 * W), Vec3(0, 1, 0), 0);
    addQuad(tris, Vec3(-W, H, -W), Vec3(2 * W, 0, 0), Vec3(0, 0, 2 * W), Vec3(0, -1, 0), 0);
    addQuad(tris, Vec3(-W, 0, -W), Vec3(2 * W, 0, 0), Vec3(0, H, 0), Vec3(0, 0, 1), 0);
    addQuad(tris, Vec3(-W, 0, W), Vec3(0, 0, -2 * W), Vec3(0, H, 0), Vec3(1, 0, 0), 1);
    addQuad(tris, Vec3(W, 0, -W), Vec3(0, 0, 2 * W), Vec3(0, H, 0), Vec3(-1, 0, 0), 2);
    addQuad(tris, Vec3(-0.25f, H - 0.01f, -0.25f), Vec3(0.5f, 0, 0), Vec3(0, 0, 0.5f), Vec3(0, -1, 0), 3);
    addBox(tris, Vec3(0.15f, 0.0f, 0.1f), Vec3(0.75f, 0.6f, 0.7f), 0);
    addBox(tris, Vec3(-0.75f, 0.0f, -0.75f), Vec3(-0.15f, 1.2f, -0.15f), 0);
    rt_background bg; memset(&bg, 0, sizeof(bg)); bg.env_tex = -1;   // BackgroundColour(0,0,0) (SceneLoader.h:278)
    scene->init(tris, mats, tex, bg);
    return scene;
}
// Host-side half of a scene: what Scene::init + Camera::init/updateView take, no device involved yet.
inline Scene* readRTSC(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("loadScene: cannot open " + path);
    auto rd = [&](void* p, size_t n) { if (n && fread(p, 1, n, f) != n) { fclose(f); throw std::runtime_error("loadScene: truncated " + path); } };
    uint32_t head[5]; rd(head, sizeof(head));
    if (head[0] != 0x43535452u || head[1] != 1u) { fclose(f); throw std::runtime_error("loadScene: not an RTSC v1 file: " + path); }
    int32_t wh[2]; rd(wh, sizeof(wh));
    Scene* scene = new Scene();
    Matrix P, V; rd(P.m, sizeof(P.m)); rd(V.m, sizeof(V.m));
    scene->camera.init(P, wh[0], wh[1]);
    scene->camera.updateView(V);
    rt_background bg; memset(&bg, 0, sizeof(bg));
    rd(&bg.kind, 4); rd(bg.colour, 12); rd(&bg.env_tex, 4);
    std::vector<rt_material> mats(head[3]); rd(mats.data(), sizeof(rt_material) * mats.size());
    std::vector<Texture> tex(head[4]);
    for (auto& t : tex) {
        int32_t h3[3]; rd(h3, sizeof(h3));
        if (h3[0] <= 0 || h3[1] <= 0 || h3[2] != 0) { fclose(f); throw std::runtime_error("loadScene: bad texture header in " + path); }
        t.width = h3[0]; t.height = h3[1]; t.rgb.resize((size_t)t.width * t.height * 3); rd(t.rgb.data(), t.rgb.size() * 4);
    }
    std::vector<uint32_t> tm(head[2]); rd(tm.data(), tm.size() * 4);
    std::vector<float> verts((size_t)head[2] * 24); rd(verts.data(), verts.size() * 4);
    fclose(f);
    std::vector<Triangle> tris(head[2]);
    for (size_t i = 0; i < tris.size(); i++) {
        tris[i].materialIndex = tm[i];
        for (int k = 0; k < 3; k++) {
            const float* d = &verts[i * 24 + k * 8]; Vertex& v = tris[i].vertices[k];
            v.p = Vec3(d[0], d[1], d[2]); v.normal = Vec3(d[3], d[4], d[5]); v.u = d[6]; v.v = d[7];
        }
    }
    scene->init(std::move(tris), mats, tex, bg);
    return scene;
}
// The same flat file raytracer_amd/abi.py:save_rtsc writes.
inline void saveRTSC(const Scene& s, const std::string& path) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("saveRTSC: cannot open " + path);
    uint32_t head[5] = { 0x43535452u, 1u, (uint32_t)s.triangles.size(), (uint32_t)s.materials.size(), (uint32_t)s.textures.size() };
    int32_t wh[2] = { (int32_t)s.camera.width, (int32_t)s.camera.height };
    fwrite(head, 4, 5, f); fwrite(wh, 4, 2, f); fwrite(s.camera.projectionMatrix.m, 4, 16, f); fwrite(s.camera.camera.m, 4, 16, f);
    fwrite(&s.background.kind, 4, 1, f); fwrite(s.background.colour, 4, 3, f); fwrite(&s.background.env_tex, 4, 1, f);
    fwrite(s.materials.data(), sizeof(rt_material), s.materials.size(), f);
    for (const Texture& t : s.textures) { int32_t h3[3] = { t.width, t.height, 0 }; fwrite(h3, 4, 3, f); fwrite(t.rgb.data(), 4, t.rgb.size(), f); }
    for (const Triangle& t : s.triangles) fwrite(&t.materialIndex, 4, 1, f);
    for (const Triangle& t : s.triangles) for (int k = 0; k < 3; k++) {
        const Vertex& v = t.vertices[k];
        float d[8] = { v.p.x, v.p.y, v.p.z, v.normal.x, v.normal.y, v.normal.z, v.u, v.v };
        fwrite(d, 4, 8, f);
    }
    fclose(f);
}

}  // namespace rtamd
