// The reference's own unit test (RTTest/RTTest.cpp: TestRayInterset, the triangle above which a ray is shot), restated
// against the C++ host mirror: the same triangle and ray go through rtamd::Scene::traverse / visible, i.e. through the
// C ABI and the GPU kernels.  Known answer of Triangle::rayIntersect for it: t = 1, u = 0.5, v = 0.25 (weights of v0, v1).
#include <cfloat>
#include <cstdio>

#include "../../raytracer_amd/host/rtamd.hpp"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL line %d: %s\n", __LINE__, #c); fails++; } } while (0)

int main() {
    using namespace rtamd;
    try {
        Scene scene;
        setCamera(&scene, Vec3(0, 0, 5), Vec3(0, 0, 0), Vec3(0, 1, 0), 45.0f, 16, 16);
        Triangle tri;
        tri.vertices[0].p = Vec3(0, 0, 0); tri.vertices[1].p = Vec3(1.0f, 0, 0); tri.vertices[2].p = Vec3(0, 1.0f, 0);
        for (int k = 0; k < 3; k++) tri.vertices[k].normal = Vec3(1.0f, 0, 0);
        rt_background bg; memset(&bg, 0, sizeof(bg)); bg.env_tex = -1;
        scene.init({ tri }, { diffuseMaterial(0) }, { solidTexture(Colour(1, 1, 1)) }, bg);
        scene.build();

        Ray r(Vec3(0.25f, 0.25f, 1.0f), Vec3(0, 0, -1));   // above the triangle
        IntersectionData h = scene.traverse(r);
        CHECK(h.ID == 0u); CHECK(h.t == 1.0f); CHECK(h.alpha == 0.5f); CHECK(h.beta == 0.25f); CHECK(h.gamma == 1.0f - (0.5f + 0.25f));
        IntersectionData m = scene.traverse(Ray(Vec3(2.0f, 2.0f, 1.0f), Vec3(0, 0, -1)));
        CHECK(m.t == FLT_MAX);                              // miss <=> t == FLT_MAX (Geometry.h:468)
        IntersectionData back = scene.traverse(Ray(Vec3(0.25f, 0.25f, 1.0f), Vec3(0, 0, 1)));
        CHECK(back.t == FLT_MAX);                           // t < 0 is rejected (Geometry.h:120)
        CHECK(!scene.visible(Vec3(0.25f, 0.25f, 1.0f), Vec3(0.25f, 0.25f, -1.0f)));   // the segment crosses the triangle
        CHECK(scene.visible(Vec3(0.25f, 0.25f, 1.0f), Vec3(0.25f, 0.25f, 0.5f)));     // it stops short of it
        CHECK(scene.visible(Vec3(2.0f, 2.0f, 1.0f), Vec3(2.0f, 2.0f, -1.0f)));
        std::vector<Ray> many(1000, r);
        many[7] = Ray(Vec3(2.0f, 2.0f, 1.0f), Vec3(0, 0, -1));
        std::vector<IntersectionData> hs = scene.traverse(many);
        int hits = 0;
        for (const auto& x : hs) hits += x.t < FLT_MAX;
        CHECK(hits == 999 && hs[7].t == FLT_MAX && hs[999].t == 1.0f);
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
    std::printf(fails ? "rttest: %d failure(s)\n" : "rttest: all checks passed\n", fails);
    return fails ? 1 : 0;
}
