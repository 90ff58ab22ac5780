// Headless counterpart of RTBase's Main.cpp: same arguments (-scene, -outputFilename, -SPP; Main.cpp:56-70), same loop
// (render() until getSPP() == SPP, print the seconds each call took, then saveHDR; Main.cpp:79-137), no window.
// Extra arguments: -type (RayTracer::type, Renderer.h:372), -width/-height (built-in cornell-box only), -seed,
// -sppPerCall (samples added per render() call), -vpls (traceVPLs before type 4), -png (also write the tonemapped PNG),
// -filter box|gaussian|mitchell (the film's ImageFilter, Renderer.h:61),
// -keys WSADEQ... (scripted key presses of Main.cpp:87-116, one per frame: move the fly camera and clear the film),
// -dump file.rtsc (load the scene, write it as a flat file and stop: checks the loaders without a GPU).
#include <chrono>
#include <fstream>
#include <iostream>

#include "gem.hpp"

int main(int argc, char* argv[]) {
    std::string sceneName = "cornell-box";
    std::string filename = "radiosity.hdr";   // upstream's default name (Main.cpp:29)
    unsigned int SPP = 8192;
    int type = 0, width = 1024, height = 1024, perCall = 1, vpls = 0, png = 0, filter = RT_FILTER_BOX;
    std::string dump, keys, aov;
    unsigned long long seed = 1;
    try {
        // name / value pairs, later ones win; a name without a value or a bare word is reported and skipped
        for (int i = 1; i < argc; i++) {
            const std::string name = argv[i];
            if (name.size() < 2 || name[0] != '-') { std::cerr << "rtamd_cli: skipping stray word '" << name << "'\n"; continue; }
            if (i + 1 >= argc) { std::cerr << "rtamd_cli: option " << name << " needs a value\n"; break; }
            const std::string value = argv[++i];
            if (name == "-scene") sceneName = value;
            else if (name == "-outputFilename") filename = value;
            else if (name == "-SPP") SPP = (unsigned int)std::stoi(value);
            else if (name == "-type") type = std::stoi(value);
            else if (name == "-width") width = std::stoi(value);
            else if (name == "-height") height = std::stoi(value);
            else if (name == "-seed") seed = std::stoull(value);
            else if (name == "-sppPerCall") perCall = std::stoi(value);
            else if (name == "-vpls") vpls = std::stoi(value);
            else if (name == "-keys") keys = value;
            else if (name == "-filter") {
                if (value == "box") filter = RT_FILTER_BOX;
                else if (value == "gaussian") filter = RT_FILTER_GAUSSIAN;
                else if (value == "mitchell") filter = RT_FILTER_MITCHELL;
                else throw std::runtime_error("-filter: box, gaussian or mitchell");
            }
            else if (name == "-png") png = std::stoi(value);
            else if (name == "-dump") dump = value;
            else if (name == "-aov") aov = value;
            else std::cerr << "rtamd_cli: unknown option " << name << "\n";
        }
        if (perCall < 1) perCall = 1;
        auto press = [](char key) {   // Main.cpp:87-116
            switch (key) {
            case 'W': rtamd::viewcamera.forward(); break;
            case 'S': rtamd::viewcamera.back(); break;
            case 'A': rtamd::viewcamera.left(); break;
            case 'D': rtamd::viewcamera.right(); break;
            case 'E': rtamd::viewcamera.flyUp(); break;
            case 'Q': rtamd::viewcamera.flyDown(); break;
            default: throw std::runtime_error("-keys: W, S, A, D, E or Q");
            }
        };
        if (!dump.empty()) {   // loader check: parse the scene and write it back as a flat RTSC file; no device involved
            rtamd::Scene* host = rtamd::loadScene(sceneName, width, height, false);
            for (char key : keys) press(key);   // the dumped camera is where the key presses leave it
            rtamd::saveRTSC(*host, dump);
            std::cout << "number of triangle: " << host->triangles.size() << "\n";
            delete host;
            return 0;
        }
        rtamd::Scene* scene = rtamd::loadScene(sceneName, width, height);
        rt_scene_info info = scene->info();
        std::cout << "number of triangle: " << info.n_tris << "\n";   // Scene::build prints this (Scene.h:101)
        {
        rtamd::RayTracer rt;
        rt.init(scene);
        rt.type = type;
        rt.seed = seed;
        if (filter != RT_FILTER_BOX) rt.setFilter(filter);
        if (vpls > 0) rt.traceVPLs(vpls);
        size_t frame = 0;
        while ((unsigned int)rt.getSPP() < SPP) {
            if (frame < keys.size()) { press(keys[frame]); rt.clear(); }
            frame++;
            rt.samplesPerCall = (int)std::min<unsigned int>((unsigned int)perCall, SPP - (unsigned int)rt.getSPP());
            auto t0 = std::chrono::steady_clock::now();
            rt.render();
            rt.sync();
            std::cout << std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count() << std::endl;   // Main.cpp:122
        }
        if (!aov.empty()) {   // what getTileFilms hands the denoiser (Renderer.h:577-603): raw fp32 RGB, <prefix>.albedo.f32 / .normal.f32
            std::vector<float> albedo, normal;
            rt.getAOVFilms(albedo, normal);
            for (int k = 0; k < 2; k++) {
                std::ofstream f(aov + (k ? ".normal.f32" : ".albedo.f32"), std::ios::binary);
                const std::vector<float>& v = k ? normal : albedo;
                f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(float)));
                if (!f) throw std::runtime_error("-aov: cannot write " + aov);
            }
        }
        rt.saveHDR(filename);   // Main.cpp:133-136
        if (png) {
            size_t pos = filename.find_last_of('.');
            rt.savePNG(filename.substr(0, pos) + ".png");   // Main.cpp:129-131
        }
        }   // the renderer goes before its scene
        delete scene;
    } catch (const std::exception& e) {
        std::cerr << "rtamd: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
