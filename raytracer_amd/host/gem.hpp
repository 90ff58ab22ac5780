// gem.hpp — scene.json + .gem loader for the C++ host: what loadScene / loadInstance / GEMModelLoader / GEMScene do
// (SceneLoader.h:92-289, GEMLoader.h:218-411, :420-750), restated from their behaviour, plus the image decoding
// Texture::load gets from stb_image (Imaging.h:31-70): 8-bit PNG and Radiance .hdr are decoded here, nothing else.
//
// Behaviour kept on purpose (each is what upstream does, cited where it happens):
//  * JSON numbers become strings through std::to_string(float) ("45.000000"), booleans "1"/"0", arrays/dicts ""
//    (GEMJson::asStr, GEMLoader.h:459-480); every non-array top-level value is a scene property, every top-level array
//    is a list of instances (GEMScene::load, :722-737), visited in std::map (sorted-key) order.
//  * Per instance one material is appended (SceneLoader.h:111-171); an unknown / missing "bsdf" prints
//    "Error in loading" and drops the instance (:188-193).  "coatingThickness" wraps the BSDF in a LayeredBSDF that is
//    never stored in the material table (:178-187, :194), so it has no effect on the triangles: ignored here too.
//  * Mesh index offsets are the running *index* count, not the vertex count (:220-224) — exact only for the first
//    mesh of a file or un-indexed meshes; an index that lands outside the vertex array is an error here (UB upstream).
//  * Triangles with area == 0 are dropped (:229), area as Triangle::init computes it (Geometry.h:76-81).
//  * Normals go through transpose(invert(world)) and are re-normalised (:199-215); tangents are ignored.
//  * A texture that cannot be read becomes the 1x1 white default (Texture::loadDefault, Imaging.h:23-30, :37-41).
#pragma once
#include <cctype>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

#include "rtamd.hpp"

namespace rtamd {
namespace gem {

// ------------------------------------------------------------------------------------------------ properties
struct Property {
    std::string name, value;
    std::string getValue(const std::string& = "") const { return value; }
    float getValue(float d) const { try { return std::stof(value); } catch (...) { return d; } }
    int getValue(int d) const { try { return std::stoi(value); } catch (...) { return d; } }
    void getValuesAsVector3(float& x, float& y, float& z, char sep = ' ', float d = 0) const {
        std::vector<float> v;
        std::stringstream ss(value);
        std::string word;
        while (std::getline(ss, word, sep)) { float f; try { f = std::stof(word); } catch (...) { f = d; } v.push_back(f); }
        while (v.size() < 3) v.push_back(d);
        x = v[0]; y = v[1]; z = v[2];
    }
};
struct PropertyList {
    std::vector<Property> properties;
    Property find(const std::string& name) const {
        for (const auto& p : properties) if (p.name == name) return p;
        Property e; e.name = name; return e;
    }
};

// ------------------------------------------------------------------------------------------------ JSON (GEMLoader.h:420-669)
struct Json {
    enum Type { Null, Boolean, Number, String, Array, Dict } type = Null;
    bool b = false; float f = 0; std::string s; std::vector<Json> arr; std::map<std::string, Json> dict;
    std::string asStr() const {
        switch (type) {
        case Boolean: return std::to_string(b);
        case Number: return std::to_string(f);
        case String: return s;
        default: return "";
        }
    }
};
class JsonParser {
    std::string s; size_t pos = 0;
    void ws() { while (pos < s.size() && std::isspace((unsigned char)s[pos])) pos++; }
    char peek() const { return pos < s.size() ? s[pos] : 0; }
    char get() { if (pos >= s.size()) throw std::runtime_error("scene.json: unexpected end of file"); return s[pos++]; }
    Json value() {
        ws();
        char c = peek();
        Json j;
        if (c == 'n') { pos += 4; return j; }
        if (c == 't') { pos += 4; j.type = Json::Boolean; j.b = true; return j; }
        if (c == 'f') { pos += 5; j.type = Json::Boolean; j.b = false; return j; }
        if (c == '-' || std::isdigit((unsigned char)c)) {
            size_t start = pos;
            if (peek() == '-') get();
            if (peek() == '0') get(); else while (std::isdigit((unsigned char)peek())) get();
            if (peek() == '.') { get(); while (std::isdigit((unsigned char)peek())) get(); }
            if (peek() == 'e' || peek() == 'E') { get(); if (peek() == '+' || peek() == '-') get(); while (std::isdigit((unsigned char)peek())) get(); }
            j.type = Json::Number; j.f = std::stof(s.substr(start, pos - start));
            return j;
        }
        if (c == '"') { j.type = Json::String; j.s = str(); return j; }
        if (c == '[') {
            get(); ws(); j.type = Json::Array;
            if (peek() == ']') { get(); return j; }
            for (;;) { j.arr.push_back(value()); ws(); if (get() == ']') break; ws(); }
            return j;
        }
        if (c == '{') {
            get(); ws(); j.type = Json::Dict;
            if (peek() == '}') { get(); return j; }
            for (;;) { ws(); std::string key = str(); ws(); get(); ws(); j.dict[key] = value(); ws(); if (get() == '}') break; ws(); }
            return j;
        }
        return j;
    }
    std::string str() { get(); std::string r; for (;;) { char c = get(); if (c == '"') break; r.push_back(c); } return r; }  // no escapes upstream either
public:
    Json parse(const std::string& text) { s = text; pos = 0; ws(); return value(); }
};

struct Instance { float w[16]; std::string meshFilename; PropertyList material; };
struct GEMScene {  // GEMLoader.h:679
    std::vector<Instance> instances;
    PropertyList sceneProperties;
    void load(const std::string& filename) {
        std::ifstream file(filename);
        if (!file) throw std::runtime_error("loadScene: cannot open " + filename);
        std::stringstream buffer; buffer << file.rdbuf();
        Json data = JsonParser().parse(buffer.str());
        for (const auto& item : data.dict) {
            if (item.second.type != Json::Array) { sceneProperties.properties.push_back({ item.first, item.second.asStr() }); continue; }
            for (const auto& inst : item.second.arr) {
                Instance in; memset(in.w, 0, sizeof(in.w));
                for (const auto& kv : inst.dict) {
                    if (kv.first == "filename") in.meshFilename = kv.second.asStr();
                    else if (kv.first == "world") {
                        if (kv.second.arr.size() < 16) throw std::runtime_error("scene.json: \"world\" needs 16 numbers");
                        for (int i = 0; i < 16; i++) in.w[i] = kv.second.arr[i].f;
                    } else in.material.properties.push_back({ kv.first, kv.second.asStr() });
                }
                instances.push_back(in);
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------ .gem meshes (GEMLoader.h:218-365)
struct StaticVertex { float position[3], normal[3], tangent[3], u, v; };   // GEMStaticVertex, 44 bytes
struct Mesh { PropertyList material; std::vector<StaticVertex> verticesStatic; std::vector<uint32_t> indices; };
inline void loadGEM(const std::string& filename, std::vector<Mesh>& meshes) {
    std::ifstream file(filename, std::ios::binary);
    if (!file) throw std::runtime_error("loadScene: cannot open " + filename);
    file.seekg(0, std::ios::end);
    const uint64_t fileSize = (uint64_t)file.tellg();
    file.seekg(0, std::ios::beg);
    // counts come from the file: check them against what is left of it before sizing anything by them
    auto need = [&](uint64_t count, uint64_t each) {
        const uint64_t at = (uint64_t)file.tellg();
        if (at > fileSize || count > (fileSize - at) / each) throw std::runtime_error("truncated GE model file " + filename);
    };
    auto u32 = [&]() { uint32_t v = 0; file.read(reinterpret_cast<char*>(&v), 4); if (!file) throw std::runtime_error("truncated GE model file " + filename); return v; };
    auto str = [&]() { uint32_t l = u32(); if (l > (1u << 20)) throw std::runtime_error("bad string length in " + filename); std::string t(l, '\0'); if (l) file.read(&t[0], l); return std::string(t.c_str()); };
    if (u32() != 4058972161u) throw std::runtime_error(filename + " is not a GE Model File");   // upstream: message + exit(0)
    uint32_t isAnimated = u32(), n = u32();
    static_assert(sizeof(StaticVertex) == 44, "GEMStaticVertex layout");
    for (uint32_t i = 0; i < n; i++) {
        Mesh mesh;
        uint32_t np = u32();
        need(np, 8);
        for (uint32_t k = 0; k < np; k++) { Property p; p.name = str(); p.value = str(); mesh.material.properties.push_back(p); }
        uint32_t nv = u32();
        need(nv, isAnimated == 0 ? 44 : 76);
        if (isAnimated == 0) {
            mesh.verticesStatic.resize(nv);
            if (nv) file.read(reinterpret_cast<char*>(mesh.verticesStatic.data()), (std::streamsize)nv * 44);
        } else file.seekg((std::streamoff)nv * 76, std::ios::cur);   // GEMAnimatedVertex (+4 bone ids, 4 weights): loadInstance never reads them
        if (!file) throw std::runtime_error("truncated GE model file " + filename);
        uint32_t ni = u32();
        need(ni, 4);
        mesh.indices.resize(ni);
        if (ni) file.read(reinterpret_cast<char*>(mesh.indices.data()), (std::streamsize)ni * 4);
        if (!file) throw std::runtime_error("truncated GE model file " + filename);
        meshes.push_back(std::move(mesh));
    }
}

// ------------------------------------------------------------------------------------------------ images (what stb_image gives Texture::load)
namespace img {
struct Inflate {   // RFC 1951
    const uint8_t* in; size_t n, pos = 0; uint32_t bitbuf = 0; int bitcnt = 0; std::vector<uint8_t> out;
    Inflate(const uint8_t* p, size_t len) : in(p), n(len) {}
    uint32_t bits(int need) {
        uint32_t v = bitbuf;
        while (bitcnt < need) { if (pos >= n) throw std::runtime_error("inflate: out of input"); v |= (uint32_t)in[pos++] << bitcnt; bitcnt += 8; }
        bitbuf = need < 32 ? v >> need : 0; bitcnt -= need;
        return need < 32 ? v & ((1u << need) - 1u) : v;
    }
    struct Huff { uint16_t count[16]; uint16_t symbol[288]; };
    static void build(Huff& h, const uint8_t* len, int n) {
        memset(h.count, 0, sizeof(h.count));
        for (int i = 0; i < n; i++) h.count[len[i]]++;
        uint16_t offs[16]; offs[1] = 0;
        for (int i = 1; i < 15; i++) offs[i + 1] = offs[i] + h.count[i];
        for (int i = 0; i < n; i++) if (len[i]) h.symbol[offs[len[i]]++] = (uint16_t)i;
        h.count[0] = 0;
    }
    int decode(const Huff& h) {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; len++) {
            code |= (int)bits(1);
            int count = h.count[len];
            if (code - count < first) return h.symbol[index + (code - first)];
            index += count; first += count; first <<= 1; code <<= 1;
        }
        throw std::runtime_error("inflate: bad code");
    }
    void codes(const Huff& lc, const Huff& dc) {
        static const uint16_t lbase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
        static const uint16_t lext[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
        static const uint16_t dbase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
        static const uint16_t dext[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
        for (;;) {
            int sym = decode(lc);
            if (sym < 256) out.push_back((uint8_t)sym);
            else if (sym == 256) return;
            else {
                sym -= 257;
                if (sym >= 29) throw std::runtime_error("inflate: bad length symbol");
                size_t len = lbase[sym] + bits(lext[sym]);
                int ds = decode(dc);
                if (ds >= 30) throw std::runtime_error("inflate: bad distance symbol");
                size_t dist = dbase[ds] + bits(dext[ds]);
                if (dist > out.size()) throw std::runtime_error("inflate: distance too far back");
                for (size_t k = 0; k < len; k++) out.push_back(out[out.size() - dist]);
            }
        }
    }
    void run() {
        for (;;) {
            int last = (int)bits(1), type = (int)bits(2);
            if (type == 0) {
                bitbuf = 0; bitcnt = 0;
                if (pos + 4 > n) throw std::runtime_error("inflate: truncated stored block");
                size_t len = in[pos] | (in[pos + 1] << 8); pos += 4;
                if (pos + len > n) throw std::runtime_error("inflate: truncated stored block");
                out.insert(out.end(), in + pos, in + pos + len); pos += len;
            } else if (type == 1) {
                uint8_t l[320]; Huff lc, dc;
                for (int i = 0; i < 144; i++) l[i] = 8;
                for (int i = 144; i < 256; i++) l[i] = 9;
                for (int i = 256; i < 280; i++) l[i] = 7;
                for (int i = 280; i < 288; i++) l[i] = 8;
                build(lc, l, 288);
                for (int i = 0; i < 30; i++) l[i] = 5;
                build(dc, l, 30);
                codes(lc, dc);
            } else if (type == 2) {
                static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
                int nlen = (int)bits(5) + 257, ndist = (int)bits(5) + 1, ncode = (int)bits(4) + 4;
                uint8_t l[320]; memset(l, 0, sizeof(l));
                for (int i = 0; i < ncode; i++) l[order[i]] = (uint8_t)bits(3);
                Huff cl; build(cl, l, 19);
                uint8_t lens[320]; int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = decode(cl);
                    if (sym < 16) lens[idx++] = (uint8_t)sym;
                    else {
                        int rep; uint8_t v = 0;
                        if (sym == 16) { if (!idx) throw std::runtime_error("inflate: repeat without a length"); v = lens[idx - 1]; rep = 3 + (int)bits(2); }
                        else if (sym == 17) rep = 3 + (int)bits(3);
                        else rep = 11 + (int)bits(7);
                        if (idx + rep > nlen + ndist) throw std::runtime_error("inflate: too many lengths");
                        while (rep--) lens[idx++] = v;
                    }
                }
                Huff lc, dc; build(lc, lens, nlen); build(dc, lens + nlen, ndist);
                codes(lc, dc);
            } else throw std::runtime_error("inflate: bad block type");
            if (last) return;
        }
    }
};
struct Image { int width = 0, height = 0, channels = 0; std::vector<float> data; };   // data: width*height*channels, already /255 for 8-bit sources

inline bool readFile(const std::string& path, std::vector<uint8_t>& b) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    b.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}
inline Image decodePNG(const std::vector<uint8_t>& b, const std::string& name) {
    auto be32 = [&](size_t o) { return ((uint32_t)b[o] << 24) | ((uint32_t)b[o + 1] << 16) | ((uint32_t)b[o + 2] << 8) | b[o + 3]; };
    uint32_t W = 0, H = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    for (size_t o = 8; o + 12 <= b.size();) {
        uint32_t len = be32(o); std::string tag(b.begin() + o + 4, b.begin() + o + 8);
        if (o + 12 + len > b.size()) throw std::runtime_error(name + ": truncated PNG chunk");
        const uint8_t* d = b.data() + o + 8;
        if (tag == "IHDR") { if (len < 13) throw std::runtime_error(name + ": short IHDR"); W = be32(o + 8); H = be32(o + 12); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (tag == "PLTE") plte.assign(d, d + len);
        else if (tag == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (tag == "IEND") break;
        o += 12 + len;
    }
    if (!W || !H || idat.size() < 2) throw std::runtime_error(name + ": not a usable PNG");
    if (W > (1u << 16) || H > (1u << 16)) throw std::runtime_error(name + ": PNG larger than 65536 pixels a side");   // keeps (stride + 1) * H far from wrapping
    if (ctype == 3 && depth == 16) throw std::runtime_error(name + ": a palette PNG cannot be 16 bits deep");
    if (interlace) throw std::runtime_error(name + ": interlaced PNG is not supported by this host");
    int comps = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!comps || (depth != 8 && depth != 16 && !((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) throw std::runtime_error(name + ": unsupported PNG colour type / bit depth");
    Inflate z(idat.data() + 2, idat.size() - 2); z.run();
    size_t bpp = (size_t)(comps * depth + 7) / 8, stride = ((size_t)W * comps * depth + 7) / 8;
    if (z.out.size() < (stride + 1) * H) throw std::runtime_error(name + ": PNG data too short");
    std::vector<uint8_t> raw(stride * H);
    for (uint32_t y = 0; y < H; y++) {
        const uint8_t* src = &z.out[(stride + 1) * y]; uint8_t* cur = &raw[stride * y]; const uint8_t* up = y ? cur - stride : nullptr;
        int ft = src[0];
        for (size_t x = 0; x < stride; x++) {
            int a = x >= bpp ? cur[x - bpp] : 0, bb = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0, v = src[1 + x];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += bb; break;
            case 3: v += (a + bb) / 2; break;
            case 4: { int p = a + bb - c, pa = abs(p - a), pb = abs(p - bb), pc = abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c); break; }
            default: throw std::runtime_error(name + ": bad PNG filter");
            }
            cur[x] = (uint8_t)v;
        }
    }
    Image im; im.width = (int)W; im.height = (int)H;
    im.channels = ctype == 3 ? 3 : comps;   // stb expands palettes to RGB
    im.data.resize((size_t)W * H * im.channels);
    const float scale = 1.0f / 255.0f; (void)scale;
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) for (int c = 0; c < comps; c++) {
        const uint8_t* row = &raw[stride * y];
        uint32_t v8;
        if (depth == 8) v8 = row[x * comps + c];
        else if (depth == 16) v8 = row[(x * comps + c) * 2];   // stb keeps the high byte
        else { uint32_t bit = x * depth, v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u); v8 = ctype == 3 ? v : v * (255u / ((1u << depth) - 1u)); }
        if (ctype == 3) {
            if ((size_t)v8 * 3 + 2 >= plte.size()) throw std::runtime_error(name + ": palette index out of range");
            for (int k = 0; k < 3; k++) im.data[((size_t)y * W + x) * 3 + k] = plte[v8 * 3 + k] / 255.0f;
        } else im.data[((size_t)y * W + x) * comps + c] = v8 / 255.0f;   // Imaging.h:59
    }
    return im;
}
inline Image decodeHDR(const std::vector<uint8_t>& b, const std::string& name) {
    size_t p = 0;
    auto line = [&]() { std::string l; while (p < b.size() && b[p] != '\n') l.push_back((char)b[p++]); p++; return l; };
    std::string l = line();
    if (l.compare(0, 2, "#?") != 0) throw std::runtime_error(name + ": not a Radiance file");
    while (p < b.size()) { l = line(); if (l.empty()) break; }
    l = line();
    int H = 0, W = 0;
    if (sscanf(l.c_str(), "-Y %d +X %d", &H, &W) != 2 || W <= 0 || H <= 0) throw std::runtime_error(name + ": unsupported Radiance orientation");
    std::vector<uint8_t> rgbe((size_t)W * H * 4);
    for (int y = 0; y < H; y++) {
        uint8_t* row = &rgbe[(size_t)y * W * 4];
        if (W >= 8 && W < 32768 && p + 4 <= b.size() && b[p] == 2 && b[p + 1] == 2 && !(b[p + 2] & 0x80)) {
            if (((b[p + 2] << 8) | b[p + 3]) != W) throw std::runtime_error(name + ": bad RLE scanline width");
            p += 4;
            for (int c = 0; c < 4; c++) for (int x = 0; x < W;) {
                if (p >= b.size()) throw std::runtime_error(name + ": truncated");
                int cnt = b[p++];
                if (cnt > 128) { cnt -= 128; if (p >= b.size() || x + cnt > W) throw std::runtime_error(name + ": bad run"); uint8_t v = b[p++]; while (cnt--) row[4 * x++ + c] = v; }
                else { if (!cnt || p + cnt > b.size() || x + cnt > W) throw std::runtime_error(name + ": bad run"); while (cnt--) row[4 * x++ + c] = b[p++]; }
            }
        } else {
            if (p + (size_t)W * 4 > b.size()) throw std::runtime_error(name + ": truncated");
            memcpy(row, &b[p], (size_t)W * 4); p += (size_t)W * 4;
        }
    }
    Image im; im.width = W; im.height = H; im.channels = 3; im.data.resize((size_t)W * H * 3);
    for (size_t i = 0; i < (size_t)W * H; i++) {
        const uint8_t* q = &rgbe[i * 4];
        if (q[3]) { float f = ldexpf(1.0f, (int)q[3] - 136); for (int k = 0; k < 3; k++) im.data[i * 3 + k] = q[k] * f; }   // stbi__hdr_convert
    }
    return im;
}
}  // namespace img

// Texture::load (Imaging.h:31-70)
inline Texture loadTextureFile(const std::string& filename) {
    Texture t;   // loadDefault: 1x1 white
    t.rgb = { 1.0f, 1.0f, 1.0f };
    std::vector<uint8_t> b;
    if (!img::readFile(filename, b) || b.size() < 8) return t;
    img::Image im;
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (filename.find(".hdr") != std::string::npos) im = img::decodeHDR(b, filename);
    else if (!memcmp(b.data(), sig, 8)) im = img::decodePNG(b, filename);
    else throw std::runtime_error(filename + ": only PNG and Radiance .hdr textures are decoded by this host (convert JPEG/TGA/BMP assets to PNG)");
    t.width = im.width; t.height = im.height;
    t.rgb.resize((size_t)im.width * im.height * 3);
    const size_t total = im.data.size();
    for (size_t i = 0; i < (size_t)im.width * im.height; i++)   // Imaging.h:45, :59 index by `channels` whatever it is
        for (size_t k = 0; k < 3; k++) { size_t j = i * im.channels + k; t.rgb[i * 3 + k] = j < total ? im.data[j] : 0.0f; }
    return t;
}

struct Loaded {   // everything Scene::init / Camera::init need, before any device is involved
    std::vector<Triangle> triangles; std::vector<rt_material> materials; std::vector<Texture> textures; rt_background background;
    Matrix P, V; int width = 0, height = 0;
    Vec3 from, to, up;
};
inline rt_material baseMaterial(uint32_t type, int tex) {
    rt_material m; memset(&m, 0, sizeof(m));
    m.type = type; m.albedo_tex = tex; m.int_ior = 1.33f; m.ext_ior = 1.0f; m.roughness = 1.0f; m.sigma = 1.0f;
    return m;
}
inline Loaded loadGEMScene(const std::string& sceneName) {
    Loaded L;
    GEMScene gemscene; gemscene.load(sceneName + "/scene.json");
    const PropertyList& sp = gemscene.sceneProperties;
    L.width = sp.find("width").getValue(1920); L.height = sp.find("height").getValue(1080);   // SceneLoader.h:241-243
    float fov = sp.find("fov").getValue(45.0f);
    L.P = Matrix::perspective(0.001f, 10000.0f, (float)L.width / (float)L.height, fov);
    Vec3 from, to, up;
    sp.find("from").getValuesAsVector3(from.x, from.y, from.z);
    sp.find("to").getValuesAsVector3(to.x, to.y, to.z);
    sp.find("up").getValuesAsVector3(up.x, up.y, up.z);
    L.V = Matrix::lookAt(from, to, up).invert();
    L.from = from; L.to = to; L.up = up;
    if (sp.find("flipX").getValue(0) == 1) L.P.a(0, 0) = -L.P.a(0, 0);
    std::map<std::string, int> textureManager;
    auto loadTexture = [&](const std::string& filename) {   // SceneLoader.h:92-102
        auto it = textureManager.find(filename);
        if (it != textureManager.end()) return it->second;
        L.textures.push_back(loadTextureFile(filename));
        int id = (int)L.textures.size() - 1;
        textureManager[filename] = id;
        return id;
    };
    for (const Instance& inst : gemscene.instances) {   // loadInstance (SceneLoader.h:104-234)
        std::vector<Mesh> meshes;
        loadGEM(sceneName + "/" + inst.meshFilename, meshes);
        const PropertyList& mp = inst.material;
        const std::string bsdf = mp.find("bsdf").getValue("");
        int have = 0;
        auto tex = [&]() { return loadTexture(sceneName + "/" + mp.find("reflectance").getValue("")); };
        auto ior = [&](rt_material& m) { m.int_ior = mp.find("intIOR").getValue(1.33f); m.ext_ior = mp.find("extIOR").getValue(1.0f); };
        if (bsdf == "diffuse") { L.materials.push_back(baseMaterial(RT_MAT_DIFFUSE, tex())); have = 1; }
        if (bsdf == "orennayar") { rt_material m = baseMaterial(RT_MAT_ORENNAYAR, tex()); m.sigma = mp.find("alpha").getValue(1.0f); L.materials.push_back(m); have = 1; }
        if (bsdf == "glass") { rt_material m = baseMaterial(RT_MAT_GLASS, tex()); ior(m); L.materials.push_back(m); have = 1; }
        if (bsdf == "mirror") { L.materials.push_back(baseMaterial(RT_MAT_MIRROR, tex())); have = 1; }
        if (bsdf == "plastic") { rt_material m = baseMaterial(RT_MAT_PLASTIC, tex()); ior(m); m.roughness = mp.find("roughness").getValue(1.0f); L.materials.push_back(m); have = 1; }
        if (bsdf == "dielectric") {
            rt_material m = baseMaterial(RT_MAT_DIELECTRIC, tex()); ior(m); m.roughness = mp.find("roughness").getValue(1.0f);
            if (m.roughness < 0.001f) m.type = RT_MAT_GLASS;   // SceneLoader.h:152-155
            L.materials.push_back(m); have = 1;
        }
        if (bsdf == "conductor") {
            rt_material m = baseMaterial(RT_MAT_CONDUCTOR, tex());
            mp.find("eta").getValuesAsVector3(m.eta[0], m.eta[1], m.eta[2]);
            mp.find("k").getValuesAsVector3(m.k[0], m.k[1], m.k[2]);
            m.roughness = mp.find("roughness").getValue(1.0f);
            L.materials.push_back(m); have = 1;
        }
        if (!have) { std::cout << "Error in loading" << std::endl; continue; }   // SceneLoader.h:188-193 (emission on a NULL material is UB upstream)
        if (mp.find("emission").getValue("") != "") { rt_material& m = L.materials.back(); mp.find("emission").getValuesAsVector3(m.emission[0], m.emission[1], m.emission[2]); }
        const unsigned int materialIndex = (unsigned int)L.materials.size() - 1;
        Matrix transform; memcpy(transform.m, inst.w, sizeof(inst.w));
        Matrix inv = transform.invert(), vecTransform;
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) vecTransform.m[r * 4 + c] = inv.m[c * 4 + r];
        const float* M = transform.m; const float* N = vecTransform.m;
        std::vector<Vertex> vertices; std::vector<uint32_t> indices;
        for (const Mesh& mesh : meshes) {
            for (const StaticVertex& sv : mesh.verticesStatic) {
                Vertex v;
                Vec3 p(sv.position[0], sv.position[1], sv.position[2]), n(sv.normal[0], sv.normal[1], sv.normal[2]);
                v.p = Vec3((p.x * M[0] + p.y * M[1] + p.z * M[2]) + M[3], (p.x * M[4] + p.y * M[5] + p.z * M[6]) + M[7], (p.x * M[8] + p.y * M[9] + p.z * M[10]) + M[11]);   // Matrix::mulPoint (Core.h:302)
                Vec3 t((n.x * N[0] + n.y * N[1] + n.z * N[2]), (n.x * N[4] + n.y * N[5] + n.z * N[6]), (n.x * N[8] + n.y * N[9] + n.z * N[10]));          // Matrix::mulVec (Core.h:295)
                float l = 1.0f / sqrtf((t.x * t.x) + (t.y * t.y) + (t.z * t.z));   // Vec3::normalize (Core.h:161)
                v.normal = Vec3(t.x * l, t.y * l, t.z * l);
                v.u = sv.u; v.v = sv.v;
                vertices.push_back(v);
            }
            const uint32_t offset = (uint32_t)indices.size();   // sic (SceneLoader.h:220)
            for (uint32_t idx : mesh.indices) indices.push_back(offset + idx);
        }
        for (size_t i = 0; i + 2 < indices.size(); i += 3) {
            if (indices[i] >= vertices.size() || indices[i + 1] >= vertices.size() || indices[i + 2] >= vertices.size())
                throw std::runtime_error(inst.meshFilename + ": vertex index out of range (multi-mesh files are offset by the index count upstream)");
            Triangle t; t.vertices[0] = vertices[indices[i]]; t.vertices[1] = vertices[indices[i + 1]]; t.vertices[2] = vertices[indices[i + 2]]; t.materialIndex = materialIndex;
            Vec3 e1 = t.vertices[2].p - t.vertices[1].p, e2 = t.vertices[0].p - t.vertices[2].p;   // Triangle::init (Geometry.h:70-81)
            Vec3 c = e1.cross(e2);
            float area = sqrtf((c.x * c.x) + (c.y * c.y) + (c.z * c.z)) * 0.5f;
            if (area > 0) L.triangles.push_back(t);
        }
    }
    memset(&L.background, 0, sizeof(L.background)); L.background.env_tex = -1;
    if (sp.find("envmap").getValue("") != "") {   // SceneLoader.h:272-280
        L.background.kind = 1;
        L.background.env_tex = loadTexture(sceneName + "/" + sp.find("envmap").getValue(""));
    }
    return L;
}

}  // namespace gem

// Scene* loadScene(std::string sceneName) (SceneLoader.h:236).  `<name>/scene.json` -> upstream's GEM scene directory;
// `<name>/scene.rtsc` or an .rtsc file -> flat export; "cornell-box" with no such directory -> the built-in box.
// `build = false` stops before Scene::build (no device is touched): used by the driver's -dump.
inline Scene* loadScene(std::string sceneName, int width = 1024, int height = 1024, bool build = true) {
    auto exists = [](const std::string& p) { FILE* f = fopen(p.c_str(), "rb"); if (f) fclose(f); return f != nullptr; };
    Scene* scene = nullptr;
    if (exists(sceneName + "/scene.json")) {
        gem::Loaded L = gem::loadGEMScene(sceneName);
        scene = new Scene();
        scene->camera.init(L.P, L.width, L.height);
        scene->camera.updateView(L.V);
        viewcamera.from = L.from; viewcamera.to = L.to; viewcamera.up = L.up;   // SceneLoader.h:261-264
        viewcamera.camera = &scene->camera; viewcamera.scene = scene;
        scene->init(std::move(L.triangles), L.materials, L.textures, L.background);
    } else if (exists(sceneName + "/scene.rtsc")) scene = readRTSC(sceneName + "/scene.rtsc");
    else if (sceneName.size() > 5 && sceneName.compare(sceneName.size() - 5, 5, ".rtsc") == 0 && exists(sceneName)) scene = readRTSC(sceneName);
    else if (sceneName == "cornell-box") scene = cornellBox(width, height);
    else throw std::runtime_error("loadScene: no scene '" + sceneName + "' (expected <name>/scene.json, <name>/scene.rtsc, an .rtsc file, or cornell-box)");
    if (viewcamera.scene != scene) {   // a flat export carries the matrices only: V's columns are left, up', back, from (Core.h:460)
        const Matrix& V = scene->camera.camera;
        viewcamera.from = Vec3(V.m[3], V.m[7], V.m[11]);
        viewcamera.to = viewcamera.from - Vec3(V.m[2], V.m[6], V.m[10]);
        viewcamera.up = Vec3(V.m[1], V.m[5], V.m[9]);
        viewcamera.camera = &scene->camera; viewcamera.scene = scene;
    }
    viewcamera.movespeed = scene->boundsDiagonal() * 0.05f;   // SceneLoader.h:284
    if (build) {
        try { scene->build(); } catch (...) { delete scene; throw; }
    }
    return scene;
}
}  // namespace rtamd
