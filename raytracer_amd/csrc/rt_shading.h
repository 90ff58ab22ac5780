// rt_shading.h — device restatement of the per-sample shading model: ShadingData (Scene.h:185), the BSDF
// family (Materials.h), lights (Lights.h), sampling distributions (Sampling.h:143) and bilinear textures
// (Imaging.h:71).  Formulas are kept verbatim, quirks included; expressions that
// the reference evaluates in double because of M_PI are evaluated in double here too.
#pragma once
#include "../../include/rtamd.h"
#include "rt_device.h"

#define RT_PI 3.14159265358979323846

struct Frame3 { f3 u, v, w; };
__device__ __forceinline__ Frame3 frame_from(f3 n) {  // Frame::fromVector (Core.h:513)
    Frame3 f;
    f.w = norm3(n);
    if (fabsf(f.w.x) > fabsf(f.w.y)) {
        float l = 1.0f / sqrtf(f.w.x * f.w.x + f.w.z * f.w.z);
        f.u = mk3(f.w.z * l, 0.0f, -f.w.x * l);
    } else {
        float l = 1.0f / sqrtf(f.w.y * f.w.y + f.w.z * f.w.z);
        f.u = mk3(0, f.w.z * l, -f.w.y * l);
    }
    f.v = cross3(f.w, f.u);
    return f;
}
__device__ __forceinline__ f3 to_local(const Frame3& f, f3 a) { return mk3(dot3(a, f.u), dot3(a, f.v), dot3(a, f.w)); }  // Core.h:534
__device__ __forceinline__ f3 to_world(const Frame3& f, f3 a) { return ((f.u * a.x) + (f.v * a.y)) + (f.w * a.z); }      // Core.h:538

struct Shading {   // ShadingData (Materials.h:14)
    f3 x, wo, sN;
    float tu, tv;
    Frame3 frame;
    uint32_t mat;
    col3 albedo;   // Texture::sample of the material's texture at (tu, tv): every BSDF reads it, evaluate() and sample() alike (Materials.h), so it is fetched once per shading point (shading_albedo)
};

// ---- Texture::sample (Imaging.h:71-93)
__device__ inline col3 tex_sample(const DScene& S, int tex, float tu, float tv) {
    DTexture t = S.tex[tex];
    const float* px = S.texels + t.offset;
    if (t.width == 1 && t.height == 1) {
        // every weight pair sums over the same texel: (s*w0 + s*w1) + ... with w's from frac(u),frac(v)
        float u = smaxf(0.0f, fabsf(tu)) * 1, v = smaxf(0.0f, fabsf(tv)) * 1;
        float fu = u - (int)floorf(u), fv = v - (int)floorf(v);
        float w0 = (1.0f - fu) * (1.0f - fv), w1 = fu * (1.0f - fv), w2 = (1.0f - fu) * fv, w3 = fu * fv;
        col3 s = mkc(px[0], px[1], px[2]);
        return (((s * w0) + (s * w1)) + (s * w2)) + (s * w3);
    }
    float u = smaxf(0.0f, fabsf(tu)) * t.width;
    float v = smaxf(0.0f, fabsf(tv)) * t.height;
    int x = (int)floorf(u), y = (int)floorf(v);
    float fu = u - x, fv = v - y;
    float w0 = (1.0f - fu) * (1.0f - fv), w1 = fu * (1.0f - fv), w2 = (1.0f - fu) * fv, w3 = fu * fv;
    x = x % t.width; y = y % t.height;
    int x1 = (x + 1) % t.width, y1 = (y + 1) % t.height;
    const float* a = px + ((size_t)y * t.width + x) * 3;
    const float* b = px + ((size_t)y * t.width + x1) * 3;
    const float* c = px + ((size_t)y1 * t.width + x) * 3;
    const float* d = px + ((size_t)y1 * t.width + x1) * 3;
    return (((mkc(a[0], a[1], a[2]) * w0) + (mkc(b[0], b[1], b[2]) * w1)) + (mkc(c[0], c[1], c[2]) * w2)) + (mkc(d[0], d[1], d[2]) * w3);
}

// ---- Sampling.h:143-197
__device__ __forceinline__ f3 cosine_sample_hemisphere(float r1, float r2) {
    float theta = acosf(sqrtf(r1));
    float phi = (float)(2 * RT_PI * r2);
    return mk3(sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta));
}
__device__ __forceinline__ float cosine_hemisphere_pdf(f3 wi) { return (float)(wi.z / RT_PI); }
__device__ __forceinline__ f3 uniform_sample_sphere(float r1, float r2) {
    float theta = acosf(1 - (2 * r1));
    float phi = (float)(2 * RT_PI * r2);
    return mk3(sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta));
}
__device__ __forceinline__ float uniform_sphere_pdf() { return (float)(1.0f / 4 * RT_PI); }  // pi/4, sic (Sampling.h:192)

// ---- ShadingHelper (Materials.h:36-173)
__device__ __forceinline__ float clampf_(float x, float lo, float hi) { return smaxf(lo, sminf(x, hi)); }
__device__ inline float fresnel_dielectric(float cosTheta, float iorInt, float iorExt) {
    cosTheta = clampf_(cosTheta, -1.0f, 1.0f);
    bool entering = cosTheta > 0.f;
    if (entering) { float t = iorInt; iorInt = iorExt; iorExt = t; cosTheta = fabsf(cosTheta); }
    float ri = iorExt / iorInt;
    float cosI = cosTheta;
    float sinI = sqrtf(smaxf(0.0f, 1 - powf(cosI, 2)));
    float sinT = ri * sinI;
    if (sinT >= 1.f) return 1.f;
    float cosT = sqrtf(smaxf(0.0f, 1 - powf(sinT, 2)));
    float par = (cosI - (ri * cosT)) / (cosI + (ri * cosT));
    float per = ((cosI * ri) - cosT) / ((cosI * ri) + cosT);
    return 0.5f * ((par * par) + (per * per));
}
__device__ inline col3 fresnel_conductor(float cosTheta, col3 ior, col3 k) {
    cosTheta = fabsf(cosTheta);
    float c2 = powf(cosTheta, 2);
    float s2 = 1 - c2;
    col3 nk2 = (ior * ior) + (k * k);
    col3 two = (ior * cosTheta) * 2.0f;
    col3 a = nk2 * c2;
    col3 num = mkc((a.r - two.r) + s2, (a.g - two.g) + s2, (a.b - two.b) + s2);
    col3 den = mkc((a.r + two.r) + s2, (a.g + two.g) + s2, (a.b + two.b) + s2);
    col3 par = mkc(num.r / den.r, num.g / den.g, num.b / den.b);
    col3 num2 = mkc((nk2.r - two.r) + c2, (nk2.g - two.g) + c2, (nk2.b - two.b) + c2);
    col3 den2 = mkc((nk2.r + two.r) + c2, (nk2.g + two.g) + c2, (nk2.b + two.b) + c2);
    col3 per = mkc(num2.r / den2.r, num2.g / den2.g, num2.b / den2.b);
    return ((par * par) + (per * per)) * 0.5f;
}
__device__ __forceinline__ float lambda_ggx(f3 wi, float alpha) {
    float sin2 = smaxf(0.0f, 1 - wi.z * wi.z);
    float tanT = sqrtf(sin2) / wi.z;
    float a = fabsf(tanT);
    float a2t2 = (alpha * a) * (alpha * a);
    return (-1 + sqrtf(1 + a2t2)) * 0.5f;
}
__device__ __forceinline__ float g_ggx(f3 wi, f3 wo, float alpha) { return 1.0f / (1.0f + lambda_ggx(wi, alpha) + lambda_ggx(wo, alpha)); }
__device__ __forceinline__ float d_ggx(f3 h, float alpha) {
    float c2 = h.z * h.z;
    float a2 = powf(alpha, 2);
    float den = (c2 * (a2 - 1)) + 1;
    return (float)(a2 / (den * den * RT_PI));
}
__device__ __forceinline__ void set_ior(float cosTheta, float& iorI, float& iorT, float intIOR, float extIOR) {
    bool entering = cosTheta > 0;
    iorI = entering ? extIOR : intIOR;
    iorT = entering ? intIOR : extIOR;
}
__device__ __forceinline__ col3 mat_emission(const DMaterial& m) { return mkc(m.emission[0], m.emission[1], m.emission[2]); }

// ---- BSDF::evaluate
__device__ inline col3 bsdf_eval(const DScene& S, const Shading& sd, f3 wi) {
    const DMaterial& m = S.mats[sd.mat];
    switch (m.type) {
    case RT_MAT_DIFFUSE:
    case RT_MAT_ORENNAYAR:  // Materials.h:777 returns Lambert before the Oren-Nayar terms
        return sd.albedo / (float)RT_PI;
    case RT_MAT_MIRROR: case RT_MAT_GLASS:
        return mkc(0, 0, 0);
    case RT_MAT_CONDUCTOR: {  // Materials.h:360
        f3 wo = to_local(sd.frame, sd.wo), wl = to_local(sd.frame, wi);
        if (wo.z <= 0.0f || wl.z <= 0.0f) return mkc(0, 0, 0);
        f3 wm = norm3(wl + wo);
        if (wm.z <= 0.0f) return mkc(0, 0, 0);
        if (m.alpha < 0.001f) return mkc(0, 0, 0);
        float D = d_ggx(wm, m.alpha), G = g_ggx(wl, wo, m.alpha);
        col3 F = fresnel_conductor(dot3(wo, wm), mkc(m.eta[0], m.eta[1], m.eta[2]), mkc(m.k[0], m.k[1], m.k[2]));
        col3 brdf = ((F * D) * G) / (4.0f * wo.z * wl.z);
        return sd.albedo * brdf;
    }
    case RT_MAT_DIELECTRIC: {  // Materials.h:638
        f3 wo = to_local(sd.frame, sd.wo), wl = to_local(sd.frame, wi);
        if (wo.z <= 0.0f) return mkc(0, 0, 0);
        float iorT, iorI; set_ior(wo.z, iorI, iorT, m.int_ior, m.ext_ior);
        float ri = iorI / iorT;
        f3 wm;
        bool reflect = wl.z * wo.z > 0;
        if (reflect) wm = norm3(wl + wo);
        else { wm = norm3((wl * ri) + wo); if (wm.z < 0) wm = -wm; }
        if (wm.z <= 0.0f) return mkc(0, 0, 0);
        float F = fresnel_dielectric(dot3(wo, wm), iorI, iorT);
        float D = d_ggx(wm, m.alpha), G = g_ggx(wl, wo, m.alpha);
        float result;
        if (reflect) result = F * D * G / (4.0f * wo.z * wl.z);
        else {
            float den = ri * dot3(wl, wm) + dot3(wo, wm);
            result = (1 - F) * D * G * dot3(wl, wm) * dot3(wo, wm) / (powf(den, 2) * wl.z * wo.z);
        }
        return sd.albedo * result;
    }
    case RT_MAT_PLASTIC: {  // Materials.h:931
        f3 wo = to_local(sd.frame, sd.wo), wl = to_local(sd.frame, wi);
        if (wl.z <= 0.0f || wo.z <= 0.0f) return mkc(0, 0, 0);
        float iorT, iorI; set_ior(wo.z, iorI, iorT, m.int_ior, m.ext_ior);
        float ks = fresnel_dielectric(wo.z, iorI, iorT), kd = 1.0f - ks;
        float e = m.phong_e;
        col3 diffuse = (sd.albedo * kd) / (float)RT_PI;
        f3 wr = mk3(-wo.x, -wo.y, wo.z);
        float cosAlpha = smaxf(0.0f, dot3(wr, wl));
        float gloss = (float)(ks * ((e + 2) / (2 * RT_PI)) * powf(cosAlpha, e));
        return diffuse + mkc(gloss, gloss, gloss);
    }
    }
    return mkc(0, 0, 0);
}

// ---- BSDF::PDF (used by the debug probe; the integrators take the pdf from sample())
__device__ inline float bsdf_pdf(const DScene& S, const Shading& sd, f3 wi) {
    const DMaterial& m = S.mats[sd.mat];
    f3 wo = to_local(sd.frame, sd.wo), wl = to_local(sd.frame, wi);
    switch (m.type) {
    case RT_MAT_DIFFUSE: case RT_MAT_ORENNAYAR: return cosine_hemisphere_pdf(wl);
    case RT_MAT_MIRROR: case RT_MAT_GLASS: return 0;
    case RT_MAT_CONDUCTOR: {
        if (wo.z <= 0.0f || wl.z <= 0.0f) return 0.0f;
        f3 wm = norm3(wl + wo);
        if (wm.z <= 0.0f) return 0.0f;
        if (m.alpha < 0.001f) return 0.0f;
        return d_ggx(wm, m.alpha) * wm.z / (4 * dot3(wo, wm));
    }
    case RT_MAT_DIELECTRIC: {
        if (wo.z <= 0.0f) return 0.0f;
        float iorT, iorI; set_ior(wo.z, iorI, iorT, m.int_ior, m.ext_ior);
        float ri = iorI / iorT;
        f3 wm; bool reflect = wl.z * wo.z > 0;
        if (reflect) wm = norm3(wl + wo);
        else { wm = norm3((wl * ri) + wo); if (wm.z < 0) wm = -wm; }
        if (wm.z <= 0.0f) return 0.0f;
        float F = fresnel_dielectric(dot3(wo, wm), iorI, iorT);
        float D = d_ggx(wm, m.alpha);
        float pr = D * wm.z / (4 * dot3(wo, wm)), pt = 0.0f;
        if (!reflect) { float den = ri * dot3(wl, wm) + dot3(wo, wm); pt = (D * dot3(wl, wm)) / powf(den, 2); }
        return F * pr + (1 - F) * pt;
    }
    case RT_MAT_PLASTIC: {
        if (wl.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
        float iorT, iorI; set_ior(wo.z, iorI, iorT, m.int_ior, m.ext_ior);
        float ks = fresnel_dielectric(wo.z, iorI, iorT), kd = 1.0f - ks;
        float e = m.phong_e;
        float pd = (float)(wl.z / RT_PI);
        f3 wr = mk3(-wo.x, -wo.y, wo.z);
        float cosAlpha = smaxf(0.0f, dot3(wr, wl));
        float pg = (float)((e + 1) * powf(cosAlpha, e) / (2 * RT_PI));
        return (kd * pd) + (ks * pg);
    }
    }
    return 0;
}

// ---- BSDF::sample; draws in statement order, first draw of a two-draw call expression is r1
__device__ inline f3 bsdf_sample(const DScene& S, const Shading& sd, Rng& rng, col3& f, float& pdf) {
    const DMaterial& m = S.mats[sd.mat];
    switch (m.type) {
    case RT_MAT_DIFFUSE: {  // Materials.h:208
        float r1 = rt_rng_next(rng), r2 = rt_rng_next(rng);
        f3 wi = cosine_sample_hemisphere(r1, r2);
        pdf = (float)(wi.z / RT_PI);
        f = sd.albedo / (float)RT_PI;
        return to_world(sd.frame, wi);
    }
    case RT_MAT_ORENNAYAR: {  // Materials.h:759
        float r1 = rt_rng_next(rng), r2 = rt_rng_next(rng);
        f3 wl = cosine_sample_hemisphere(r1, r2);
        pdf = (float)(wl.z / RT_PI);
        f3 wi = to_world(sd.frame, wl);
        f = bsdf_eval(S, sd, wi);
        return wi;
    }
    case RT_MAT_MIRROR: {  // Materials.h:257
        f3 wo = to_local(sd.frame, sd.wo);
        f3 wi = mk3(-wo.x, -wo.y, wo.z);
        pdf = 1.0f;
        f = sd.albedo / wi.z;
        return to_world(sd.frame, wi);
    }
    case RT_MAT_CONDUCTOR: {  // Materials.h:309
        col3 eta = mkc(m.eta[0], m.eta[1], m.eta[2]), kk = mkc(m.k[0], m.k[1], m.k[2]);
        f3 wo = to_local(sd.frame, sd.wo);
        if (wo.z <= 0.0f) { pdf = 0.0f; f = mkc(0, 0, 0); return mk3(0, 0, 0); }
        if (m.alpha < 0.001f) {
            f3 wi = mk3(-wo.x, -wo.y, wo.z);
            pdf = 1.0f;
            f = sd.albedo * fresnel_conductor(wi.z, eta, kk);
            return to_world(sd.frame, wi);
        }
        float r1 = rt_rng_next(rng), r2 = rt_rng_next(rng);
        float a2 = powf(m.alpha, 2);
        float cosTheta = sqrtf((1 - r1) / (r1 * ((a2 - 1) + 1)));  // sic (Materials.h:342)
        float sinTheta = sqrtf(smaxf(0.0f, 1 - powf(cosTheta, 2)));
        float phi = (float)(2 * RT_PI * r2);
        f3 wm = mk3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
        f3 wl = ((wm * dot3(wo, wm)) * 2.0f) - wo;
        float D = d_ggx(wm, m.alpha), G = g_ggx(wl, wo, m.alpha);
        col3 F = fresnel_conductor(dot3(wl, wm), eta, kk);
        col3 brdf = ((F * D) * G) / (4.0f * wo.z * wl.z);
        pdf = D * wm.z / (4 * dot3(wo, wm));
        f = ((sd.albedo * brdf) * fabsf(wl.z)) / pdf;
        return to_world(sd.frame, wl);
    }
    case RT_MAT_GLASS: {  // Materials.h:447
        f3 wo = to_local(sd.frame, sd.wo);
        float cosTheta = wo.z, iorT, iorI;
        set_ior(cosTheta, iorI, iorT, m.int_ior, m.ext_ior);
        float ri = iorI / iorT;
        float F = fresnel_dielectric(cosTheta, iorI, iorT);
        f3 wi = mk3(0, 0, 0);
        bool reflect = rt_rng_next(rng) < F;
        if (reflect) {
            wi = mk3(-wo.x, -wo.y, wo.z);
            f = sd.albedo;
            pdf = F;
        } else {
            float sinI = sqrtf(smaxf(0.0f, 1.0f - powf(cosTheta, 2)));
            float sinT = ri * sinI;
            if (sinT >= 1.0f) {
                wi = mk3(-wo.x, -wo.y, wo.z);
                f = sd.albedo;
                pdf = 1.0f;
            } else {
                float cosT = sqrtf(smaxf(0.0f, 1 - powf(sinT, 2)));
                wi = mk3(-wo.x * ri, -wo.y * ri, 0);
                wi.z = cosTheta > 0 ? -cosT : cosT;
                wi = norm3(wi);
                pdf = (1.0f - F);
                f = sd.albedo * (pdf * ri * ri);
            }
        }
        return to_world(sd.frame, norm3(wi));
    }
    case RT_MAT_DIELECTRIC: {  // Materials.h:545
        f3 wo = to_local(sd.frame, sd.wo);
        float cosTheta = wo.z;
        if (cosTheta <= 0.0f) { pdf = 0.0f; f = mkc(0, 0, 0); return mk3(0, 0, 0); }
        float iorT, iorI; set_ior(cosTheta, iorI, iorT, m.int_ior, m.ext_ior);
        float ri = iorI / iorT;
        float r1 = rt_rng_next(rng), r2 = rt_rng_next(rng);
        float a2 = powf(m.alpha, 2);
        float cosM = sqrtf((1 - r1) / (r1 * ((a2 - 1) + 1)));
        float sinM = sqrtf(smaxf(0.0f, 1 - powf(cosM, 2)));
        float phi = (float)(2 * RT_PI * r2);
        f3 wm = mk3(sinM * cosf(phi), sinM * sinf(phi), cosM);
        f3 wl = mk3(0, 0, 0);  // Materials.h:577-580: Fresnel is taken at a default-constructed wiLocal
        float pr, pt;
        float F = fresnel_dielectric(dot3(wl, wm), iorI, iorT);
        float D = d_ggx(wm, m.alpha);
        bool reflect = false;
        if (rt_rng_next(rng) < F) {
            reflect = true;
            wl = (-wo) + ((wm * dot3(wo, wm)) * 2.0f);
            pr = D * cosM / (4 * dot3(wo, wm)); pt = 0;
        } else {
            float cosI = dot3(wo, wm);
            float sinI = sqrtf(smaxf(0.0f, 1 - powf(cosI, 2)));
            float sinT = ri * sinI;
            if (sinT >= 1.0f) {
                reflect = true;
                wl = (-wo) + ((wm * dot3(wo, wm)) * 2.0f);
                pr = D * cosM / (4 * dot3(wo, wm)); pt = 0; F = 1.0f;
            } else {
                float cosT = sqrtf(smaxf(0.0f, 1 - powf(sinT, 2)));
                wl = ((-wo) * ri) + (wm * (ri * cosI - cosT));
                if (wo.z > 0) wl = -wl;
                pr = 0.0f;
                float den = ri * dot3(wl, wm) + dot3(wo, wm);
                pt = (D * dot3(wl, wm)) / powf(den, 2);
            }
        }
        pdf = F * pr + (1 - F) * pt;
        float G = g_ggx(wl, wo, m.alpha);
        if (reflect) {
            pdf = D * wm.z / (4 * dot3(wo, wm));
            float brdf = F * D * G / (4.0f * wo.z * wl.z);
            f = ((sd.albedo * brdf) * fabsf(wl.z)) / pdf;
        } else {
            float den = ri * dot3(wl, wm) + dot3(wo, wm);
            float bsdf = ((1 - F) * D * G * dot3(wl, wm) * dot3(wo, wm)) / (powf(den, 2) * wl.z * wo.z);
            f = ((sd.albedo * bsdf) * fabsf(wl.z)) / pdf;
        }
        return to_world(sd.frame, wl);
    }
    case RT_MAT_PLASTIC: {  // Materials.h:861
        f3 wo = to_local(sd.frame, sd.wo);
        float cosTheta = wo.z;
        if (cosTheta <= 0.0f) { pdf = 1.0f; f = mkc(0, 0, 0); return mk3(0, 0, 0); }
        float iorT, iorI; set_ior(cosTheta, iorI, iorT, m.int_ior, m.ext_ior);
        float ks = fresnel_dielectric(cosTheta, iorI, iorT), kd = 1.0f - ks;
        f3 wl;
        float e = m.phong_e;
        float s1 = rt_rng_next(rng), s2 = rt_rng_next(rng);
        f3 wr = mk3(-wo.x, -wo.y, wo.z);
        if (rt_rng_next(rng) > ks) {
            wl = cosine_sample_hemisphere(s1, s2);
        } else {
            float ct = powf(s1, 1.0f / (e + 1.f));
            float st = sqrtf(smaxf(0.0f, 1.0f - powf(ct, 2)));
            float phi = (float)(2 * RT_PI * s2);
            f3 lobe = mk3(st * cosf(phi), st * sinf(phi), ct);
            Frame3 fr = frame_from(wr);
            wl = to_local(fr, lobe);  // sic: toLocal (Materials.h:914)
        }
        float cosAlpha = smaxf(0.0f, dot3(wr, wl));
        pdf = (float)((kd * (wl.z / RT_PI)) + (ks * (e + 1) * powf(cosAlpha, e) / (2 * RT_PI)));
        col3 diffuse = (sd.albedo * kd) / (float)RT_PI;
        float gloss = (float)(ks * ((e + 2) / (2 * RT_PI)) * powf(cosAlpha, e));
        f = diffuse + mkc(gloss, gloss, gloss);
        return to_world(sd.frame, wl);
    }
    }
    pdf = 0; f = mkc(0, 0, 0);
    return mk3(0, 0, 0);
}

// ---- background (Lights.h:102, :162)
__device__ inline col3 bg_evaluate(const DScene& S, f3 wi) {
    if (S.bg_kind == 0) return mkc(S.bg_colour[0], S.bg_colour[1], S.bg_colour[2]);
    float u = atan2f(wi.z, wi.x);
    u = (u < 0.0f) ? (float)(u + (2.0f * RT_PI)) : u;
    u = (float)(u / (2.0f * RT_PI));
    float v = (float)(acosf(wi.y) / RT_PI);
    return tex_sample(S, S.bg_tex, u, v);
}

// ---- EnvironmentMap's tabulated importance sampling (Lights.h:306-325, :347-424) over the tables of EnvironmentMap::build
// (scene_build.cpp).  Upstream reads every coordinate the tables produce through Vec2::operator[], which returns int
// (Sampling.h:204; pinned from the compiled reference in tests/golden/scalars_r2.npz): for coordinates in [0, 1) that is 0,
// so sample() / sampleDirectionFromLight() / PDF() end at theta = 0 -- wi = (0, 1, 0), pdf = (table pdf) * 2 pi^2 / 0 = +inf.
// The steps are kept so that the draws that do not collapse (a coordinate of exactly 1, a walk that runs off the end of
// a table whose pdfs sum to less than the draw) come out as upstream's.
__device__ inline float env_sample1d(const float* __restrict__ dist, int n, float u, float* pdf, int* offset) {
    float cdf = 0.0f;
    int index = 0;
    for (int i = 0; i < n; i++) {
        cdf += dist[i];
        if (u <= cdf) { index = i; break; }
    }
    index = index < n - 1 ? index : n - 1;
    if (offset) *offset = index;
    if (pdf) *pdf = dist[index] / (float)n;
    float t = (u - (cdf - dist[index])) / dist[index];
    return (index + t) / (float)n;
}
__device__ inline void env_sample_continuous(const DScene& S, float u0, float u1, float& pdf, float& d0, float& d1) {
    float p0, p1;
    int v;
    d1 = env_sample1d(S.env_marg, S.env_nv, u1, &p1, &v);
    d0 = env_sample1d(S.env_cond + (size_t)v * S.env_nu, S.env_nu, u0, &p0, nullptr);
    pdf = p0 * p1;
}
__device__ __forceinline__ int vec2_index(float x) { return (x != x) ? 0 : (int)x; }   // NaN: undefined upstream, 0 here and in the oracle
__device__ inline f3 env_sample_dir(const DScene& S, float u0, float u1, float& pdf) {
    float d0, d1;
    env_sample_continuous(S, u0, u1, pdf, d0, d1);
    const int iu = vec2_index(d0), iv = vec2_index(d1);
    float theta = (float)((double)(float)iv * RT_PI);              // UVToDirection (Lights.h:366-374)
    float phi = (float)((double)((float)iu * 2.0f) * RT_PI);
    float sinTheta = sinf(theta), cosTheta = cosf(theta), sinPhi = sinf(phi), cosPhi = cosf(phi);
    f3 wi = mk3(sinTheta * cosPhi, cosTheta, sinTheta * sinPhi);
    float theta2 = (float)((double)iv * RT_PI);
    float sinTheta2 = sinf(theta2);
    pdf = (float)((double)pdf * ((double)(2.0f) * RT_PI * RT_PI / (double)sinTheta2));   // Jacobian (Lights.h:396)
    return wi;
}
__device__ inline float env_pdf(const DScene& S, f3 wi) {   // EnvironmentMap::PDF (Lights.h:402-409) with PdfUV / DirectionToUV
    float u = atan2f(wi.z, wi.x);
    u = (u < 0.0f) ? (float)(u + (2.0f * RT_PI)) : u;
    u = (float)(u / (2.0f * RT_PI));
    float v = (float)(acosf(wi.y) / RT_PI);
    const int p0 = vec2_index(u), p1 = vec2_index(v);
    int iu = p0 * S.env_nu, iv = p1 * S.env_nv;
    iu = iu < 0 ? 0 : (iu > S.env_nu - 1 ? S.env_nu - 1 : iu);
    iv = iv < 0 ? 0 : (iv > S.env_nv - 1 ? S.env_nv - 1 : iv);
    float pdfUV = (S.env_cond[(size_t)iv * S.env_nu + iu] * S.env_nu * S.env_nv) / S.env_func_int;
    float theta = (float)((double)p1 * RT_PI);
    float sinTheta = sinf(theta);
    return (float)((double)pdfUV * ((double)(2.0f) * RT_PI * RT_PI) / (double)sinTheta);
}

// ---- triangle helpers on device records
__device__ __forceinline__ f3 tri_gnormal(const DTri& t) {  // Geometry.h:175
    f3 n = ld3(t.n);
    return (t.mat_gsign & 0x80000000u) ? n * -1.0f : n * 1.0f;
}
__device__ __forceinline__ f3 tri_sample_point(const DTri& t, float r1, float r2) {  // Geometry.h:162
    float alpha = 1 - sqrtf(r1);
    float beta = r2 * sqrtf(r1);
    float gamma = 1 - (alpha + beta);
    return ((ld3(t.v0) * alpha) + (ld3(t.v1) * beta)) + (ld3(t.v2) * gamma);
}
// A whole 64-B record as four 16-B rows: a shade lane gathers its records with dwordx4 loads only (fetching the 12-byte
// fields one by one is a dwordx3 / dwordx2 / dword instruction each, and every divergent load instruction costs the texture
// path about a cycle per lane whatever it fetches: k_shade's limiter).
struct TriRows { float4 a, b, c, e; };
__device__ __forceinline__ TriRows tri_rows(const DTri* t) {
    const float4* q = reinterpret_cast<const float4*>(t);
    TriRows r; r.a = q[0]; r.b = q[1]; r.c = q[2]; r.e = q[3];
    return r;
}
__device__ __forceinline__ uint32_t tri_mat_gsign(const TriRows& r) { return __float_as_uint(r.e.w); }
__device__ __forceinline__ f3 tri_gnormal(const TriRows& r) {
    f3 n = mk3(r.c.y, r.c.z, r.c.w);
    return (tri_mat_gsign(r) & 0x80000000u) ? n * -1.0f : n * 1.0f;
}
__device__ __forceinline__ f3 tri_sample_point(const TriRows& r, float r1, float r2) {
    float alpha = 1 - sqrtf(r1);
    float beta = r2 * sqrtf(r1);
    float gamma = 1 - (alpha + beta);
    return ((mk3(r.a.x, r.a.y, r.a.z) * alpha) + (mk3(r.a.w, r.b.x, r.b.y) * beta)) + (mk3(r.b.z, r.b.w, r.c.x) * gamma);
}

// Scene::calculateShadingData (Scene.h:185-215) for a hit
__device__ inline Shading shading_data(const DScene& S, f3 o, f3 dir, float t, float alpha, float beta, uint32_t rank) {
    Shading sd;
    // the shading record whole, and nothing of the intersection record: its material word is repeated here (one cache line per
    // shading point instead of two).  ShadingData::gNormal (Scene.h:191, :201-203) is not kept: nothing on the path reads it
    // (Renderer.h:138 is in an unused helper); lights take their triangle's gNormal from the triangle (Lights.h:44-82).
    const float4* sq = reinterpret_cast<const float4*>(S.trishade + rank);
    const float4 s0 = sq[0], s1 = sq[1], s2 = sq[2], s3 = sq[3];
    const uint32_t mat_gsign = __float_as_uint(s3.w);
    float gamma = 1.0f - (alpha + beta);
    sd.x = o + (dir * t);
    f3 n = ((mk3(s0.x, s0.y, s0.z) * alpha) + (mk3(s0.w, s1.x, s1.y) * beta)) + (mk3(s1.z, s1.w, s2.x) * gamma);
    sd.sN = norm3(n);
    sd.tu = s2.y * alpha + s2.w * beta + s3.y * gamma;
    sd.tv = s2.z * alpha + s3.x * beta + s3.z * gamma;
    sd.mat = mat_gsign & 0x7FFFFFFFu;
    sd.wo = -dir;
    if (S.mats[sd.mat].flags & RT_MF_TWOSIDED) {
        if (dot3(sd.wo, sd.sN) < 0) sd.sN = -sd.sN;
    }
    sd.frame = frame_from(sd.sN);
    sd.albedo = tex_sample(S, S.mats[sd.mat].tex, sd.tu, sd.tv);
    return sd;
}
