// oracle/sro_post.h — CPU restatement of the output stage.
// TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). PARITY UNPINNED: the reference holds no fixture for this stage and
// its own code cannot be built here (ImageProcessing.h:3-11 needs OpenColorIO and a config file outside the repository).
//
// Follows ImageHandler::post_process_pixel (ImageProcessing.h:137-151) with its helpers luminance (:115-118),
// apply_gamma (:128-133), rgb2hsv (:12-55), hsv2rgb (:58-114), XYZ::clamp (XYZ.cpp:306-315) and the 8-bit conversion of
// GUI::commit_pixel (GUI.h:104-107: float -> sf::Uint8, truncation). Float/double promotions are kept as written.
// The OpenColorIO step (:141-143, "Linear Rec.709" -> "AGX Base sRGB" of an AgX config that is not in the reference's
// repository) is a hook: tonemap 0 leaves it out, tonemap 1 is the public analytic AgX approximation (inset matrix,
// log2 encoding over [-12.47393, 4.026069] EV, 6th-order sigmoid fit, outset matrix) shared with the product; it stands in
// for the view transform and is NOT the reference's LUT.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>

namespace sro {

struct RGB { float X, Y, Z; };

inline float post_luminance(RGB v) { return v.X * 0.2126f + v.Y * 0.7152f + v.Z * 0.0722f; }   // :115-118, XYZ.cpp:279-281

inline RGB post_apply_gamma(RGB in, float gamma) {                                              // :128-133
    float lums = post_luminance(in);
    if (lums == 0) return RGB{0, 0, 0};
    float post_lums = (float)std::pow((double)lums, 1.0 / (double)gamma);
    return RGB{in.X * post_lums / lums, in.Y * post_lums / lums, in.Z * post_lums / lums};
}

inline RGB post_rgb2hsv(RGB in) {                                                               // :12-55
    RGB out;
    double min, max, delta;
    min = in.X < in.Y ? in.X : in.Y;
    min = min < in.Z ? min : in.Z;
    max = in.X > in.Y ? in.X : in.Y;
    max = max > in.Z ? max : in.Z;
    out.Z = (float)max;
    delta = max - min;
    if (delta < 0.00001) { out.Y = 0; out.X = 0; return out; }
    if (max > 0.0) out.Y = (float)(delta / max);
    else { out.Y = 0.0f; out.X = NAN; return out; }
    if (in.X >= max) out.X = (float)((in.Y - in.Z) / delta);
    else if (in.Y >= max) out.X = (float)(2.0 + (in.Z - in.X) / delta);
    else out.X = (float)(4.0 + (in.X - in.Y) / delta);
    out.X = (float)(out.X * 60.0);
    if (out.X < 0.0) out.X = (float)(out.X + 360.0);
    return out;
}

inline RGB post_hsv2rgb(RGB in) {                                                               // :58-114
    double hh, p, q, t, ff;
    long i;
    RGB out;
    if (in.Y <= 0.0) return RGB{in.Z, in.Z, in.Z};
    hh = in.X;
    if (hh >= 360.0) hh = 0.0;
    hh /= 60.0;
    i = (long)hh;
    ff = hh - i;
    p = in.Z * (1.0 - in.Y);
    q = in.Z * (1.0 - (in.Y * ff));
    t = in.Z * (1.0 - (in.Y * (1.0 - ff)));
    switch (i) {
    case 0: out = RGB{in.Z, (float)t, (float)p}; break;
    case 1: out = RGB{(float)q, in.Z, (float)p}; break;
    case 2: out = RGB{(float)p, in.Z, (float)t}; break;
    case 3: out = RGB{(float)p, (float)q, in.Z}; break;
    case 4: out = RGB{(float)t, (float)p, in.Z}; break;
    case 5:
    default: out = RGB{in.Z, (float)p, (float)q}; break;
    }
    return out;
}

// stand-in for the OCIO view transform (see the header); display-encoded output like "AGX Base sRGB"
inline RGB post_agx_approx(RGB v) {
    const float min_ev = -12.47393f, max_ev = 4.026069f;
    float c[3] = {0.842479062253094f * v.X + 0.0784335999999992f * v.Y + 0.0792237451477643f * v.Z,
                  0.0423282422610123f * v.X + 0.878468636469772f * v.Y + 0.0791661274605434f * v.Z,
                  0.0423756549057051f * v.X + 0.0784336f * v.Y + 0.879142973793104f * v.Z};
    for (int k = 0; k < 3; k++) {
        float x = std::log2(std::max(c[k], 1e-10f));
        x = std::min(std::max(x, min_ev), max_ev);
        x = (x - min_ev) / (max_ev - min_ev);
        const float x2 = x * x, x4 = x2 * x2;
        c[k] = 15.5f * x4 * x2 - 40.14f * x4 * x + 31.96f * x4 - 6.868f * x2 * x + 0.4298f * x2 + 0.1191f * x - 0.00232f;
    }
    return RGB{1.19687900512017f * c[0] - 0.0980208811401368f * c[1] - 0.0990297440797205f * c[2],
               -0.0528968517574562f * c[0] + 1.15190312990417f * c[1] - 0.0989611768448433f * c[2],
               -0.0529716355144438f * c[0] - 0.0980434501171241f * c[1] + 1.15107367264116f * c[2]};
}

inline float post_clamp01(float v) { return std::min(std::max(v, 0.0f), 1.0f); }               // XYZ.cpp:306-315

// ImageProcessing.h:137-151; out = clamp(.., 0, 1) * 255 as floats
inline RGB post_process_pixel(RGB lums, int tonemap) {
    lums = post_apply_gamma(lums, 1);
    if (tonemap == 1) lums = post_agx_approx(lums);
    lums = post_apply_gamma(lums, 2.2f);
    RGB hsv = post_rgb2hsv(lums);
    lums = post_hsv2rgb(hsv);
    return RGB{post_clamp01(lums.X) * 255, post_clamp01(lums.Y) * 255, post_clamp01(lums.Z) * 255};
}

// GUI.h:104-107: sf::Color(float, float, float) converts each to sf::Uint8 (truncation; a NaN is stored as 0 here, the
// low byte of what x86's cvttss2si returns for it)
inline uint8_t post_to_u8(float v) { return (v != v) ? (uint8_t)0 : (uint8_t)(int)v; }

}  // namespace sro
