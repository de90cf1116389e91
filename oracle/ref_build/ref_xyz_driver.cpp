// oracle/ref_build/ref_xyz_driver.cpp — extern "C" wrappers around the reference's own XYZ.cpp object code,
// compiled against the reference's XYZ.h where it lies. Test infrastructure only: tests use it to pin the
// restated vector math (oracle/sro_math.h) bit for bit.
#include "XYZ.h"
#include <cstdint>

extern "C" {
void ref_vec_ops(const float* a3, const float* b3, uint32_t n, float* reflect3, float* cross3, float* norm3, float* dot1) {
    for (uint32_t i = 0; i < n; i++) {
        XYZ a(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]), b(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]);
        XYZ r = XYZ::reflect(a, b), c = XYZ::cross(a, b), nn = XYZ::normalize(a);
        reflect3[3 * i] = r.X; reflect3[3 * i + 1] = r.Y; reflect3[3 * i + 2] = r.Z;
        cross3[3 * i] = c.X; cross3[3 * i + 1] = c.Y; cross3[3 * i + 2] = c.Z;
        norm3[3 * i] = nn.X; norm3[3 * i + 1] = nn.Y; norm3[3 * i + 2] = nn.Z;
        dot1[i] = XYZ::dot(a, b);
    }
}
void ref_apply_rotation(const float* p3, const float* q4, uint32_t n, float* out3) {
    for (uint32_t i = 0; i < n; i++) {
        Quat q(q4[4 * i], q4[4 * i + 1], q4[4 * i + 2], q4[4 * i + 3]);
        XYZ r = Quat::applyRotation(XYZ(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]), q);
        out3[3 * i] = r.X; out3[3 * i + 1] = r.Y; out3[3 * i + 2] = r.Z;
    }
}
// quirk Q4: `float - XYZ` adds (XYZ.cpp:431-433); linear_mix and cdot are used by the BSDF
void ref_misc(const float* a3, const float* b3, float c, uint32_t n, float* fminus3, float* mix3, float* cdot1, float* slope3) {
    for (uint32_t i = 0; i < n; i++) {
        XYZ a(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]), b(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]);
        XYZ f = c - a, m = XYZ::linear_mix(c, a, b), s = XYZ::slope(a, b);
        fminus3[3 * i] = f.X; fminus3[3 * i + 1] = f.Y; fminus3[3 * i + 2] = f.Z;
        mix3[3 * i] = m.X; mix3[3 * i + 1] = m.Y; mix3[3 * i + 2] = m.Z;
        cdot1[i] = XYZ::cdot(a, b);
        slope3[3 * i] = s.X; slope3[3 * i + 1] = s.Y; slope3[3 * i + 2] = s.Z;
    }
}
// scene ingestion (oracle/sro_gltf.py, csrc/srx_gltf.cpp): the swizzle constructors the loader feeds glTF node data
// through (XYZ.cpp:97,115-121,613) and Quat::multiply (XYZ.cpp:626-633) as Transformation::stack uses it
void ref_swizzles(const double* attr4, uint32_t n, float* pos3, float* scale3, float* rot4, float* vswz3) {
    std::vector<char> swizzle = {0, -2, 1}, rot_swizzle = {0, -2, 1, 3}, scale_swizzle = {0, 2, 1};   // FileLoading.h:450-452
    for (uint32_t i = 0; i < n; i++) {
        std::vector<double> a4(attr4 + 4 * i, attr4 + 4 * i + 4), a3(attr4 + 4 * i, attr4 + 4 * i + 3);
        XYZ p(a3, swizzle), sc(a3, scale_swizzle);
        Quat q(a4, rot_swizzle);
        XYZ v = XYZ((float)a3[0], (float)a3[1], (float)a3[2]).swizzle(swizzle);
        pos3[3 * i] = p.X; pos3[3 * i + 1] = p.Y; pos3[3 * i + 2] = p.Z;
        scale3[3 * i] = sc.X; scale3[3 * i + 1] = sc.Y; scale3[3 * i + 2] = sc.Z;
        rot4[4 * i] = q.X; rot4[4 * i + 1] = q.Y; rot4[4 * i + 2] = q.Z; rot4[4 * i + 3] = q.W;
        vswz3[3 * i] = v.X; vswz3[3 * i + 1] = v.Y; vswz3[3 * i + 2] = v.Z;
    }
}
void ref_quat_multiply(const float* a4, const float* b4, uint32_t n, float* out4) {
    for (uint32_t i = 0; i < n; i++) {
        Quat a(a4[4 * i], a4[4 * i + 1], a4[4 * i + 2], a4[4 * i + 3]), b(b4[4 * i], b4[4 * i + 1], b4[4 * i + 2], b4[4 * i + 3]);
        Quat r = Quat::multiply(a, b);
        out4[4 * i] = r.X; out4[4 * i + 1] = r.Y; out4[4 * i + 2] = r.Z; out4[4 * i + 3] = r.W;
    }
}
}
