// oracle/ref_build/ref_matrix_driver.cpp — extern "C" wrappers around the reference's own Matrix.h and Transformation.h,
// compiled where they lie (both are header-only and build unmodified with this directory's flags; XYZ.o supplies the
// vector / quaternion code they call). Test infrastructure only: tests use it to pin, bit for bit,
//   * Matrix3x3::applyTBNMatrix (Matrix.h:62-64): every bounce direction goes through it (Core.h:510-547)
//     — the oracle's restatement (oracle/sro_math.h) and the DEVICE's apply_tbn (csrc/srx_device.h, via srx_math_probe);
//   * Transformation::stack / collpase / apply (Transformation.h:24-57): how the loader flattens a node's transform chain
//     before Scene::package (Primitives.h:423-434) — the oracle's loader (oracle/sro_gltf.py), which the
//     product's loader (csrc/srx_gltf.cpp) equals float for float (tests/test_gltf_cpu.py).
#include "Matrix.h"
#include "Transformation.h"
#include <cstdint>

extern "C" {
void ref_apply_tbn(const float* p3, const float* t3, const float* b3, const float* n3, uint32_t n, float* out3) {
    for (uint32_t i = 0; i < n; i++) {
        const XYZ p(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]), T(t3[3 * i], t3[3 * i + 1], t3[3 * i + 2]);
        const XYZ B(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]), N(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]);
        const XYZ r = Matrix3x3::applyTBNMatrix(p, T, B, N);
        out3[3 * i] = r.X; out3[3 * i + 1] = r.Y; out3[3 * i + 2] = r.Z;
    }
}
// kinds[k]: 0 = translation (vals4[4k .. 4k+2]), 1 = rotation quaternion (vals4[4k .. 4k+3]), 2 = both (rotation here,
// the translation in the NEXT vals4 row, which is then skipped). The chain is collapsed with Transformation::collpase and
// applied to the points (rotate, then translate: Transformation::apply) and, rotation only, to the normals.
void ref_collapse_apply(const int32_t* kinds, const float* vals4, uint32_t n_transforms, const float* pts3, const float* nrm3, uint32_t n_pts,
                        float* out_pts3, float* out_nrm3) {
    std::vector<Transformation> chain;
    for (uint32_t k = 0; k < n_transforms; k++) {
        const float* v = vals4 + 4 * k;
        if (kinds[k] == 0) chain.push_back(Transformation(XYZ(v[0], v[1], v[2])));
        else if (kinds[k] == 1) chain.push_back(Transformation(Quat(v[0], v[1], v[2], v[3])));
        else { chain.push_back(Transformation(XYZ(v[4], v[5], v[6]), Quat(v[0], v[1], v[2], v[3]))); k++; }
    }
    Transformation T = Transformation::collpase(chain);
    Transformation R = T.rotation();
    for (uint32_t i = 0; i < n_pts; i++) {
        XYZ p(pts3[3 * i], pts3[3 * i + 1], pts3[3 * i + 2]);
        T.apply(p);
        out_pts3[3 * i] = p.X; out_pts3[3 * i + 1] = p.Y; out_pts3[3 * i + 2] = p.Z;
        XYZ q(nrm3[3 * i], nrm3[3 * i + 1], nrm3[3 * i + 2]);
        R.apply(q);
        out_nrm3[3 * i] = q.X; out_nrm3[3 * i + 1] = q.Y; out_nrm3[3 * i + 2] = q.Z;
    }
}
}
