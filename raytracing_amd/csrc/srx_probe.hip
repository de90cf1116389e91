// raytracing_amd/csrc/srx_probe.hip — C-ABI: the shading path's device math on arrays, for parity tests.
//
// The helpers of srx_device.h restate XYZ.cpp (reflect :178-186, dot :279, cdot :282, cross :292, normalize :237,
// linear_mix :397, Quat::applyRotation :679-683), Matrix3x3::applyTBNMatrix (Matrix.h:62-64) and VecLib's polynomial
// trig (VecLib.cpp:186-267). k_shade uses them inline; this kernel runs the SAME inline functions over arrays so that a
// test can hold the DEVICE's results against the reference's own object code (oracle/_ref/libref_xyz.so,
// libref_matrix.so, ref_veclib_probe), bit for bit, not only against the CPU restatement. Nothing in the render path calls it.
#include "srx_host.h"

#define SRX_PROBE_FLOATS 26

__global__ __launch_bounds__(256) void k_math_probe(const float* a3, const float* b3, const float* c3, const float* q4, float s, uint32_t n, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 a(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]), b(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]), c(c3[3 * i], c3[3 * i + 1], c3[3 * i + 2]);
    const float qx = q4[4 * i], qy = q4[4 * i + 1], qz = q4[4 * i + 2], qw = q4[4 * i + 3];
    float* o = out + (size_t)i * SRX_PROBE_FLOATS;
    const f3 r = reflect3(a, b), x = cross3(a, b), nn = normalize3(a), rot = apply_rotation(a, qx, qy, qz, qw), mix = linear_mix3(s, a, b);
    const f3 tbn = apply_tbn(a, b, c, f3(qx, qy, qz));
    o[0] = r.x; o[1] = r.y; o[2] = r.z;
    o[3] = x.x; o[4] = x.y; o[5] = x.z;
    o[6] = nn.x; o[7] = nn.y; o[8] = nn.z;
    o[9] = dot3(a, b); o[10] = cdot3(a, b);
    o[11] = rot.x; o[12] = rot.y; o[13] = rot.z;
    o[14] = mix.x; o[15] = mix.y; o[16] = mix.z;
    o[17] = tbn.x; o[18] = tbn.y; o[19] = tbn.z;
    o[20] = full_sin(a.x); o[21] = full_cos(a.x); o[22] = vtan(a.x);
    // quirk Q4 as the shade kernel writes it (`float - XYZ` adds, XYZ.cpp:431-433)
    o[23] = s + a.x; o[24] = s + a.y; o[25] = s + a.z;
}

extern "C" int srx_math_probe(const float* a3, const float* b3, const float* c3, const float* q4, float s, uint32_t n, float* out26) {
    if (!a3 || !b3 || !c3 || !q4 || !out26) SRX_FAIL("srx_math_probe: null argument");
    if (n == 0) return 0;
    if (n > (1u << 24)) SRX_FAIL("srx_math_probe: at most 2^24 elements");
    float* d = nullptr;
    const size_t in_floats = (size_t)n * 13, out_floats = (size_t)n * SRX_PROBE_FLOATS;
    SRX_HIP(hipMalloc((void**)&d, (in_floats + out_floats) * sizeof(float)));
    float *da = d, *db = d + 3 * (size_t)n, *dc = d + 6 * (size_t)n, *dq = d + 9 * (size_t)n, *dout = d + in_floats;
    hipError_t e = hipMemcpy(da, a3, (size_t)n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db, b3, (size_t)n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dc, c3, (size_t)n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dq, q4, (size_t)n * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_math_probe, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dc, dq, s, n, dout); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out26, dout, out_floats * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) SRX_FAIL(std::string("srx_math_probe: ") + hipGetErrorString(e));
    return 0;
}
