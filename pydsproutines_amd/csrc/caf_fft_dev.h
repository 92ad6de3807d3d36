// Device-side complex helpers and the in-register inverse DFT butterflies shared by the LDS-resident FFT kernels
// (caf_fused.hip: 16384-point hypothesis transform; caf_perdelay.hip: per-delay row transforms).
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

namespace caf {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// (spelled with explicit fused multiply-adds: left to the contraction pass, the same source line rounds differently in
// different surroundings, and the engines' modes -- surface, no surface, rows -- must agree bit for bit)
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 mulj(float2 a) { return make_float2(-a.y, a.x); }  // * (+j)
// inverse 4-point DFT (kernel e^{+j 2 pi m n / 4}), in place
__device__ __forceinline__ void idft4(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 s02 = cadd(a0, a2), d02 = csub(a0, a2);
    const float2 s13 = cadd(a1, a3), d13 = mulj(csub(a1, a3));
    a0 = cadd(s02, s13);
    a1 = cadd(d02, d13);
    a2 = csub(s02, s13);
    a3 = csub(d02, d13);
}

// inverse 16-point DFT in registers: v[k] <- sum_m v[m] W^{mk}, W = e^{+j 2 pi / 16}
__device__ __forceinline__ void idft16(float2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f;  // cos(pi/8)
    constexpr float S1 = 0.38268343236508977f;  // sin(pi/8)
    constexpr float R2 = 0.70710678118654752f;  // 1/sqrt(2)
    // stage 1: for each m2 in 0..3, DFT4 over m1 of v[4 m1 + m2]  ->  v[4 n1 + m2] = u[m2][n1]
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) idft4(v[m2], v[4 + m2], v[8 + m2], v[12 + m2]);
    // internal twiddles W^{m2 n1}
    {
        const float2 w1 = make_float2(C1, S1), w3 = make_float2(S1, C1);
        v[4 * 1 + 1] = cmul(v[4 * 1 + 1], w1);
        v[4 * 2 + 1] = make_float2((v[4 * 2 + 1].x - v[4 * 2 + 1].y) * R2, (v[4 * 2 + 1].x + v[4 * 2 + 1].y) * R2);  // W^2
        v[4 * 3 + 1] = cmul(v[4 * 3 + 1], w3);
        v[4 * 1 + 2] = make_float2((v[4 * 1 + 2].x - v[4 * 1 + 2].y) * R2, (v[4 * 1 + 2].x + v[4 * 1 + 2].y) * R2);  // W^2
        v[4 * 2 + 2] = mulj(v[4 * 2 + 2]);                                                                             // W^4
        v[4 * 3 + 2] = make_float2((-v[4 * 3 + 2].x - v[4 * 3 + 2].y) * R2, (v[4 * 3 + 2].x - v[4 * 3 + 2].y) * R2);  // W^6
        v[4 * 1 + 3] = cmul(v[4 * 1 + 3], w3);
        v[4 * 2 + 3] = make_float2((-v[4 * 2 + 3].x - v[4 * 2 + 3].y) * R2, (v[4 * 2 + 3].x - v[4 * 2 + 3].y) * R2);  // W^6
        v[4 * 3 + 3] = cmul(v[4 * 3 + 3], make_float2(-C1, -S1));                                                      // W^9
    }
    // stage 2: for each n1, DFT4 over m2 of v[4 n1 + m2] -> Y[n1 + 4 n2] at v[4 n1 + n2]
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) idft4(v[4 * n1 + 0], v[4 * n1 + 1], v[4 * n1 + 2], v[4 * n1 + 3]);
    // 4x4 transpose of register names so that v[k] = Y[k]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) {
            const float2 t = v[4 * a + b];
            v[4 * a + b] = v[4 * b + a];
            v[4 * b + a] = t;
        }
}

// inverse 2-point and 8-point DFTs (kernel e^{+j 2 pi m n / R}), in place, natural order in and out
__device__ __forceinline__ void idft2(float2& a0, float2& a1) {
    const float2 s = cadd(a0, a1), d = csub(a0, a1);
    a0 = s;
    a1 = d;
}
__device__ __forceinline__ void idft8(float2 (&v)[8]) {
    constexpr float R2 = 0.70710678118654752f;
    // two 4-point transforms over the even / odd inputs, then the radix-2 combination with W8^k = e^{+j pi k / 4}
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    idft4(e0, e1, e2, e3);
    idft4(o0, o1, o2, o3);
    o1 = make_float2((o1.x - o1.y) * R2, (o1.x + o1.y) * R2);    // * W8^1
    o2 = mulj(o2);                                                 // * W8^2
    o3 = make_float2((-o3.x - o3.y) * R2, (o3.x - o3.y) * R2);   // * W8^3
    v[0] = cadd(e0, o0);
    v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1);
    v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2);
    v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3);
    v[7] = csub(e3, o3);
}

// inverse 3-point DFT (kernel e^{+j 2 pi n k / 3}), in place
__device__ __forceinline__ void idft3(float2& x0, float2& x1, float2& x2) {
    constexpr float S = 0.86602540378443865f;  // sin(2 pi / 3)
    const float2 t = cadd(x1, x2), d = csub(x1, x2);
    const float2 a = make_float2(x0.x - 0.5f * t.x, x0.y - 0.5f * t.y);
    const float2 b = make_float2(S * d.x, S * d.y);
    x0 = cadd(x0, t);
    x1 = make_float2(a.x - b.y, a.y + b.x);  // a + j b
    x2 = make_float2(a.x + b.y, a.y - b.x);  // a - j b
}
// inverse 5-point DFT (kernel e^{+j 2 pi n k / 5})
__device__ __forceinline__ void idft5(float2& x0, float2& x1, float2& x2, float2& x3, float2& x4) {
    constexpr float C1 = 0.30901699437494742f, C2 = -0.80901699437494742f;  // cos(2 pi / 5), cos(4 pi / 5)
    constexpr float S1 = 0.95105651629515357f, S2 = 0.58778525229247313f;   // sin(2 pi / 5), sin(4 pi / 5)
    const float2 t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
    const float2 a1 = make_float2(x0.x + C1 * t1.x + C2 * t2.x, x0.y + C1 * t1.y + C2 * t2.y);
    const float2 a2 = make_float2(x0.x + C2 * t1.x + C1 * t2.x, x0.y + C2 * t1.y + C1 * t2.y);
    const float2 b1 = make_float2(S1 * t3.x + S2 * t4.x, S1 * t3.y + S2 * t4.y);
    const float2 b2 = make_float2(S2 * t3.x - S1 * t4.x, S2 * t3.y - S1 * t4.y);
    x0 = make_float2(x0.x + t1.x + t2.x, x0.y + t1.y + t2.y);
    x1 = make_float2(a1.x - b1.y, a1.y + b1.x);  // a1 + j b1
    x4 = make_float2(a1.x + b1.y, a1.y - b1.x);  // a1 - j b1
    x2 = make_float2(a2.x - b2.y, a2.y + b2.x);
    x3 = make_float2(a2.x + b2.y, a2.y - b2.x);
}
// inverse 10-point DFT in place: X[k] = E[k mod 5] + W10^k O[k mod 5]
__device__ __forceinline__ void idft10(float2 (&v)[10]) {
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], e4 = v[8];
    float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7], o4 = v[9];
    idft5(e0, e1, e2, e3, e4);
    idft5(o0, o1, o2, o3, o4);
    // W10^k = e^{+j 2 pi k / 10}, k = 1 .. 4
    o1 = cmul(o1, make_float2(0.80901699437494742f, 0.58778525229247313f));
    o2 = cmul(o2, make_float2(0.30901699437494742f, 0.95105651629515357f));
    o3 = cmul(o3, make_float2(-0.30901699437494742f, 0.95105651629515357f));
    o4 = cmul(o4, make_float2(-0.80901699437494742f, 0.58778525229247313f));
    v[0] = cadd(e0, o0), v[5] = csub(e0, o0);
    v[1] = cadd(e1, o1), v[6] = csub(e1, o1);
    v[2] = cadd(e2, o2), v[7] = csub(e2, o2);
    v[3] = cadd(e3, o3), v[8] = csub(e3, o3);
    v[4] = cadd(e4, o4), v[9] = csub(e4, o4);
}
// Lane mapping of the planar in-LDS engines (caf_fused.hip, "Thread <-> data of the four passes"): the pass-1 butterfly
// of thread tid, and its inverse (the "butterfly order" in which template-spectrum rows and 32768-point block spectra
// are stored: element m at (m & ~1023) + fp_tid_of(m & 1023)).
__device__ __forceinline__ uint32_t fp_m2(uint32_t tid) {
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t b = (lane & 3) + 4 * (wave & 3), c = ((lane >> 2) & 3) + 4 * (wave >> 2), d = lane >> 4;
    return 64 * b + 4 * c + d;
}
__host__ __device__ __forceinline__ uint32_t fp_tid_of(uint32_t m2) {
    const uint32_t b = m2 >> 6, c = (m2 >> 2) & 15, d = m2 & 3;
    return (b & 3) | ((c & 3) << 2) | (d << 4) | ((b >> 2) << 6) | ((c >> 2) << 8);
}

}  // namespace caf
