// Mixed-radix inverse DFT butterflies in registers, radices 2 .. 20 and 23, 25 (composite ones as Cooley-Tukey butterflies, primes from 7 up in the direct form) (kernel e^{+j 2 pi n k / R}): shared
// by the plan-driven kernel of caf_perdelay_mr.hip and the run-time-specialised kernel of caf_perdelay_jit.h (this header
// is also compiled by hiprtc: no standard-library includes).
#pragma once
#include "caf_fft_dev.h"

namespace caf {

// e^{+j 2 pi q / R} as compile-time constants (the butterflies below index them with unrolled loop counters)
constexpr double mr_poly_sin(double x) {  // |x| <= pi / 4
    const double x2 = x * x;
    return x * (1.0 + x2 * (-1.0 / 6 + x2 * (1.0 / 120 + x2 * (-1.0 / 5040 + x2 * (1.0 / 362880 + x2 * (-1.0 / 39916800 + x2 * (1.0 / 6227020800.0)))))));
}
constexpr double mr_poly_cos(double x) {  // |x| <= pi / 4
    const double x2 = x * x;
    return 1.0 + x2 * (-0.5 + x2 * (1.0 / 24 + x2 * (-1.0 / 720 + x2 * (1.0 / 40320 + x2 * (-1.0 / 3628800 + x2 * (1.0 / 479001600 + x2 * (-1.0 / 87178291200.0)))))));
}
constexpr double MR_PI = 3.14159265358979323846;
// cos / sin of 2 pi q / r through the octant of q / r: exact zeros, halves and signs where they belong
constexpr double mr_cos_frac(int q, int r) {
    q %= r;
    if (2 * q > r) return mr_cos_frac(r - q, r);               // cos(2 pi - t) = cos t
    if (4 * q > r) return -mr_cos_frac(r - 2 * q, 2 * r);      // cos(t) = -cos(pi - t); pi - t = 2 pi (r - 2q) / (2r)
    if (8 * q > r) return mr_poly_sin(MR_PI / 2 - 2 * MR_PI * q / r);
    return mr_poly_cos(2 * MR_PI * q / r);
}
constexpr double mr_sin_frac(int q, int r) {
    q %= r;
    if (2 * q > r) return -mr_sin_frac(r - q, r);
    if (4 * q > r) return mr_sin_frac(r - 2 * q, 2 * r);
    if (8 * q > r) return mr_poly_cos(MR_PI / 2 - 2 * MR_PI * q / r);
    return mr_poly_sin(2 * MR_PI * q / r);
}
template <int R>
struct MrW {
    float c[R], s[R];
    constexpr MrW() : c(), s() {
        for (int q = 0; q < R; ++q) {
            c[q] = (float)mr_cos_frac(q, R);
            s[q] = (float)mr_sin_frac(q, R);
        }
    }
};
template <int R>
__device__ __forceinline__ void mr_idft(float2* v);

// inverse DFT of a prime number of points R (kernel e^{+j 2 pi n k / R}) in the direct form on pair sums and differences:
// X[k], X[R - k] = a_k +- j b_k with a_k = x0 + sum_m cos(2 pi m k / R) (x_m + x_{R-m}), b_k = sum_m sin(2 pi m k / R) (x_m - x_{R-m}):
// (R - 1)^2 real multiply-adds, constants folded at compile time.  7 in the mixed-radix plans; 11, 13, 17, 19, 23 make cutouts
// with those prime factors one-kernel lengths too (24 .. 44 flop per point where a 16-point butterfly costs 11: still far
// below the three trips through HBM of product rows -> rocFFT rows -> argmax).
template <int R>
__device__ __forceinline__ void idft_prime(float2* v) {
    constexpr MrW<R> W = MrW<R>();
    constexpr int H = (R - 1) / 2;
    const float2 x0 = v[0];
    float2 t[H], d[H];
#pragma unroll
    for (int m = 1; m <= H; ++m) t[m - 1] = cadd(v[m], v[R - m]), d[m - 1] = csub(v[m], v[R - m]);
    float2 s0 = x0;
#pragma unroll
    for (int m = 0; m < H; ++m) s0 = cadd(s0, t[m]);
    v[0] = s0;
#pragma unroll
    for (int k = 1; k <= H; ++k) {
        float2 a = x0, b = make_float2(0.f, 0.f);
#pragma unroll
        for (int m = 1; m <= H; ++m) {
            const float c = W.c[(m * k) % R], sn = W.s[(m * k) % R];
            a.x = __builtin_fmaf(c, t[m - 1].x, a.x), a.y = __builtin_fmaf(c, t[m - 1].y, a.y);
            b.x = __builtin_fmaf(sn, d[m - 1].x, b.x), b.y = __builtin_fmaf(sn, d[m - 1].y, b.y);
        }
        v[k] = make_float2(a.x - b.y, a.y + b.x);      // a + j b
        v[R - k] = make_float2(a.x + b.y, a.y - b.x);  // a - j b
    }
}
__device__ __forceinline__ void idft7(float2* v) { idft_prime<7>(v); }

// Cooley-Tukey butterfly of R = A B points in registers: n = B a + b, k = k1 + A k2;
//   X[k1 + A k2] = sum_b W_B^{b k2} ( W_R^{b k1} sum_a W_A^{a k1} x[B a + b] )
template <int A, int B>
__device__ __forceinline__ void idft_ct(float2* v) {
    constexpr int R = A * B;
    constexpr MrW<R> W = MrW<R>();
    float2 y[R];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        float2 t[A];
#pragma unroll
        for (int a = 0; a < A; ++a) t[a] = v[B * a + b];
        mr_idft<A>(t);
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1)
            y[b * A + k1] = (b * k1) ? cmul(t[k1], make_float2(W.c[(b * k1) % R], W.s[(b * k1) % R])) : t[k1];
    }
#pragma unroll
    for (int k1 = 0; k1 < A; ++k1) {
        float2 u[B];
#pragma unroll
        for (int b = 0; b < B; ++b) u[b] = y[b * A + k1];
        mr_idft<B>(u);
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2) v[k1 + A * k2] = u[k2];
    }
}

template <int R>
__device__ __forceinline__ void mr_idft(float2* v) {
    if constexpr (R == 2) idft2(v[0], v[1]);
    if constexpr (R == 3) idft3(v[0], v[1], v[2]);
    if constexpr (R == 4) idft4(v[0], v[1], v[2], v[3]);
    if constexpr (R == 5) idft5(v[0], v[1], v[2], v[3], v[4]);
    if constexpr (R == 6) idft_ct<3, 2>(v);
    if constexpr (R == 7) idft7(v);
    if constexpr (R == 8) idft8(*reinterpret_cast<float2(*)[8]>(v));
    if constexpr (R == 9) idft_ct<3, 3>(v);
    if constexpr (R == 10) idft10(*reinterpret_cast<float2(*)[10]>(v));
    if constexpr (R == 12) idft_ct<4, 3>(v);
    if constexpr (R == 14) idft_ct<7, 2>(v);
    if constexpr (R == 15) idft_ct<5, 3>(v);
    if constexpr (R == 16) idft16(*reinterpret_cast<float2(*)[16]>(v));
    if constexpr (R == 18) idft_ct<3, 6>(v);
    if constexpr (R == 20) idft_ct<5, 4>(v);
    if constexpr (R == 25) idft_ct<5, 5>(v);
    if constexpr (R == 11 || R == 13 || R == 17 || R == 19 || R == 23) idft_prime<R>(v);
}

}  // namespace caf
