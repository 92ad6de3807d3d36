// Per-delay correlator for ONE cutout length, compiled at run time (hiprtc, caf_jit.hip) with every size a constant:
//   rx window x conj(cutout) -> N-point FFT in LDS -> |.|^2 -> (max, first argmax) [+ optional |.|^2 / complex planes]
// -- the reference's literal per-delay algorithm (fastXcorr branches B / C, xcorrRoutines.py:511-566; cp_fastXcorr :29-167, whose
// cutout length is a free argument, benchmark_xcorrs.py:62-71; multiplySlices.cu:113-216 + cuFFT + argmax.cu:93-153;
// IppXcorrFFT.cpp:94-178) -- for N = 2^a 3^b 5^c 7^d.  The reference compiles its kernels at run time too (cupy RawModule / NVRTC).
//
// What a length-specialised kernel buys over the plan-driven one (caf_perdelay_mr.hip: radices, strides, butterfly counts and
// divisors are kernel arguments there): every LDS access is `base + immediate`, divisions are by constants, no clamped or
// switch-selected butterflies -- and a transform that needs no index arithmetic per element at all:
//
//   * IN-PLACE decimation in frequency over the mixed-radix index space n = sum_d n_d M_d (M_d = prod_{q > d} R_q): pass p
//     transforms dimension p (R_p points, stride STR_p) and multiplies output k_p by W_N^{k_p m_p K_p} (m_p = the index within
//     the remaining dimensions, K_p = prod_{q < p} R_q).  A butterfly reads and writes the SAME R_p positions, so a pass needs
//     one workgroup barrier, not two, and no second image.  The spectrum ends up digit-reversed in LDS -- which costs nothing:
//     the last pass enumerates its butterflies in natural order of (k_0, k_1, ...), so that register (b, t) IS spectrum index
//     b + t N / R_last: maxima are taken from the registers in ascending index order and the optional planes leave as
//     contiguous runs, straight from the registers.
//   * Layout: position = sum_d idx_d STR_d with per-dimension pads and per-pass lane orders that the host picked by
//     SIMULATING the bank rules of ds_read_b64 / ds_write_b64 (MI355X_MICROARCH.md, LDS) over every access of a row group
//     (caf_jit.hip, pdj_layout): conflicts are designed out per length instead of padded against one radix.
//
// Macros (all integers): PDJ_N, PDJ_NP (passes, 2 .. 5), PDJ_R0 .. PDJ_R4 (1 beyond the last pass), PDJ_S0 .. PDJ_S4 (position
// strides), PDJ_ORD0 .. PDJ_ORD4 (brace lists: for pass p the other dimensions, fastest lane digit first, -1 padded), PDJ_TPR
// (threads per row), PDJ_RPW (rows per workgroup), PDJ_WG (workgroup size), PDJ_IMG (row image, elements), PDJ_XREG (cutout
// held in registers across rows), PDJ_P0_LINEAR (pass 0: position == butterfly index), PDJ_Q (below; 1 when absent).
//
// Cutouts longer than one LDS image (PDJ_Q = Q > 1, cutout length NT = Q N): one decimation-in-frequency step in front of the
// transform, taken one output residue at a time.  With n = j + m N and k = tq + Q k',
//   Z[tq + Q k'] = sum_{j < N} g_tq[j] W_N^{j k'},     g_tq[j] = W_NT^{j tq} sum_{m < Q} u[j + m N] W_Q^{m tq},     u = conj(x y)
// so a row is Q transforms of N points in the same image: pass 0 of residue tq forms g_tq on the fly -- Q loads of the cutout and
// of the window per point instead of one, out of the L2 (a row's window and the cutout are re-read Q times, and consecutive rows
// overlap in all but `step` samples) --, the constants W_Q^{m tq} and the step of the input twiddle are uniform (scalar loads
// from the NT-entry table `twq`), the per-point twiddle W_NT^{j tq} is one table entry per butterfly times powers of that step.
// Nothing goes through HBM between the product and the maximum, which is what the rows path (product rows -> rocFFT -> argmax)
// does for these lengths.
//
// Lengths with a prime factor above 23 (PDJ_BLU = 1: the cutout has PDJ_NX samples, the image PDJ_N = M >= 2 NX - 1): Bluestein.
//   Z[k] = c[k] sum_n (u[n] c[n]) conj(c[k - n]),   c[i] = e^{+j pi i^2 / NX}
// is a circular convolution of length M: forward transform of a = u c (zero beyond NX) -- the passes above --, product with the
// transformed chirp `bhat` (host, float64, 1 / M folded in), transform back.  The way back needs no reordering and no new
// butterflies: the forward flow graph is F = Perm . Pass_{P-1} ... Pass_0 with Pass_p = (twiddles) . (butterflies); F is symmetric,
// so F = Pass_0^T ... Pass_{P-1}^T Perm^T, and Pass_p^T = (butterflies) . (twiddles) on the same positions.  Applied to
// d = conj(A bhat), which already sits where Perm^T puts it, the transposed passes P-1 ... 0 leave conj(convolution) in natural
// order: the last forward pass and the first transposed one share their registers (the last pass has no twiddles), the middle
// ones read, multiply by the SAME twiddles, transform and write back, and the transposed pass 0 ends in registers holding
// samples b + t NB0 in ascending order.  |Z[k]|^2 = |conv[k]|^2 -- the outer chirp only matters to the complex plane.
#ifdef __HIPCC_RTC__
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
#endif
#include "caf_mr_dev.h"
#include "caf_energy.h"

namespace caf {
namespace pdj {

#ifndef PDJ_Q
#define PDJ_Q 1
#endif
constexpr int P = PDJ_NP, N = PDJ_N, TPR = PDJ_TPR, RPW = PDJ_RPW, WG = PDJ_WG, IMG = PDJ_IMG;
#ifndef PDJ_BLU
#define PDJ_BLU 0
#endif
#ifndef PDJ_NX
#define PDJ_NX (PDJ_N * PDJ_Q)
#endif
constexpr int Q = PDJ_Q, NT = PDJ_NX;  // NT: the cutout's length; N: the transform the image holds
constexpr bool BLU = PDJ_BLU != 0;
static_assert((Q == 1 && !BLU) || !PDJ_XREG, "the cutout stays in registers only in the one-image form");
static_assert(BLU ? (Q == 1 && N >= 2 * NT - 1) : NT == N * Q, "lengths");
// (as constexpr functions over local tables: namespace-scope arrays would be host variables to the device pass)
constexpr int cRAD(int d) {
    constexpr int T[5] = {PDJ_R0, PDJ_R1, PDJ_R2, PDJ_R3, PDJ_R4};
    return T[d];
}
constexpr int cSTR(int d) {
    constexpr int T[5] = {PDJ_S0, PDJ_S1, PDJ_S2, PDJ_S3, PDJ_S4};
    return T[d];
}
constexpr int cORD(int p, int i) {
    constexpr int T[5][4] = {PDJ_ORD0, PDJ_ORD1, PDJ_ORD2, PDJ_ORD3, PDJ_ORD4};
    return T[p][i];
}
constexpr int cM(int d) {
    int m = 1;
    for (int q = d + 1; q < P; ++q) m *= cRAD(q);
    return m;
}
constexpr int cK(int d) {
    int k = 1;
    for (int q = 0; q < d; ++q) k *= cRAD(q);
    return k;
}
constexpr int cNB(int p) { return N / cRAD(p); }                        // butterflies of pass p
constexpr int cCNT(int p) { return (cNB(p) + TPR - 1) / TPR; }          // ... per thread
constexpr bool cFULL(int p) { return cCNT(p) * TPR == cNB(p); }        // every thread's every butterfly exists

// LDS accesses of the transform: volatile, so that they stay single ds_read_b64 / ds_write_b64 -- merged into ds_read2_b64 a
// pair costs 8 LDS cycles against 2 + 2 (MI355X_MICROARCH.md, LDS), and the layout was chosen for the b64 bank rules
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
__device__ __forceinline__ float2 lds_ld(const float2* p) {
    const unsigned long long u = *(const volatile lds_u64*)p;
    float2 r;
    __builtin_memcpy(&r, &u, 8);
    return r;
}
__device__ __forceinline__ void lds_st(float2* p, float2 v) {
    unsigned long long u;
    __builtin_memcpy(&u, &v, 8);
    *(volatile lds_u64*)p = u;
}
// element `elem` of a complex64 array with a wave-uniform base: scalar base + 32-bit byte offset, no 64-bit address per load
#define PDJ_AS1 __attribute__((address_space(1)))
__device__ __forceinline__ float2 gld(const float2* base, uint32_t elem) {
    const unsigned long long u = *reinterpret_cast<const PDJ_AS1 unsigned long long*>((const PDJ_AS1 char*)base + (elem << 3));
    float2 r;
    __builtin_memcpy(&r, &u, 8);
    return r;
}

// Rows that live inside one wave (TPR divides 64) need no workgroup barrier between the passes: a wave's LDS instructions are
// served in the order they are issued, the accesses below are volatile (the compiler keeps that order), and no other wave
// touches the row's image, key slot or factor.
constexpr bool WAVE_ROWS = (64 % TPR) == 0;
__device__ __forceinline__ void row_sync() {
    if constexpr (WAVE_ROWS)
        __builtin_amdgcn_wave_barrier();
    else
        __syncthreads();
}

// (value, first index) maximum over the aligned group of G = gcd(TPR, 64) lanes a thread sits in (G a power of two <= 16 uses
// DPP only: quad permutes, row_half_mirror, row_mirror; such a group never straddles two rows or two waves).  One lane per
// group then offers the result to the row's slot -- with one LDS atomic per THREAD the 64 same-address operations of a wave
// were served one after the other and showed as two thirds of the kernel's bank-conflict cycles.
constexpr int cGCD(int a, int b) { return b == 0 ? a : cGCD(b, a % b); }
constexpr int GRP = cGCD(TPR, 64) > 16 ? 16 : cGCD(TPR, 64);
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ void group_max(uint32_t& vbits, uint32_t& nidx) {  // nidx = ~index: the larger, the earlier
    auto step = [&](uint32_t ov, uint32_t oi) {
        const bool take = ov > vbits || (ov == vbits && oi > nidx);
        vbits = take ? ov : vbits;
        nidx = take ? oi : nidx;
    };
    if constexpr (GRP >= 2) step(dpp<0xB1>(vbits), dpp<0xB1>(nidx));    // quad_perm [1,0,3,2]
    if constexpr (GRP >= 4) step(dpp<0x4E>(vbits), dpp<0x4E>(nidx));    // quad_perm [2,3,0,1]
    if constexpr (GRP >= 8) step(dpp<0x141>(vbits), dpp<0x141>(nidx));  // row_half_mirror
    if constexpr (GRP >= 16) step(dpp<0x140>(vbits), dpp<0x140>(nidx));  // row_mirror
}

// butterfly b of pass p -> its base position in the row image and m_p (index within the dimensions after p)
template <int p>
__device__ __forceinline__ void decode(int b, int& base, int& m) {
    if (p == 0 && PDJ_P0_LINEAR) {
        base = b, m = b;
        return;
    }
    int r = b;
    base = 0, m = 0;
#pragma unroll
    for (int i = 0; i < P - 1; ++i) {
        const int d = cORD(p, i);
        const int dig = (i == P - 2) ? r : r % cRAD(d);
        r /= cRAD(d);
        base += dig * cSTR(d);
        if (d > p) m += dig * cM(d);
    }
}

// twiddles of pass p on the butterfly's outputs 1 .. R - 1: W_N^{k m K_p} = w1^k, w1 from the table, powers by recurrence
template <int p>
__device__ __forceinline__ void twiddle(float2* v, float2 w1, int row_it) {
    constexpr int R = cRAD(p);
    // (opaque: w1 does not change from row to row, and its R - 2 powers -- of every butterfly of every pass -- would be hoisted out
    //  of the row loop and spilled.  Tied to the row counter instead of `volatile`: a volatile asm keeps its place among the
    //  volatile LDS accesses, i.e. right behind the reads, and made the wave wait for the twiddle load BEFORE its butterfly)
    asm("" : "+v"(w1.x), "+v"(w1.y) : "s"(row_it));
    float2 pw = w1;
    v[1] = cmul(v[1], pw);
#pragma unroll
    for (int t = 2; t < R; ++t) {
        pw = cmul(pw, w1);
        v[t] = cmul(v[t], pw);
    }
}

// a middle pass (0 < p < P - 1): image -> registers -> butterfly, twiddles -> the same positions
template <int p>
__device__ __forceinline__ void middle_pass(float2* __restrict__ buf, const float2* __restrict__ tw, int l, bool active, int row_it) {
    constexpr int R = cRAD(p), CNT = cCNT(p), NB = cNB(p);
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        const int b = l + c * TPR;
        const bool ok = active && (cFULL(p) || b < NB);
        if (ok) {
            int base, m;
            decode<p>(b, base, m);
            const float2 w1 = gld(tw, (uint32_t)(m * cK(p)));
            float2 v[R];
#pragma unroll
            for (int t = 0; t < R; ++t) v[t] = lds_ld(&buf[base + t * cSTR(p)]);
            mr_idft<R>(v);
            twiddle<p>(v, w1, row_it);
#pragma unroll
            for (int t = 0; t < R; ++t) lds_st(&buf[base + t * cSTR(p)], v[t]);
        }
    }
    row_sync();
}

// the transposed middle pass (Bluestein's way back): image -> registers -> the same twiddles, butterfly -> the same positions
template <int p>
__device__ __forceinline__ void middle_pass_t(float2* __restrict__ buf, const float2* __restrict__ tw, int l, bool active, int row_it) {
    constexpr int R = cRAD(p), CNT = cCNT(p), NB = cNB(p);
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        const int b = l + c * TPR;
        const bool ok = active && (cFULL(p) || b < NB);
        if (ok) {
            int base, m;
            decode<p>(b, base, m);
            const float2 w1 = gld(tw, (uint32_t)(m * cK(p)));
            float2 v[R];
#pragma unroll
            for (int t = 0; t < R; ++t) v[t] = lds_ld(&buf[base + t * cSTR(p)]);
            twiddle<p>(v, w1, row_it);
            mr_idft<R>(v);
#pragma unroll
            for (int t = 0; t < R; ++t) lds_st(&buf[base + t * cSTR(p)], v[t]);
        }
    }
    row_sync();
}

}  // namespace pdj
}  // namespace caf

extern "C" __global__ __launch_bounds__(PDJ_WG, 4) void k_pdj(const float2* __restrict__ x, const float2* __restrict__ y, int64_t ylen,
                                                              const float2* __restrict__ tw, const float2* __restrict__ twq,
                                                              const float2* __restrict__ bhat, const double* __restrict__ prefix,
                                                              const double* __restrict__ xnorm, int64_t start, int64_t step,
                                                              int64_t num, int32_t rows_per_wg, int32_t zero_oor,
                                                              float* __restrict__ qf2, uint32_t* __restrict__ fidx,
                                                              float* __restrict__ plane, float2* __restrict__ cplane) {
    using namespace caf;
    using namespace caf::pdj;
    __shared__ __attribute__((aligned(16))) float2 s_buf[RPW * IMG];
    __shared__ unsigned long long s_key[2 * RPW];
    __shared__ float s_inv[RPW];
    constexpr int R0 = cRAD(0), CNT0 = cCNT(0), NB0 = cNB(0);
    constexpr int RL = cRAD(P - 1), CNTL = cCNT(P - 1), NBL = cNB(P - 1);
    const int tid = threadIdx.x;
    const bool active = tid < RPW * TPR;  // (the last threads of the workgroup only keep the barriers company)
    const int rl = active ? tid / TPR : 0, l_fixed = active ? tid - rl * TPR : 0;
    float2* buf = s_buf + rl * IMG;
    const double xn = *xnorm;
    if (tid < 2 * RPW) s_key[tid] = 0ull;
#if PDJ_XREG
    // the cutout stays in registers across the workgroup's rows
    float2 xr[CNT0][R0];
#pragma unroll
    for (int c = 0; c < CNT0; ++c) {
        const int b = min(l_fixed + c * TPR, NB0 - 1);
#pragma unroll
        for (int t = 0; t < R0; ++t) xr[c][t] = x[b + t * NB0];
    }
#endif
    __syncthreads();
    const int64_t row0 = (int64_t)blockIdx.x * rows_per_wg * RPW;
    // The window energy of a row comes from two entries of the float64 prefix, fetched by ONE thread per row -- a row AHEAD: the
    // loads of row it + 1 go out at the end of row it and are consumed behind the first pass of row it + 1, so no wave ever waits
    // for them (fetched where they are used, the row's other waves stood at the barrier for two dependent memory round trips).
    const bool norm_lane = active && l_fixed == 0;
    double pa_n = 0.0, pb_n = 0.0;
    auto energy_bounds = [&](int64_t s_, int64_t& a_, int64_t& b_) {
        a_ = s_ < 0 ? 0 : (s_ > ylen ? ylen : s_);
        b_ = s_ + NT;
        b_ = b_ < 0 ? 0 : (b_ > ylen ? ylen : b_);
    };
    if (norm_lane) {
        int64_t a_, b_;
        energy_bounds(start + (row0 + rl) * step, a_, b_);
        pa_n = prefix[a_], pb_n = prefix[b_];
    }
    for (int it = 0; it < rows_per_wg; ++it) {
        const int64_t row = row0 + (int64_t)it * RPW + rl;
        const bool live = active && row < num;
        const int64_t s = start + row * step;
        const bool oor = (s < 0) || (s + NT > ylen);
        const bool zero = !live || (oor && zero_oor);
        // (an opaque copy of the row-local thread index: everything derived from it -- sixteen 64-bit cutout addresses, plane
        //  indices, image positions -- is the same for every row, gets hoisted out of the row loop and is spilled there)
        int l = l_fixed;
        asm volatile("" : "+v"(l));
        const bool norm_thread = norm_lane && !zero;
        const double pa = pa_n, pb = pb_n;
        float bv = -1.f;  // the row's maximum as this thread sees it (over its bins of every residue)
        uint32_t bi = 0;
#pragma unroll 1
        for (int tq = 0; tq < Q; ++tq) {
        const int pass_it = it * Q + tq;
        if constexpr (Q > 1) asm volatile("" : "+v"(l));  // (as above, per residue: Q x R0 x CNT0 load offsets would be hoisted and spilled)
        // ---- pass 0: global loads, product with the cutout, butterfly, twiddles, into the image
        {
            const bool inside = !zero && !oor;
            const float2* yrow = y + s;
#pragma unroll
            for (int c = 0; c < CNT0; ++c) {
                const int b = l + c * TPR;
                const bool ok = active && (cFULL(0) || b < NB0);
                const int bb = cFULL(0) ? b : min(b, NB0 - 1);
                float2 v[R0];
                if constexpr (BLU) {
                    // a[n] = conj(x y) c[n] for n < NT, zero beyond (clamped loads, the zero selected afterwards)
                    float2 xa[R0], yq[R0];
#pragma unroll
                    for (int t = 0; t < R0; ++t) xa[t] = gld(x, (uint32_t)min(bb + t * NB0, NT - 1));
                    if (inside) {
#pragma unroll
                        for (int t = 0; t < R0; ++t) yq[t] = gld(yrow, (uint32_t)min(bb + t * NB0, NT - 1));
                    } else {
#pragma unroll
                        for (int t = 0; t < R0; ++t) {
                            const int64_t g = s + min(bb + t * NB0, NT - 1);
                            yq[t] = (!zero && g >= 0 && g < ylen) ? y[g] : make_float2(0.f, 0.f);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < R0; ++t) v[t] = make_float2(xa[t].x * yq[t].x - xa[t].y * yq[t].y, -(xa[t].x * yq[t].y + xa[t].y * yq[t].x));
                    // (the chirp in a batch of its own: three operands of R0 points in flight at once spill)
#pragma unroll
                    for (int t = 0; t < R0; ++t) xa[t] = gld(twq, (uint32_t)min(bb + t * NB0, NT - 1));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < R0; ++t) v[t] = (bb + t * NB0 < NT) ? cmul(v[t], xa[t]) : make_float2(0.f, 0.f);
                } else if constexpr (Q == 1) {
                    // every load of the butterfly goes out before the first product is formed (left to itself the scheduler pairs
                    // each cutout / window load with its product and waits for memory sixteen times per butterfly)
                    float2 xa[R0], yq[R0];
#pragma unroll
                    for (int t = 0; t < R0; ++t) {
#if PDJ_XREG
                        xa[t] = xr[c][t];
#else
                        xa[t] = gld(x, (uint32_t)(bb + t * NB0));
#endif
                    }
                    if (inside) {
#pragma unroll
                        for (int t = 0; t < R0; ++t) yq[t] = yrow[bb + t * NB0];
                    } else {
#pragma unroll
                        for (int t = 0; t < R0; ++t) {
                            const int64_t g = s + bb + t * NB0;
                            yq[t] = (!zero && g >= 0 && g < ylen) ? y[g] : make_float2(0.f, 0.f);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < R0; ++t)  // conj(x y): forward = conj(IDFT(conj .))
                        v[t] = make_float2(xa[t].x * yq[t].x - xa[t].y * yq[t].y, -(xa[t].x * yq[t].y + xa[t].y * yq[t].x));
                } else {
                    // g_tq of the butterfly's R0 points: Q batches of loads, each folded in with its uniform constant W_Q^{m tq}
#pragma unroll
                    for (int t = 0; t < R0; ++t) v[t] = make_float2(0.f, 0.f);
#pragma unroll 1
                    for (int m = 0; m < Q; ++m) {
                        float2 xa[R0], yq[R0];
#pragma unroll
                        for (int t = 0; t < R0; ++t) xa[t] = gld(x, (uint32_t)(bb + t * NB0 + m * N));
                        if (inside) {
#pragma unroll
                            for (int t = 0; t < R0; ++t) yq[t] = gld(yrow, (uint32_t)(bb + t * NB0 + m * N));
                        } else {
#pragma unroll
                            for (int t = 0; t < R0; ++t) {
                                const int64_t g = s + bb + t * NB0 + m * N;
                                yq[t] = (!zero && g >= 0 && g < ylen) ? y[g] : make_float2(0.f, 0.f);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const float2 cm = twq[((m * tq) % Q) * N];  // (uniform: a scalar load)
#pragma unroll
                        for (int t = 0; t < R0; ++t) {
                            const float2 u = make_float2(xa[t].x * yq[t].x - xa[t].y * yq[t].y, -(xa[t].x * yq[t].y + xa[t].y * yq[t].x));
                            v[t].x = __builtin_fmaf(u.x, cm.x, __builtin_fmaf(-u.y, cm.y, v[t].x));
                            v[t].y = __builtin_fmaf(u.x, cm.y, __builtin_fmaf(u.y, cm.x, v[t].y));
                        }
                    }
                    if (tq) {  // W_NT^{(bb + t NB0) tq}: one entry per butterfly, times powers of the uniform step W_NT^{NB0 tq}
                        float2 pw = gld(twq, (uint32_t)(bb * tq));
                        const float2 ws = twq[NB0 * tq];
                        v[0] = cmul(v[0], pw);
#pragma unroll
                        for (int t = 1; t < R0; ++t) {
                            pw = cmul(pw, ws);
                            v[t] = cmul(v[t], pw);
                        }
                    }
                }
                mr_idft<R0>(v);
                int base, m;
                decode<0>(bb, base, m);
                twiddle<0>(v, gld(tw, (uint32_t)m), pass_it);  // (K_0 = 1)
                if (ok) {
#pragma unroll
                    for (int t = 0; t < R0; ++t) lds_st(&buf[base + t * cSTR(0)], v[t]);
                }
            }
            if (active && l_fixed == 0 && tq == 0) {
                // rsq(E ||x||^2) + one Newton step (2^-45 or better before the rounding to float32).  E = 0: rsq = inf, 0 * inf = NaN row.
                float inv0 = 0.f;
                if (norm_thread) {
                    double e = pb - pa;
                    if (!(e > CAF_ENERGY_RESOLVED * pb)) {  // (not resolved by the prefix: summed again, caf_energy.h)
                        int64_t ea, eb;
                        energy_bounds(s, ea, eb);
                        e = window_energy_direct(y, energy_chunks(prefix, ylen), ea, eb);
                    }
                    const double en = e * (xn * xn);
                    const double y0 = __builtin_amdgcn_rsq(en);
                    inv0 = (float)__builtin_fma(__builtin_fma(-(en * y0), 0.5 * y0, 0.5), y0, y0);
                }
                *(volatile float*)&s_inv[rl] = inv0;
            }
            row_sync();
        }
        if constexpr (P > 2) middle_pass<1>(buf, tw, l, active, pass_it);
        if constexpr (P > 3) middle_pass<2>(buf, tw, l, active, pass_it);
        if constexpr (P > 4) middle_pass<3>(buf, tw, l, active, pass_it);
        if constexpr (BLU) {
            // ---- last forward pass, product with the transformed chirp, conjugate, first transposed pass: in registers
#pragma unroll
            for (int c = 0; c < CNTL; ++c) {
                const int b = l + c * TPR;
                const bool ok = active && (cFULL(P - 1) || b < NBL);
                if (ok) {
                    int base, m;
                    decode<P - 1>(b, base, m);
                    float2 v[RL];
#pragma unroll
                    for (int t = 0; t < RL; ++t) v[t] = lds_ld(&buf[base + t * cSTR(P - 1)]);
                    mr_idft<RL>(v);
#pragma unroll
                    for (int t = 0; t < RL; ++t) {
                        const float2 z = cmul(v[t], gld(bhat, (uint32_t)(b + t * NBL)));
                        v[t] = make_float2(z.x, -z.y);
                    }
                    mr_idft<RL>(v);
#pragma unroll
                    for (int t = 0; t < RL; ++t) lds_st(&buf[base + t * cSTR(P - 1)], v[t]);
                }
            }
            row_sync();
            if constexpr (P > 4) middle_pass_t<3>(buf, tw, l, active, pass_it);
            if constexpr (P > 3) middle_pass_t<2>(buf, tw, l, active, pass_it);
            if constexpr (P > 2) middle_pass_t<1>(buf, tw, l, active, pass_it);
            // ---- transposed pass 0: register (b, t) is conj(convolution)[b + t NB0]; the first NT samples are the row's bins
            float2 v[CNT0][R0];
            bool okb[CNT0];
#pragma unroll
            for (int c = 0; c < CNT0; ++c) {
                const int b = l + c * TPR;
                okb[c] = active && (cFULL(0) || b < NB0);
                const int bb = cFULL(0) ? b : min(b, NB0 - 1);
                int base, m;
                decode<0>(bb, base, m);
#pragma unroll
                for (int t = 0; t < R0; ++t) v[c][t] = lds_ld(&buf[base + t * cSTR(0)]);
                twiddle<0>(v[c], gld(tw, (uint32_t)m), pass_it);
                mr_idft<R0>(v[c]);
            }
            const float inv = *(volatile float*)&s_inv[rl];
#pragma unroll
            for (int t = 0; t < R0; ++t)
#pragma unroll
                for (int c = 0; c < CNT0; ++c) {
                    const int k = l + c * TPR + t * NB0;
                    okb[c] = okb[c] && k < NT;  // (k grows with t: once beyond the cutout's bins, always)
                    const float zr = v[c][t].x * inv, zi = v[c][t].y * inv;
                    v[c][t] = make_float2(zr, zi);
                    const float val = __builtin_fmaf(zr, zr, zi * zi);
                    const bool up = okb[c] && val > bv;  // (ascending index: the strict comparison keeps the first maximum)
                    bv = up ? val : bv;
                    bi = up ? (uint32_t)k : bi;
                }
            // (okb now says: butterfly c exists and its LAST bin is inside -- the stores test every bin)
            if (plane && live) {
                float* prow = plane + row * NT;
#pragma unroll
                for (int t = 0; t < R0; ++t)
#pragma unroll
                    for (int c = 0; c < CNT0; ++c) {
                        const int k = l + c * TPR + t * NB0;
                        if ((cFULL(0) || l + c * TPR < NB0) && k < NT) prow[k] = __builtin_fmaf(v[c][t].x, v[c][t].x, v[c][t].y * v[c][t].y);
                    }
            }
            if (cplane && live) {
                float2* crow = cplane + row * NT;
#pragma unroll
                for (int t = 0; t < R0; ++t)
#pragma unroll
                    for (int c = 0; c < CNT0; ++c) {
                        const int k = l + c * TPR + t * NB0;
                        if ((cFULL(0) || l + c * TPR < NB0) && k < NT) {
                            const float2 ck = gld(twq, (uint32_t)k);  // conj(Z[k]) = conj(c[k]) conj(conv[k])
                            crow[k] = make_float2(v[c][t].x * ck.x + v[c][t].y * ck.y, v[c][t].y * ck.x - v[c][t].x * ck.y);
                        }
                    }
            }
        } else
        // ---- last pass: butterflies in natural order of (k_0, k_1, ...): register (b, t) is spectrum index b + t NBL of the
        // image's transform, i.e. bin tq + Q (b + t NBL) of the row
        {
            float2 v[CNTL][RL];
            bool okc[CNTL];
#pragma unroll
            for (int c = 0; c < CNTL; ++c) {
                const int b = l + c * TPR;
                okc[c] = active && (cFULL(P - 1) || b < NBL);
                int base, m;
                decode<P - 1>(cFULL(P - 1) ? b : min(b, NBL - 1), base, m);
#pragma unroll
                for (int t = 0; t < RL; ++t) v[c][t] = lds_ld(&buf[base + t * cSTR(P - 1)]);
                mr_idft<RL>(v[c]);
            }
            const float inv = *(volatile float*)&s_inv[rl];  // (written before the barrier of pass 0)
            float lv = -1.f;
            uint32_t li = 0;
#pragma unroll
            for (int t = 0; t < RL; ++t)
#pragma unroll
                for (int c = 0; c < CNTL; ++c) {
                    const float zr = v[c][t].x * inv, zi = v[c][t].y * inv;
                    v[c][t] = make_float2(zr, zi);
                    const float val = __builtin_fmaf(zr, zr, zi * zi);
                    const bool up = okc[c] && val > lv;  // (ascending index: the strict comparison keeps the first maximum)
                    lv = up ? val : lv;
                    li = up ? (uint32_t)(l + c * TPR + t * NBL) : li;
                }
            if constexpr (Q == 1) {
                bv = lv, bi = li;
            } else {  // (residues are not visited in ascending bin order: equal values keep the lower bin)
                li = (uint32_t)tq + (uint32_t)Q * li;
                const bool take = lv > bv || (lv == bv && li < bi);
                bv = take ? lv : bv;
                bi = take ? li : bi;
            }
            // the optional planes leave from the registers as contiguous runs (Q > 1: every Q-th bin), behind ONE uniform branch each
            if (plane && live) {
                float* prow = plane + row * NT + tq;
#pragma unroll
                for (int t = 0; t < RL; ++t)
#pragma unroll
                    for (int c = 0; c < CNTL; ++c)
                        if (cFULL(P - 1) || okc[c]) prow[Q * (l + c * TPR + t * NBL)] = __builtin_fmaf(v[c][t].x, v[c][t].x, v[c][t].y * v[c][t].y);
            }
            if (cplane && live) {
                float2* crow = cplane + row * NT + tq;
#pragma unroll
                for (int t = 0; t < RL; ++t)
#pragma unroll
                    for (int c = 0; c < CNTL; ++c)
                        if (cFULL(P - 1) || okc[c]) crow[Q * (l + c * TPR + t * NBL)] = make_float2(v[c][t].x, -v[c][t].y);
            }
        }
        if (Q > 1 && tq + 1 < Q) row_sync();  // the image is read to the end before the next residue's first pass overwrites it
        }  // residues
        if (qf2 || fidx) {
            // one 64-bit maximum of (value bits, ~index) per row: values are >= +0, whose bit patterns order like the numbers; a
            // thread that saw only NaNs offers nothing, so an all-NaN row -- a zero-energy window -- keeps key 0 -> (NaN, 0)
            unsigned long long* slot = &s_key[(it & 1) * RPW + rl];
            // (a thread that saw nothing -- inactive, or only NaNs -- offers (0, 0): below every real candidate)
            uint32_t vb = (live && !(bv < 0.f)) ? __float_as_uint(bv) : 0u, ni = (live && !(bv < 0.f)) ? ~bi : 0u;
            group_max(vb, ni);
            const unsigned long long key = ((unsigned long long)vb << 32) | ni;
            if (live && key && (tid & (GRP - 1)) == 0) atomicMax(slot, key);
            row_sync();
            if (live && l_fixed == 0) {
                const unsigned long long kk = *(volatile unsigned long long*)slot;
                if (qf2) qf2[row] = kk ? __uint_as_float((uint32_t)(kk >> 32)) : __builtin_nanf("");
                if (fidx) fidx[row] = kk ? ~(uint32_t)kk : 0u;
                *(volatile unsigned long long*)slot = 0ull;  // (next used two rows from now, behind the barriers of the row in between)
            }
        } else {
            row_sync();  // the image is read to the end before the next row's first pass overwrites it
        }
        if (norm_lane && it + 1 < rows_per_wg) {  // (rows past `num` clamp to valid entries: their factor is never used)
            int64_t a_, b_;
            energy_bounds(start + (row + RPW) * step, a_, b_);
            pa_n = prefix[a_], pb_n = prefix[b_];
        }
    }
}
