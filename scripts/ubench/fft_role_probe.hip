// Removal profiling of the FFT role (fused_item<1024> of pydsproutines_amd/csrc/caf_fused.hip) at the C2 shape:
// the hypothesis loop is restated here with compile-time switches that each REMOVE one cost (results are wrong by
// construction; only the time matters), so that the difference to the full loop prices that cost:
//   1  no |y|^2 global stores (values folded into one register, stored once)
//   2  no template-spectrum row loads per hypothesis (the first row is reused)
//   4  workgroup barriers -> wave barriers
//   8  no twiddle-table LDS reads in passes 2/3 (a register constant instead)
//   16 no pass-1 twiddle recurrence (one constant)
//   32 no LDS data traffic in passes 1-3 (registers kept; arithmetic unchanged)
//   64 no pass-4 LDS reads
//  128 no idft16 arithmetic (passes 1-3 butterflies skipped)
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -I../../include -I../../pydsproutines_amd/csrc
#include "caf_fused.hip"
// (round 3: the product kernel moved to a planar LDS image; this probe keeps the ROUND-2 structure it was written to
// measure -- complex image, pitches 68 / 1090 -- as a historical harness; its numbers describe that structure only)
namespace caf {
constexpr int F_ROW = 68;
constexpr int F_N1 = 16 * F_ROW + 2;
}

#include <cstdio>
#include <type_traits>
#include <vector>

namespace caf {
__device__ __forceinline__ void gst1(float* base, uint32_t byteoff, float v) {
    *reinterpret_cast<CAF_AS1 float*>((CAF_AS1 char*)base + byteoff) = v;
}
// flag 65536: wave stagger.  g_st[0] / g_st[1] = units of 64 cycles that the second wave group sleeps after B2 / B3,
// g_st[2] = 1: the first group runs at priority 3, g_st[3] = how the groups are formed (0: waves 8..15, 1: waves with bit 2)
__device__ int g_st[8];
// flag 262144: in-kernel stamps (s_memtime, shader cycles) of one transform per workgroup: 12 points per wave, read at the
// end of the iteration (one lgkmcnt wait there), written by lane 0 of every wave of workgroups 0..63
__device__ unsigned long long* g_stamps;
// flag 524288: progress-based issue priority (s_setprio): a wave that is ahead inside a barrier-free stretch yields to
// the ones behind it.  g_st[5] selects the variant (see PRIO below).
#define PRIO(point)                                                                              \
    {                                                                                            \
        constexpr int pv = (FLAGS >> 19) & 15; /* compile-time variant: flags 524288 * variant */ \
        if (pv == 1) {                                                                           \
            if (point == 0) __builtin_amdgcn_s_setprio(3);                                       \
            if (point == 1) __builtin_amdgcn_s_setprio(2);                                       \
            if (point == 2) __builtin_amdgcn_s_setprio(1);                                       \
            if (point == 3) __builtin_amdgcn_s_setprio(0);                                       \
        } else if (pv == 2) { /* only the tail: low priority once pass 3 is computed */          \
            if (point == 0) __builtin_amdgcn_s_setprio(2);                                       \
            if (point == 3) __builtin_amdgcn_s_setprio(0);                                       \
        } else if (pv == 3) { /* reversed static order: youngest waves first */                  \
            if (point == 0) {                                                                    \
                if ((wave_id >> 2) == 1) __builtin_amdgcn_s_setprio(1);                          \
                if ((wave_id >> 2) == 2) __builtin_amdgcn_s_setprio(2);                          \
                if ((wave_id >> 2) == 3) __builtin_amdgcn_s_setprio(3);                          \
            }                                                                                    \
        } else if (pv == 4) { /* as 1, and the same ladder over pass 4 / pass 1 */               \
            if (point == 0 || point == 4) __builtin_amdgcn_s_setprio(3);                         \
            if (point == 1 || point == 5) __builtin_amdgcn_s_setprio(2);                         \
            if (point == 2) __builtin_amdgcn_s_setprio(1);                                       \
            if (point == 3 || point == 6) __builtin_amdgcn_s_setprio(0);                         \
        } else if (pv == 6) { /* static order, oldest waves first */                             \
            if (point == 0) {                                                                    \
                if ((wave_id >> 2) == 0) __builtin_amdgcn_s_setprio(3);                          \
                if ((wave_id >> 2) == 1) __builtin_amdgcn_s_setprio(2);                          \
                if ((wave_id >> 2) == 2) __builtin_amdgcn_s_setprio(1);                          \
            }                                                                                    \
        } else if (pv == 5) { /* LDS-write phases at high priority (they feed the barrier), arithmetic low */ \
            if (point == 1 || point == 3 || point == 6) __builtin_amdgcn_s_setprio(3);           \
            if (point == 0 || point == 2 || point == 4) __builtin_amdgcn_s_setprio(0);           \
        }                                                                                        \
    }
#define STAMP(i)                                        \
    if (FLAGS & 262144) {                               \
        __builtin_amdgcn_sched_barrier(0);              \
        tstamp[i] = __builtin_amdgcn_s_memtime();       \
        __builtin_amdgcn_sched_barrier(0);              \
    }

template <int FLAGS>
__global__ __launch_bounds__(1024) void k_probe(const float2* __restrict__ xb, const float2* __restrict__ hc,
                                                const int32_t* __restrict__ shifts, const float2* __restrict__ tw1,
                                                const float2* __restrict__ tw23, int32_t nfreq, int32_t nhyp,
                                                int32_t hyp_per_wg, int32_t nblk, int32_t tiles_per_blk,
                                                float* __restrict__ vt) {
    __shared__ __attribute__((aligned(16))) float2 s_d[F_LDS_DATA];
    __shared__ float2 s_tw2[16 * 64];
    __shared__ float2 s_tw3[16 * 4];
    const int tid = threadIdx.x;
    const int ngroups = (nhyp + hyp_per_wg - 1) / hyp_per_wg;
    const int lin = blockIdx.x;
    const int q = lin >> 3;
    const int blk = (q / ngroups) * 8 + (lin & 7);
    const int grp = q - (q / ngroups) * ngroups;
    if (blk >= nblk) return;
    const int h0 = grp * hyp_per_wg;
    const int h1 = min(h0 + hyp_per_wg, nhyp);
    s_tw2[tid] = tw23[tid];
    if (tid < 64) s_tw3[tid] = tw23[1024 + tid];

    auto bar = [&](int which = 0) {
        if ((FLAGS & 4) || ((FLAGS & 256) && which == 3) || ((FLAGS & 512) && which == 1))
            __builtin_amdgcn_wave_barrier();
        else
            __syncthreads();
    };
    const float2 w = ld2(tw1, (uint32_t)(1024 + tid));
    const float2* xp = xb + (int64_t)blk * FB;
    float* vt_blk = vt + (int64_t)blk * tiles_per_blk * nhyp * 64;
    const float2* hrow_cur;
    int sh_cur;
    auto row_of = [&](int h) {
        const int t = h / nfreq;
        sh_cur = *((const CAF_AS1 int32_t*)shifts + (h - t * nfreq));
        hrow_cur = hc + (int64_t)t * FB;
    };
    float2 pr[16], xr[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) xr[a] = ld2(xp, (uint32_t)(1024 * a + tid));
    row_of(h0);
#pragma unroll
    for (int a = 0; a < 16; ++a) pr[a] = cmul(xr[a], ld2(hrow_cur, (uint32_t)((1024 * a + tid - sh_cur) & (FB - 1))));
    float acc = 0.f;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int st_a = (FLAGS & 65536) ? g_st[0] : 0, st_b = (FLAGS & 65536) ? g_st[1] : 0;
    const bool grp_b = (FLAGS & 65536) && (g_st[3] ? ((wave_id >> 2) & 1) : (wave_id >> 3));
    if ((FLAGS & 65536) && g_st[2] && !grp_b) __builtin_amdgcn_s_setprio(3);

    __syncthreads();

    for (int h = h0; h < h1; ++h) {
        const bool more = h + 1 < h1;
        int64_t hoff = (int64_t)h * 64;
        asm volatile("" : "+s"(hoff));
        int lz = 0;
        asm volatile("" : "+v"(lz));
        unsigned long long tstamp[12];
        STAMP(0)
        float2 v1[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) v1[a] = pr[a];
        if (!(FLAGS & 128)) idft16(v1);
        {
            float2 p = w;
            asm volatile("" : "+v"(p.x), "+v"(p.y));
            const float2 wj = p;
            v1[1] = cmul(v1[1], p);
#pragma unroll
            for (int n1 = 2; n1 < 16; ++n1) {
                if (!(FLAGS & 16)) p = cmul(p, wj);
                v1[n1] = cmul(v1[n1], p);
            }
        }
        STAMP(1)
        PRIO(6)
        if (!(FLAGS & (131072 | 8388608))) bar(1);
        STAMP(2)
        {
            const int off = (tid >> 6) * F_ROW + (tid & 63);
            if (!(FLAGS & 32)) {
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) s_d[n1 * F_N1 + off] = v1[n1];
            }
        }
        STAMP(3)
        bar(2);
        STAMP(4)
        PRIO(0)
        if ((FLAGS & 65536) && grp_b)
            for (int i = 0; i < st_a; ++i) __builtin_amdgcn_s_sleep(1);
        row_of(more ? h + 1 : h);
        float2 v[16];
        {
            const int base = (tid >> 6) * F_N1 + (tid & 63);
            if (!(FLAGS & 32)) {
#pragma unroll
                for (int b = 0; b < 16; ++b) v[b] = s_d[base + b * F_ROW];
            } else {
#pragma unroll
                for (int b = 0; b < 16; ++b) v[b] = v1[b];
            }
            if (!(FLAGS & 128)) idft16(v);
#pragma unroll
            for (int n2 = 1; n2 < 16; ++n2) v[n2] = cmul(v[n2], (FLAGS & 8) ? w : s_tw2[n2 * 64 + (tid & 63) + lz]);
            STAMP(5)
            PRIO(1)
            if (!(FLAGS & 32)) {
#pragma unroll
                for (int n2 = 0; n2 < 16; ++n2) s_d[base + n2 * F_ROW] = v[n2];
            }
        }
        __builtin_amdgcn_wave_barrier();
        STAMP(6)
        PRIO(2)
        float2 hn[16];
        auto hload = [&](int a) {
            const uint32_t e = (uint32_t)(((1024 * a + tid - sh_cur) & (FB - 1)) + lz);
            if (FLAGS & 16777216) {
                // device-scope load (sc1): skips the CU's L1, which a 128 KB row per transform cannot use anyway
                const uint64_t u = __hip_atomic_load(reinterpret_cast<const CAF_AS1 uint64_t*>((const CAF_AS1 char*)hrow_cur + (e << 3)),
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float2 r;
                __builtin_memcpy(&r, &u, 8);
                hn[a] = r;
            } else if (FLAGS & 16384) {
                const uint64_t u = __builtin_nontemporal_load(reinterpret_cast<const CAF_AS1 uint64_t*>((const CAF_AS1 char*)hrow_cur + (e << 3)));
                float2 r;
                __builtin_memcpy(&r, &u, 8);
                hn[a] = r;
            } else {
                hn[a] = ld2(hrow_cur, e);
            }
        };
        if (FLAGS & 4096) {
            const uint32_t off = (uint32_t)(((tid - sh_cur) & (FB - 1)) + lz);
#pragma unroll
            for (int a = 0; a < 16; ++a) hn[a] = ld2(hrow_cur + 1024 * a, off);
        } else if (FLAGS & 2) {
#pragma unroll
            for (int a = 0; a < 16; ++a) hn[a] = make_float2(w.x + a, w.y);
        } else if (FLAGS & 32768) {
#pragma unroll
            for (int a = 0; a < 8; ++a) hload(a);
        } else if (!(FLAGS & 8192)) {
#pragma unroll
            for (int a = 0; a < 16; ++a) hload(a);
        }
        {
            const int base = (tid >> 6) * F_N1 + ((tid >> 2) & 15) * F_ROW + (tid & 3);
            if (!(FLAGS & 32)) {
#pragma unroll
                for (int c = 0; c < 16; ++c) v[c] = s_d[base + 4 * c];
            }
            // flag 8388608: pass 4 across the four lanes of a quad (DPP) instead of through LDS.  The last LDS reads of a
            // transform are these: B1 can follow them directly, and B3 disappears.
            if (FLAGS & 8388608) bar(1);
            if (!(FLAGS & 128)) idft16(v);
#pragma unroll
            for (int n3 = 1; n3 < 16; ++n3) v[n3] = cmul(v[n3], (FLAGS & 8) ? w : s_tw3[n3 * 4 + (tid & 3) + lz]);
            STAMP(7)
            PRIO(3)
            if (FLAGS & 8388608) {
                const int lane4 = tid & 3;
                const float sg_a = (lane4 & 2) ? -1.f : 1.f, sg_b = (lane4 & 1) ? -1.f : 1.f;
                const bool rot = lane4 == 3;
                const int n1w = __builtin_amdgcn_readfirstlane(tid >> 6);
                const uint32_t lane_off = (uint32_t)(tid & 63) << 2;
#pragma unroll
                for (int n3 = 0; n3 < 16; ++n3) {
                    const float2 m = v[n3];
                    // lanes (0,2) and (1,3): s0 = x0 + x2, s2 = x1 + x3, s1 = x0 - x2, s3 = x1 - x3
                    const float smx = sg_a * m.x, smy = sg_a * m.y;
                    float2 t;
                    asm volatile("v_add_f32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(t.x) : "v"(m.x), "v"(smx));
                    asm volatile("v_add_f32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(t.y) : "v"(m.y), "v"(smy));
                    // lane 3: times j
                    const float2 r = rot ? make_float2(-t.y, t.x) : t;
                    // lanes (0,1) and (2,3): y0 = s0 + s2, y2 = s0 - s2, y1 = s1 + j s3, y3 = s1 - j s3
                    const float srx = sg_b * r.x, sry = sg_b * r.y;
                    float2 y;
                    asm volatile("v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(y.x) : "v"(r.x), "v"(srx));
                    asm volatile("v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(y.y) : "v"(r.y), "v"(sry));
                    const float val = y.x * y.x + y.y * y.y;
                    // probe layout: the 64 lanes of a wave contiguous per (n3, n1, hypothesis); uniform base + lane offset
                    // (tile index folded into the 193 tiles a block owns in this harness: 256 would run past the buffer)
                    const int tile_p = ((n3 * 16 + n1w) * 193) >> 8;
                    float* pu = vt_blk + (int64_t)(tile_p * nhyp) * 64 + hoff;
                    if (FLAGS & 1)
                        acc += val;
                    else
                        gst1(pu, lane_off, val);
                }
#pragma unroll
                for (int a = 0; a < 16; ++a) pr[a] = cmul(xr[a], hn[a]);
                continue;
            }
            if (!(FLAGS & 32)) {
#pragma unroll
                for (int n3 = 0; n3 < 16; ++n3) s_d[base + 4 * n3] = v[n3];
            }
        }
        STAMP(8)
        if (FLAGS & 8192) {
#pragma unroll
            for (int a = 0; a < 16; ++a) hload(a);
        }
        if (FLAGS & 32768) {
#pragma unroll
            for (int a = 8; a < 16; ++a) hload(a);
        }
        bar(3);
        STAMP(9)
        PRIO(4)
        if ((FLAGS & 65536) && grp_b)
            for (int i = 0; i < st_b; ++i) __builtin_amdgcn_s_sleep(1);
        {
            const int n1 = tid & 15, n2 = (tid >> 4) & 15, q4 = tid >> 8;
            const int base = n1 * F_N1 + n2 * F_ROW;
            __builtin_amdgcn_sched_barrier(0);
            // flag 131072: B1 directly after the pass-4 reads (all 16 values in registers), so that the stretch up to B2
            // holds pass-4 arithmetic + stores + next products + pass-1 butterfly + pass-1 writes with no barrier between
            float4 lo_all[4], hi_all[4];
            if (FLAGS & 131072) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n3 = q4 + 4 * i;
                    lo_all[i] = *reinterpret_cast<const float4*>(&s_d[base + 4 * n3 + lz]);
                    hi_all[i] = *reinterpret_cast<const float4*>(&s_d[base + 4 * n3 + 2 + lz]);
                }
                bar(1);
                if ((FLAGS & 65536) && grp_b)
                    for (int i = 0; i < g_st[4]; ++i) __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n3 = q4 + 4 * i;
                int lzi = 0;
                asm volatile("" : "+v"(lzi));
                float4 lo, hi;
                if (FLAGS & 131072) {
                    lo = lo_all[i];
                    hi = hi_all[i];
                } else if (!(FLAGS & 64)) {
                    lo = *reinterpret_cast<const float4*>(&s_d[base + 4 * n3 + lzi]);
                    hi = *reinterpret_cast<const float4*>(&s_d[base + 4 * n3 + 2 + lzi]);
                } else {
                    lo = make_float4(v[4 * i].x, v[4 * i].y, v[4 * i + 1].x, v[4 * i + 1].y);
                    hi = make_float4(v[4 * i + 2].x, v[4 * i + 2].y, v[4 * i + 3].x, v[4 * i + 3].y);
                }
                float2 a0 = make_float2(lo.x, lo.y), a1 = make_float2(lo.z, lo.w);
                float2 a2 = make_float2(hi.x, hi.y), a3 = make_float2(hi.z, hi.w);
                idft4(a0, a1, a2, a3);
                const float2 y[4] = {a0, a1, a2, a3};
#pragma unroll
                for (int n4 = 0; n4 < ((FLAGS & 1024) ? 3 : 4); ++n4) {
                    const int tile_u = 16 * i + 64 * n4;
                    const int tile_t = (n2 >> 2) + 4 * q4;
                    float* pu = vt_blk + (int64_t)tile_u * nhyp * 64 + hoff;
                    const uint32_t voff = ((uint32_t)tile_t * (uint32_t)nhyp * 64u + (uint32_t)(n1 + 16 * (n2 & 3))) << 2;
                    const float val = y[n4].x * y[n4].x + y[n4].y * y[n4].y;
                    if (FLAGS & 1)
                        acc += val;
                    else if ((FLAGS & 1024) || tile_u + tile_t < tiles_per_blk)
                        gst1(pu, voff, val);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            STAMP(10)
            PRIO(5)
#pragma unroll
            for (int a = 0; a < 16; ++a) pr[a] = cmul(xr[a], hn[a]);
            __builtin_amdgcn_sched_barrier(0);
            STAMP(11)
        }
        if ((FLAGS & 262144) && h == h0 + 20 && blockIdx.x < 64 && (tid & 63) == 0) {
            unsigned long long* d = g_stamps + ((size_t)blockIdx.x * 16 + (tid >> 6)) * 12;
#pragma unroll
            for (int i = 0; i < 12; ++i) d[i] = tstamp[i];
        }
    }
    if (FLAGS & 1) vt_blk[tid] = acc;
}


// ---- Variant 3: the same loop with hand-packed complex arithmetic (v_pk_add/mul/fma_f32 on (re, im) register pairs,
// op_sel / neg modifiers instead of moves): packed instructions issue in ~4.2 cycles for two lanes of work where the
// scalar forms take 2.4-2.9 each (pk_rate.hip).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_addj(v2f a, v2f b) {  // a + j b
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_subj(v2f a, v2f b) {  // a - j b
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f b) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
__device__ __forceinline__ v2f pk_cmul_s(v2f a, v2f b) {  // b uniform (SGPR pair)
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "s"(b), "v"(t));
    return r;
}
__device__ __forceinline__ void pk_idft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
    const v2f s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, e13 = a1 - a3;
    a0 = s02 + s13;
    a1 = pk_addj(d02, e13);
    a2 = s02 - s13;
    a3 = pk_subj(d02, e13);
}
__device__ __forceinline__ void pk_idft16(v2f (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) pk_idft4(v[m2], v[4 + m2], v[8 + m2], v[12 + m2]);
    const v2f w1 = {C1, S1}, w2 = {R2, R2}, w3 = {S1, C1}, w6 = {-R2, R2}, w9 = {-C1, -S1}, zero = {0.f, 0.f};
    v[5] = pk_cmul_s(v[5], w1);
    v[9] = pk_cmul_s(v[9], w2);
    v[13] = pk_cmul_s(v[13], w3);
    v[6] = pk_cmul_s(v[6], w2);
    v[10] = pk_addj(zero, v[10]);
    v[14] = pk_cmul_s(v[14], w6);
    v[7] = pk_cmul_s(v[7], w3);
    v[11] = pk_cmul_s(v[11], w6);
    v[15] = pk_cmul_s(v[15], w9);
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) pk_idft4(v[4 * n1 + 0], v[4 * n1 + 1], v[4 * n1 + 2], v[4 * n1 + 3]);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) {
            const v2f t = v[4 * a + b];
            v[4 * a + b] = v[4 * b + a];
            v[4 * b + a] = t;
        }
}
__device__ __forceinline__ v2f ldv(const float2* base, uint32_t elem) {
    const float2 f = ld2(base, elem);
    return (v2f){f.x, f.y};
}

template <int FLAGS>
__global__ __launch_bounds__(1024) void k_probe_pk(const float2* __restrict__ xb, const float2* __restrict__ hc,
                                                   const int32_t* __restrict__ shifts, const float2* __restrict__ tw1,
                                                   const float2* __restrict__ tw23, int32_t nfreq, int32_t nhyp,
                                                   int32_t hyp_per_wg, int32_t nblk, int32_t tiles_per_blk,
                                                   float* __restrict__ vt) {
    __shared__ __attribute__((aligned(16))) v2f s_d[F_LDS_DATA];
    __shared__ v2f s_tw2[16 * 64];
    __shared__ v2f s_tw3[16 * 4];
    const int tid = threadIdx.x;
    const int ngroups = (nhyp + hyp_per_wg - 1) / hyp_per_wg;
    const int lin = blockIdx.x;
    const int q = lin >> 3;
    const int blk = (q / ngroups) * 8 + (lin & 7);
    const int grp = q - (q / ngroups) * ngroups;
    if (blk >= nblk) return;
    const int h0 = grp * hyp_per_wg;
    const int h1 = min(h0 + hyp_per_wg, nhyp);
    s_tw2[tid] = ldv(tw23, tid);
    if (tid < 64) s_tw3[tid] = ldv(tw23, 1024 + tid);
    const bool nosync = FLAGS & 4;
    const v2f w = ldv(tw1, (uint32_t)(1024 + tid));
    const float2* xp = xb + (int64_t)blk * FB;
    float* vt_blk = vt + (int64_t)blk * tiles_per_blk * nhyp * 64;
    const float2* hrow_cur;
    int sh_cur;
    auto row_of = [&](int h) {
        const int t = h / nfreq;
        sh_cur = *((const CAF_AS1 int32_t*)shifts + (h - t * nfreq));
        hrow_cur = hc + (int64_t)t * FB;
    };
    v2f pr[16], xr[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) xr[a] = ldv(xp, (uint32_t)(1024 * a + tid));
    row_of(h0);
#pragma unroll
    for (int a = 0; a < 16; ++a) pr[a] = pk_cmul(xr[a], ldv(hrow_cur, (uint32_t)((1024 * a + tid - sh_cur) & (FB - 1))));
    float acc = 0.f;
    __syncthreads();
    for (int h = h0; h < h1; ++h) {
        const bool more = h + 1 < h1;
        int64_t hoff = (int64_t)h * 64;
        asm volatile("" : "+s"(hoff));
        int lz = 0;
        asm volatile("" : "+v"(lz));
        v2f v1[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) v1[a] = pr[a];
        pk_idft16(v1);
        {
            v2f p = w;
            asm volatile("" : "+v"(p));
            const v2f wj = p;
            v1[1] = pk_cmul(v1[1], p);
#pragma unroll
            for (int n1 = 2; n1 < 16; ++n1) {
                p = pk_cmul(p, wj);
                v1[n1] = pk_cmul(v1[n1], p);
            }
        }
        if (nosync) __builtin_amdgcn_wave_barrier(); else __syncthreads();
        if (!(FLAGS & 32)) {
            const int off = (tid >> 6) * F_ROW + (tid & 63);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) s_d[n1 * F_N1 + off] = v1[n1];
        }
        if (nosync) __builtin_amdgcn_wave_barrier(); else __syncthreads();
        row_of(more ? h + 1 : h);
        v2f v[16];
        {
            const int base = (tid >> 6) * F_N1 + (tid & 63);
            if (!(FLAGS & 32)) {
#pragma unroll
                for (int b = 0; b < 16; ++b) v[b] = s_d[base + b * F_ROW];
            } else {
#pragma unroll
                for (int b = 0; b < 16; ++b) v[b] = v1[b];
            }
            pk_idft16(v);
#pragma unroll
            for (int n2 = 1; n2 < 16; ++n2) v[n2] = pk_cmul(v[n2], (FLAGS & 8) ? w : s_tw2[n2 * 64 + (tid & 63) + lz]);
            if (!(FLAGS & 32)) {
#pragma unroll
                for (int n2 = 0; n2 < 16; ++n2) s_d[base + n2 * F_ROW] = v[n2];
            }
        }
        __builtin_amdgcn_wave_barrier();
        v2f hn[16];
        if (!(FLAGS & 2)) {
#pragma unroll
            for (int a = 0; a < 16; ++a) hn[a] = ldv(hrow_cur, (uint32_t)(((1024 * a + tid - sh_cur) & (FB - 1)) + lz));
        } else {
#pragma unroll
            for (int a = 0; a < 16; ++a) hn[a] = (v2f){w.x + a, w.y};
        }
        {
            const int base = (tid >> 6) * F_N1 + ((tid >> 2) & 15) * F_ROW + (tid & 3);
            if (!(FLAGS & 32)) {
#pragma unroll
                for (int c = 0; c < 16; ++c) v[c] = s_d[base + 4 * c];
            }
            pk_idft16(v);
#pragma unroll
            for (int n3 = 1; n3 < 16; ++n3) v[n3] = pk_cmul(v[n3], (FLAGS & 8) ? w : s_tw3[n3 * 4 + (tid & 3) + lz]);
            if (!(FLAGS & 32)) {
#pragma unroll
                for (int n3 = 0; n3 < 16; ++n3) s_d[base + 4 * n3] = v[n3];
            }
        }
        if (nosync) __builtin_amdgcn_wave_barrier(); else __syncthreads();
        {
            const int n1 = tid & 15, n2 = (tid >> 4) & 15, q4 = tid >> 8;
            const int base = n1 * F_N1 + n2 * F_ROW;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n3 = q4 + 4 * i;
                int lzi = 0;
                asm volatile("" : "+v"(lzi));
                v2f a0, a1, a2, a3;
                if (!(FLAGS & 64)) {
                    const float4 lo = *reinterpret_cast<const float4*>(&s_d[base + 4 * n3 + lzi]);
                    const float4 hi = *reinterpret_cast<const float4*>(&s_d[base + 4 * n3 + 2 + lzi]);
                    a0 = (v2f){lo.x, lo.y}; a1 = (v2f){lo.z, lo.w}; a2 = (v2f){hi.x, hi.y}; a3 = (v2f){hi.z, hi.w};
                } else {
                    a0 = v[4 * i]; a1 = v[4 * i + 1]; a2 = v[4 * i + 2]; a3 = v[4 * i + 3];
                }
                pk_idft4(a0, a1, a2, a3);
                const v2f y[4] = {a0, a1, a2, a3};
#pragma unroll
                for (int n4 = 0; n4 < 4; ++n4) {
                    const int tile_u = 16 * i + 64 * n4;
                    const int tile_t = (n2 >> 2) + 4 * q4;
                    float* pu = vt_blk + (int64_t)tile_u * nhyp * 64 + hoff;
                    const uint32_t voff = ((uint32_t)tile_t * (uint32_t)nhyp * 64u + (uint32_t)(n1 + 16 * (n2 & 3))) << 2;
                    const v2f sq = y[n4] * y[n4];
                    const float val = sq.x + sq.y;
                    if (FLAGS & 1)
                        acc += val;
                    else if (tile_u + tile_t < tiles_per_blk)
                        gst1(pu, voff, val);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int a = 0; a < 16; ++a) pr[a] = pk_cmul(xr[a], hn[a]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (FLAGS & 1) vt_blk[tid] = acc;
}

// Variant 2: the next hypothesis' pass-1 butterfly (X*H product, DFT16, twiddle recurrence: VALU only) is placed
// INSIDE the LDS-bound stretch, between passes 2 and 3 of the current hypothesis, instead of after pass 4.
// ORDER 0: P2, P1c(next), P3.   ORDER 1: P1c(next) split: product+DFT16 after P2, recurrence twiddles after P3.
template <int FLAGS, int ORDER>
__global__ __launch_bounds__(1024) void k_probe2(const float2* __restrict__ xb, const float2* __restrict__ hc,
                                                 const int32_t* __restrict__ shifts, const float2* __restrict__ tw1,
                                                 const float2* __restrict__ tw23, int32_t nfreq, int32_t nhyp,
                                                 int32_t hyp_per_wg, int32_t nblk, int32_t tiles_per_blk,
                                                 float* __restrict__ vt) {
    __shared__ __attribute__((aligned(16))) float2 s_d[F_LDS_DATA];
    __shared__ float2 s_tw2[16 * 64];
    __shared__ float2 s_tw3[16 * 4];
    const int tid = threadIdx.x;
    const int ngroups = (nhyp + hyp_per_wg - 1) / hyp_per_wg;
    const int lin = blockIdx.x;
    const int q = lin >> 3;
    const int blk = (q / ngroups) * 8 + (lin & 7);
    const int grp = q - (q / ngroups) * ngroups;
    if (blk >= nblk) return;
    const int h0 = grp * hyp_per_wg;
    const int h1 = min(h0 + hyp_per_wg, nhyp);
    s_tw2[tid] = tw23[tid];
    if (tid < 64) s_tw3[tid] = tw23[1024 + tid];
    auto bar = [&]() {
        if (FLAGS & 4)
            __builtin_amdgcn_wave_barrier();
        else
            __syncthreads();
    };
    const float2 w = ld2(tw1, (uint32_t)(1024 + tid));
    const float2* xp = xb + (int64_t)blk * FB;
    float* vt_blk = vt + (int64_t)blk * tiles_per_blk * nhyp * 64;
    const float2* hrow_cur;
    int sh_cur;
    auto row_of = [&](int h) {
        const int t = h / nfreq;
        sh_cur = *((const CAF_AS1 int32_t*)shifts + (h - t * nfreq));
        hrow_cur = hc + (int64_t)t * FB;
    };
    float2 xr[16], v1[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) xr[a] = ld2(xp, (uint32_t)(1024 * a + tid));
    auto p1_dft = [&]() {  // v1 holds the template-spectrum row: product, DFT16
#pragma unroll
        for (int a = 0; a < 16; ++a) v1[a] = cmul(xr[a], v1[a]);
        idft16(v1);
    };
    auto p1_tw = [&]() {
        float2 p = w;
        asm volatile("" : "+v"(p.x), "+v"(p.y));
        const float2 wj = p;
        v1[1] = cmul(v1[1], p);
#pragma unroll
        for (int n1 = 2; n1 < 16; ++n1) {
            p = cmul(p, wj);
            v1[n1] = cmul(v1[n1], p);
        }
    };
    row_of(h0);
#pragma unroll
    for (int a = 0; a < 16; ++a) v1[a] = ld2(hrow_cur, (uint32_t)((1024 * a + tid - sh_cur) & (FB - 1)));
    p1_dft();
    p1_tw();
    float acc = 0.f;
    __syncthreads();

    for (int h = h0; h < h1; ++h) {
        const bool more = h + 1 < h1;
        int64_t hoff = (int64_t)h * 64;
        asm volatile("" : "+s"(hoff));
        int lz = 0;
        asm volatile("" : "+v"(lz));
        bar();  // everybody has finished the pass-4 reads of the previous hypothesis
        {
            const int off = (tid >> 6) * F_ROW + (tid & 63);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) s_d[n1 * F_N1 + off] = v1[n1];
        }
        bar();
        row_of(more ? h + 1 : h);
        // next hypothesis' template-spectrum row straight into the (now free) v1 registers
        if (!(FLAGS & 2)) {
#pragma unroll
            for (int a = 0; a < 16; ++a) v1[a] = ld2(hrow_cur, (uint32_t)(((1024 * a + tid - sh_cur) & (FB - 1)) + lz));
        }
        float2 v[16];
        {
            const int base = (tid >> 6) * F_N1 + (tid & 63);
#pragma unroll
            for (int b = 0; b < 16; ++b) v[b] = s_d[base + b * F_ROW];
            idft16(v);
#pragma unroll
            for (int n2 = 1; n2 < 16; ++n2) v[n2] = cmul(v[n2], (FLAGS & 8) ? w : s_tw2[n2 * 64 + (tid & 63) + lz]);
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) s_d[base + n2 * F_ROW] = v[n2];
        }
        __builtin_amdgcn_wave_barrier();
        {
            const int base = (tid >> 6) * F_N1 + ((tid >> 2) & 15) * F_ROW + (tid & 3);
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = s_d[base + 4 * c];
            // pass-1 butterfly of the next hypothesis: arithmetic only, flies under the LDS traffic around it
            p1_dft();
            if (ORDER == 0) p1_tw();
            idft16(v);
#pragma unroll
            for (int n3 = 1; n3 < 16; ++n3) v[n3] = cmul(v[n3], (FLAGS & 8) ? w : s_tw3[n3 * 4 + (tid & 3) + lz]);
#pragma unroll
            for (int n3 = 0; n3 < 16; ++n3) s_d[base + 4 * n3] = v[n3];
            if (ORDER == 1) p1_tw();
        }
        bar();
        {
            const int n1 = tid & 15, n2 = (tid >> 4) & 15, q4 = tid >> 8;
            const int base = n1 * F_N1 + n2 * F_ROW;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n3 = q4 + 4 * i;
                int lzi = 0;
                asm volatile("" : "+v"(lzi));
                const float4 lo = *reinterpret_cast<const float4*>(&s_d[base + 4 * n3 + lzi]);
                const float4 hi = *reinterpret_cast<const float4*>(&s_d[base + 4 * n3 + 2 + lzi]);
                float2 a0 = make_float2(lo.x, lo.y), a1 = make_float2(lo.z, lo.w);
                float2 a2 = make_float2(hi.x, hi.y), a3 = make_float2(hi.z, hi.w);
                idft4(a0, a1, a2, a3);
                const float2 y[4] = {a0, a1, a2, a3};
#pragma unroll
                for (int n4 = 0; n4 < 4; ++n4) {
                    const int tile_u = 16 * i + 64 * n4;
                    const int tile_t = (n2 >> 2) + 4 * q4;
                    float* pu = vt_blk + (int64_t)tile_u * nhyp * 64 + hoff;
                    const uint32_t voff = ((uint32_t)tile_t * (uint32_t)nhyp * 64u + (uint32_t)(n1 + 16 * (n2 & 3))) << 2;
                    const float val = y[n4].x * y[n4].x + y[n4].y * y[n4].y;
                    if (FLAGS & 1)
                        acc += val;
                    else if (tile_u + tile_t < tiles_per_blk)
                        gst1(pu, voff, val);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (FLAGS & 1) vt_blk[tid] = acc;
}

}  // namespace caf

static int g_threads = 1024;  // (arithmetic-only variants also run with fewer waves per workgroup)
template <int FLAGS>
static void run(const char* what, float2* xb, float2* hc, int32_t* sh, float2* tw1, float2* tw23, float* vt, int nblk) {
    const int nfreq = 256, nhyp = 256, hpw = 64, tiles = 193;
    const int ngroups = nhyp / hpw;
    const dim3 grid((unsigned)(ngroups * 8 * ((nblk + 7) / 8)));
    auto launch = [&]() {
        hipLaunchKernelGGL((caf::k_probe<FLAGS>), grid, dim3(g_threads), 0, 0, xb, hc, sh, tw1, tw23, nfreq, nhyp, hpw, nblk, tiles, vt);
    };
    launch();
    (void)hipDeviceSynchronize();
    float ms = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float m1;
        (void)hipEventElapsedTime(&m1, e0, e1);
        ms = m1 < ms ? m1 : ms;
    }
    const double transforms = (double)nblk * nhyp;
    // every CU runs transforms/256 of them back to back
    printf("flags %3d  %-58s %8.3f ms   %6.2f us per transform per CU\n", FLAGS, what, ms, ms * 1e3 / (transforms / 256.0));
    fflush(stdout);
}

template <int FLAGS, int ORDER>
static void run2(const char* what, float2* xb, float2* hc, int32_t* sh, float2* tw1, float2* tw23, float* vt, int nblk) {
    const int nfreq = 256, nhyp = 256, hpw = 64, tiles = 193;
    const int ngroups = nhyp / hpw;
    const dim3 grid((unsigned)(ngroups * 8 * ((nblk + 7) / 8)));
    auto launch = [&]() {
        hipLaunchKernelGGL((caf::k_probe2<FLAGS, ORDER>), grid, dim3(1024), 0, 0, xb, hc, sh, tw1, tw23, nfreq, nhyp, hpw, nblk, tiles, vt);
    };
    launch();
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double transforms = (double)nblk * nhyp;
    printf("v2 order %d flags %3d  %-46s %8.3f ms   %6.2f us per transform per CU (best of 3)\n", ORDER, FLAGS, what, best,
           best * 1e3 / (transforms / 256.0));
    fflush(stdout);
}

template <int FLAGS>
static void run_pk(const char* what, float2* xb, float2* hc, int32_t* sh, float2* tw1, float2* tw23, float* vt, int nblk) {
    const int nfreq = 256, nhyp = 256, hpw = 64, tiles = 193;
    const int ngroups = nhyp / hpw;
    const dim3 grid((unsigned)(ngroups * 8 * ((nblk + 7) / 8)));
    auto launch = [&]() {
        hipLaunchKernelGGL((caf::k_probe_pk<FLAGS>), grid, dim3(1024), 0, 0, xb, hc, sh, tw1, tw23, nfreq, nhyp, hpw, nblk, tiles, vt);
    };
    launch();
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("PACKED flags %3d  %-50s %8.3f ms   %6.2f us per transform per CU\n", FLAGS, what, best, best * 1e3 / ((double)nblk * nhyp / 256.0));
    fflush(stdout);
}

int main() {
    const int nblk = 512;  // 2048 workgroups = 8 rounds over 256 CUs
    const size_t FBs = 16384;
    float2 *xb, *hc, *tw1, *tw23;
    int32_t* sh;
    float* vt;
    (void)hipMalloc(&xb, nblk * FBs * 8);
    (void)hipMalloc(&hc, 2 * FBs * 8);
    (void)hipMalloc(&tw1, 16 * 1024 * 8);
    (void)hipMalloc(&tw23, (1024 + 64) * 8);
    (void)hipMalloc(&sh, 256 * 4);
    const size_t vt_bytes = (size_t)nblk * 193 * 256 * 64 * 4;
    (void)hipMalloc(&vt, vt_bytes);
    std::vector<float2> h(nblk * FBs);
    uint32_t s = 12345;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
    };
    for (auto& e : h) e = make_float2(rnd(), rnd());
    (void)hipMemcpy(xb, h.data(), nblk * FBs * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(hc, h.data(), 2 * FBs * 8, hipMemcpyHostToDevice);
    std::vector<float2> t(16 * 1024);
    for (int n1 = 0; n1 < 16; ++n1)
        for (int m2 = 0; m2 < 1024; ++m2) {
            const double a = 2.0 * M_PI * n1 * m2 / 16384.0;
            t[n1 * 1024 + m2] = make_float2((float)cos(a), (float)sin(a));
        }
    (void)hipMemcpy(tw1, t.data(), 16 * 1024 * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(tw23, t.data(), (1024 + 64) * 8, hipMemcpyHostToDevice);
    std::vector<int32_t> shv(256);
    for (int k = 0; k < 256; ++k) shv[k] = 4 * (k - 128);
    (void)hipMemcpy(sh, shv.data(), 256 * 4, hipMemcpyHostToDevice);

    // the chip's clock sags over the first seconds of load: warm up, then measure every variant right after a baseline
    for (int r = 0; r < 12; ++r) run<0>("full loop (warm-up)", xb, hc, sh, tw1, tw23, vt, nblk);
    if (getenv("PROBE_WAVES")) {
        // does the butterfly arithmetic need four waves per SIMD?  (no memory, no LDS, no barriers; time is per workgroup
        // of g_threads threads: at a wave-count-independent issue rate it scales with the thread count)
        for (int t : {1024, 768, 512, 256}) {
            g_threads = t;
            char what[96];
            snprintf(what, sizeof(what), "arithmetic only, %d threads per workgroup (%d waves per SIMD)", t, t / 256);
            run<111>(what, xb, hc, sh, tw1, tw23, vt, nblk);
        }
        g_threads = 1024;
        return 0;
    }
    if (getenv("PROBE_HLOAD")) {
        for (int rep = 0; rep < 3; ++rep) {
            run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
            run<16777216>("template-spectrum row loads at device scope (sc1: no L1)", xb, hc, sh, tw1, tw23, vt, nblk);
        }
        return 0;
    }
    if (getenv("PROBE_DPP")) {
        for (int rep = 0; rep < 3; ++rep) {
            run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
            run<8388608>("pass 4 across quad lanes (DPP), two barriers", xb, hc, sh, tw1, tw23, vt, nblk);
        }
        return 0;
    }
    if (getenv("PROBE_PRIO")) {
        for (int rep = 0; rep < 2; ++rep) {
            run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
            run<524288 * 1>("priorities by progress, passes 2-3 (3,2,1,0)", xb, hc, sh, tw1, tw23, vt, nblk);
            run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
            run<524288 * 2>("priority 2 from B2, 0 once pass 3 is computed", xb, hc, sh, tw1, tw23, vt, nblk);
            run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
            run<524288 * 3>("static priorities, youngest waves first", xb, hc, sh, tw1, tw23, vt, nblk);
            run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
            run<524288 * 4>("priorities by progress, all four passes", xb, hc, sh, tw1, tw23, vt, nblk);
            run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
            run<524288 * 5>("LDS-write phases high, arithmetic low", xb, hc, sh, tw1, tw23, vt, nblk);
            run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
            run<131072>("early B1", xb, hc, sh, tw1, tw23, vt, nblk);
            run<131072 + 524288 * 3>("early B1 + static priorities, youngest first", xb, hc, sh, tw1, tw23, vt, nblk);
            run<131072 + 524288 * 6>("early B1 + static priorities, oldest first", xb, hc, sh, tw1, tw23, vt, nblk);
            run<524288 * 6>("static priorities, oldest first", xb, hc, sh, tw1, tw23, vt, nblk);
        }
        return 0;
    }
    if (getenv("PROBE_STAMPS")) {
        unsigned long long* d_st;
        const size_t nst = 64 * 16 * 12;
        (void)hipMalloc(&d_st, nst * 8);
        (void)hipMemset(d_st, 0, nst * 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(caf::g_stamps), &d_st, sizeof(d_st));
        run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
        run<262144>("full loop with stamps", xb, hc, sh, tw1, tw23, vt, nblk);
        std::vector<unsigned long long> st(nst);
        (void)hipMemcpy(st.data(), d_st, nst * 8, hipMemcpyDeviceToHost);
        static const char* names[11] = {"pass-1 butterfly + twiddles", "wait B1", "pass-1 writes (issue)", "wait B2",
                                        "pass 2: reads + butterfly + twiddles", "pass-2 writes (issue) + wave barrier",
                                        "row loads (issue) + pass 3: reads + butterfly + twiddles", "pass-3 writes (issue)", "wait B3",
                                        "pass 4: reads + radix 4 + |y|^2 + stores", "next products X * H"};
        // one workgroup in detail (every wave), then the mean over 64 workgroups x 16 waves
        printf("workgroup 0, cycles per phase for each of its 16 waves (s_memtime):\n");
        for (int i = 0; i < 11; ++i) {
            printf("  %-58s", names[i]);
            for (int w = 0; w < 16; ++w) printf(" %5lld", (long long)(st[(0 * 16 + w) * 12 + i + 1] - st[(0 * 16 + w) * 12 + i]));
            printf("\n");
        }
        printf("  %-58s", "whole iteration");
        for (int w = 0; w < 16; ++w) printf(" %5lld", (long long)(st[w * 12 + 11] - st[w * 12]));
        printf("\nmean over 64 workgroups x 16 waves (min .. max):\n");
        double tot = 0;
        for (int i = 0; i < 11; ++i) {
            double sum = 0, mn = 1e18, mx = 0;
            for (int g = 0; g < 64 * 16; ++g) {
                const double dlt = (double)(st[g * 12 + i + 1] - st[g * 12 + i]);
                sum += dlt;
                mn = dlt < mn ? dlt : mn;
                mx = dlt > mx ? dlt : mx;
            }
            tot += sum / (64 * 16);
            printf("  %-58s %8.0f  (%6.0f .. %6.0f)\n", names[i], sum / (64 * 16), mn, mx);
        }
        printf("  %-58s %8.0f\n", "sum", tot);
        return 0;
    }
    if (getenv("PROBE_EARLY_B1")) {
        for (int rep = 0; rep < 2; ++rep) {
            run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
            run<131072>("B1 right after the pass-4 reads", xb, hc, sh, tw1, tw23, vt, nblk);
        }
        for (int grp = 0; grp < 2; ++grp)
            for (int prio = 0; prio < 2; ++prio)
                for (int k1 : {4, 8, 12, 16, 24})
                    for (int ka : {0, 8}) {
                        const int st[8] = {ka, 0, prio, grp, k1, 0, 0, 0};
                        (void)hipMemcpyToSymbol(HIP_SYMBOL(caf::g_st), st, sizeof(st));
                        char what[96];
                        snprintf(what, sizeof(what), "early B1 + stagger: groups %d prio %d sleep %2d after B1, %2d after B2", grp, prio, k1, ka);
                        run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
                        run<131072 + 65536>(what, xb, hc, sh, tw1, tw23, vt, nblk);
                    }
        return 0;
    }
    if (getenv("PROBE_STAGGER")) {
        // wave stagger sweep: sleep units after B2 / B3 for the second wave group, priorities, group shapes
        for (int grp = 0; grp < 2; ++grp)
            for (int prio = 0; prio < 2; ++prio)
                for (int ka : {0, 4, 8, 12, 16, 24, 32})
                    for (int kb : {0, 8, 16}) {
                        if (ka == 0 && kb == 0 && !prio) continue;
                        const int st[8] = {ka, kb, prio, grp, 0, 0, 0, 0};
                        (void)hipMemcpyToSymbol(HIP_SYMBOL(caf::g_st), st, sizeof(st));
                        char what[96];
                        snprintf(what, sizeof(what), "stagger: groups %d prio %d sleep %2d after B2, %2d after B3", grp, prio, ka, kb);
                        run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
                        run<65536>(what, xb, hc, sh, tw1, tw23, vt, nblk);
                    }
        return 0;
    }
#define AB(F, WHAT)                                              \
    run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);       \
    run<F>(WHAT, xb, hc, sh, tw1, tw23, vt, nblk);
    run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
    run_pk<0>("full loop, packed arithmetic", xb, hc, sh, tw1, tw23, vt, nblk);
    run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
    run_pk<0>("full loop, packed arithmetic", xb, hc, sh, tw1, tw23, vt, nblk);
    run<111>("arithmetic only", xb, hc, sh, tw1, tw23, vt, nblk);
    run_pk<111>("arithmetic only, packed", xb, hc, sh, tw1, tw23, vt, nblk);
    run<7>("- stores - rows - barriers", xb, hc, sh, tw1, tw23, vt, nblk);
    run_pk<7>("- stores - rows - barriers, packed", xb, hc, sh, tw1, tw23, vt, nblk);
    AB(8192, "row loads issued after pass 3")
    AB(32768, "row loads: 8 before pass 3, 8 after")
    AB(16384, "row loads nontemporal")
    AB(8192, "row loads issued after pass 3 (again)")
    AB(1, "- global |y|^2 stores")
    AB(2, "- template-spectrum row loads")
    AB(3, "- stores - row loads")
    AB(4, "- workgroup barriers (all three)")
    AB(256, "- B3 only (before pass 4)")
    AB(512, "- B1 only (before pass-1 writes)")
    AB(1024, "12 unconditional stores, n4 = 3 dropped")
    AB(1024 + 256, "12 stores, - B3")
    AB(8, "- twiddle-table LDS reads")
    AB(16, "- pass-1 twiddle recurrence")
    AB(64, "- pass-4 LDS reads")
    AB(128, "- idft16 arithmetic")
    AB(7, "- stores - rows - barriers")
    AB(111, "arithmetic only (no memory, LDS, barriers)")
    AB(155, "LDS + barriers only (no butterflies, no global memory)")
    run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
    run2<0, 0>("pipelined pass 1 (P2, P1c, P3)", xb, hc, sh, tw1, tw23, vt, nblk);
    run<0>("full loop", xb, hc, sh, tw1, tw23, vt, nblk);
    run2<7, 0>("pipelined, - stores - rows - barriers", xb, hc, sh, tw1, tw23, vt, nblk);
    return 0;
}
