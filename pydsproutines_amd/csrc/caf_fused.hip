// Fused hypothesis kernel for gfx950: spectral conjugate-multiply -> 16384-point inverse FFT held
// entirely in LDS -> |.|^2, one workgroup per (rx block, group of hypotheses).
//
// Why: with rocFFT in the middle, every hypothesis product makes four HBM passes
// (multiply write, two FFT passes read+write, |.|^2 read) = ~55 B per CAF cell.  Here the product
// never leaves the CU: the block spectrum X stays in registers for the whole hypothesis loop, the
// template-spectrum row is read from L2 and multiplied one hypothesis ahead, the 16384-point transform runs
// in the 160 KB LDS of a CDNA4 CU, and only |y|^2 (4 B per cell, coalesced 256-B rows) goes to HBM.
// A second kernel (k_transpose_norm_argmax) turns the hypothesis-major |y|^2 tiles into the
// delay-major QF^2 surface + per-delay argmax + peak.  Replaces the same reference stages as
// caf_kernels.hip (multiplySlices.cu:206-211, cuFFT, complex_magn.cu:8-19, argmax.cu:93-153).
//
// Transform: y[n] = sum_m P[m] e^{+j 2 pi m n / B},  B = 16384 = 16 * 16 * 16 * 4, decimation as
//   m = 1024 a + 64 b + 4 c + d      (a,b,c in [0,16), d in [0,4))   input index
//   n = n1 + 16 n2 + 256 n3 + 4096 n4 (n1,n2,n3 in [0,16), n4 in [0,4)) output index
// pass 1: DFT16 over a  (butterfly <-> m2 = 64 b + 4 c + d), twiddle e^{j2pi m2 n1 / 16384}
// pass 2: DFT16 over b  (butterfly <-> n1, 4c+d),            twiddle e^{j2pi (4c+d) n2 / 1024}
// pass 3: DFT16 over c  (butterfly <-> n1, n2, d),           twiddle e^{j2pi d n3 / 64}
// pass 4: DFT4  over d  (butterfly <-> n1, n2, n3) -> y, lanes <-> consecutive n
// Passes 2 and 3 read and write the same LDS addresses per butterfly (in place), so only one
// barrier per pass is needed.  LDS image: element (n1, row, col) at n1*1090 + row*68 + col
// (complex64); the 68/1090 pitches keep the strided reads of passes 3 and 4 off the same banks.
// Default: 1024 threads (4 waves per SIMD, 102 VGPRs), one butterfly per thread and pass; the
// 512-thread / two-butterfly variant is kept as an A/B switch (measured 20 % slower).
#include <cstdlib>

#include "caf_internal.h"

namespace caf {

constexpr int FB = 16384;              // fused block size
constexpr int F_ROW = 68;              // sub-row pitch (64 used)
constexpr int F_N1 = 16 * F_ROW + 2;   // pitch between n1 planes (1090)
constexpr int F_LDS_DATA = 16 * F_N1;  // complex elements

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 mulj(float2 a) { return make_float2(-a.y, a.x); }  // * (+j)
// uniform base + 32-bit BYTE offset: lets the compiler pick the scalar-base (saddr) addressing form instead
// of building a 64-bit address per access
__device__ __forceinline__ float2 ld2(const float2* base, uint32_t elem) {
    return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base) + (elem << 3));
}

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

// |y|^2 tiles: vt[blk_local][s_tile][h][64]  (s_tile = delay/64 inside the block)
// FT threads per workgroup (512: 2 waves/SIMD, 256-VGPR budget; 1024: 4 waves/SIMD, 128 VGPRs);
// BPT = 1024 / FT radix-16 butterflies per thread and pass.
template <int FT>
__global__ __launch_bounds__(FT) void k_fused_caf(const float2* __restrict__ xb,       // [blocks][FB] spectra
                                                  const float2* __restrict__ hc,       // [T][FB] or [T*F][FB]
                                                  const int32_t* __restrict__ shifts,  // [F] (shift modes)
                                                  const float2* __restrict__ tw1,      // [16][1024]
                                                  const float2* __restrict__ tw23,     // [16][64] then [16][4]
                                                  int32_t table_mode, int32_t nfreq, int32_t nhyp, int32_t hyp_per_wg,
                                                  int32_t nblk, int32_t tiles_per_blk, float* __restrict__ vt) {
    constexpr int BPT = 1024 / FT;
    __shared__ __attribute__((aligned(16))) float2 s_d[F_LDS_DATA];
    __shared__ float2 s_tw2[16 * 64];
    __shared__ float2 s_tw3[16 * 4];
    const int tid = threadIdx.x;
    // XCD-aware mapping (speed only): workgroups are dealt round-robin over the 8 XCDs, so linear id L runs
    // on XCD L % 8.  All hypothesis groups of one rx block are given to the SAME XCD (block b -> XCD b % 8,
    // its groups on consecutive slots of that XCD), so the block spectrum X that every group re-reads per
    // hypothesis is served by one 4 MiB L2 instead of being replicated (and thrashed) in several.
    const int ngroups = (nhyp + hyp_per_wg - 1) / hyp_per_wg;
    const int lin = blockIdx.x;
    const int q = lin >> 3;
    const int blk = (q / ngroups) * 8 + (lin & 7);
    const int grp = q - (q / ngroups) * ngroups;
    if (blk >= nblk) return;
    const int h0 = grp * hyp_per_wg;
    const int h1 = min(h0 + hyp_per_wg, nhyp);

    for (int i = tid; i < 1024; i += FT) s_tw2[i] = tw23[i];
    if (tid < 64) s_tw3[tid] = tw23[1024 + tid];

    // hypothesis-independent per-thread state: the pass-1 twiddle base e^{+j 2 pi m2 / 16384}
    float2 w[BPT];
    const float2* xp = xb + (int64_t)blk * FB;  // uniform base; per-thread offsets stay 32-bit (saddr loads)
#pragma unroll
    for (int j = 0; j < BPT; ++j) w[j] = tw1[1024 + tid + j * FT];
    float* vt_blk = vt + (int64_t)blk * tiles_per_blk * nhyp * 64;

    // row of the template-spectrum table used by hypothesis h (uniform) and its circular shift
    const float2* hrow_cur;
    int sh_cur;
    auto row_of = [&](int h) {
        if (table_mode) {
            hrow_cur = hc + (int64_t)h * FB;
            sh_cur = 0;
        } else {
            const int t = h / nfreq;
            sh_cur = shifts[h - t * nfreq];
            hrow_cur = hc + (int64_t)t * FB;
        }
    };
    // pr[j][a] = X[1024 a + m2] * Hc_h[1024 a + m2]: the input of pass 1, produced one hypothesis ahead
    // (during pass 4 of the previous one) from the persistent X registers and a freshly loaded Hc row.
    float2 pr[BPT][16];
    float2 xr[BPT][16];  // block spectrum, persistent
#pragma unroll
    for (int j = 0; j < BPT; ++j)
#pragma unroll
        for (int a = 0; a < 16; ++a) xr[j][a] = ld2(xp, (uint32_t)(1024 * a + j * FT + tid));
    row_of(h0);
#pragma unroll
    for (int j = 0; j < BPT; ++j)
#pragma unroll
        for (int a = 0; a < 16; ++a)
            pr[j][a] = cmul(xr[j][a],
                            ld2(hrow_cur, (uint32_t)((1024 * a + tid + j * FT - sh_cur) & (FB - 1))));

    for (int h = h0; h < h1; ++h) {
        const bool more = h + 1 < h1;
        // h*64 as an opaque scalar: otherwise loop-strength-reduction turns the 32 store addresses of pass 4
        // into 32 64-bit induction variables (64 VGPRs + 32 adds per hypothesis)
        int64_t hoff = (int64_t)h * 64;
        asm volatile("" : "+s"(hoff));
        int lz = 0;  // an opaque zero added to loop-invariant LDS table indices (see pass 1)
        asm volatile("" : "+v"(lz));
        // ---- pass 1: P = X * Hc_h ; DFT16 over a ; twiddle w^n1 ; write A[n1][m2] ----
        // One butterfly at a time (32 live data registers).  pr is dead after this pass and is refilled
        // with the next hypothesis' products during pass 4, the low-pressure phase.
        // The register-only part (DFT16 + twiddles) runs BEFORE the barrier that protects the LDS image, so it
        // overlaps with slower waves still finishing pass 4 of the previous hypothesis.
        float2 v1[BPT][16];
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
#pragma unroll
            for (int a = 0; a < 16; ++a) v1[j][a] = pr[j][a];
            idft16(v1[j]);
            float2 p = w[j];
            // opaque to the optimiser: otherwise the 15 powers (and the LDS twiddles below) are hoisted out
            // of the hypothesis loop as loop invariants and cost ~180 persistent registers
            asm volatile("" : "+v"(p.x), "+v"(p.y));
            const float2 wj = p;
            v1[j][1] = cmul(v1[j][1], p);
#pragma unroll
            for (int n1 = 2; n1 < 16; ++n1) {
                p = cmul(p, wj);
                v1[j][n1] = cmul(v1[j][n1], p);
            }
        }
        __syncthreads();  // previous hypothesis' pass-4 reads are done (and the LDS tables are in place)
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            const int m2 = tid + j * FT;
            const int off = (m2 >> 6) * F_ROW + (m2 & 63);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) s_d[n1 * F_N1 + off] = v1[j][n1];
        }
        __syncthreads();
        row_of(more ? h + 1 : h);  // unconditional refill (the last one is redundant): no select keeps pr alive
        // ---- pass 2: DFT16 over b, in place (n1 = idx >> 6, col = idx & 63) ----
        // all of a thread's butterflies are read first, so the LDS reads of butterfly j+1 fly under the
        // arithmetic of butterfly j (they touch disjoint addresses)
        {
            float2 v[BPT][16];
#pragma unroll
            for (int j = 0; j < BPT; ++j) {
                const int idx = tid + j * FT;
                const int base = (idx >> 6) * F_N1 + (idx & 63);
#pragma unroll
                for (int b = 0; b < 16; ++b) v[j][b] = s_d[base + b * F_ROW];
            }
#pragma unroll
            for (int j = 0; j < BPT; ++j) {
                const int idx = tid + j * FT;
                const int base = (idx >> 6) * F_N1 + (idx & 63);
                idft16(v[j]);
#pragma unroll
                for (int n2 = 1; n2 < 16; ++n2) v[j][n2] = cmul(v[j][n2], s_tw2[n2 * 64 + (idx & 63) + lz]);
#pragma unroll
                for (int n2 = 0; n2 < 16; ++n2) s_d[base + n2 * F_ROW] = v[j][n2];
            }
        }
        // No workgroup barrier here: plane n1 = idx >> 6 is written in pass 2 and read in pass 3 by the SAME
        // wave (wave w owns planes w, w + FT/64, ...), and a wave's LDS operations complete in order.
        __builtin_amdgcn_wave_barrier();
        // next hypothesis' template-spectrum row: issued before pass 3 so that the L2 latency is covered by
        // the pass-3 butterfly and the pass-4 work
        float2 hn[BPT][16];
#pragma unroll
        for (int j = 0; j < BPT; ++j)
#pragma unroll
            for (int a = 0; a < 16; ++a)
                hn[j][a] = ld2(hrow_cur, (uint32_t)(((1024 * a + tid + j * FT - sh_cur) & (FB - 1)) + lz));
        // ---- pass 3: DFT16 over c, in place (n1 = idx >> 6, n2 = (idx >> 2) & 15, d = idx & 3) ----
        {
            float2 v[BPT][16];
#pragma unroll
            for (int j = 0; j < BPT; ++j) {
                const int idx = tid + j * FT;
                const int base = (idx >> 6) * F_N1 + ((idx >> 2) & 15) * F_ROW + (idx & 3);
#pragma unroll
                for (int c = 0; c < 16; ++c) v[j][c] = s_d[base + 4 * c];
            }
#pragma unroll
            for (int j = 0; j < BPT; ++j) {
                const int idx = tid + j * FT;
                const int base = (idx >> 6) * F_N1 + ((idx >> 2) & 15) * F_ROW + (idx & 3);
                idft16(v[j]);
#pragma unroll
                for (int n3 = 1; n3 < 16; ++n3) v[j][n3] = cmul(v[j][n3], s_tw3[n3 * 4 + (idx & 3) + lz]);
#pragma unroll
                for (int n3 = 0; n3 < 16; ++n3) s_d[base + 4 * n3] = v[j][n3];
            }
        }
        __syncthreads();
        // ---- pass 4: DFT4 over d ; |y|^2 -> vt tiles (lanes <-> consecutive delays) ----
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            const int idx = tid + j * FT;
            const int n1 = idx & 15, n2 = (idx >> 4) & 15, q = idx >> 8;
            const int base = n1 * F_N1 + n2 * F_ROW;
            // next hypothesis' template-spectrum row for butterfly j: the loads fly while this butterfly's
            // pass-4 work runs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n3 = q + 4 * i;
                int lzi = 0;
                asm volatile("" : "+v"(lzi));
                const float4 lo = *reinterpret_cast<const float4*>(&s_d[base + 4 * n3 + lzi]);
                const float4 hi = *reinterpret_cast<const float4*>(&s_d[base + 4 * n3 + 2 + lzi]);
                float2 a0 = make_float2(lo.x, lo.y), a1 = make_float2(lo.z, lo.w);
                float2 a2 = make_float2(hi.x, hi.y), a3 = make_float2(hi.z, hi.w);
                idft4(a0, a1, a2, a3);
                // n = n1 + 16 n2 + 256 n3 + 4096 n4  ->  tile = n >> 6 = (n2 >> 2) + 4 n3 + 64 n4, lane = n & 63.
                // Uniform (scalar) part of the address + one 32-bit per-thread offset, so that no per-store
                // 64-bit address is kept alive across the hypothesis loop.
                const float2 y[4] = {a0, a1, a2, a3};
#pragma unroll
                for (int n4 = 0; n4 < 4; ++n4) {
                    const int tile_u = 16 * i + 64 * n4;                      // uniform part of the tile index
                    const int tile_t = (n2 >> 2) + 4 * q;                     // per-thread part
                    float* pu = vt_blk + (int64_t)tile_u * nhyp * 64 + hoff;  // uniform (scalar) base
                    const uint32_t voff = ((uint32_t)tile_t * (uint32_t)nhyp * 64u + (uint32_t)(n1 + 16 * (n2 & 3))) << 2;
                    if (tile_u + tile_t < tiles_per_blk)
                        *reinterpret_cast<float*>(reinterpret_cast<char*>(pu) + voff) = y[n4].x * y[n4].x + y[n4].y * y[n4].y;
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the four sub-steps from being co-scheduled (registers)
            }
#pragma unroll
            for (int a = 0; a < 16; ++a) pr[j][a] = cmul(xr[j][a], hn[j][a]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ----------------------------------------------------------------------------------------
// |y|^2 tiles -> delay-major QF^2 surface + per-delay argmax + tile peak record.
//   in : vt[blk][s_tile][t*F+f][64] float32   (contiguous 64*F*4 bytes per (blk, s_tile, t))
//   out: as k_magsq_norm_argmax.
// One workgroup per (s_tile, template, block): streams its contiguous tile in chunks of TR_F
// hypotheses (32 KiB, 16-byte loads, 8 in flight per lane) through an LDS transpose; surface rows
// leave as 512-byte segments.
// ----------------------------------------------------------------------------------------
// TR_F = hypotheses per LDS chunk: 256 (66 KB tile, whole 1-KiB surface rows at F = 256) or 128 (33 KB).
template <int TR_F, bool NT_STORE>
__global__ __launch_bounds__(256) void k_transpose_norm_argmax(
    const float* __restrict__ vt, int32_t ntmpl, int32_t nfreq, const float* __restrict__ tscale,
    const float* __restrict__ inv_e, int64_t num_shifts, int64_t shift_start, int32_t step, int32_t blk0,
    int32_t tiles_per_blk, float* __restrict__ surface, float* __restrict__ row_max, int32_t* __restrict__ row_arg,
    PeakRec* __restrict__ partial, int64_t partial_per_tmpl) {
    __shared__ float s_tile[64][TR_F + 1];
    __shared__ float s_rowv[64];
    __shared__ int32_t s_rowi[64];
    const int z = blockIdx.z, t = blockIdx.y, tile = blockIdx.x;
    const int blk = blk0 + z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl0 = tile * 64;
    const int64_t rel0 = (int64_t)blk * step + sl0;
    int64_t nv = num_shifts - (int64_t)blk * step;
    if (nv > step) nv = step;
    const int64_t pidx = (int64_t)blk * tiles_per_blk + tile;
    if (sl0 >= nv) {
        if (threadIdx.x == 0 && partial) {
            PeakRec r;
            r.v = -1.f;
            r.delay = 0x7fffffff;
            r.f = 0;
            partial[(int64_t)t * partial_per_tmpl + pidx] = r;
        }
        return;
    }
    const int nrows = (int)min((int64_t)64, nv - sl0);
    const int nhyp = ntmpl * nfreq;
    const float* vin = vt + (((int64_t)z * tiles_per_blk + tile) * nhyp + (int64_t)t * nfreq) * 64;
    // load mapping: float4 number (i*256 + tid) of the chunk -> hypothesis fl = i*16 + (tid >> 4),
    // delays s4 .. s4+3 with s4 = 4*(tid & 15) (the same four delays for every i)
    const int s4 = 4 * (threadIdx.x & 15);
    const float ts = tscale[t];
    float g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = (s4 + k < nrows) ? inv_e[rel0 + s4 + k] * ts : -1.f;

    float bv[16];
    int32_t bi[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bv[r] = -1.f;
        bi[r] = 0;
    }
    // Per-lane running best over the columns this lane visits (lane + 64 c of every chunk); the
    // cross-lane reduction is done once per row after the last chunk.  The next chunk's loads are issued
    // before the store phase of the current one, so HBM reads stay in flight while rows are written.
    float4 q[TR_F / 16];
    auto load_chunk = [&](int f0) {
        const int nf = min(TR_F, nfreq - f0);
#pragma unroll
        for (int i = 0; i < TR_F / 16; ++i) {
            const int fl = i * 16 + (threadIdx.x >> 4);
            q[i] = (fl < nf) ? *reinterpret_cast<const float4*>(vin + (int64_t)(f0 + fl) * 64 + s4)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_chunk(0);
    for (int f0 = 0; f0 < nfreq; f0 += TR_F) {
        const int nf = min(TR_F, nfreq - f0);
#pragma unroll
        for (int i = 0; i < TR_F / 16; ++i) {
            const int fl = i * 16 + (threadIdx.x >> 4);
            // invalid delays (g < 0) are stored as -1 so that they never win the argmax
            s_tile[s4 + 0][fl] = g[0] < 0.f ? -1.f : q[i].x * g[0];
            s_tile[s4 + 1][fl] = g[1] < 0.f ? -1.f : q[i].y * g[1];
            s_tile[s4 + 2][fl] = g[2] < 0.f ? -1.f : q[i].z * g[2];
            s_tile[s4 + 3][fl] = g[3] < 0.f ? -1.f : q[i].w * g[3];
        }
        __syncthreads();
        if (f0 + TR_F < nfreq) load_chunk(f0 + TR_F);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave + 4 * r;
            if (row < nrows) {
                float* srow = surface ? surface + ((int64_t)t * num_shifts + rel0 + row) * nfreq + f0 : nullptr;
#pragma unroll
                for (int c = 0; c < TR_F / 64; ++c) {
                    const int fl = lane + 64 * c;
                    if (fl < nf) {
                        const float v = s_tile[row][fl];
                        if (srow) {
                            if (NT_STORE)
                                __builtin_nontemporal_store(v, &srow[fl]);  // write-once stream: keep it out of L2
                            else
                                srow[fl] = v;
                        }
                        if (v > bv[r]) {  // columns are visited in increasing order: first maximum wins
                            bv[r] = v;
                            bi[r] = f0 + fl;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // one cross-lane reduction per row: highest value, lowest frequency index on ties
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv[r], o, 64);
            const int32_t oi = __shfl_xor(bi[r], o, 64);
            if (ov > bv[r] || (ov == bv[r] && oi < bi[r])) {
                bv[r] = ov;
                bi[r] = oi;
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s_rowv[wave + 4 * r] = bv[r];
            s_rowi[wave + 4 * r] = bi[r];
        }
    }
    __syncthreads();
    if (wave == 0) {
        float v = -1.f;
        if (lane < nrows) {
            v = s_rowv[lane];
            const int64_t o = (int64_t)t * num_shifts + rel0 + lane;
            if (row_max) row_max[o] = v;
            if (row_arg) row_arg[o] = s_rowi[lane];
        }
        if (partial) {
            float b = v;
            int32_t bidx = lane;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(b, o, 64);
                const int32_t oi = __shfl_xor(bidx, o, 64);
                if (ov > b || (ov == b && oi < bidx)) {
                    b = ov;
                    bidx = oi;
                }
            }
            if (lane == 0) {
                PeakRec r;
                r.v = b;
                r.delay = (int32_t)(shift_start + rel0 + bidx);
                r.f = s_rowi[bidx];
                partial[(int64_t)t * partial_per_tmpl + pidx] = r;
            }
        }
    }
}

void launch_fused_caf(const float2* xb, const float2* hc, const int32_t* shifts, const float2* tw1,
                      const float2* tw23, int32_t table_mode, int32_t nfreq, int32_t nhyp, int32_t hyp_per_wg,
                      int32_t nblk, int32_t tiles_per_blk, float* vt, hipStream_t st) {
    const int ngroups = (nhyp + hyp_per_wg - 1) / hyp_per_wg;
    const dim3 grid((unsigned)(ngroups * 8 * ((nblk + 7) / 8)));  // 1-D, XCD-aware mapping inside the kernel
    // CAF_FUSED_THREADS=512 selects the 2-waves/SIMD variant (A/B switch; 1024 measured faster)
    static const int threads = [] {
        const char* e = getenv("CAF_FUSED_THREADS");
        return (e && atoi(e) == 512) ? 512 : 1024;
    }();
    if (threads == 512)
        hipLaunchKernelGGL(k_fused_caf<512>, grid, dim3(512), 0, st, xb, hc, shifts, tw1, tw23, table_mode, nfreq,
                           nhyp, hyp_per_wg, nblk, tiles_per_blk, vt);
    else
        hipLaunchKernelGGL(k_fused_caf<1024>, grid, dim3(1024), 0, st, xb, hc, shifts, tw1, tw23, table_mode, nfreq,
                           nhyp, hyp_per_wg, nblk, tiles_per_blk, vt);
}

void launch_transpose_norm_argmax(const float* vt, int32_t ntmpl, int32_t nfreq, const float* tscale,
                                  const float* inv_e, int64_t num_shifts, int64_t shift_start, int32_t step,
                                  int32_t blk0, int32_t nblk, int32_t tiles_per_blk, float* surface, float* row_max,
                                  int32_t* row_arg, PeakRec* partial, int64_t partial_per_tmpl, hipStream_t st) {
    // CAF_TR_F=64|256 selects another chunk width (A/B switch; 128 measured fastest at F = 256:
    // 256 halves the occupancy, 64 halves the store segments)
    static const int trf = [] {
        const char* e = getenv("CAF_TR_F");
        return e ? atoi(e) : 128;
    }();
    static const int nts = [] {
        const char* e = getenv("CAF_NT_STORE");
        return e ? atoi(e) : 0;
    }();
    const dim3 grid(tiles_per_blk, ntmpl, nblk);
    if (trf == 256 && nfreq > 128)
        hipLaunchKernelGGL((k_transpose_norm_argmax<256, false>), grid, dim3(256), 0, st, vt, ntmpl, nfreq, tscale,
                           inv_e, num_shifts, shift_start, step, blk0, tiles_per_blk, surface, row_max, row_arg,
                           partial, partial_per_tmpl);
    else if (trf == 64)
        hipLaunchKernelGGL((k_transpose_norm_argmax<64, false>), grid, dim3(256), 0, st, vt, ntmpl, nfreq, tscale,
                           inv_e, num_shifts, shift_start, step, blk0, tiles_per_blk, surface, row_max, row_arg,
                           partial, partial_per_tmpl);
    else if (nts)
        hipLaunchKernelGGL((k_transpose_norm_argmax<128, true>), grid, dim3(256), 0, st, vt, ntmpl, nfreq, tscale,
                           inv_e, num_shifts, shift_start, step, blk0, tiles_per_blk, surface, row_max, row_arg,
                           partial, partial_per_tmpl);
    else
        hipLaunchKernelGGL((k_transpose_norm_argmax<128, false>), grid, dim3(256), 0, st, vt, ntmpl, nfreq, tscale,
                           inv_e, num_shifts, shift_start, step, blk0, tiles_per_blk, surface, row_max, row_arg,
                           partial, partial_per_tmpl);
}

}  // namespace caf
