// Hand-written gfx950 (CDNA4, wave64) kernels of the CAF hypothesis engine.
//
// Replaces, with a different algorithm (frequency-domain overlap-save per hypothesis
// instead of one DFT per delay), what the reference does with
//   custom_kernels/multiplySlices.cu:113-216   sliding conjugate-multiply + window energy
//   custom_kernels/complex_magn.cu:8-19        |.|^2
//   custom_kernels/argmax.cu:93-153            per-row argmax
//   custom_kernels/filter.cu:291-347           sliding energy (moving sum, double accumulate)
// These kernels are HBM-bound elementwise / transpose / reduction work: no MFMA.
#include "caf_internal.h"
#include "caf_energy.h"

namespace caf {

// ----------------------------------------------------------------------------------------
// Sliding energy.  P[i] = sum_{j<i} |rx[j]|^2 in float64 for i in [0, M]; the window energy
// of any support is then a difference of two prefix values (exact to ~1e-16 relative), the
// f64 counterpart of the reference's double-accumulated moving sum (filter.cu:324-339,
// multiplySlices.cu:153,201).
// ----------------------------------------------------------------------------------------
constexpr int PFX_THREADS = 256;
constexpr int PFX_PER_THREAD = 4;
constexpr int PFX_TILE = PFX_THREADS * PFX_PER_THREAD;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Exclusive scan of the tile sums in place, one workgroup: a thread owns up to 16 consecutive entries, so up to 16384
// tiles take one trip through the wave scan and the two barriers (one entry per thread and a trip per 1024 tiles took
// 17.6 us on the 16384 tiles of a 1.7e7-sample record); longer arrays carry the total from chunk to chunk.
__global__ __launch_bounds__(1024) void k_scan_tile_sums(double* __restrict__ tile_sums, int64_t ntiles) {
    constexpr int MAXPER = 16;
    __shared__ double s_wave[16];
    __shared__ double s_carry;
    if (threadIdx.x == 0) s_carry = 0.0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (int)((ntiles + 1023) / 1024 < MAXPER ? (ntiles + 1023) / 1024 : MAXPER);
    for (int64_t c = 0; c < ntiles; c += (int64_t)1024 * per) {
        const int64_t i0 = c + (int64_t)threadIdx.x * per;
        double v[MAXPER];
        double tot = 0.0;
#pragma unroll
        for (int j = 0; j < MAXPER; ++j) {
            v[j] = (j < per && i0 + j < ntiles) ? tile_sums[i0 + j] : 0.0;
            tot += v[j];
        }
        double incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        __syncthreads();  // (s_carry of the previous chunk is in place; s_wave is free)
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        double run = s_carry + (incl - tot);
        for (int w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
        for (int j = 0; j < MAXPER; ++j) {
            if (j < per && i0 + j < ntiles) tile_sums[i0 + j] = run;
            run += v[j];
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = run;
    }
}

// prefix[i] for i in [0, m] (m + 1 entries) in two launches of ONE kernel body: WRITE = false leaves the total of every
// tile in tile_sums, WRITE = true adds up the totals of the tiles before its own (a fixed summation order; tile_sums
// already scanned when there are more than PFX_DIRECT_TILES of them) and writes the tile's prefix; the values do not
// depend on timing.  A tile = 256 threads x 8 samples x SUB sub-tiles, a thread's samples staying in registers (16-byte loads, 16-byte
// stores).  24 B of traffic per sample against the 16 B of a one-launch scan -- which was built and measured
// (scripts/ubench/tilescan_model.hip, profiles/r04/ubench/tilescan_model.log: ticketed tiles + look-back through
// device-scope atomics, 57 us for 10^7 samples however the tiles were sized) and is not used.
constexpr int PF_NT = 256, PF_PER = 8;
constexpr int PFX_DIRECT_TILES = 1024;

template <int SUB, bool ALIGNED, bool WRITE>
__global__ __launch_bounds__(PF_NT) void k_prefix_tiles(const float2* __restrict__ rx, int64_t m, double* __restrict__ tile_sums,
                                                        int32_t scanned, double* __restrict__ prefix) {
    constexpr int NW = PF_NT / 64;
    __shared__ double s_wave[SUB][NW];  // wave totals, then (WRITE, wave 0) what precedes each wave in the whole record
    __shared__ double s_part[NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tbase = (int64_t)blockIdx.x * (PF_NT * PF_PER * SUB) + (int64_t)threadIdx.x * PF_PER;
    float2 v[SUB][PF_PER];
#pragma unroll
    for (int k = 0; k < SUB; ++k) {
        const int64_t base = tbase + (int64_t)k * PF_NT * PF_PER;
        if (base + PF_PER <= m) {
            if (ALIGNED) {
#pragma unroll
                for (int j = 0; j < PF_PER; j += 2) {
                    const float4 q = *reinterpret_cast<const float4*>(rx + base + j);
                    v[k][j] = make_float2(q.x, q.y);
                    v[k][j + 1] = make_float2(q.z, q.w);
                }
            } else {
#pragma unroll
                for (int j = 0; j < PF_PER; ++j) v[k][j] = rx[base + j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < PF_PER; ++j) v[k][j] = base + j < m ? rx[base + j] : make_float2(0.f, 0.f);
        }
    }
    // (WRITE) the totals of the preceding tiles, while the loads are in flight: thread t adds tiles t, t + 256, ...
    double before = 0.0;
    if (WRITE) {
        if (scanned) {
            before = tile_sums[blockIdx.x];
        } else {
            double c = 0.0;
            for (int t = threadIdx.x; t < (int)blockIdx.x; t += PF_NT) c += tile_sums[t];
            c = wave_sum_f64(c);
            if (lane == 0) s_part[wave] = c;
        }
    }
    double tot[SUB], incl[SUB];
#pragma unroll
    for (int k = 0; k < SUB; ++k) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < PF_PER; ++j) t += (double)v[k][j].x * (double)v[k][j].x + (double)v[k][j].y * (double)v[k][j].y;
        tot[k] = t;
        double in = t;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double u = __shfl_up(in, o, 64);
            if (lane >= o) in += u;
        }
        incl[k] = in;
        if (lane == 63) s_wave[k][wave] = in;
    }
    __syncthreads();
    if (wave == 0) {
        if (WRITE && !scanned) {
#pragma unroll
            for (int w = 0; w < NW; ++w) before += s_part[w];
        }
        double run = before;  // sequential over sub-tiles and waves, the same order in every lane and in both launches
#pragma unroll
        for (int k = 0; k < SUB; ++k)
            for (int w = 0; w < NW; ++w) {
                const double t = s_wave[k][w];
                if (WRITE && lane == w) s_wave[k][w] = run;
                run += t;
            }
        if (!WRITE && lane == 0) tile_sums[blockIdx.x] = run;
    }
    if (!WRITE) return;
    // the energies of the 64-sample chunks (8 lanes x 8 samples), plain sums: what window_energy_direct adds up (caf_energy.h)
    {
        double* chunks = prefix + ((m + 2) & ~(int64_t)1);
        const int64_t nchunks = (m + 63) / 64;
#pragma unroll
        for (int k = 0; k < SUB; ++k) {
            double c = tot[k];
            c += __shfl_xor(c, 1, 64);
            c += __shfl_xor(c, 2, 64);
            c += __shfl_xor(c, 4, 64);
            const int64_t ci = (tbase + (int64_t)k * PF_NT * PF_PER) >> 6;
            if ((lane & 7) == 0 && ci < nchunks) chunks[ci] = c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SUB; ++k) {
        const int64_t base = tbase + (int64_t)k * PF_NT * PF_PER;
        const double off = s_wave[k][wave] + (incl[k] - tot[k]);
        double p[PF_PER];
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < PF_PER; ++j) {
            p[j] = off + t;  // (t: the exclusive sum within the thread, the same additions as tot[k])
            t += (double)v[k][j].x * (double)v[k][j].x + (double)v[k][j].y * (double)v[k][j].y;
        }
        if (base + PF_PER <= m + 1) {
#pragma unroll
            for (int j = 0; j < PF_PER; j += 2) *reinterpret_cast<double2*>(prefix + base + j) = make_double2(p[j], p[j + 1]);
        } else {
#pragma unroll
            for (int j = 0; j < PF_PER; ++j)
                if (base + j <= m) prefix[base + j] = p[j];
        }
    }
}

// inv_e[i] = 1 / sum_g E(s + st_g, len_g),  s = shift_start + i; E from the prefix (window_energy, caf_energy.h).
// A window of zeros (a gap in a recording) is the reference's 0 / 0 = NaN, and only that is.
__global__ __launch_bounds__(256) void k_inv_energy(const float2* __restrict__ rx, int64_t rx_len, const double* __restrict__ prefix,
                                                    int64_t shift_start, int64_t num_shifts, const int32_t* __restrict__ gstart,
                                                    const int32_t* __restrict__ glen, int32_t ngroups,
                                                    float* __restrict__ inv_e) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= num_shifts) return;
    const int64_t s = shift_start + i;
    double e = 0.0;
    for (int g = 0; g < ngroups; ++g) {
        const int64_t a = s + gstart[g];
        e += window_energy(prefix, rx, rx_len, a, a + glen[g]);
    }
    inv_e[i] = e > 0.0 ? (float)(1.0 / e) : __builtin_nanf("");
}

// ----------------------------------------------------------------------------------------
// Overlap-save block gather: xb[b][m] = rx[src0 + b*step + m] (0 past the end of rx).
// ----------------------------------------------------------------------------------------
// (two consecutive points per thread: one 16-byte load -- 8-byte aligned, which global loads allow -- and one aligned 16-byte
//  store; 8 bytes per lane left the copy at 2.0 TB/s, a quarter of the peak)
typedef float v4f_a8_t __attribute__((ext_vector_type(4), aligned(8)));
__global__ __launch_bounds__(256) void k_gather_blocks(const float2* __restrict__ rx, int64_t rx_len,
                                                       int64_t src0, int32_t step, int32_t log2_bsz,
                                                       int64_t total, float2* __restrict__ xb) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;  // (block sizes are even: a pair never straddles two blocks)
    if (i >= total) return;
    const int64_t b = i >> log2_bsz;
    const int64_t m = i & (((int64_t)1 << log2_bsz) - 1);
    const int64_t src = src0 + b * step + m;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (src + 1 < rx_len) {
        const v4f_a8_t u = *reinterpret_cast<const v4f_a8_t*>(rx + src);
        v = make_float4(u.x, u.y, u.z, u.w);
    } else if (src < rx_len) {
        const float2 u = rx[src];
        v.x = u.x, v.y = u.y;
    }
    *reinterpret_cast<float4*>(xb + i) = v;
}

// hc[i] = conj(h[i]) * scale   (template spectra -> pre-conjugated, 1/B folded in)
__global__ __launch_bounds__(256) void k_conj_scale(float2* __restrict__ h, int64_t n, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float2 v = h[i];
    h[i] = make_float2(v.x * scale, -v.y * scale);
}

// ----------------------------------------------------------------------------------------
// Spectral conjugate-multiply across hypotheses (the frequency-domain counterpart of the
// reference's sliding conjugate-multiply, multiplySlices.cu:206-211):
//     P[z][h][m] = X[z][m] * Hc_h[m],     h = t*F + f
// SHIFT mode: Hc_h[m] = Hc0_t[(m - shift_f) mod B]  (on-grid frequency = circular shift of
// one template spectrum, so all F hypotheses read one B-point table that stays in L2);
// TABLE mode: Hc_h given explicitly.
// One thread owns two consecutive spectral points (16-byte loads/stores) and walks HG
// hypotheses with X held in registers: per element written, 8 B go to HBM and the X read is
// amortised 1/HG.  Write-bound.
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ float4 cmul2(const float4 x, const float4 h) {
    float4 r;
    r.x = x.x * h.x - x.y * h.y;
    r.y = x.x * h.y + x.y * h.x;
    r.z = x.z * h.z - x.w * h.w;
    r.w = x.z * h.w + x.w * h.z;
    return r;
}

template <int MODE>  // 0: shift, even shifts (16-B H loads) ; 1: shift, any ; 2: table
__global__ __launch_bounds__(MUL_THREADS) void k_spectral_mul(const float2* __restrict__ xb,
                                                             const float2* __restrict__ hc,
                                                             const int32_t* __restrict__ shifts, int32_t bsz,
                                                             int32_t pitch, int32_t nfreq, int32_t nhyp,
                                                             int32_t hyp_per_wg, float2* __restrict__ pbuf) {
    const int m = (blockIdx.x * MUL_THREADS + threadIdx.x) * 2;
    if (m >= bsz) return;
    const int z = blockIdx.z;
    const float4 x = *reinterpret_cast<const float4*>(xb + (int64_t)z * bsz + m);
    const int h0 = blockIdx.y * hyp_per_wg;
    const int h1 = min(h0 + hyp_per_wg, nhyp);
    const int mask = bsz - 1;
    float2* prow = pbuf + ((int64_t)z * nhyp + h0) * pitch + m;
#pragma unroll 4
    for (int h = h0; h < h1; ++h, prow += pitch) {
        float4 hv;
        if (MODE == 2) {
            hv = *reinterpret_cast<const float4*>(hc + (int64_t)h * bsz + m);
        } else {
            const int t = h / nfreq;
            const int f = h - t * nfreq;
            const int i0 = (m - shifts[f]) & mask;
            const float2* hrow = hc + (int64_t)t * bsz;
            if (MODE == 0) {
                hv = *reinterpret_cast<const float4*>(hrow + i0);
            } else {
                const float2 a = hrow[i0];
                const float2 b = hrow[(i0 + 1) & mask];
                hv = make_float4(a.x, a.y, b.x, b.y);
            }
        }
        *reinterpret_cast<float4*>(prow) = cmul2(x, hv);
    }
}

// ----------------------------------------------------------------------------------------
// |.|^2 + normalise + transpose + per-delay argmax + partial global peak, on the IFFT output
// (the fused counterpart of complex_magn.cu:8-19 + argmax.cu:93-153 + the QF^2 division of
// xcorrRoutines.py:528-529).
//   in : P[z][t*F+f][s_local]  complex64, hypothesis-major, row pitch `pitch`
//   out: surface[t][s][f]      float32, delay-major (the reference's CAF layout)
//        row_max / row_arg [t][s], and one (value, delay, f) record per workgroup.
// A workgroup owns MAG_S consecutive delays of one (rx block, template) and walks the F axis in
// chunks of MAG_F through an LDS tile: reads are MAG_S*8 B contiguous per hypothesis row,
// writes are MAG_F*4 B contiguous per delay row.
// ----------------------------------------------------------------------------------------
struct Best {
    float v;
    int32_t i;
};

__device__ __forceinline__ Best wave_best(Best b) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(b.v, o, 64);
        const int32_t oi = __shfl_xor(b.i, o, 64);
        if (ov > b.v || (ov == b.v && oi < b.i)) {
            b.v = ov;
            b.i = oi;
        }
    }
    return b;
}

__global__ __launch_bounds__(MAG_THREADS) void k_magsq_norm_argmax(
    const float2* __restrict__ pbuf, int32_t pitch, int32_t ntmpl, int32_t nfreq, const float* __restrict__ tscale,
    const float* __restrict__ inv_e, int64_t num_shifts, int64_t shift_start, int32_t step, int32_t blk0,
    int32_t tiles_per_blk, float* __restrict__ surface, float* __restrict__ row_max, int32_t* __restrict__ row_arg,
    PeakRec* __restrict__ partial, int64_t partial_per_tmpl) {
    __shared__ float s_tile[MAG_S][MAG_F + 1];
    __shared__ float s_rowv[MAG_S];
    __shared__ int32_t s_rowi[MAG_S];

    const int z = blockIdx.z, t = blockIdx.y;
    const int blk = blk0 + z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl0 = blockIdx.x * MAG_S;                     // first local delay of this tile
    const int64_t rel0 = (int64_t)blk * step + sl0;         // index relative to shift_start
    int64_t nv = num_shifts - (int64_t)blk * step;          // valid delays in this rx block
    if (nv > step) nv = step;
    const int64_t pidx = ((int64_t)blk * tiles_per_blk + blockIdx.x);
    if (sl0 >= nv) {  // whole tile past the end (last, partial rx block): neutral record
        if (threadIdx.x == 0 && partial) {
            PeakRec r;
            r.v = -1.f;
            r.delay = 0x7fffffff;
            r.f = 0;
            partial[(int64_t)t * partial_per_tmpl + pidx] = r;
        }
        return;
    }
    const int nrows = (int)min((int64_t)MAG_S, nv - sl0);
    const bool lane_ok = lane < nrows;
    const float ie = lane_ok ? inv_e[rel0 + lane] : 0.f;
    const float ts = tscale[t];
    const float2* pin = pbuf + ((int64_t)z * ntmpl + t) * nfreq * (int64_t)pitch + sl0 + lane;

    constexpr int ROWS_PER_WAVE = MAG_S / (MAG_THREADS / 64);
    float bv[ROWS_PER_WAVE];
    int32_t bi[ROWS_PER_WAVE];
#pragma unroll
    for (int r = 0; r < ROWS_PER_WAVE; ++r) {
        bv[r] = -1.f;
        bi[r] = 0;
    }

    // A wave owns the hypothesis rows fl = wave + 4 k of a chunk: all MAG_F/4 loads of a chunk are issued
    // before any is used, and the next chunk's loads are issued before the store phase of the current
    // one, so HBM reads stay in flight while rows are written.  Each lane keeps a running best over the
    // columns it visits; the cross-lane reduction happens once per row after the last chunk.
    constexpr int LOADS = MAG_F / (MAG_THREADS / 64);
    float2 q[LOADS];
    auto load_chunk = [&](int f0) {
        const int nf = min(MAG_F, nfreq - f0);
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const int fl = wave + k * (MAG_THREADS / 64);
            q[k] = (lane_ok && fl < nf) ? pin[(int64_t)(f0 + fl) * pitch] : make_float2(0.f, 0.f);
        }
    };
    load_chunk(0);
    const float gsc = ts * ie;
    for (int f0 = 0; f0 < nfreq; f0 += MAG_F) {
        const int nf = min(MAG_F, nfreq - f0);
        // phase 1: hypothesis rows -> LDS tile (transposed)
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const int fl = wave + k * (MAG_THREADS / 64);
            s_tile[lane][fl] = lane_ok ? (q[k].x * q[k].x + q[k].y * q[k].y) * gsc : -1.f;
        }
        __syncthreads();
        if (f0 + MAG_F < nfreq) load_chunk(f0 + MAG_F);
        // phase 2: delay rows -> surface, per-lane running argmax
#pragma unroll
        for (int r = 0; r < ROWS_PER_WAVE; ++r) {
            const int row = wave + r * (MAG_THREADS / 64);
            if (row < nrows) {
                float* srow = surface ? surface + ((int64_t)t * num_shifts + rel0 + row) * nfreq + f0 : nullptr;
#pragma unroll
                for (int c = 0; c < MAG_F / 64; ++c) {
                    const int fl = lane + 64 * c;
                    if (fl < nf) {
                        const float v = s_tile[row][fl];
                        if (srow) srow[fl] = v;
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
#pragma unroll
    for (int r = 0; r < ROWS_PER_WAVE; ++r) {
        Best b;
        b.v = bv[r];
        b.i = bi[r];
        b = wave_best(b);
        bv[r] = b.v;
        bi[r] = b.i;
    }
    // per-row results -> LDS -> coalesced stores; tile best -> partial record
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < ROWS_PER_WAVE; ++r) {
            const int row = wave + r * (MAG_THREADS / 64);
            s_rowv[row] = bv[r];
            s_rowi[row] = bi[r];
        }
    }
    __syncthreads();
    if (wave == 0) {
        float v = -1.f;
        int32_t fi = 0;
        if (lane_ok) {
            v = s_rowv[lane];
            fi = s_rowi[lane];
            const int64_t o = (int64_t)t * num_shifts + rel0 + lane;
            if (row_max) row_max[o] = v < 0.f ? __builtin_nanf("") : v;  // (nothing beat the initial value: a zero-energy window, all NaN)
            if (row_arg) row_arg[o] = fi;
        }
        if (partial) {
            // best over the tile: highest value, then lowest delay
            Best b;
            b.v = v;
            b.i = lane;
            b = wave_best(b);
            if (lane == 0) {
                PeakRec r;
                r.v = b.v;
                r.delay = (int32_t)(shift_start + rel0 + b.i);
                r.f = s_rowi[b.i];
                partial[(int64_t)t * partial_per_tmpl + pidx] = r;
            }
        }
    }
}

// Reduce the per-tile records (highest value, then lowest delay).  Grid (templates, parts): workgroup (t, y)
// reduces slice y of template t's records; with one part it writes the template's result, with several it
// writes one record per part to `scratch` for a second, single-part launch (a single workgroup walking
// 2.6e5 records took 0.14 ms at config C2).
__global__ __launch_bounds__(1024) void k_peak_reduce(const PeakRec* __restrict__ partial, int64_t per_tmpl,
                                                      int64_t stride, PeakRec* __restrict__ scratch,
                                                      float* __restrict__ peak_val, int32_t* __restrict__ peak_delay,
                                                      int32_t* __restrict__ peak_freq) {
    __shared__ PeakRec s_w[16];
    const int t = blockIdx.x;
    const int64_t slice = (per_tmpl + gridDim.y - 1) / gridDim.y;
    const int64_t i0 = (int64_t)blockIdx.y * slice, i1 = min(per_tmpl, i0 + slice);
    const PeakRec* p = partial + (int64_t)t * stride;
    PeakRec b;
    b.v = -2.f;
    b.delay = 0x7fffffff;
    b.f = 0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 1024) {
        const PeakRec r = p[i];
        if (r.v > b.v || (r.v == b.v && r.delay < b.delay)) b = r;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        PeakRec r;
        r.v = __shfl_xor(b.v, o, 64);
        r.delay = __shfl_xor(b.delay, o, 64);
        r.f = __shfl_xor(b.f, o, 64);
        if (r.v > b.v || (r.v == b.v && r.delay < b.delay)) b = r;
    }
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            const PeakRec r = s_w[w];
            if (r.v > b.v || (r.v == b.v && r.delay < b.delay)) b = r;
        }
        if (scratch) {
            scratch[(int64_t)t * gridDim.y + blockIdx.y] = b;
        } else {
            if (peak_val) peak_val[t] = b.v;
            if (peak_delay) peak_delay[t] = b.delay;
            if (peak_freq) peak_freq[t] = b.f;
        }
    }
}

// Peak records from finished per-delay rows (T, S) float32 (the no-frequency-scan mode of the persistent engine writes
// the rows directly): workgroup (chunk, t) scans RP_CHUNK delays of row t -- highest value, lowest delay on ties, NaN
// never -- and leaves one record; k_peak_reduce finishes.
constexpr int RP_CHUNK = 16384;
__global__ __launch_bounds__(256) void k_rows_peak(const float* __restrict__ rows, int64_t num_shifts, int64_t shift_start,
                                                   PeakRec* __restrict__ partial, int64_t partial_per_tmpl) {
    __shared__ PeakRec s_w[4];
    const int t = blockIdx.y;
    const int64_t i0 = (int64_t)blockIdx.x * RP_CHUNK, i1 = min(num_shifts, i0 + RP_CHUNK);
    const float* r = rows + (int64_t)t * num_shifts;
    PeakRec b;
    b.v = -1.f;
    b.delay = 0x7fffffff;
    b.f = 0;
    typedef float rp_v4 __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte loads at 4-byte alignment
    int64_t i = i0 + 4 * (int64_t)threadIdx.x;
    for (; i + 3 < i1; i += 1024) {
        const rp_v4 q = __builtin_nontemporal_load(reinterpret_cast<const rp_v4*>(r + i));  // read once
        const float x[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (x[k] > b.v || (x[k] == b.v && (int32_t)(shift_start + i + k) < b.delay)) {
                b.v = x[k];
                b.delay = (int32_t)(shift_start + i + k);
            }
    }
    for (; i < i1; ++i) {  // (at most three values, of one thread)
        const float v = r[i];
        if (v > b.v || (v == b.v && (int32_t)(shift_start + i) < b.delay)) {
            b.v = v;
            b.delay = (int32_t)(shift_start + i);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(b.v, o, 64);
        const int32_t od = __shfl_xor(b.delay, o, 64);
        if (ov > b.v || (ov == b.v && od < b.delay)) {
            b.v = ov;
            b.delay = od;
        }
    }
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (s_w[w].v > b.v || (s_w[w].v == b.v && s_w[w].delay < b.delay)) b = s_w[w];
        partial[(int64_t)t * partial_per_tmpl + blockIdx.x] = b;
    }
}

// ----------------------------------------------------------------------------------------
// Launch wrappers (host)
// ----------------------------------------------------------------------------------------
int64_t rows_peak_chunks(int64_t num_shifts) { return (num_shifts + RP_CHUNK - 1) / RP_CHUNK; }
void launch_rows_peak(const float* rows, int32_t ntmpl, int64_t num_shifts, int64_t shift_start, PeakRec* partial,
                      int64_t partial_per_tmpl, hipStream_t st) {
    hipLaunchKernelGGL(k_rows_peak, dim3((unsigned)rows_peak_chunks(num_shifts), (unsigned)ntmpl), dim3(256), 0, st, rows,
                       num_shifts, shift_start, partial, partial_per_tmpl);
}

// doubles of scratch the prefix needs (one total per tile of the smaller tile size, + the scan's own total)
int64_t prefix_num_tiles(int64_t m) { return (m + 1 + PF_NT * PF_PER - 1) / (PF_NT * PF_PER) + 1; }

template <int SUB>
static void launch_prefix_tiles(const float2* rx, int64_t m, double* tile_sums, double* prefix, hipStream_t st) {
    const int64_t nt = (m + 1 + PF_NT * PF_PER * SUB - 1) / (PF_NT * PF_PER * SUB);
    const int scanned = nt > PFX_DIRECT_TILES;
    const dim3 g((unsigned)nt), b(PF_NT);
    if ((reinterpret_cast<uintptr_t>(rx) & 15) == 0) {
        hipLaunchKernelGGL((k_prefix_tiles<SUB, true, false>), g, b, 0, st, rx, m, tile_sums, 0, prefix);
        if (scanned) hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, st, tile_sums, nt);
        hipLaunchKernelGGL((k_prefix_tiles<SUB, true, true>), g, b, 0, st, rx, m, tile_sums, scanned, prefix);
    } else {
        hipLaunchKernelGGL((k_prefix_tiles<SUB, false, false>), g, b, 0, st, rx, m, tile_sums, 0, prefix);
        if (scanned) hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, st, tile_sums, nt);
        hipLaunchKernelGGL((k_prefix_tiles<SUB, false, true>), g, b, 0, st, rx, m, tile_sums, scanned, prefix);
    }
}

void launch_energy_prefix(const float2* rx, int64_t m, double* tile_sums, double* prefix, hipStream_t st) {
    launch_prefix_tiles<1>(rx, m, tile_sums, prefix, st);
}

void scan_tiles(double* tile_sums, int64_t ntiles, hipStream_t st) {
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, st, tile_sums, ntiles);
}

void launch_inv_energy(const float2* rx, int64_t rx_len, const double* prefix, int64_t shift_start, int64_t num_shifts,
                       const int32_t* gstart, const int32_t* glen, int32_t ngroups, float* inv_e, hipStream_t st) {
    const unsigned g = (unsigned)((num_shifts + 255) / 256);
    hipLaunchKernelGGL(k_inv_energy, dim3(g), dim3(256), 0, st, rx, rx_len, prefix, shift_start, num_shifts, gstart, glen,
                       ngroups, inv_e);
}

void launch_gather_blocks(const float2* rx, int64_t rx_len, int64_t src0, int32_t step, int32_t bsz, int32_t nblk,
                          float2* xb, hipStream_t st) {
    int lb = 0;
    while ((1 << lb) < bsz) ++lb;
    const int64_t total = (int64_t)nblk * bsz;
    hipLaunchKernelGGL(k_gather_blocks, dim3((unsigned)((total / 2 + 255) / 256)), dim3(256), 0, st, rx, rx_len, src0, step,
                       lb, total, xb);
}

// Explicit-frequency hypotheses in the time domain (plan creation, CAF_FREQ_NORM): row (t, f) of the B-point
// spectrum table gets u_t[n] * exp(+j 2 pi nu_f n) for n < N and zeros up to B, u = tmpl or conj(tmpl).
// The phase is reduced in cycles in float64 before the trig call (full accuracy for large nu * n), the product
// is formed in float64 and rounded once -- what the host loop this replaces did, T*F*N sincos calls faster.
// (npart > 1: row (t, f, q) holds the samples q * plen .. q * plen + plen - 1 of the modulated template at its start -- the
//  partitions of a template longer than what one block correlates; the phase runs over the sample's place in the template)
__global__ __launch_bounds__(256) void k_build_hyp_time(const float2* __restrict__ tm, const double* __restrict__ nu,
                                                        int32_t n_tmpl, int32_t bsz, int32_t nfreq, int32_t conj_u,
                                                        float2* __restrict__ hc, int32_t npart, int32_t plen) {
    const int nbx = (bsz + 255) / 256;
    const int q = blockIdx.x / nbx;
    const int m = (blockIdx.x - q * nbx) * 256 + threadIdx.x;  // place in the row
    if (m >= bsz) return;
    const int f = blockIdx.y, t = blockIdx.z;
    const int64_t n = (int64_t)q * plen + m;  // place in the template
    float2 o = make_float2(0.f, 0.f);
    if (m < plen && n < n_tmpl) {
        const float2 u = tm[(int64_t)t * n_tmpl + n];
        const double ur = u.x, ui = conj_u ? -(double)u.y : (double)u.y;
        double cyc = nu[f] * (double)n;
        cyc -= floor(cyc);
        double s, c;
        sincos(2.0 * M_PI * cyc, &s, &c);
        o = make_float2((float)(ur * c - ui * s), (float)(ur * s + ui * c));
    }
    hc[(((int64_t)t * nfreq + f) * npart + q) * bsz + m] = o;
}

void launch_build_hyp_time(const float2* tm, const double* nu, int32_t n_tmpl, int32_t bsz, int32_t nfreq, int32_t ntmpl,
                           int32_t conj_u, float2* hc, hipStream_t st, int32_t npart, int32_t plen) {
    if (npart <= 1) {
        npart = 1;
        plen = bsz;
    }
    for (int f0 = 0; f0 < nfreq; f0 += 65535) {  // grid.y limit
        const int nf = std::min(65535, nfreq - f0);
        hipLaunchKernelGGL(k_build_hyp_time, dim3((unsigned)(npart * ((bsz + 255) / 256)), (unsigned)nf, (unsigned)ntmpl), dim3(256),
                           0, st, tm, nu + f0, n_tmpl, bsz, nfreq, conj_u, hc + (int64_t)f0 * npart * bsz, npart, plen);
    }
}

void launch_conj_scale(float2* h, int64_t n, float scale, hipStream_t st) {
    hipLaunchKernelGGL(k_conj_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h, n, scale);
}

void launch_spectral_mul(int mode, const float2* xb, const float2* hc, const int32_t* shifts, int32_t bsz,
                         int32_t pitch, int32_t nfreq, int32_t nhyp, int32_t hyp_per_wg, int32_t nblk, float2* pbuf,
                         hipStream_t st) {
    const dim3 grid((bsz / 2 + MUL_THREADS - 1) / MUL_THREADS, (nhyp + hyp_per_wg - 1) / hyp_per_wg, nblk);
    if (mode == 0)
        hipLaunchKernelGGL(k_spectral_mul<0>, grid, dim3(MUL_THREADS), 0, st, xb, hc, shifts, bsz, pitch, nfreq, nhyp,
                           hyp_per_wg, pbuf);
    else if (mode == 1)
        hipLaunchKernelGGL(k_spectral_mul<1>, grid, dim3(MUL_THREADS), 0, st, xb, hc, shifts, bsz, pitch, nfreq, nhyp,
                           hyp_per_wg, pbuf);
    else
        hipLaunchKernelGGL(k_spectral_mul<2>, grid, dim3(MUL_THREADS), 0, st, xb, hc, shifts, bsz, pitch, nfreq, nhyp,
                           hyp_per_wg, pbuf);
}

void launch_magsq(const float2* pbuf, int32_t pitch, int32_t ntmpl, int32_t nfreq, const float* tscale,
                  const float* inv_e, int64_t num_shifts, int64_t shift_start, int32_t step, int32_t blk0,
                  int32_t nblk, int32_t tiles_per_blk, float* surface, float* row_max, int32_t* row_arg,
                  PeakRec* partial, int64_t partial_per_tmpl, hipStream_t st) {
    hipLaunchKernelGGL(k_magsq_norm_argmax, dim3(tiles_per_blk, ntmpl, nblk), dim3(MAG_THREADS), 0, st, pbuf, pitch,
                       ntmpl, nfreq, tscale, inv_e, num_shifts, shift_start, step, blk0, tiles_per_blk, surface,
                       row_max, row_arg, partial, partial_per_tmpl);
}

void launch_peak_reduce(const PeakRec* partial, int64_t count, int64_t stride, int32_t ntmpl, PeakRec* scratch,
                        float* pv, int32_t* pd, int32_t* pf, hipStream_t st) {
    // scratch: PEAK_PARTS records per template (may be NULL: single stage)
    if (scratch && count > 16384) {
        hipLaunchKernelGGL(k_peak_reduce, dim3(ntmpl, PEAK_PARTS), dim3(1024), 0, st, partial, count, stride, scratch,
                           nullptr, nullptr, nullptr);
        hipLaunchKernelGGL(k_peak_reduce, dim3(ntmpl, 1), dim3(1024), 0, st, (const PeakRec*)scratch, (int64_t)PEAK_PARTS,
                           (int64_t)PEAK_PARTS, nullptr, pv, pd, pf);
    } else {
        hipLaunchKernelGGL(k_peak_reduce, dim3(ntmpl, 1), dim3(1024), 0, st, partial, count, stride, nullptr, pv, pd, pf);
    }
}

}  // namespace caf
