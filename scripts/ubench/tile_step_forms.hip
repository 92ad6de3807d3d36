// What bounds one CU of the tile role?  The role's step -- 8 x 16-byte tile loads per lane, scale, per-delay running
// maximum, transposition through a 64 x 33 float LDS patch per wave, surface-row stores -- with the row stores in three
// forms (4 / 8 / 16 bytes per lane = 32 / 16 / 8 store instructions per step), with and without the arithmetic, on few
// workgroups (the CU's own limit) and on 256 (HBM's).  Every workgroup streams tiles of its own.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tile_step_forms.hip -o tile_step_forms
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));
constexpr int AUX_NT = 2, AUX_SC1 = 16;  // gfx942+: bit1 = nt, bit4 = sc1
constexpr int TW_H = 32, TW_PITCH = 33, TW_LDS = 64 * TW_PITCH;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_of(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// tiles: [tile][nfreq][64] float (hypothesis-major, 64 delays contiguous); out: [tile][64 delays][nfreq]
// FORM 3: loads + LDS only (no row stores); 4: LDS + row stores only (no tile loads); 5: loads and 4-byte stores without the
// LDS patch (the values a lane loaded, stored where the transposed ones would go: timing only)
template <int FORM, bool MATH>
__global__ __launch_bounds__(1024) void k_tiles(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ rmax,
                                                int tiles_per_wg, int nfreq) {
    __shared__ float lds[16 * TW_LDS];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    float* s_w = lds + wave * TW_LDS;
    const int s4 = 4 * (lane & 15), fq = lane >> 4;
    const int nsteps = nfreq / TW_H;
    for (int it = wave; it < tiles_per_wg; it += 16) {
        const size_t tile = (size_t)blockIdx.x * tiles_per_wg + it;
        const float* vin = in + tile * nfreq * 64;
        float* srow0 = out + tile * 64 * nfreq;
        const __amdgpu_buffer_rsrc_t rin = buf_of(vin, (uint32_t)nfreq * 256u);
        const __amdgpu_buffer_rsrc_t rout = buf_of(srow0, (uint32_t)(64 * nfreq) * 4u);
        float g[4], bv[4];
        int bi[4];
        for (int k = 0; k < 4; ++k) g[k] = 1.0f + 0.125f * k, bv[k] = -1.f, bi[k] = 0;
        v4f_t qa[8], qb[8];
        auto load_step = [&](v4f_t(&q)[8], int s) {
            if (FORM == 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = v4f_t{1.f + s, 2.f, 3.f, 4.f + i};
                return;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                q[i] = __builtin_bit_cast(v4f_t, __builtin_amdgcn_raw_buffer_load_b128(rin, (fq * 64 + s4) * 4, (s * TW_H + 4 * i) * 256, AUX_SC1 | AUX_NT));
        };
        const int half = lane >> 5, col = lane & 31;
        float* swr = s_w + s4 * TW_PITCH + fq;
        auto do_step = [&](v4f_t(&q)[8], int s) {
            const int f0 = s * TW_H;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int hyp = f0 + 4 * i + fq;
                float x[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (MATH) {
                        x[k] *= g[k];
                        const bool up = x[k] > bv[k];
                        bv[k] = up ? x[k] : bv[k];
                        bi[k] = up ? hyp : bi[k];
                    }
                    if (FORM != 5) swr[k * TW_PITCH + 4 * i] = x[k];
                }
            }
            if (FORM == 5) {
                const int voff5 = (half * nfreq + col) * 4;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, q[i].x), rout, voff5, ((8 * i) * nfreq + f0) * 4, AUX_NT);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, q[i].y), rout, voff5, ((8 * i + 2) * nfreq + f0) * 4, AUX_NT);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, q[i].z), rout, voff5, ((8 * i + 4) * nfreq + f0) * 4, AUX_NT);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, q[i].w), rout, voff5, ((8 * i + 6) * nfreq + f0) * 4, AUX_NT);
                }
                load_step(q, s + 2);
                return;
            }
            load_step(q, s + 2);
            __builtin_amdgcn_wave_barrier();
            if (FORM == 3) {
                const float* srd = s_w + half * TW_PITCH + col;
                float acc = 0.f;
#pragma unroll
                for (int r = 0; r < 64; r += 2) acc += srd[r * TW_PITCH];
                if (acc == 12345.678f) rmax[0] = acc;
            } else if (FORM == 0 || FORM == 4) {
                const float* srd = s_w + half * TW_PITCH + col;
                const int voff = (half * nfreq + col) * 4;
#pragma unroll
                for (int r = 0; r < 64; r += 2)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, srd[r * TW_PITCH]), rout, voff, (r * nfreq + f0) * 4, AUX_NT);
            } else if (FORM == 1) {
                // four rows x 128 bytes per instruction: rows 2 j' + {0, 1} and 32 + 2 j' + {0, 1}? (pitch 33: rows r, r + 1 differ by
                // one bank; the pairs of a lane are consecutive floats) -- lane: row = (lane >> 4) -> {0, 1, 32, 33}, col2 = 2 (lane & 15)
                const int rsel = lane >> 4, row = (rsel & 1) + 32 * (rsel >> 1), col2 = 2 * (lane & 15);
                const float* srd = s_w + row * TW_PITCH + col2;
                const int voff = (row * nfreq + col2) * 4;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float* sp = srd + 2 * j * TW_PITCH;
                    const v2f_t o = {sp[0], sp[1]};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i_t, o), rout, voff, (2 * j * nfreq + f0) * 4, AUX_NT);
                }
            } else {
                const int row4 = ((lane >> 3) & 3) + 32 * half, col4 = 4 * (lane & 7);
                const float* srd = s_w + row4 * TW_PITCH + col4;
                const int voff = (row4 * nfreq + col4) * 4;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float* sp = srd + 4 * j * TW_PITCH;
                    const v4f_t o = {sp[0], sp[1], sp[2], sp[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_t, o), rout, voff, (4 * j * nfreq + f0) * 4, AUX_NT);
                }
            }
            __builtin_amdgcn_wave_barrier();
        };
        load_step(qa, 0);
        load_step(qb, 1);
        for (int s = 0; s < nsteps; s += 2) {
            do_step(qa, s);
            do_step(qb, s + 1);
        }
        if (MATH && fq == 0) {
            for (int k = 0; k < 4; ++k) rmax[tile * 64 + s4 + k] = bv[k] + (float)bi[k];
        }
    }
}

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

template <typename K>
static int run(const char* name, K kern, int wgs, int tiles_per_wg, const float* in, float* out, float* rmax, int nfreq, double* sum) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(1024), 0, 0, in, out, rmax, tiles_per_wg, nfreq);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    const double bytes = 2.0 * wgs * (double)tiles_per_wg * nfreq * 64 * 4;
    // spot check of the transposition: out[tile 1][delay d][hyp h] == in[tile 1][h][d] (* scale if MATH; compare ratios)
    std::vector<float> ho(64 * nfreq), hi(64 * nfreq);
    CK(hipMemcpy(ho.data(), out + (size_t)64 * nfreq, ho.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hi.data(), in + (size_t)64 * nfreq, hi.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    double s = 0;
    for (int d = 0; d < 64; ++d)
        for (int h = 0; h < nfreq; ++h) {
            const float a = ho[d * nfreq + h], b = hi[h * 64 + d];
            const float g = 1.0f + 0.125f * (d & 3);
            if (!(a == b || a == b * g)) ++bad;
            s += a;
        }
    *sum = s;
    printf("%-34s %4d workgroups  %8.3f ms  %7.1f GB/s per CU  %6.2f TB/s total  (%d wrong)\n", name, wgs, best, bytes / best / 1e6 / wgs, bytes / best / 1e9, bad);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const int nfreq = 256;
    const int big = argc > 1 ? atoi(argv[1]) : 256;  // workgroups of the HBM-bound run
    const size_t max_tiles = (size_t)256 * 128;      // 2 GB in + 2 GB out
    float *in, *out, *rmax;
    CK(hipMalloc(&in, max_tiles * nfreq * 64 * 4));
    CK(hipMalloc(&out, max_tiles * nfreq * 64 * 4));
    CK(hipMalloc(&rmax, max_tiles * 64 * 4));
    {
        std::vector<float> h((size_t)4 * nfreq * 64);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000003u) * 1e-6f;
        for (size_t t = 0; t < max_tiles; t += 4) CK(hipMemcpy(in + t * nfreq * 64, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    double s;
    for (int rep = 0; rep < 2; ++rep) {
        for (int wgs : {8, 32, big}) {
            const int tpw = wgs <= 32 ? 1024 : 128;
            if ((size_t)wgs * tpw > max_tiles) continue;
            if (run("4 B/lane stores (shipped), math", k_tiles<0, true>, wgs, tpw, in, out, rmax, nfreq, &s)) return 1;
            if (run("4 B/lane stores, no math", k_tiles<0, false>, wgs, tpw, in, out, rmax, nfreq, &s)) return 1;
            if (run("8 B/lane stores, math", k_tiles<1, true>, wgs, tpw, in, out, rmax, nfreq, &s)) return 1;
            if (run("8 B/lane stores, no math", k_tiles<1, false>, wgs, tpw, in, out, rmax, nfreq, &s)) return 1;
            if (run("16 B/lane stores, math", k_tiles<2, true>, wgs, tpw, in, out, rmax, nfreq, &s)) return 1;
            if (run("16 B/lane stores, no math", k_tiles<2, false>, wgs, tpw, in, out, rmax, nfreq, &s)) return 1;
            if (run("loads + LDS, no stores (x2 bytes)", k_tiles<3, false>, wgs, tpw, in, out, rmax, nfreq, &s)) return 1;
            if (run("LDS + stores, no loads (x2 bytes)", k_tiles<4, false>, wgs, tpw, in, out, rmax, nfreq, &s)) return 1;
            if (run("loads + stores, no LDS", k_tiles<5, false>, wgs, tpw, in, out, rmax, nfreq, &s)) return 1;
        }
    }
    return 0;
}
