// Structural model of the FFT role: the per-transform instruction mix of a 16384-point in-LDS transform (VALU
// instruction counts, LDS reads / writes, workgroup barriers, template-spectrum row loads, |y|^2 stores) issued in the
// order each candidate structure would issue it, with filler arithmetic instead of butterflies.  Results are
// meaningless; only the time per transform matters.  It prices a restructuring BEFORE the real kernel is written:
//   C0  the shipped structure: 1024 threads x 16 points, 16.16.16.4, complex ds_write_b64 exchange, 3 barriers
//   C0v C0 with unmerged LDS reads (volatile: ds_read_b64 instead of ds_read2_b64 / ds_read2st64_b64)
//   C1  C0 with planar exchanges: ds_write_addtid_b32 rows + ds_read_b128
//   C1w C1 with the first exchange wave-local as well (two workgroup barriers per transform instead of three)
//   A1  512 threads x 32 points (256 VGPRs), 32.32.16, complex exchange, 3 barriers (second pass in place)
//   A2  A1 with planar addtid exchanges (4 barriers)
//   B2  two co-resident 512-thread workgroups (128 VGPRs), 32.32.16, block spectrum streamed, planar exchange one plane
//       at a time through a 64 KB buffer (8 barriers)
//   (a parity-split variant -- two workgroups, one 8192-point DIF half each, block spectrum resident -- does not fit 128
//   VGPRs: 64 for the resident spectrum + 64 for a template-spectrum row in flight, before any butterfly)
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 fft_struct_model.hip -o fft_struct_model
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define AS1 __attribute__((address_space(1)))
__device__ __forceinline__ float2 ld2(const float2* base, uint32_t elem) {
    const uint64_t u = *reinterpret_cast<const AS1 uint64_t*>((const AS1 char*)base + (elem << 3));
    float2 r;
    __builtin_memcpy(&r, &u, 8);
    return r;
}
// N filler instructions (v_fma_f32, three VGPR operands) over the NR live registers, coefficients from cf[NC]
template <int N, int NR, int NC>
__device__ __forceinline__ void valu(float (&r)[NR], const float (&cf)[NC]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int d = (i * 7) % NR, a = (i * 7 + 3) % NR;  // every register is read before it is rewritten
        r[d] = __builtin_fmaf(r[a], cf[i % NC], r[d]);
    }
}
template <int NP>
__device__ __forceinline__ void wr_c64(float2* s, int pitch, int pos, const float (&r)[2 * NP]) {
#pragma unroll
    for (int j = 0; j < NP; ++j) s[j * pitch + pos] = make_float2(r[2 * j], r[2 * j + 1]);
}
// VOL: volatile accesses are not merged into ds_read2_b64 / ds_read2st64_b64 (8 LDS cycles per pair against 2 + 2)
template <int NP, bool VOL = false>
__device__ __forceinline__ void rd_c64(const float2* s, int pitch, int pos, float (&r)[2 * NP]) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        uint64_t u;
        if (VOL)
            u = *(const volatile __attribute__((address_space(3))) uint64_t*)(&s[j * pitch + pos]);
        else
            u = *reinterpret_cast<const uint64_t*>(&s[j * pitch + pos]);
        float2 v;
        __builtin_memcpy(&v, &u, 8);
        r[2 * j] = v.x;
        r[2 * j + 1] = v.y;
    }
}
// one plane (NP floats per thread) as rows of 64 floats: address = m0 + j*pitchB + 4*lane
template <int NP, int OFF0>
__device__ __forceinline__ void wr_plane(uint32_t m0, const float* r) {
    static_assert(NP == 16 || NP == 32, "");
    if constexpr (NP == 16) {
        asm volatile(
            "s_mov_b32 m0, %0\n\t"
            "ds_write_addtid_b32 %1 offset:%17\n\tds_write_addtid_b32 %2 offset:%17+4096\n\t"
            "ds_write_addtid_b32 %3 offset:%17+8192\n\tds_write_addtid_b32 %4 offset:%17+12288\n\t"
            "ds_write_addtid_b32 %5 offset:%17+16384\n\tds_write_addtid_b32 %6 offset:%17+20480\n\t"
            "ds_write_addtid_b32 %7 offset:%17+24576\n\tds_write_addtid_b32 %8 offset:%17+28672\n\t"
            "ds_write_addtid_b32 %9 offset:%17+32768\n\tds_write_addtid_b32 %10 offset:%17+36864\n\t"
            "ds_write_addtid_b32 %11 offset:%17+40960\n\tds_write_addtid_b32 %12 offset:%17+45056\n\t"
            "ds_write_addtid_b32 %13 offset:%17+49152\n\tds_write_addtid_b32 %14 offset:%17+53248\n\t"
            "ds_write_addtid_b32 %15 offset:%17+57344\n\tds_write_addtid_b32 %16 offset:%17+61440" ::"s"(m0),
            "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(r[8]), "v"(r[9]),
            "v"(r[10]), "v"(r[11]), "v"(r[12]), "v"(r[13]), "v"(r[14]), "v"(r[15]), "n"(OFF0)
            : "memory");
    } else {
        wr_plane<16, OFF0>(m0, r);
        wr_plane<16, OFF0 + 2048>(m0, r + 16);
    }
}
template <int NP>
__device__ __forceinline__ void rd_plane(const float* s, int pitch4, int pos4, float* r) {
#pragma unroll
    for (int j = 0; j < NP / 4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(s + (j * pitch4 + pos4) * 4);
        r[4 * j] = v.x;
        r[4 * j + 1] = v.y;
        r[4 * j + 2] = v.z;
        r[4 * j + 3] = v.w;
    }
}
template <int NT, bool VOL = false>
__device__ __forceinline__ void rd_tw(const float2* tw, int lane, float (&cf)[2 * NT]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        uint64_t u;
        if (VOL)
            u = *(const volatile __attribute__((address_space(3))) uint64_t*)(&tw[j * 64 + lane]);
        else
            u = *reinterpret_cast<const uint64_t*>(&tw[j * 64 + lane]);
        float2 v;
        __builtin_memcpy(&v, &u, 8);
        cf[2 * j] = v.x;
        cf[2 * j + 1] = v.y;
    }
}
__device__ __forceinline__ void st_nt(float* p, uint32_t byteoff, float v) {
    __builtin_nontemporal_store(v, reinterpret_cast<AS1 float*>((AS1 char*)p + byteoff));
}

typedef __attribute__((address_space(3))) char lds_char;
template <typename T>
__device__ __forceinline__ uint32_t lds_addr(T* p) {
    return (uint32_t)(uintptr_t)(lds_char*)p;
}

struct Args {
    const float2* xb;  // [16384]
    const float2* hc;  // [16384]
    float* vt;         // [wgs][64][16384]
    int nt;
};

// ------------------------------------------------------------------ C0 / C1: 1024 threads x 16 points
template <bool PLANAR, bool VOL, bool WAVE1 = false>
__global__ __launch_bounds__(1024) void k_c0(Args a) {
    __shared__ __attribute__((aligned(16))) float2 s_d[16 * 1090];
    __shared__ float2 s_tw[16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s_tw[tid] = make_float2(0.999f, 0.001f);
    float xr[32], pr[32], cw[2] = {0.9990234f + tid * 1e-9f, 0.0441f};
    float* vt = a.vt + (size_t)blockIdx.x * 64 * 16384;
    const uint32_t m0 = __builtin_amdgcn_readfirstlane(lds_addr(s_d) + (uint32_t)wave * 256u);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float2 x = ld2(a.xb, 1024 * i + tid), h = ld2(a.hc, 1024 * i + tid);
        xr[2 * i] = x.x, xr[2 * i + 1] = x.y;
        pr[2 * i] = x.x * h.x, pr[2 * i + 1] = x.y * h.y;
    }
    __syncthreads();
    for (int t = 0; t < a.nt; ++t) {
        uint32_t hoff = (uint32_t)(t & 63) * 65536u;
        asm volatile("" : "+s"(hoff));
        int lz = 0;
        asm volatile("" : "+v"(lz));
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = pr[i];
        valu<296, 32, 2>(v, cw);  // DFT16 + pass-1 twiddle recurrence
        __syncthreads();
        if (PLANAR) {
            float re[16], im[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) re[i] = v[2 * i], im[i] = v[2 * i + 1];
            wr_plane<16, 0>(m0, re);
            wr_plane<16, 4092>(m0 + 61444u, im);
        } else {
            wr_c64<16>(s_d, 1090, wave * 68 + lane, v);
        }
        // WAVE1: the first exchange wave-local too (16384 = 1024 per wave x 16 across waves: passes 1-3 and both of their
        // exchanges inside one barrier-free stretch, the only workgroup-wide exchange in front of the last pass)
        if (WAVE1)
            __builtin_amdgcn_wave_barrier();
        else
            __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            float cf[30];
            if (PLANAR) {
                rd_plane<16>((const float*)s_d, 1024, tid + lz, v);
                rd_plane<16>((const float*)s_d + 16384, 1024, tid + lz, v + 16);
            } else {
                rd_c64<16, VOL>(s_d, 68, wave * 1090 + lane + lz, v);
            }
            rd_tw<15, VOL>(s_tw, lane + lz, cf);
            valu<236, 32, 30>(v, cf);
            if (PLANAR) {
                float re[16], im[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) re[i] = v[2 * i], im[i] = v[2 * i + 1];
                wr_plane<16, 0>(m0, re);
                wr_plane<16, 4092>(m0 + 61444u, im);
            } else {
                wr_c64<16>(s_d, 68, wave * 1090 + lane + lz, v);
            }
            if (pass == 0) __builtin_amdgcn_wave_barrier();
        }
        float hn[32];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float2 h = ld2(a.hc, ((1024 * i + tid - 4 * t) & 16383) + lz);
            hn[2 * i] = h.x, hn[2 * i + 1] = h.y;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float y[8];
            if (PLANAR) {
                rd_plane<4>((const float*)s_d, 1024, tid + 1024 * i + lz, y);
                rd_plane<4>((const float*)s_d + 16384, 1024, tid + 1024 * i + lz, y + 4);
            } else {
                const float4 lo = *reinterpret_cast<const float4*>(&s_d[(i * 4 + (lane & 3)) * 1090 + wave * 68 + (lane >> 2) * 4 + 2 * lz]);
                const float4 hi = *reinterpret_cast<const float4*>(&s_d[(i * 4 + (lane & 3)) * 1090 + wave * 68 + (lane >> 2) * 4 + 2 + 2 * lz]);
                y[0] = lo.x, y[1] = lo.y, y[2] = lo.z, y[3] = lo.w, y[4] = hi.x, y[5] = hi.y, y[6] = hi.z, y[7] = hi.w;
            }
            valu<16, 8, 2>(y, cw);
#pragma unroll
            for (int k = 0; k < 4; ++k) st_nt(vt, hoff + (uint32_t)((4 * i + k) * 1024 + tid) * 4u, y[2 * k] * y[2 * k] + y[2 * k + 1] * y[2 * k + 1]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            pr[2 * i] = xr[2 * i] * hn[2 * i] - xr[2 * i + 1] * hn[2 * i + 1];
            pr[2 * i + 1] = xr[2 * i] * hn[2 * i + 1] + xr[2 * i + 1] * hn[2 * i];
        }
    }
}

// ------------------------------------------------------------------ A1 / A2: 512 threads x 32 points, 256 VGPRs
template <bool PLANAR>
__global__ __launch_bounds__(512) void k_a(Args a) {
    __shared__ __attribute__((aligned(16))) float2 s_d[16384 + 16 * 66];
    __shared__ float2 s_tw[32 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2048; i += 512) s_tw[i] = make_float2(0.999f, 0.001f);
    float xr[64], pr[64], cw[2] = {0.9990234f + tid * 1e-9f, 0.0441f};
    float* vt = a.vt + (size_t)blockIdx.x * 64 * 16384;
    const uint32_t m0 = __builtin_amdgcn_readfirstlane(lds_addr(s_d) + (uint32_t)wave * 256u);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float2 x = ld2(a.xb, 512 * i + tid), h = ld2(a.hc, 512 * i + tid);
        xr[2 * i] = x.x, xr[2 * i + 1] = x.y;
        pr[2 * i] = x.x * h.x, pr[2 * i + 1] = x.y * h.y;
    }
    __syncthreads();
    for (int t = 0; t < a.nt; ++t) {
        uint32_t hoff = (uint32_t)(t & 63) * 65536u;
        asm volatile("" : "+s"(hoff));
        int lz = 0;
        asm volatile("" : "+v"(lz));
        float v[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = pr[i];
        valu<432 + 248, 64, 2>(v, cw);  // DFT32 + twiddle recurrence
        __syncthreads();
        auto put = [&](int pitch_c64) {
            if (PLANAR) {
                float re[32], im[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) re[i] = v[2 * i], im[i] = v[2 * i + 1];
                wr_plane<32, 0>(m0, re);
                wr_plane<32, 2044>(m0 + 63492u, im);
            } else {
                wr_c64<32>(s_d, pitch_c64, tid + lz, v);
            }
        };
        auto get = [&]() {
            if (PLANAR) {
                rd_plane<32>((const float*)s_d, 512, tid + lz, v);
                rd_plane<32>((const float*)s_d + 16384, 512, tid + lz, v + 32);
            } else {
                rd_c64<32>(s_d, 514, tid + lz, v);
            }
        };
        put(514);
        __syncthreads();
        {
            float cf[62];
            get();
            rd_tw<31>(s_tw, lane + lz, cf);
            valu<432 + 124, 64, 62>(v, cf);
            if (PLANAR) __syncthreads();  // not in place: every wave's reads before anybody's writes
            put(514);
        }
        __syncthreads();
        // last pass: two DFT16 per thread; the next row arrives in two halves under them
        float hn[32];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float2 h = ld2(a.hc, ((512 * (16 * half + i) + tid - 4 * t) & 16383) + lz);
                hn[2 * i] = h.x, hn[2 * i + 1] = h.y;
            }
            float y[32];
            if (PLANAR) {
                rd_plane<16>((const float*)s_d, 512, tid + 4096 * half + lz, y);
                rd_plane<16>((const float*)s_d + 16384, 512, tid + 4096 * half + lz, y + 16);
            } else {
                rd_c64<16>(s_d, 514, tid + 8192 * half + lz, y);
            }
            valu<176, 32, 2>(y, cw);
#pragma unroll
            for (int k = 0; k < 16; ++k) st_nt(vt, hoff + (uint32_t)((16 * half + k) * 512 + tid) * 4u, y[2 * k] * y[2 * k] + y[2 * k + 1] * y[2 * k + 1]);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int o = 32 * half + 2 * i;
                pr[o] = xr[o] * hn[2 * i] - xr[o + 1] * hn[2 * i + 1];
                pr[o + 1] = xr[o] * hn[2 * i + 1] + xr[o + 1] * hn[2 * i];
            }
        }
    }
}

// ------------------------------------------------------------------ B2: two workgroups per CU, everything streamed
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_b2(Args a) {
    __shared__ __attribute__((aligned(16))) float s_p[16384 + 512];
    __shared__ float2 s_tw[16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 1024; i += 512) s_tw[i] = make_float2(0.999f, 0.001f);
    float cw[2] = {0.9990234f + tid * 1e-9f, 0.0441f};
    float* vt = a.vt + (size_t)blockIdx.x * 64 * 16384;
    const uint32_t m0 = __builtin_amdgcn_readfirstlane(lds_addr(s_p) + (uint32_t)wave * 256u);
    __syncthreads();
    for (int t = 0; t < a.nt; ++t) {
        uint32_t hoff = (uint32_t)(t & 63) * 65536u;
        asm volatile("" : "+s"(hoff));
        int lz = 0;
        asm volatile("" : "+v"(lz));
        float re[32], im[32];
        // products from two streamed rows, eight points at a time
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float2 x[8], h[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                x[i] = ld2(a.xb, 512 * (8 * c + i) + tid + lz);
                h[i] = ld2(a.hc, ((512 * (8 * c + i) + tid - 4 * t) & 16383) + lz);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                re[8 * c + i] = x[i].x * h[i].x - x[i].y * h[i].y;
                im[8 * c + i] = x[i].x * h[i].y + x[i].y * h[i].x;
            }
        }
        float cf[32];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            {
                float v[64];
#pragma unroll
                for (int i = 0; i < 32; ++i) v[2 * i] = re[i], v[2 * i + 1] = im[i];
                if (pass == 0)
                    valu<432 + 248, 64, 2>(v, cw);
                else
                    valu<432 + 124, 64, 32>(v, cf);
#pragma unroll
                for (int i = 0; i < 32; ++i) re[i] = v[2 * i], im[i] = v[2 * i + 1];
            }
            __syncthreads();  // the buffer's previous readers are done
            wr_plane<32, 0>(m0, re);
            __syncthreads();
            rd_plane<32>(s_p, 512, tid + lz, re);
            __syncthreads();
            wr_plane<32, 0>(m0, im);
            __syncthreads();
            rd_plane<32>(s_p, 512, tid + lz, im);
            if (pass == 0) rd_tw<16>(s_tw, lane + lz, cf);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float y[32];
#pragma unroll
            for (int i = 0; i < 16; ++i) y[2 * i] = re[16 * half + i], y[2 * i + 1] = im[16 * half + i];
            valu<176, 32, 2>(y, cw);
#pragma unroll
            for (int k = 0; k < 16; ++k) st_nt(vt, hoff + (uint32_t)((16 * half + k) * 512 + tid) * 4u, y[2 * k] * y[2 * k] + y[2 * k + 1] * y[2 * k + 1]);
        }
    }
}

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

template <typename K>
static int run(const char* name, K kern, int threads, int wgs_per_cu, double units_per_wg_transform, Args a, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * wgs_per_cu;
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    // 16384-point transforms finished per CU: nt * wgs_per_cu * units
    const double us = best * 1e3 / (a.nt * wgs_per_cu * units_per_wg_transform);
    printf("%-4s %4d thr x %d wg/CU  %8.3f ms  -> %6.2f us per 16384-point transform and CU\n", name, threads, wgs_per_cu, best, us);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const int nt = argc > 1 ? atoi(argv[1]) : 256, reps = argc > 2 ? atoi(argv[2]) : 3;
    Args a;
    float2 *xb, *hc;
    float* vt;
    CK(hipMalloc(&xb, 16384 * 8));
    CK(hipMalloc(&hc, 16384 * 8));
    CK(hipMalloc(&vt, (size_t)512 * 64 * 16384 * 4));
    std::vector<float2> h(16384);
    for (int i = 0; i < 16384; ++i) h[i] = make_float2(0.5f + 1e-5f * i, 0.25f - 1e-5f * i);
    CK(hipMemcpy(xb, h.data(), 16384 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(hc, h.data(), 16384 * 8, hipMemcpyHostToDevice));
    a.xb = xb, a.hc = hc, a.vt = vt, a.nt = nt;
    for (int rep = 0; rep < 2; ++rep) {  // twice: the clock settles under load
        if (run("C0", k_c0<false, false>, 1024, 1, 1.0, a, reps)) return 1;
        if (run("C0v", k_c0<false, true>, 1024, 1, 1.0, a, reps)) return 1;
        if (run("C1", k_c0<true, false>, 1024, 1, 1.0, a, reps)) return 1;
        if (run("C1w", k_c0<true, false, true>, 1024, 1, 1.0, a, reps)) return 1;
        if (run("A1", k_a<false>, 512, 1, 1.0, a, reps)) return 1;
        if (run("A2", k_a<true>, 512, 1, 1.0, a, reps)) return 1;
        if (run("B2", k_b2, 512, 2, 1.0, a, reps)) return 1;
    }
    return 0;
}
