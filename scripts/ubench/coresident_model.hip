// Can the HBM-bound tile role share a CU with the ALU/LDS-bound FFT role?  Today k_caf_persistent gives each role CUs
// of its own (160 + 96): an FFT workgroup owns all 512 VGPRs per SIMD lane and 148 of the 160 KB of LDS, so nothing can
// be co-resident.  This model prices the alternative BEFORE the real kernels are written:
//   k_fft  : the FFT-role model of fft_struct_model.hip (C0 / C1 instruction mix) with the block spectrum STREAMED per
//            transform instead of register-resident, compiled to <= 104 VGPRs (4 waves per SIMD leave 96 registers free);
//   k_tile : a real |y|^2-tile -> surface transposer that needs NO LDS and <= 96 VGPRs: 16-byte loads of four delays per
//            hypothesis, 4 x 4 transposes across lanes (v_permlane32_swap / v_permlane16_swap), 16-byte stores of four
//            hypotheses per delay (eight consecutive stores of a lane complete a 128-byte surface segment), per-delay
//            running maximum in the lane that owns the delay.
// Each is timed alone and then side by side on two streams (k_fft launched first: one 1024-thread workgroup per CU,
// then 256 four-wave k_tile workgroups that only fit one per CU beside it).
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 coresident_model.hip -o coresident_model
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define AS1 __attribute__((address_space(1)))
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 ld2(const float2* base, uint32_t elem) {
    const uint64_t u = *reinterpret_cast<const AS1 uint64_t*>((const AS1 char*)base + (elem << 3));
    float2 r;
    __builtin_memcpy(&r, &u, 8);
    return r;
}
template <int N, int NR, int NC>
__device__ __forceinline__ void valu(float (&r)[NR], const float (&cf)[NC]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int d = (i * 7) % NR, a = (i * 7 + 3) % NR;
        r[d] = __builtin_fmaf(r[a], cf[i % NC], r[d]);
    }
}
template <int NP>
__device__ __forceinline__ void wr_c64(float2* s, int pitch, int pos, const float (&r)[2 * NP]) {
#pragma unroll
    for (int j = 0; j < NP; ++j) s[j * pitch + pos] = make_float2(r[2 * j], r[2 * j + 1]);
}
template <int NP>
__device__ __forceinline__ void rd_c64(const float2* s, int pitch, int pos, float (&r)[2 * NP]) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const float2 v = s[j * pitch + pos];
        r[2 * j] = v.x;
        r[2 * j + 1] = v.y;
    }
}
template <int OFF0>
__device__ __forceinline__ void wr_plane16(uint32_t m0, const float* r) {
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
}
template <int NP>
__device__ __forceinline__ void rd_plane(const float* s, int pitch4, int pos4, float* r) {
#pragma unroll
    for (int j = 0; j < NP / 4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(s + (j * pitch4 + pos4) * 4);
        r[4 * j] = v.x, r[4 * j + 1] = v.y, r[4 * j + 2] = v.z, r[4 * j + 3] = v.w;
    }
}
template <int NT>
__device__ __forceinline__ void rd_tw(const float2* tw, int lane, float (&cf)[2 * NT]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const float2 v = tw[j * 64 + lane];
        cf[2 * j] = v.x, cf[2 * j + 1] = v.y;
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
    const float2* xb;
    const float2* hc;
    float* vt;
    int nt;
};

// FFT-role model: XRES = block spectrum resident in registers (the shipped form) or streamed per transform
template <bool PLANAR, bool XRES>
__device__ __forceinline__ void fft_model(Args a) {
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
        if (XRES) xr[2 * i] = x.x, xr[2 * i + 1] = x.y;
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
        valu<296, 32, 2>(v, cw);
        __syncthreads();
        auto put = [&](int pitch, int pos) {
            if (PLANAR) {
                float re[16], im[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) re[i] = v[2 * i], im[i] = v[2 * i + 1];
                wr_plane16<0>(m0, re);
                wr_plane16<4092>(m0 + 61444u, im);
            } else {
                wr_c64<16>(s_d, pitch, pos, v);
            }
        };
        put(1090, wave * 68 + lane);
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            float cf[30];
            if (PLANAR) {
                rd_plane<16>((const float*)s_d, 1024, tid + lz, v);
                rd_plane<16>((const float*)s_d + 16384, 1024, tid + lz, v + 16);
            } else {
                rd_c64<16>(s_d, 68, wave * 1090 + lane + lz, v);
            }
            rd_tw<15>(s_tw, lane + lz, cf);
            valu<236, 32, 30>(v, cf);
            put(68, wave * 1090 + lane + lz);
            if (pass == 0) __builtin_amdgcn_wave_barrier();
        }
        float hn[32];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float2 h = ld2(a.hc, ((1024 * i + tid - 4 * t) & 16383) + lz);
            hn[2 * i] = h.x, hn[2 * i + 1] = h.y;
        }
        __syncthreads();
        if (!XRES) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float2 x = ld2(a.xb, 1024 * i + tid + lz);
                xr[2 * i] = x.x, xr[2 * i + 1] = x.y;
            }
        }
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
__global__ __launch_bounds__(1024) void k_fft_res(Args a) { fft_model<false, true>(a); }
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(104))) void k_fft_str(Args a) { fft_model<false, false>(a); }
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(104))) void k_fft_str_planar(Args a) { fft_model<true, false>(a); }

// ---------------------------------------------------------------------------------------------------------------
// |y|^2 tiles [tile][256 hypotheses][64 delays] -> surface [tile*64 + delay][256] + per-delay maximum / argument.
// One tile per wave and turn; lane = (fq = lane >> 4, dl = lane & 15): loads hypothesis 4i + fq, delays 4 dl .. 4 dl + 3;
// after the 4 x 4 transposes it holds delay 4 dl + fq, hypotheses 4 i .. 4 i + 3.
__device__ __forceinline__ void xpose4(v4f& q) {
    // in place on the four registers of q (the builtins return fresh pairs and cost a copy per operand)
    float x = q.x, y = q.y, z = q.z, w = q.w;
    // one block with explicit wait states: a VGPR written by a VALU instruction (a swap included) must not be read by a
    // permlane swap within two wait states, and the compiler does not look inside inline assembly
    asm volatile(
        "s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
        : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
    q.x = x, q.y = y, q.z = z, q.w = w;
}
struct TArgs {
    const float* vt;   // [ntiles][256][64]
    float* surf;       // [ntiles*64][256]
    float* rmax;       // [ntiles*64]
    int* rarg;
    int ntiles;
    int check;         // 1: plain stores (so that the host can verify), 0: nt
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_of(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
template <typename Tp>
__device__ __forceinline__ Tp* uniform_ptr(Tp* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<Tp*>(((uint64_t)hi << 32) | lo);
}
typedef int v4i __attribute__((ext_vector_type(4)));
template <bool CHECK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_tile(TArgs a) {
    const int lane = threadIdx.x & 63, fq = lane >> 4, dl = lane & 15;
    const int wave_g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), nwaves = gridDim.x * 4;
    const int vin = (fq * 64 + 4 * dl) * 4, vout = (4 * dl + fq) * 1024;
    const float g = 1.0f + 0.0009765625f * (float)(4 * dl + fq);
#pragma unroll 1
    for (int tile = wave_g; tile < a.ntiles; tile += nwaves) {
        const __amdgpu_buffer_rsrc_t rin = buf_of(uniform_ptr(a.vt + (size_t)tile * 16384), 65536u);
        const __amdgpu_buffer_rsrc_t rout = buf_of(uniform_ptr(a.surf + (size_t)tile * 16384), 65536u);
        v4f qa[8], qb[8];
        auto load = [&](v4f(&q)[8], int s) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                q[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rin, vin, (32 * s + 4 * i) * 256, 2));
        };
        float bv = -1.f;
        int bi = 0;
        auto work = [&](v4f(&q)[8], int s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xpose4(q[i]);
                q[i] *= g;
                const int h0 = 32 * s + 4 * i;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool up = q[i][k] > bv;  // increasing hypothesis order: the first maximum stays
                    bv = up ? q[i][k] : bv;
                    bi = up ? h0 + k : bi;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, q[i]), rout, vout, h0 * 4, CHECK ? 0 : 2);  // nt
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        load(qa, 0);
        load(qb, 1);
        // (steps past the end read beyond the descriptor: zeros, no memory traffic -- the loop body stays straight-line)
#pragma unroll 1
        for (int s = 0; s < 8; s += 2) {
            work(qa, s);
            load(qa, s + 2);
            __builtin_amdgcn_sched_barrier(0);
            work(qb, s + 1);
            load(qb, s + 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        a.rmax[(size_t)tile * 64 + 4 * dl + fq] = bv;
        a.rarg[(size_t)tile * 64 + 4 * dl + fq] = bi;
    }
}

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

int main(int argc, char** argv) {
    const int nt = argc > 1 ? atoi(argv[1]) : 1024;        // transforms per FFT workgroup
    const int ntiles = argc > 2 ? atoi(argv[2]) : 131072;  // 64 KB each: 8 GB in, 8 GB out
    Args a;
    float2 *xb, *hc;
    float* vt;
    CK(hipMalloc(&xb, 16384 * 8));
    CK(hipMalloc(&hc, 16384 * 8));
    CK(hipMalloc(&vt, (size_t)256 * 64 * 16384 * 4));
    std::vector<float2> h(16384);
    for (int i = 0; i < 16384; ++i) h[i] = make_float2(0.5f + 1e-5f * i, 0.25f - 1e-5f * i);
    CK(hipMemcpy(xb, h.data(), 16384 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(hc, h.data(), 16384 * 8, hipMemcpyHostToDevice));
    a.xb = xb, a.hc = hc, a.vt = vt, a.nt = nt;
    TArgs ta;
    float *tin, *surf, *rmax;
    int* rarg;
    CK(hipMalloc(&tin, (size_t)ntiles * 65536));
    CK(hipMalloc(&surf, (size_t)ntiles * 65536));
    CK(hipMalloc(&rmax, (size_t)ntiles * 256));
    CK(hipMalloc(&rarg, (size_t)ntiles * 256));
    ta.vt = tin, ta.surf = surf, ta.rmax = rmax, ta.rarg = rarg, ta.ntiles = ntiles, ta.check = 0;
    // correctness of the register transposer on two tiles of known content
    {
        std::vector<float> t2(2 * 16384);
        for (int i = 0; i < 2 * 16384; ++i) t2[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
        CK(hipMemcpy(tin, t2.data(), t2.size() * 4, hipMemcpyHostToDevice));
        TArgs tc = ta;
        tc.ntiles = 2, tc.check = 1;
        hipLaunchKernelGGL(k_tile<true>, dim3(1), dim3(256), 0, 0, tc);
        CK(hipDeviceSynchronize());
        std::vector<float> s2(2 * 16384), rm(128);
        std::vector<int> ra(128);
        CK(hipMemcpy(s2.data(), surf, s2.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(rm.data(), rmax, 128 * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ra.data(), rarg, 128 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int tl = 0; tl < 2; ++tl)
            for (int d = 0; d < 64; ++d) {
                const float g = 1.0f + 0.0009765625f * (float)d;
                float bv = -1.f;
                int bi = 0;
                for (int hh = 0; hh < 256; ++hh) {
                    const float v = t2[tl * 16384 + hh * 64 + d] * g;
                    if (s2[(tl * 64 + d) * 256 + hh] != v) {
                        if (bad < 12) printf("  tile %d delay %d hyp %d: got %g want %g\n", tl, d, hh, s2[(tl * 64 + d) * 256 + hh], v);
                        ++bad;
                    }
                    if (v > bv) bv = v, bi = hh;
                }
                if (rm[tl * 64 + d] != bv || ra[tl * 64 + d] != bi) {
                    if (bad < 24) printf("  tile %d delay %d: max %g arg %d want %g %d\n", tl, d, rm[tl * 64 + d], ra[tl * 64 + d], bv, bi);
                    ++bad;
                }
            }
        printf("register transposer: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
        if (bad) return 1;
    }
    CK(hipMemset(tin, 0, (size_t)ntiles * 65536));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    hipEvent_t e[4];
    for (auto& ev : e) CK(hipEventCreate(&ev));
    auto ms = [&](hipEvent_t x, hipEvent_t y) {
        float v = 0.f;
        (void)hipEventElapsedTime(&v, x, y);
        return v;
    };
    const double tile_bytes = (double)ntiles * 131072.0;
    for (int rep = 0; rep < 2; ++rep) {
        printf("-- repetition %d\n", rep);
        void (*ffts[3])(Args) = {k_fft_res, k_fft_str, k_fft_str_planar};
        const char* names[3] = {"resident X, 128 VGPRs", "streamed X, 104 VGPRs", "streamed X, planar exchange, 104 VGPRs"};
        float alone[3];
        for (int v = 0; v < 3; ++v) {
            CK(hipEventRecord(e[0], s1));
            hipLaunchKernelGGL(ffts[v], dim3(256), dim3(1024), 0, s1, a);
            CK(hipEventRecord(e[1], s1));
            CK(hipEventSynchronize(e[1]));
            alone[v] = ms(e[0], e[1]);
            printf("FFT model alone (%s): %.3f ms = %.2f us per transform\n", names[v], alone[v], alone[v] * 1e3 / nt);
        }
        for (int wgs : {512, 1024, 2048}) {
            CK(hipEventRecord(e[2], s2));
            hipLaunchKernelGGL(k_tile<true>, dim3(wgs), dim3(256), 0, s2, ta);
            CK(hipEventRecord(e[3], s2));
            CK(hipEventSynchronize(e[3]));
            printf("tile streamer alone, %d workgroups, plain (write-back) stores: %.3f ms = %.2f TB/s (read + write)\n", wgs, ms(e[2], e[3]), tile_bytes / ms(e[2], e[3]) / 1e9);
        }
        {
            CK(hipEventRecord(e[2], s2));
            hipLaunchKernelGGL(k_tile<true>, dim3(256), dim3(256), 0, s2, ta);
            CK(hipEventRecord(e[3], s2));
            CK(hipEventSynchronize(e[3]));
            printf("tile streamer alone, 256 workgroups, plain (write-back) stores: %.3f ms = %.2f TB/s (read + write)\n", ms(e[2], e[3]), tile_bytes / ms(e[2], e[3]) / 1e9);
        }
        for (int wgs : {256, 1024}) {
            CK(hipEventRecord(e[2], s2));
            hipLaunchKernelGGL(k_tile<false>, dim3(wgs), dim3(256), 0, s2, ta);
            CK(hipEventRecord(e[3], s2));
            CK(hipEventSynchronize(e[3]));
            printf("tile streamer alone, %4d workgroups: %.3f ms = %.2f TB/s (read + write)\n", wgs, ms(e[2], e[3]), tile_bytes / ms(e[2], e[3]) / 1e9);
        }
        for (int v = 1; v < 3; ++v) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e[0], s1));
            hipLaunchKernelGGL(ffts[v], dim3(256), dim3(1024), 0, s1, a);
            CK(hipEventRecord(e[1], s1));
            CK(hipEventRecord(e[2], s2));
            hipLaunchKernelGGL(k_tile<true>, dim3(256), dim3(256), 0, s2, ta);
            CK(hipEventRecord(e[3], s2));
            CK(hipEventSynchronize(e[1]));
            CK(hipEventSynchronize(e[3]));
            printf("side by side (%s): FFT %.3f ms = %.2f us per transform (alone %.2f); tiles %.3f ms = %.2f TB/s\n", names[v],
                   ms(e[0], e[1]), ms(e[0], e[1]) * 1e3 / nt, alone[v] * 1e3 / nt, ms(e[2], e[3]), tile_bytes / ms(e[2], e[3]) / 1e9);
        }
        fflush(stdout);
    }
    return 0;
}
