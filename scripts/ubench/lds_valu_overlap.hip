// Microbenchmark (gfx950): can the LDS exchange traffic of an FFT pass overlap with its butterfly arithmetic?
// One "pass" per wave and iteration = 16 ds_read_b64 + 256 dependent-ish v_fma_f32 + 16 ds_write_b64 (the shape of
// pass 2/3 of fused_item in caf_fused.hip).  Variants differ only in synchronisation / placement:
//   valu      arithmetic only                      lds       reads + writes only
//   bar       both, __syncthreads per pass          nobar     both, wave-private LDS region, no workgroup barrier
//   stagger   nobar + waves 8..15 start half a pass late       prio   nobar + s_setprio by wave
//   pipe      nobar, two register sets software-pipelined inside one wave (reads of set B fly under the math of A)
//   2wg/4wg   bar, 512 / 256 threads per workgroup, 2 / 4 workgroups per CU (same waves per CU)
// Prints ns per pass per CU and the ratio to valu+lds (1.0 = fully serialised, max(valu,lds)/(valu+lds) = ideal).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

enum { M_VALU, M_LDS, M_BAR, M_NOBAR, M_STAGGER, M_PRIO, M_PIPE, M_LDS128, M_LDS32 };
// ACC 0: plain accesses (the compiler merges pairs into ds_read2st64_b64 / ds_write2st64_b64);
// ACC 1: volatile 8-byte accesses (one ds_read_b64 / ds_write_b64 each)
template <int ACC>
__device__ __forceinline__ float2 lld(const float2* p) {
    if (ACC == 0) return *p;
    const unsigned long long u = *(const volatile __attribute__((address_space(3))) unsigned long long*)p;
    float2 r;
    __builtin_memcpy(&r, &u, 8);
    return r;
}
template <int ACC>
__device__ __forceinline__ void lst(float2* p, float2 v) {
    if (ACC == 0) {
        *p = v;
        return;
    }
    unsigned long long u;
    __builtin_memcpy(&u, &v, 8);
    *(volatile __attribute__((address_space(3))) unsigned long long*)p = u;
}


template <int R, int NE = 16>
__device__ __forceinline__ void math(float2 (&v)[NE], float c) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            v[k].x = __builtin_fmaf(v[k].x, c, v[(k + 1) & (NE - 1)].y);
            v[k].y = __builtin_fmaf(v[k].y, c, v[(k + 3) & (NE - 1)].x);
        }
}

template <int MODE, int NT, int ACC = 0>
__global__ __launch_bounds__(NT) void k(float* out, int iters, float c) {
    extern __shared__ float2 lds[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    float2* mine = lds + wave * 1024 + lane;  // wave-private 8 KB region, element k at mine[64 k]: conflict-free
    float2 v[16], u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v[i] = make_float2(tid * 1e-3f + i, 1.f - i * 0.01f);
        u[i] = make_float2(tid * 2e-3f - i, 0.5f + i * 0.02f);
        mine[64 * i] = v[i];
    }
    __syncthreads();
    if (MODE == M_STAGGER && wave >= NT / 128) math<4>(v, c);
    if (MODE == M_PRIO) {
        if ((wave >> 2) & 1) __builtin_amdgcn_s_setprio(1);
        if ((wave >> 3) & 1) __builtin_amdgcn_s_setprio(2);
    }
    float2 pa[8], pb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        pa[i] = v[i];
        pb[i] = v[8 + i];
    }
    for (int it = 0; it < iters; ++it) {
        if (MODE == M_VALU) {
            math<8>(v, c);
        } else if (MODE == M_LDS) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = lld<ACC>(&mine[64 * i]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 16; ++i) lst<ACC>(&mine[64 * ((i + 1) & 15)], v[i]);
            __builtin_amdgcn_wave_barrier();
        } else if (MODE == M_LDS128) {
            float4* m4 = reinterpret_cast<float4*>(lds + wave * 1024) + lane;  // element k at m4[64 k]
            float4 q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = m4[64 * i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) m4[64 * ((i + 1) & 7)] = q[i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = make_float2(q[i].x, q[i].y);
        } else if (MODE == M_LDS32) {
            float* m1 = reinterpret_cast<float*>(lds + wave * 1024) + lane;  // element k at m1[64 k]
            float q[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) q[i] = m1[64 * i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 32; ++i) m1[64 * ((i + 1) & 31)] = q[i];
            __builtin_amdgcn_wave_barrier();
            v[0].x = q[0];
        } else if (MODE == M_PIPE) {
            // two half-size register sets with LDS regions of their own (elements 0..7 and 8..15): the reads of one
            // set are in flight, and the writes of the other drain, under the arithmetic of the other set
#pragma unroll
            for (int i = 0; i < 8; ++i) pa[i] = lld<ACC>(&mine[64 * i]);
            __builtin_amdgcn_sched_barrier(0);
            math<8, 8>(pb, c);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) lst<ACC>(&mine[64 * (8 + ((i + 1) & 7))], pb[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) pb[i] = lld<ACC>(&mine[64 * (8 + i)]);
            __builtin_amdgcn_sched_barrier(0);
            math<8, 8>(pa, c);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) lst<ACC>(&mine[64 * ((i + 1) & 7)], pa[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = lld<ACC>(&mine[64 * i]);
            math<8>(v, c);
#pragma unroll
            for (int i = 0; i < 16; ++i) lst<ACC>(&mine[64 * ((i + 1) & 15)], v[i]);
            if (MODE == M_BAR)
                __syncthreads();
            else
                __builtin_amdgcn_wave_barrier();
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y + u[i].x + u[i].y + pa[i & 7].x + pb[i & 7].y;
    if (s == 12345.678f) out[1] = s;
}

template <int MODE, int NT, int ACC = 0>
static double run(float* d, int iters, int wg_per_cu) {
    const size_t lds = (size_t)NT * 16 * sizeof(float2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, NT, ACC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    auto launch = [&]() { hipLaunchKernelGGL((k<MODE, NT, ACC>), dim3(256 * wg_per_cu), dim3(NT), lds, 0, d, iters, 0.4999f); };
    launch();
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const int passes = iters;
    return ms * 1e6 / passes;  // ns per pass (all 16 waves of a CU do one pass each)
}

int main() {
    float* d;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    const int iters = 4000;
    const double tv = run<M_VALU, 1024>(d, iters, 1);
    const double tl = run<M_LDS, 1024>(d, iters, 1);
    printf("valu      %8.1f ns/pass/CU   (256 fma x 16 waves; 2 cyc/instr/SIMD ideal = 2048 cyc)\n", tv);
    printf("lds b64   %8.1f ns/pass/CU   (16 rd + 16 wr b64 x 16 waves)\n", tl);
    const double tlv = run<M_LDS, 1024, 1>(d, iters, 1);
    printf("lds b64v  %8.1f ns/pass/CU   (same, unmerged ds_read_b64 / ds_write_b64)\n", tlv);
    printf("lds b128  %8.1f ns/pass/CU   (8 rd + 8 wr b128 x 16 waves)\n", run<M_LDS128, 1024>(d, iters, 1));
    printf("lds b32   %8.1f ns/pass/CU   (32 rd + 32 wr b32 x 16 waves)\n", run<M_LDS32, 1024>(d, iters, 1));
    auto rep = [&](const char* name, double t) {
        printf("%-9s %8.1f ns/pass/CU   serial-ratio %.3f (ideal %.3f)\n", name, t, t / (tv + tl),
               (tv > tl ? tv : tl) / (tv + tl));
    };
    rep("bar", run<M_BAR, 1024>(d, iters, 1));
    rep("nobar", run<M_NOBAR, 1024>(d, iters, 1));
    rep("stagger", run<M_STAGGER, 1024>(d, iters, 1));
    rep("prio", run<M_PRIO, 1024>(d, iters, 1));
    rep("pipe", run<M_PIPE, 1024>(d, iters, 1));
    auto repv = [&](const char* name, double t) {
        printf("%-9s %8.1f ns/pass/CU   serial-ratio %.3f (ideal %.3f)  [unmerged b64]\n", name, t, t / (tv + tlv),
               (tv > tlv ? tv : tlv) / (tv + tlv));
    };
    repv("bar", run<M_BAR, 1024, 1>(d, iters, 1));
    repv("nobar", run<M_NOBAR, 1024, 1>(d, iters, 1));
    repv("pipe", run<M_PIPE, 1024, 1>(d, iters, 1));
    rep("2wg bar", run<M_BAR, 512>(d, iters, 2));
    rep("4wg bar", run<M_BAR, 256>(d, iters, 4));
    rep("2wg nobar", run<M_NOBAR, 512>(d, iters, 2));
    rep("1wg512 bar", 2 * run<M_BAR, 512>(d, iters, 1));
    rep("1wg512 pipe", 2 * run<M_PIPE, 512>(d, iters, 1));
    return 0;
}
