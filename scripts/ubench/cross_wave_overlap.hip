// Can VALU work of some waves of a CU run under the LDS traffic of OTHER waves of the same CU?
// 1024 threads per workgroup (4 waves per SIMD), one workgroup per CU.  Waves 0..7 (two per SIMD) run a chain-free
// stream of v_fmac_f32; waves 8..15 (the other two per SIMD) run ds_write_b64 / ds_read_b64 streams.  Three timings:
// VALU group alone, LDS group alone, both.  both ~ max(...) -> the two pipes overlap across waves; both ~ sum -> they do not.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize cross_wave_overlap.hip -o cross_wave_overlap
#include <hip/hip_runtime.h>

#include <cstdio>

typedef __attribute__((address_space(3))) float2 lds_f2;

template <int MODE>  // bit 0: VALU group works, bit 1: LDS group works; bits 2..3: 0 = writes, 1 = reads, 2 = both
__global__ __launch_bounds__(1024) void k(float* out, int iters, int split) {
    __shared__ float2 s[16384 + 1024];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    s[tid] = make_float2((float)tid, 1.f);
    __syncthreads();
    // split = 0: groups are waves 0..7 / 8..15 (two of each per SIMD); split = 1: even / odd SIMD pairs by wave & 1
    const bool valu_grp = split ? ((wave & 1) == 0) : (wave < 8);
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (float)(tid + i);
    float2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = make_float2((float)i, (float)tid);
    lds_f2* p = (lds_f2*)s + tid;
    const int lop = (MODE >> 2) & 3;
    if (valu_grp) {
        if (MODE & 1) {
            const float a = out[0], b = out[1];
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            }
        }
    } else {
        if (MODE & 2) {
            for (int it = 0; it < iters; ++it) {
                if (lop != 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(p), "v"(v[i & 7]), "n"(8192 * 0 + 0) : "memory");
                }
                if (lop != 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1" : "=v"(v[i]) : "v"(p) : "memory");
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1" : "=v"(v[i]) : "v"(p) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += v[i].x + v[i].y;
    if (r == 12345.678f) out[tid] = r;
}

template <int MODE>
static float run(float* d, int iters, int split) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d, iters, split);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d, iters, split);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    float* d;
    (void)hipMalloc(&d, 4096);
    (void)hipMemset(d, 0, 4096);
    const int iters = 4000;
    for (int w = 0; w < 5; ++w) run<3>(d, iters, 0);
    for (int split = 0; split < 2; ++split) {
        printf("groups: %s\n", split ? "even / odd waves (each SIMD pair holds one kind only)" : "waves 0..7 VALU, 8..15 LDS (two of each per SIMD)");
        const float a = run<1>(d, iters, split);
        // per wave and iteration: 256 v_fmac; 16 ds_write_b64 and/or 16 ds_read_b64
        printf("  VALU group alone                 %7.3f ms  (%.2f cycles@2.4GHz per v_fmac per SIMD)\n", a, a * 1e-3 * 2.4e9 / (iters * 256.0 * 2));
        const float w = run<2>(d, iters, split), rd = run<2 + 4>(d, iters, split), rw = run<2 + 8>(d, iters, split);
        printf("  LDS group alone: writes %7.3f ms (%.1f cyc per ds_write_b64 per CU)   reads %7.3f ms (%.1f)   both %7.3f ms\n", w,
               w * 1e-3 * 2.4e9 / (iters * 16.0 * 8), rd, rd * 1e-3 * 2.4e9 / (iters * 16.0 * 8), rw);
        const float bw = run<3>(d, iters, split), br = run<3 + 4>(d, iters, split), brw = run<3 + 8>(d, iters, split);
        printf("  both groups:     writes %7.3f ms (sum %.3f, max %.3f)   reads %7.3f ms (sum %.3f, max %.3f)   both %7.3f ms (sum %.3f, max %.3f)\n",
               bw, a + w, a > w ? a : w, br, a + rd, a > rd ? a : rd, brw, a + rw, a > rw ? a : rw);
    }
    return 0;
}
