// Microbenchmark (gfx950): what slows scalar-f32 VALU issue below its 2-cycle peak?  Operand kinds (SGPR, literal,
// VOP2 vs VOP3) and read-after-write distance inside one wave (DIST independent chains interleaved), at 4 waves/SIMD.
// Prints ns per wave-instruction per SIMD and cycles at an assumed 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
enum { K_FMA_VVV, K_FMA_SVV, K_FMAC_VV, K_FMAC_SV, K_MUL_LIT, K_FMAMK, K_ADD_VV, K_SUB_VV, K_MIX, K_N };
static const char* names[K_N] = {"v_fma_f32 d,v,v,v",   "v_fma_f32 d,s,v,v", "v_fmac_f32 d,v,v", "v_fmac_f32 d,s,v", "v_mul_f32 d,lit,v",
                                 "v_fmamk_f32 d,v,lit,v", "v_add_f32 d,v,v",   "v_sub_f32 d,v,v",  "mix add/sub/mul/fmac"};
// DIST = number of independent dependency chains a wave interleaves (RAW distance in instructions)
template <int K, int DIST>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float c) {
    float a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = threadIdx.x * 1e-3f + i;
        b[i] = 1.f - i * 0.01f;
    }
    float cs = c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 128 / DIST; ++r) {
#pragma unroll
            for (int i = 0; i < DIST; ++i) {
                float& x = a[i];
                const float y = b[i], z = b[(i + 7) & 15];
                if (K == K_FMA_VVV) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (K == K_FMA_SVV) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x) : "s"(cs), "v"(z));
                if (K == K_FMAC_VV) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (K == K_FMAC_SV) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x) : "s"(cs), "v"(z));
                if (K == K_MUL_LIT) asm volatile("v_mul_f32_e32 %0, 0x3f7fbe77, %0" : "+v"(x));
                if (K == K_FMAMK) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(x) : "v"(z));
                if (K == K_ADD_VV) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(x) : "v"(z));
                if (K == K_SUB_VV) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(x) : "v"(z));
                if (K == K_MIX) {
                    if ((r & 3) == 0) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(x) : "v"(z));
                    if ((r & 3) == 1) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(x) : "v"(y));
                    if ((r & 3) == 2) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(x) : "v"(y));
                    if ((r & 3) == 3) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.678f) out[1] = s;
}
template <int K, int DIST>
static void run(float* d, int iters, int threads) {
    auto launch = [&]() { hipLaunchKernelGGL((k<K, DIST>), dim3(256), dim3(threads), 0, 0, d, iters, 0.999f); };
    launch();
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 128 * (threads / 64) / 4.0;
    printf("%-24s raw-dist=%2d waves/SIMD=%d  %.3f ns  %.2f cyc@2.4\n", names[K], DIST, threads / 256, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}
template <int K>
static void all(float* d, int iters) {
    run<K, 16>(d, iters, 1024);
    run<K, 8>(d, iters, 1024);
    run<K, 4>(d, iters, 1024);
    run<K, 2>(d, iters, 1024);
    run<K, 1>(d, iters, 1024);
    run<K, 4>(d, iters, 512);
    run<K, 1>(d, iters, 512);
}
int main() {
    float* d;
    (void)hipMalloc(&d, 64);
    (void)hipMemset(d, 0, 64);
    const int iters = 4000;
    all<K_FMA_VVV>(d, iters);
    all<K_FMA_SVV>(d, iters);
    all<K_FMAC_VV>(d, iters);
    all<K_FMAC_SV>(d, iters);
    all<K_MUL_LIT>(d, iters);
    all<K_FMAMK>(d, iters);
    all<K_ADD_VV>(d, iters);
    all<K_MIX>(d, iters);
    return 0;
}
