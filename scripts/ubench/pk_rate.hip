// Microbenchmark (gfx950): issue cost of packed-f32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32,
// with and without op_sel / neg modifiers and SGPR-pair operands) against the scalar v_fma_f32 / v_add_f32 forms,
// and of the cross-lane forms (DPP quad_perm, v_permlane32_swap), at 1, 2 and 4 waves per SIMD.
// 16 independent accumulators per thread, so no dependent-issue stalls.  Prints cycles per wave-instruction per SIMD
// at an assumed 2.4 GHz (relative numbers are what matters).
// Built WITHOUT SLP vectorisation (-fno-slp-vectorize) -- the scalar rows must stay scalar.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

enum { T_FMA, T_ADD, T_FMAC, T_PKFMA, T_PKFMA_MOD, T_PKMUL_SGPR, T_PKADD, T_PKADD_MOD, T_CMUL, T_DPP, T_SWAP32, T_SWAP16, T_N };
static const char* names[T_N] = {"v_fma_f32 (3 vgpr)",       "v_add_f32_e32",           "v_fmac_f32_e32",
                                 "v_pk_fma_f32",             "v_pk_fma_f32 op_sel+neg", "v_pk_mul_f32 sgpr-pair",
                                 "v_pk_add_f32",             "v_pk_add_f32 op_sel+neg", "cmul = pk_mul + pk_fma (dep.)",
                                 "v_add_f32_dpp quad_perm",  "v_permlane32_swap",       "v_permlane16_swap"};
static const int per_iter[T_N] = {16 * 8, 16 * 8, 16 * 8, 16 * 8, 16 * 8, 16 * 8, 16 * 8, 16 * 8, 16 * 8 * 2, 16 * 8, 8 * 8, 8 * 8};

template <int T>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float cx, float cy) {
    v2f a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (v2f){threadIdx.x * 1e-3f + i, 1.f - i * 0.01f};
    const v2f c = {cx, cy};
    v2f cs = c;  // copy that is forced into an SGPR pair below
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                v2f& x = a[i];
                v2f& y = a[(i + 5) & 15];
                if (T == T_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x.x) : "v"(y.x), "v"(y.y));
                if (T == T_ADD) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(x.x) : "v"(y.y));
                if (T == T_FMAC) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x.x) : "v"(y.x), "v"(y.y));
                if (T == T_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(a[(i + 9) & 15]));
                if (T == T_PKFMA_MOD)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
                                 : "+v"(x)
                                 : "v"(y), "v"(a[(i + 9) & 15]));
                if (T == T_PKMUL_SGPR) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(x) : "s"(cs));
                if (T == T_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (T == T_PKADD_MOD)
                    asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(x) : "v"(y));
                if (T == T_CMUL) {
                    v2f t;
                    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(x), "v"(y));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
                                 : "=v"(x)
                                 : "v"(x), "v"(y), "v"(t));
                }
                if (T == T_DPP)
                    asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x.x) : "v"(y.x));
                if (T == T_SWAP32 && i < 8) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x.x), "+v"(a[i + 8].y));
                if (T == T_SWAP16 && i < 8) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x.x), "+v"(a[i + 8].y));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    if (s == 12345.678f) out[1] = s;
}

template <int T>
static void run(float* d, int iters) {
    for (int threads : {256, 512, 1024}) {
        auto launch = [&]() { hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), 0, 0, d, iters, 0.99f, 0.01f); };
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
        const double instr_per_simd = (double)iters * per_iter[T] * (threads / 64) / 4.0;
        printf("%-32s waves/SIMD=%d  %7.3f ms  %.2f cyc/instr/SIMD @2.4GHz\n", names[T], threads / 256, ms,
               ms * 1e6 / instr_per_simd * 2.4);
    }
}

int main() {
    float* d;
    (void)hipMalloc(&d, 64);
    (void)hipMemset(d, 0, 64);
    const int iters = 4000;
    run<T_FMA>(d, iters);
    run<T_ADD>(d, iters);
    run<T_FMAC>(d, iters);
    run<T_PKFMA>(d, iters);
    run<T_PKFMA_MOD>(d, iters);
    run<T_PKMUL_SGPR>(d, iters);
    run<T_PKADD>(d, iters);
    run<T_PKADD_MOD>(d, iters);
    run<T_CMUL>(d, iters);
    run<T_DPP>(d, iters);
    run<T_SWAP32>(d, iters);
    run<T_SWAP16>(d, iters);
    return 0;
}
