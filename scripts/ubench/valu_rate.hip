// Microbenchmark: sustained wave64 VALU issue rate on gfx950 for v_add_f32 / v_fma_f32 / mixed add+mul,
// at 1, 2, 4 waves per SIMD (256 CUs, one workgroup per CU).  Prints cycles per wave-instruction per SIMD.
// CAUTION (round 2): built with plain -O3 the loop below is SLP-vectorised into v_pk_add_f32 / v_pk_fma_f32, so the
// "2 cycles per op" it printed in round 1 was a 4-cycle PACKED instruction doing two ops.  Build it with
// -fno-slp-vectorize, or use valu_issue.hip / pk_rate.hip (inline asm, one instruction form per row) instead.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(float* out, int iters) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    const float c = out[0] * 0.5f + 1.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = a[i] + c;
                else if (OP == 1) a[i] = __builtin_fmaf(a[i], c, c);
                else a[i] = (r & 1) ? a[i] * c : a[i] - c;
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 12345.678f) out[1] = s;
}
int main() {
    float* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int op = 0; op < 3; ++op)
        for (int threads : {256, 512, 1024}) {
            auto launch = [&]() {
                if (op == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, d, iters);
                if (op == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, d, iters);
                if (op == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, d, iters);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)iters * 128 * (threads / 64) / 4.0;
            printf("op=%d waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD (= %.2f cycles @2.1GHz, %.2f @2.4GHz)\n", op,
                   threads / 256, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.1, ms * 1e6 / instr_per_simd * 2.4);
        }
    return 0;
}
