// How far does ds_write_addtid_b32 reach?  address = M0 + 16-bit offset + 4 * lane: is M0 taken whole or only [15:0]?
// One wave writes a tag with several (M0, offset) pairs into a 160 KB LDS image; the image is dumped and searched.
// Build: hipcc --offload-arch=gfx950 -O3 addtid_reach.hip -o addtid_reach
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k(unsigned* out, unsigned m0a, unsigned m0b, unsigned m0c) {
    extern __shared__ unsigned s[];
    for (int i = threadIdx.x; i < 40960; i += 64) s[i] = 0;
    __syncthreads();
    unsigned t1 = 0x11110000u + threadIdx.x, t2 = 0x22220000u + threadIdx.x, t3 = 0x33330000u + threadIdx.x, t4 = 0x44440000u + threadIdx.x;
    asm volatile("s_mov_b32 m0, %0\n\tds_write_addtid_b32 %1 offset:0" ::"s"(m0a), "v"(t1) : "memory");
    asm volatile("s_mov_b32 m0, %0\n\tds_write_addtid_b32 %1 offset:65532" ::"s"(m0a), "v"(t2) : "memory");
    asm volatile("s_mov_b32 m0, %0\n\tds_write_addtid_b32 %1 offset:0" ::"s"(m0b), "v"(t3) : "memory");
    asm volatile("s_mov_b32 m0, %0\n\tds_write_addtid_b32 %1 offset:60000" ::"s"(m0c), "v"(t4) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 40960; i += 64) out[i] = s[i];
}
int main() {
    unsigned* d;
    hipMalloc(&d, 163840);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    const unsigned m0a = 1024, m0b = 100000, m0c = 90000;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 163840, 0, d, m0a, m0b, m0c);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<unsigned> h(40960);
    hipMemcpy(h.data(), d, 163840, hipMemcpyDeviceToHost);
    const char* what[4] = {"M0=1024 off=0", "M0=1024 off=65532", "M0=100000 off=0", "M0=90000 off=60000"};
    const unsigned want[4] = {1024, 1024 + 65532, 100000, 150000};
    for (int t = 0; t < 4; ++t) {
        const unsigned tag = 0x11110000u * (t + 1);
        long first = -1;
        int cnt = 0;
        for (int i = 0; i < 40960; ++i)
            if ((h[i] & 0xffff0000u) == tag) { if (first < 0) first = (long)i * 4 - 4 * (h[i] & 0xffff); ++cnt; }
        printf("%-20s -> lane 0 landed at byte %ld (%d lanes found); full-M0 address would be %u, M0[15:0] address %u\n", what[t], first, cnt,
               want[t], (t < 2 ? want[t] : (t == 2 ? (100000u & 0xffff) : ((90000u & 0xffff) + 60000u))));
    }
    return 0;
}
