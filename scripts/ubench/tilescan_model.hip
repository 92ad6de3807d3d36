// Where the time of a ticketed single-pass tile scan goes (tilescan.h beside this file): a read-only pass over 2^24 floats
// (count of positive samples per tile), with the ticket and the look-back switched on separately.
//   hipcc --offload-arch=gfx950 -O3 tilescan_model.hip -o /tmp/tilescan_model && /tmp/tilescan_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "tilescan.h"
using namespace caf;

template <int NT, int PER, bool TICKET, bool LOOK>
__global__ __launch_bounds__(NT) void k_model(const float* __restrict__ x, int64_t n, uint64_t* ws, uint32_t ntiles, int64_t* out) {
    __shared__ int32_t s_w[NT / 64];
    __shared__ uint32_t s_tile;
    uint32_t tile = blockIdx.x;
    if (TICKET) {
        if (threadIdx.x == 0) s_tile = ts_ticket(ws, ntiles);
        __syncthreads();
        tile = s_tile;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)tile * NT * PER + (int64_t)threadIdx.x * PER;
    int c = 0;
#pragma unroll
    for (int j = 0; j < PER; j += 4) {
        const float4 q = *reinterpret_cast<const float4*>(x + base + j);
        c += (q.x > 0.f) + (q.y > 0.f) + (q.z > 0.f) + (q.w > 0.f);
    }
    c = ts_wave_sum(c);
    if (lane == 0) s_w[wave] = c;
    __syncthreads();
    if (wave == 0) {
        int64_t total = 0;
        for (int w = 0; w < NT / 64; ++w) total += s_w[w];
        int64_t e = 0;
        if (LOOK) e = ts_exclusive<int64_t>(ws, tile, ntiles, total);
        if (lane == 0) out[tile] = e + total;
    }
}

template <int NT, int PER, bool TICKET, bool LOOK>
static void run(const float* x, int64_t n, uint64_t* ws, int64_t* out, const char* what) {
    const uint32_t nt = (uint32_t)(n / (NT * PER));
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    float best = 1e9f, sum = 0.f;
    for (int it = 0; it < 12; ++it) {
        hipMemsetAsync(ws, 0xff, ts_words(nt) * 8, nullptr);
        hipEventRecord(a, nullptr);
        hipLaunchKernelGGL((k_model<NT, PER, TICKET, LOOK>), dim3(nt), dim3(NT), 0, nullptr, x, n, ws, nt, out);
        hipEventRecord(b, nullptr);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (it >= 2) best = ms < best ? ms : best, sum += ms;
    }
    int64_t last;
    hipMemcpy(&last, out + nt - 1, 8, hipMemcpyDeviceToHost);
    printf("%-44s NT=%4d PER=%2d tiles=%5u  min %6.1f us  mean %6.1f us  (last tile's inclusive count %lld)\n", what, NT, PER, nt,
           best * 1e3, sum / 10 * 1e3, (long long)last);
}

int main() {
    const int64_t n = 1 << 24;
    std::vector<float> h(n);
    for (int64_t i = 0; i < n; ++i) h[i] = (i * 2654435761u) & 0x100 ? 1.f : -1.f;
    float* x;
    uint64_t* ws;
    int64_t* out;
    hipMalloc(&x, n * 4), hipMalloc(&ws, 1 << 22), hipMalloc(&out, 1 << 22);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
#define ALL(NT, PER)                                                        \
    run<NT, PER, false, false>(x, n, ws, out, "static tile, no look-back");   \
    run<NT, PER, true, false>(x, n, ws, out, "ticket, no look-back");         \
    run<NT, PER, false, true>(x, n, ws, out, "static tile, look-back (model only)"); \
    run<NT, PER, true, true>(x, n, ws, out, "ticket + look-back");
    ALL(256, 16)
    ALL(1024, 16)
    ALL(1024, 32)
    ALL(256, 32)
    return 0;
}
