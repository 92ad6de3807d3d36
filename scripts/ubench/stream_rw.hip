// What a streaming kernel can reach on this chip: read-only, write-only and copy over 1 GiB, 16 bytes per lane.
//   hipcc --offload-arch=gfx950 -O3 stream_rw.hip -o /tmp/stream_rw && /tmp/stream_rw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>  // 0 read (sum), 1 write, 2 copy
__global__ __launch_bounds__(256) void k_stream(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (MODE == 0) {
            const v4f v = NT ? __builtin_nontemporal_load(in + i) : in[i];
            acc += v.x + v.y + v.z + v.w;
        } else if (MODE == 1) {
            const v4f v = {(float)i, 1.f, 2.f, 3.f};
            if (NT) __builtin_nontemporal_store(v, out + i);
            else out[i] = v;
        } else {
            const v4f v = NT ? __builtin_nontemporal_load(in + i) : in[i];
            if (NT) __builtin_nontemporal_store(v, out + i);
            else out[i] = v;
        }
    }
    if (MODE == 0 && acc == 12345.678f) *sink = acc;
}

template <int MODE, bool NT>
static void run(const v4f* in, v4f* out, size_t n, float* sink, int grid, const char* what) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a), (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 8; ++it) {
        (void)hipEventRecord(a, nullptr);
        hipLaunchKernelGGL((k_stream<MODE, NT>), dim3(grid), dim3(256), 0, nullptr, in, out, n, sink);
        (void)hipEventRecord(b, nullptr);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (it >= 2 && ms < best) best = ms;
    }
    const double bytes = (double)n * 16.0 * (MODE == 2 ? 2.0 : 1.0);
    printf("%-28s grid %6d  %8.1f us  %6.2f TB/s\n", what, grid, best * 1e3, bytes / (best * 1e-3) / 1e12);
}

int main() {
    const size_t n = (size_t)1 << 26;  // 1 GiB of float4
    v4f *in, *out;
    float* sink;
    (void)hipMalloc(&in, n * 16), (void)hipMalloc(&out, n * 16), (void)hipMalloc(&sink, 4);
    (void)hipMemset(in, 1, n * 16);
    for (int grid : {2048, 8192, 65536}) {
        run<0, false>(in, out, n, sink, grid, "read");
        run<0, true>(in, out, n, sink, grid, "read, nontemporal");
        run<1, false>(in, out, n, sink, grid, "write");
        run<1, true>(in, out, n, sink, grid, "write, nontemporal");
        run<2, false>(in, out, n, sink, grid, "copy (read + write bytes)");
        run<2, true>(in, out, n, sink, grid, "copy, nontemporal");
    }
    return 0;
}
