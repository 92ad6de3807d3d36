// Does the raw-buffer range check of gfx950 include the SGPR offset (soffset)?  The tile role of k_caf_persistent and
// the |y|^2 stores of the FFT role rely on it: rows / tiles past the valid extent are addressed through soffset and
// must be dropped by the hardware.  (LLVM's intrinsic documentation calls soffset "excluded from bounds checking";
// that describes what the compiler assumes, not what gfx9-family hardware does for raw buffers.)
// Build: hipcc --offload-arch=gfx950 -O3 buffer_bounds.hip -o buffer_bounds
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void k_probe(float* out, const float* in, float* got, int records_bytes, int soff_store, int soff_load) {
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, records_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, records_bytes, 0x00020000);
    const int lane = threadIdx.x;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, 1000.f + lane), ro, lane * 4, soff_store, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, 2000.f + lane), ro, lane * 4, soff_store, 16);  // sc1
    got[lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, lane * 4, soff_load, 0));
}

int main() {
    const int n = 1024;
    float *out, *in, *got;
    hipMalloc(&out, n * 4);
    hipMalloc(&in, n * 4);
    hipMalloc(&got, 64 * 4);
    std::vector<float> h(n), g(64);
    int bad = 0;
    for (int records : {256, 384}) {
        for (int soff : {0, 128, 256, 512, 2048}) {
            for (int i = 0; i < n; ++i) h[i] = (float)i;
            hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
            hipMemset(out, 0, n * 4);
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, out, in, got, records, soff, soff);
            hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
            hipMemcpy(g.data(), got, 64 * 4, hipMemcpyDeviceToHost);
            int written = 0, wrong_place = 0, loaded = 0;
            for (int i = 0; i < n; ++i)
                if (h[i] != 0.f) {
                    ++written;
                    if (i * 4 >= records) ++wrong_place;
                }
            for (int l = 0; l < 64; ++l)
                if (g[l] != 0.f || (soff == 0 && l == 0)) ++loaded;
            // expected when soffset IS range-checked: lanes with lane*4 + soff < records are written / loaded
            int expect = 0;
            for (int l = 0; l < 64; ++l) expect += (l * 4 + soff < records);
            printf("records %4d B soffset %5d: %2d words written (%d beyond the records), %2d lanes loaded non-zero; %2d expected if soffset is checked\n",
                   records, soff, written, wrong_place, loaded, expect);
            if (written != expect || wrong_place) bad = 1;
        }
    }
    printf(bad ? "soffset is NOT (fully) range-checked\n" : "soffset is range-checked: out-of-range stores dropped, loads return 0\n");
    return bad;
}
