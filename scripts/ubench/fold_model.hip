// Prices "the FFT workgroup transposes its own tiles" BEFORE it is built into k_caf_persistent (round 4).
//
// Today the |y|^2 tiles an FFT item writes (hypothesis-major, 256-byte rows) are turned into the delay-major surface by a
// second role on 96 dedicated CUs.  The arrangement modelled here needs no second role: wave w of an FFT workgroup wrote
// the tiles t = w (mod 16) of its item itself (pass 4: wave = (n2 >> 2) + 4 q = the per-thread part of the tile index), so
// ONE ITEM LATER the same wave fetches them back by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR holds bytes in flight)
// into a private 1 KB patch, 16 hypotheses x 16 delays per round, reads the patch transposed with one ds_read_b128 and
// writes 64-byte surface row segments (4 delays x 16 hypotheses per store instruction).  Three rounds per transform and
// wave, at three fixed points of the transform (after the first exchange, before the template row is fetched, after the
// products of the last pass), each: s_waitcnt vmcnt(0) -> ds_read_b128 -> 4 stores -> next DMA.
//
// The patch is lane-linear (an LDS-DMA writes M0 + 16 * lane), so the swizzle that makes the transposed read conflict-free
// sits in the SOURCE addresses: slot p = 4 * hyp + (quad ^ (hyp >> 2)) holds delays 4 quad .. 4 quad + 3 of hypothesis hyp.
//
//   check : functional test of exactly that round (DMA to a patch above 64 KB of LDS, swizzle, transposed read, stores)
//   F0    : the shipped structure (fft_struct_model's C1) with the tile layout of the real kernel
//   Fv    : F0 + the arithmetic and registers of normalise + running maximum in pass 4 (60 VALU, 27 live registers)
//   F1    : Fv + the three rounds (real HBM traffic: 4 MB of tiles and 12 MB of surface per workgroup, 1 GB + 0.8 GB in all)
//   F1nt  : F1 with nt on the DMA loads and the surface stores
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 fold_model.hip -o fold_model
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define AS1 __attribute__((address_space(1)))
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) v4f_t lds_v4f;

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
template <int OFF0>
__device__ __forceinline__ void wr_plane16(uint32_t m0, const float* r) {
    asm volatile(
        "s_mov_b32 m0, %0\n\ts_nop 0\n\t"
        "ds_write_addtid_b32 %1 offset:%17\n\tds_write_addtid_b32 %2 offset:%17+4096\n\t"
        "ds_write_addtid_b32 %3 offset:%17+8192\n\tds_write_addtid_b32 %4 offset:%17+12288\n\t"
        "ds_write_addtid_b32 %5 offset:%17+16384\n\tds_write_addtid_b32 %6 offset:%17+20480\n\t"
        "ds_write_addtid_b32 %7 offset:%17+24576\n\tds_write_addtid_b32 %8 offset:%17+28672\n\t"
        "ds_write_addtid_b32 %9 offset:%17+32768\n\tds_write_addtid_b32 %10 offset:%17+36864\n\t"
        "ds_write_addtid_b32 %11 offset:%17+40960\n\tds_write_addtid_b32 %12 offset:%17+45056\n\t"
        "ds_write_addtid_b32 %13 offset:%17+49152\n\tds_write_addtid_b32 %14 offset:%17+53248\n\t"
        "ds_write_addtid_b32 %15 offset:%17+57344\n\tds_write_addtid_b32 %16 offset:%17+61440" ::"s"(m0),
        "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(r[8]), "v"(r[9]), "v"(r[10]),
        "v"(r[11]), "v"(r[12]), "v"(r[13]), "v"(r[14]), "v"(r[15]), "n"(OFF0)
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
template <int J0, int NJ>
__device__ __forceinline__ void rd_tw(const float2* tw, int lane, float (&cf)[2 * NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const uint64_t u = *(const volatile __attribute__((address_space(3))) uint64_t*)(&tw[(J0 + j) * 64 + lane]);
        float2 v;
        __builtin_memcpy(&v, &u, 8);
        cf[2 * j] = v.x, cf[2 * j + 1] = v.y;
    }
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_of(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
template <typename Tp>
__device__ __forceinline__ Tp* uniform_ptr(Tp* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<Tp*>(((uint64_t)hi << 32) | lo);
}
// one LDS-DMA round trip issue: 64 lanes x 16 bytes -> LDS bytes [lds_dst, lds_dst + 1024); the compiler does not see it
template <bool NT>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, uint32_t lds_dst) {
    if (NT)
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen nt lds" ::"v"(voff), "s"(r), "s"(soff), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(r), "s"(soff), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- the geometry shared by the model and the functional check ----
// tiles of one (block, group) item: vt[tile][h 0..63][64 delays] floats; surface rows: [delay][256 hypotheses]
constexpr int NH = 64, ROWB = 1024;          // hypotheses per item, bytes per surface row
__device__ __forceinline__ uint32_t dma_voff(int lane) {  // slot = lane: hyp = lane >> 2, quad = (lane & 3) ^ (lane >> 4)
    return (uint32_t)((lane >> 2) * 256 + (((lane & 3) ^ (lane >> 4)) * 16));
}
__device__ __forceinline__ uint32_t patch_rd(int lane) {  // reader: hyp = lane & 15, quad = lane >> 4
    const int hyp = lane & 15, quad = lane >> 4;
    return (uint32_t)(16 * (4 * hyp + (quad ^ (hyp >> 2))));
}
__device__ __forceinline__ uint32_t st_voff(int lane) {   // delay 4 quad (+ k), hypothesis lane & 15
    return (uint32_t)((lane >> 4) * 4 * ROWB + (lane & 15) * 4);
}
// round R of a wave's item: tile j = R >> 4 (the wave's j-th tile), then Z order over (g = 16-hypothesis group, c = 16-delay chunk)
__device__ __forceinline__ void round_of(int R, int& j, int& g, int& c) {
    j = R >> 4;
    g = (R & 1) | ((R >> 1) & 2);
    c = ((R >> 1) & 1) | ((R >> 2) & 2);
}

struct Args {
    const float2* xb;
    const float2* hc;
    float* vt;    // [wgs][256 tiles][64][64]
    float* surf;  // [wgs / 4][12288][256]
    int nt;
    uint32_t* stats;  // [wgs][16 waves][4]: 100 MHz ticks spent in the waits of round points A, B, C
};

// VAL: normalise + running maximum in pass 4; ROUNDS: the three tile rounds per transform; ORDER 0: stores, then the DMA,
// s_waitcnt vmcnt(0) at the next point / 1: DMA first, then the stores, counted waits (only the DMA is waited for);
// NT: nt on DMA loads and surface stores; SRC 1: every DMA reads the same (L2-resident) 16 KB; WAIT 0: no waits (timing only)
template <int VAL, int ROUNDS, int ORDER = 0, int NT = 0, int SRC = 0, int WAIT = 1, int RH = 16, int ORD = 2, int VS = 100>
__global__ __launch_bounds__(1024) void k_fold(Args a) {
    uint32_t st_acc[3] = {0u, 0u, 0u};
    __shared__ __attribute__((aligned(16))) float s_img[32768 + 64];  // planar image: 2 x 64 KB
    __shared__ float2 s_tw[15 * 64];
    __shared__ __attribute__((aligned(16))) float s_patch[16 * 256];  // 1 KB per wave
    float* s_d = s_img;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 960) s_tw[tid] = make_float2(0.999f, 0.001f);
    float xr[32], pr[32], cw[2] = {0.9990234f + tid * 1e-9f, 0.0441f};
    float* vt = a.vt + (size_t)blockIdx.x * 256 * NH * 64;
    float* surf = a.surf + (size_t)(blockIdx.x >> 2) * 12288 * 256 + (blockIdx.x & 3) * 64;
    const uint32_t m0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_char*)s_d + (uint32_t)wave * 256u);
    const uint32_t patch0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_char*)s_patch + (uint32_t)wave * 1024u);
    // (the read address is laundered: nothing the compiler can see ever stores to s_patch, so a plain read of it folds to undef)
    uint32_t prd_a = (uint32_t)(uintptr_t)(lds_char*)s_patch + (uint32_t)wave * 1024u + patch_rd(lane);
    asm volatile("" : "+v"(prd_a));
    const lds_char* prd = (const lds_char*)(uintptr_t)prd_a;
    const __amdgpu_buffer_rsrc_t rvt = buf_of(uniform_ptr(vt), 256u * NH * 256u);
    const __amdgpu_buffer_rsrc_t rsf = buf_of(uniform_ptr(surf), 12288u * ROWB);
    // (shapes other than 16 x 16: timing only -- plain lane order, no swizzle)
    const uint32_t dvo = RH == 16 ? dma_voff(lane) : (uint32_t)((lane / (64 / RH)) * 256 + (lane % (64 / RH)) * 16);
    const uint32_t svo = RH == 16 ? st_voff(lane) : (uint32_t)((lane / RH) * 4 * ROWB + (lane % RH) * 4);
    float keep[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) keep[i] = 1.f + i + tid * 1e-3f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float2 x = ld2(a.xb, 1024 * i + tid), h = ld2(a.hc, 1024 * i + tid);
        xr[2 * i] = x.x, xr[2 * i + 1] = x.y;
        pr[2 * i] = x.x * h.x, pr[2 * i + 1] = x.y * h.y;
    }
    __syncthreads();
    // round R: wait for the DMA of the previous round point, read the patch, store, issue this point's DMA
    auto round_pt = [&](int t, int r) {
        if (!ROUNDS) return;
        // round shape: RH hypotheses x DD = 256 / RH delays (read pieces of 4 DD bytes, write segments of 4 RH bytes)
        constexpr int DD = 256 / RH, NG = 64 / RH, NC = 64 / DD;
        auto shape_of = [&](int RR, int& jj, int& gg, int& cc) {
            if (ORD == 2 && RH == 16) {
                round_of(RR, jj, gg, cc);
            } else {
                jj = RR >> 4;
                const int r16 = RR & 15;
                if (ORD == 0) gg = r16 % NG, cc = r16 / NG; else cc = r16 % NC, gg = r16 / NC;
            }
        };
        const int R = 3 * (t & 63) + r;
        int j, g, c;
        shape_of(R, j, g, c);
        if (WAIT) {
            const uint64_t t0 = __builtin_amdgcn_s_memtime();
            if (ORDER == 0)
                wait_vm0();
            else if (r == 0)
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // younger than the DMA: its 4 stores + 12 tile stores
            else if (r == 1)
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // its 4 stores
            // (r == 2: the template row loads issued after that DMA have been waited for: it has landed)
            asm volatile("s_nop 0" ::: "memory");
            st_acc[r] += (uint32_t)(__builtin_amdgcn_s_memtime() - t0);
        }
        const v4f_t q = *reinterpret_cast<const lds_v4f*>(prd);
        // the data just read belongs to the PREVIOUS round point
        int jp, gp, cp;
        shape_of(R == 0 ? 191 : R - 1, jp, gp, cp);
        const int tile_p = wave + 16 * (jp & 3) + 64 * (jp >> 2);
        const uint32_t so = (uint32_t)((tile_p * 64 + DD * cp) * ROWB + gp * RH * 4);
        constexpr int aux = NT ? 2 : 0;
        const int tile = wave + 16 * (j & 3) + 64 * (j >> 2);
        const uint32_t dso = SRC ? 0u : (uint32_t)((tile * NH + RH * g) * 256 + DD * 4 * c);
        if (ORDER == 1) {
            float qq[4] = {q.x, q.y, q.z, q.w};
            asm volatile("" : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[2]), "+v"(qq[3]));  // the read has returned before the patch is refilled
            dma16<NT != 0>(rvt, dvo, dso, patch0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(qq[0]), rsf, (int)svo, (int)so, aux);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(qq[1]), rsf, (int)svo, (int)(so + ROWB), aux);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(qq[2]), rsf, (int)svo, (int)(so + 2 * ROWB), aux);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(qq[3]), rsf, (int)svo, (int)(so + 3 * ROWB), aux);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(q.x), rsf, (int)svo, (int)so, aux);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(q.y), rsf, (int)svo, (int)(so + ROWB), aux);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(q.z), rsf, (int)svo, (int)(so + 2 * ROWB), aux);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(q.w), rsf, (int)svo, (int)(so + 3 * ROWB), aux);
            dma16<NT != 0>(rvt, dvo, dso, patch0);
        }
    };
    for (int t = 0; t < a.nt; ++t) {
        uint32_t hoff = (uint32_t)(t & 63) * 256u;
        asm volatile("" : "+s"(hoff));
        int lz = 0;
        asm volatile("" : "+v"(lz));
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = pr[i];
        valu<296 * VS / 100, 32, 2>(v, cw);
        __syncthreads();
        {
            float re[16], im[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) re[i] = v[2 * i], im[i] = v[2 * i + 1];
            wr_plane16<0>(m0, re);
            wr_plane16<4092>(m0 + 61444u, im);
        }
        __syncthreads();
        round_pt(t, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            rd_plane<16>((const float*)s_d, 1024, tid + lz, v);
            rd_plane<16>((const float*)s_d + 16384, 1024, tid + lz, v + 16);
            {  // (twiddles a few at a time, as the real kernel consumes them)
                float cf[10];
                rd_tw<0, 5>(s_tw, lane + lz, cf);
                valu<80 * VS / 100, 32, 10>(v, cf);
                rd_tw<5, 5>(s_tw, lane + lz, cf);
                valu<78 * VS / 100, 32, 10>(v, cf);
                rd_tw<10, 5>(s_tw, lane + lz, cf);
                valu<78 * VS / 100, 32, 10>(v, cf);
            }
            float re[16], im[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) re[i] = v[2 * i], im[i] = v[2 * i + 1];
            wr_plane16<0>(m0, re);
            wr_plane16<4092>(m0 + 61444u, im);
            if (pass == 0) {
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
                round_pt(t, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float hn[32];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float2 h = ld2(a.hc, ((1024 * i + tid - 4 * t) & 16383) + lz);
            hn[2 * i] = h.x, hn[2 * i + 1] = h.y;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            pr[2 * i] = xr[2 * i] * hn[2 * i] - xr[2 * i + 1] * hn[2 * i + 1];
            pr[2 * i + 1] = xr[2 * i] * hn[2 * i + 1] + xr[2 * i + 1] * hn[2 * i];
        }
        __builtin_amdgcn_sched_barrier(0);
        round_pt(t, 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float y[8];
            rd_plane<4>((const float*)s_d, 1024, tid + 1024 * i + lz, y);
            rd_plane<4>((const float*)s_d + 16384, 1024, tid + 1024 * i + lz, y + 4);
            valu<16 * VS / 100, 8, 2>(y, cw);
#pragma unroll
            for (int k = 0; k < 3; ++k) {  // three valid output quarters (N = 4096)
                float val = y[2 * k] * y[2 * k] + y[2 * k + 1] * y[2 * k + 1];
                if (VAL) {  // normalise + running maximum with its hypothesis: 5 instructions per value
                    const int o = 3 * i + k;
                    val *= keep[o];
                    const bool up = val > keep[12 + o];
                    keep[12 + o] = up ? val : keep[12 + o];
                    const uint32_t bi = __builtin_bit_cast(uint32_t, keep[24 + (o >> 2)]);
                    const uint32_t repl = (bi & ~(0xffu << (8 * (o & 3)))) | ((uint32_t)(t & 63) << (8 * (o & 3)));
                    keep[24 + (o >> 2)] = __builtin_bit_cast(float, up ? repl : bi);
                }
                const int tile = wave + 16 * i + 64 * k;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, val), rvt, lane * 4, tile * NH * 256 + hoff, 0);
            }
        }
    }
    if (VAL) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 27; ++i) s += keep[i];
        if (s == 12345.f) a.vt[tid] = s;
    }
    wait_vm0();
    if (ROUNDS && a.stats && lane == 0) {
        uint32_t* o = a.stats + ((size_t)blockIdx.x * 16 + wave) * 4;
        o[0] = st_acc[0], o[1] = st_acc[1], o[2] = st_acc[2];
    }
}

// ---- functional check of one round: tile -> patch (above 64 KB of LDS) -> surface segments ----
__global__ __launch_bounds__(1024) void k_check(const float* tile /* [64 h][64 d] */, float* surf /* [64 d][256] */, uint32_t* where) {
    extern __shared__ __attribute__((aligned(16))) char s_all[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t patch_off = 147456u + (uint32_t)wave * 1024u;  // 144 KB + wave KB: beyond any 16-bit reach
    const uint32_t base = (uint32_t)(uintptr_t)(lds_char*)s_all;
    const uint32_t patch0 = __builtin_amdgcn_readfirstlane(base + patch_off);
    const __amdgpu_buffer_rsrc_t rt = buf_of(uniform_ptr(tile), 64u * 256u);
    const __amdgpu_buffer_rsrc_t rs = buf_of(uniform_ptr(surf), 64u * ROWB);
    // wave w takes (g, c) = (w & 3, w >> 2): the sixteen waves cover the tile
    const int g = wave & 3, c = wave >> 2;
    dma16<false>(rt, dma_voff(lane), (uint32_t)((16 * g) * 256 + 64 * c), patch0);
    wait_vm0();
    const v4f_t q = *reinterpret_cast<const lds_v4f*>((const lds_char*)s_all + patch_off + patch_rd(lane));
    const uint32_t so = (uint32_t)((16 * c) * ROWB + g * 64);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(q.x), rs, (int)st_voff(lane), (int)so, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(q.y), rs, (int)st_voff(lane), (int)(so + ROWB), 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(q.z), rs, (int)st_voff(lane), (int)(so + 2 * ROWB), 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(q.w), rs, (int)st_voff(lane), (int)(so + 3 * ROWB), 0);
    if (tid == 0) where[0] = base, where[1] = patch0;
}

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

static int check() {
    float *tile, *surf;
    uint32_t* where;
    CK(hipMalloc(&tile, 64 * 64 * 4));
    CK(hipMalloc(&surf, 64 * 256 * 4));
    CK(hipMalloc(&where, 8));
    std::vector<float> h(64 * 64), s(64 * 256, -1.f);
    for (int hh = 0; hh < 64; ++hh)
        for (int d = 0; d < 64; ++d) h[hh * 64 + d] = (float)(hh * 100 + d);
    CK(hipMemcpy(tile, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(surf, s.data(), s.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)k_check, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    hipLaunchKernelGGL(k_check, dim3(1), dim3(1024), 163840, 0, tile, surf, where);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(s.data(), surf, s.size() * 4, hipMemcpyDeviceToHost));
    uint32_t w[2];
    CK(hipMemcpy(w, where, 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int d = 0; d < 64; ++d)
        for (int hh = 0; hh < 64; ++hh)
            if (s[d * 256 + hh] != (float)(hh * 100 + d)) {
                if (bad < 8) printf("  mismatch: delay %d hyp %d: got %g want %d\n", d, hh, s[d * 256 + hh], hh * 100 + d);
                ++bad;
            }
    printf("check: LDS base %u, wave 0 patch at byte %u of LDS: %s (%d of 4096 cells wrong)\n", w[0], w[1], bad ? "FAILED" : "ok", bad);
    return bad != 0;
}

template <typename K>
static int run(const char* name, K kern, Args a, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    printf("%-10s %8.3f ms  -> %6.2f us per 16384-point transform and CU", name, best, best * 1e3 / a.nt);
    if (a.stats) {
        std::vector<uint32_t> st(256 * 16 * 4);
        CK(hipMemcpy(st.data(), a.stats, st.size() * 4, hipMemcpyDeviceToHost));
        double acc[3] = {0, 0, 0};
        for (int i = 0; i < 256 * 16; ++i)
            for (int k = 0; k < 3; ++k) acc[k] += st[i * 4 + k];
        // s_memtime ticks at 100 MHz: 10 ns each; average per wave and transform
        printf("   waits A/B/C: %.2f / %.2f / %.2f us per transform", acc[0] / (256 * 16) / a.nt * 0.01, acc[1] / (256 * 16) / a.nt * 0.01,
               acc[2] / (256 * 16) / a.nt * 0.01);
        CK(hipMemset(a.stats, 0, st.size() * 4));
    }
    printf("\n");
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const int nt = argc > 1 ? atoi(argv[1]) : 256, reps = argc > 2 ? atoi(argv[2]) : 3;
    if (check()) return 1;
    Args a;
    float2 *xb, *hc;
    CK(hipMalloc(&xb, 16384 * 8));
    CK(hipMalloc(&hc, 16384 * 8));
    CK(hipMalloc(&a.vt, (size_t)256 * 256 * NH * 64 * 4));
    CK(hipMalloc(&a.surf, (size_t)64 * 12288 * 256 * 4));
    CK(hipMemset(a.vt, 0, (size_t)256 * 256 * NH * 64 * 4));
    std::vector<float2> h(16384);
    for (int i = 0; i < 16384; ++i) h[i] = make_float2(0.5f + 1e-5f * i, 0.25f - 1e-5f * i);
    CK(hipMemcpy(xb, h.data(), 16384 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(hc, h.data(), 16384 * 8, hipMemcpyHostToDevice));
    a.xb = xb, a.hc = hc, a.nt = nt;
    CK(hipMalloc(&a.stats, 256 * 16 * 16));
    CK(hipMemset(a.stats, 0, 256 * 16 * 16));
    for (int rep = 0; rep < 2; ++rep) {  // repeated: the clock settles under load
        if (run("F0", k_fold<0, 0>, a, reps)) return 1;
        if (run("Fv", k_fold<1, 0>, a, reps)) return 1;
        if (run("F1", k_fold<1, 1, 0, 0>, a, reps)) return 1;
        if (run("F1c", k_fold<1, 1, 1, 0>, a, reps)) return 1;
        if (run("F1c-L2", k_fold<1, 1, 1, 0, 1>, a, reps)) return 1;
        if (run("16x16 g", k_fold<1, 1, 1, 0, 0, 1, 16, 0>, a, reps)) return 1;
        if (run("16x16 c", k_fold<1, 1, 1, 0, 0, 1, 16, 1>, a, reps)) return 1;
        if (run("32x8 g", k_fold<1, 1, 1, 0, 0, 1, 32, 0>, a, reps)) return 1;
        if (run("32x8 c", k_fold<1, 1, 1, 0, 0, 1, 32, 1>, a, reps)) return 1;
        if (run("32x8 c nt", k_fold<1, 1, 1, 1, 0, 1, 32, 1>, a, reps)) return 1;
        if (run("64x4 c", k_fold<1, 1, 1, 0, 0, 1, 64, 1>, a, reps)) return 1;
        if (run("8x32 g", k_fold<1, 1, 1, 0, 0, 1, 8, 0>, a, reps)) return 1;
        if (run("8x32 c", k_fold<1, 1, 1, 0, 0, 1, 8, 1>, a, reps)) return 1;
        if (run("4x64 g", k_fold<1, 1, 1, 0, 0, 1, 4, 0>, a, reps)) return 1;
        // second lever: decimation-in-time butterflies with merged twiddles = fewer vector instructions, everything else unchanged
        if (run("F0 -6%", (k_fold<0, 0, 0, 0, 0, 1, 16, 2, 94>), a, reps)) return 1;
        if (run("F0 -12%", (k_fold<0, 0, 0, 0, 0, 1, 16, 2, 88>), a, reps)) return 1;
        if (run("F0", k_fold<0, 0>, a, reps)) return 1;
    }
    return 0;
}
