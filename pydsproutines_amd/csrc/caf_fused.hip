// Fused hypothesis kernel for gfx950: spectral conjugate-multiply -> 16384-point inverse FFT held
// entirely in LDS -> |.|^2, one workgroup per (rx block, group of hypotheses).
//
// Why: with rocFFT in the middle, every hypothesis product makes four HBM passes
// (multiply write, two FFT passes read+write, |.|^2 read) = ~55 B per CAF cell.  Here the product
// never leaves the CU: the block spectrum X stays in registers for the whole hypothesis loop, the
// template-spectrum row is read from L2 and multiplied one hypothesis ahead, the 16384-point transform runs
// in the 160 KB LDS of a CDNA4 CU, and only |y|^2 (4 B per cell, coalesced 256-B rows) goes to HBM.
// A second kernel (k_transpose_norm_argmax) turns the hypothesis-major |y|^2 tiles into the
// delay-major QF^2 surface + per-delay argmax + peak.  Replaces the same reference stages as
// caf_kernels.hip (multiplySlices.cu:206-211, cuFFT, complex_magn.cu:8-19, argmax.cu:93-153).
//
// Transform: y[n] = sum_m P[m] e^{+j 2 pi m n / B},  B = 16384 = 16 * 16 * 16 * 4, decimation as
//   m = 1024 a + 64 b + 4 c + d      (a,b,c in [0,16), d in [0,4))   input index
//   n = n1 + 16 n2 + 256 n3 + 4096 n4 (n1,n2,n3 in [0,16), n4 in [0,4)) output index
// pass 1: DFT16 over a  (butterfly <-> m2 = 64 b + 4 c + d), twiddle e^{j2pi m2 n1 / 16384}
// pass 2: DFT16 over b  (butterfly <-> n1, 4c+d),            twiddle e^{j2pi (4c+d) n2 / 1024}
// pass 3: DFT16 over c  (butterfly <-> n1, n2, d),           twiddle e^{j2pi d n3 / 64}
// pass 4: DFT4  over d  (butterfly <-> n1, n2, n3) -> y, lanes <-> consecutive n
// Passes 2 and 3 read and write the same LDS addresses per butterfly (in place), so only one
// barrier per pass is needed.  LDS image: element (n1, row, col) at n1*1090 + row*68 + col
// (complex64); the 68/1090 pitches keep the strided reads of passes 3 and 4 off the same banks.
// Default: 1024 threads (4 waves per SIMD, all 128 VGPRs, no spills), one butterfly per thread and pass; the
// 512-thread / two-butterfly variant is kept as an A/B switch (measured 20 % slower).
// The same hypothesis loop (fused_item) also runs inside k_caf_persistent (end of this file), which
// overlaps it with the transpose stage on other CUs in a single work-queue launch.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "caf_internal.h"
#include "caf_fft_dev.h"

namespace caf {

constexpr int FB = 16384;              // fused block size
// LDS image of the transform: PLANAR -- a real and an imaginary plane, each cut into 16 regions (one per n1 = the wave
// that owns it in passes 2 and 3) of 16 rows of 64 floats.  Every exchange WRITES whole rows with ds_write_addtid_b32
// (address = M0 + 16-bit immediate + 4 * lane: no address register, one source dword -> 2 cycles of the SIMD's
// LDS path per instruction, against ~9 for the ds_write_b64 of the complex image this replaces: measured with
// scripts/ubench/fft_struct_model.hip, -6 % per transform) and READS 16-byte pieces (ds_read_b128, four consecutive
// values of the digit the next pass transforms).  Conflict-free by construction:
//   exchange 1 (pass 1 -> 2): row = writer wave, position = writer lane; the reader's 16-lane groups see 16 different
//                             16-byte slots of a 256-byte row;
//   exchange 2 (pass 2 -> 3, inside one wave): rows start at fp_row2(n2): readers of one group differ in n2 and d, the
//                             row starts spread n2 over the four 64-byte quarters of the bank window;
//   exchange 3 (pass 3 -> 4): readers of one group differ in n1: the region pitch is 32 mod 256 bytes.
constexpr int FP_P1 = 4384;              // bytes per (plane, n1) region: 4288 used by the skewed rows of exchange 2
constexpr int FP_IM = 16 * FP_P1;        // byte offset of the imaginary plane
constexpr int F_LDS_BYTES = 2 * FP_IM;   // 140 288
constexpr int F_LDS_DATA = F_LDS_BYTES / 8;  // in complex elements (the image is declared as float2)
__host__ __device__ constexpr int fp_row2(int n2) { return ((n2 >> 1) & 3) * 1088 + ((n2 & 1) + 2 * (n2 >> 3)) * 256; }
static_assert(fp_row2(15) + 256 <= FP_P1 && FP_P1 % 256 == 32, "planar image geometry");

// Global-memory accessors: uniform base + 32-bit BYTE offset lets the compiler pick the scalar-base (saddr)
// addressing form instead of building a 64-bit address per access.  The explicit address-space-1 casts keep
// the accesses global_* (not flat_*) even where the base pointer was rebuilt from scalar halves
// (uniform_ptr below), whose provenance the address-space inference cannot see.
#define CAF_AS1 __attribute__((address_space(1)))
// ... and the constant address space for read-only values at wave-uniform addresses: SCALAR loads (s_load, counted by lgkmcnt).
// A vector load of such a value inside a hypothesis loop is followed by s_waitcnt vmcnt(0) -- which also waits for every
// tile store still on its way to memory, and then for the load's own round trip.
#define CAF_AS4 __attribute__((address_space(4)))
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 ld2(const float2* base, uint32_t elem) {
    const uint64_t u = *reinterpret_cast<const CAF_AS1 uint64_t*>((const CAF_AS1 char*)base + (elem << 3));
    float2 r;
    __builtin_memcpy(&r, &u, 8);
    return r;
}
__device__ __forceinline__ float4 gld4(const float* base, uint32_t byteoff) {
    const v4f_t v = *reinterpret_cast<const CAF_AS1 v4f_t*>((const CAF_AS1 char*)base + byteoff);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float gld1(const float* base, uint32_t byteoff) {
    return *reinterpret_cast<const CAF_AS1 float*>((const CAF_AS1 char*)base + byteoff);
}
// write-through store (sc1): visible device-wide once it has completed, without an L2 write-back fence
__device__ __forceinline__ void gst1_wt(float* base, uint32_t byteoff, float v) {
    __hip_atomic_store(reinterpret_cast<CAF_AS1 float*>((CAF_AS1 char*)base + byteoff), v, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// ---- planar image access ----
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) v4f_t lds_v4f;
// eight rows of one plane: lane l of the wave writes v_k to byte m0 + O_k + 4 l.  m0 must be wave-uniform.  (One
// wait state between the scalar write of M0 and an add-TID LDS instruction; the compiler does not look inside.)
template <int O0, int O1, int O2, int O3, int O4, int O5, int O6, int O7>
__device__ __forceinline__ void lds_rows8(uint32_t m0, float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                          float v7) {
    asm volatile(
        "s_mov_b32 m0, %0\n\ts_nop 0\n\t"
        "ds_write_addtid_b32 %1 offset:%9\n\tds_write_addtid_b32 %2 offset:%10\n\t"
        "ds_write_addtid_b32 %3 offset:%11\n\tds_write_addtid_b32 %4 offset:%12\n\t"
        "ds_write_addtid_b32 %5 offset:%13\n\tds_write_addtid_b32 %6 offset:%14\n\t"
        "ds_write_addtid_b32 %7 offset:%15\n\tds_write_addtid_b32 %8 offset:%16"
        :
        : "s"(m0), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "n"(O0), "n"(O1), "n"(O2), "n"(O3),
          "n"(O4), "n"(O5), "n"(O6), "n"(O7)
        : "memory");
}
// sixteen complex values -> rows R(k) = B + k * S of both planes (k = register index)
template <int S>
__device__ __forceinline__ void lds_rows16c(uint32_t m0, const float2 (&v)[16]) {
    lds_rows8<0, S, 2 * S, 3 * S, 4 * S, 5 * S, 6 * S, 7 * S>(m0, v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x);
    lds_rows8<0, S, 2 * S, 3 * S, 4 * S, 5 * S, 6 * S, 7 * S>(m0 + 8 * S, v[8].x, v[9].x, v[10].x, v[11].x, v[12].x, v[13].x, v[14].x, v[15].x);
    lds_rows8<0, S, 2 * S, 3 * S, 4 * S, 5 * S, 6 * S, 7 * S>(m0 + FP_IM, v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y);
    lds_rows8<0, S, 2 * S, 3 * S, 4 * S, 5 * S, 6 * S, 7 * S>(m0 + FP_IM + 8 * S, v[8].y, v[9].y, v[10].y, v[11].y, v[12].y, v[13].y, v[14].y, v[15].y);
}
// exchange 2: row n2 starts at fp_row2(n2)
__device__ __forceinline__ void lds_rows16c_x2(uint32_t m0, const float2 (&v)[16]) {
    lds_rows8<fp_row2(0), fp_row2(1), fp_row2(2), fp_row2(3), fp_row2(4), fp_row2(5), fp_row2(6), fp_row2(7)>(
        m0, v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x);
    lds_rows8<fp_row2(8), fp_row2(9), fp_row2(10), fp_row2(11), fp_row2(12), fp_row2(13), fp_row2(14), fp_row2(15)>(
        m0, v[8].x, v[9].x, v[10].x, v[11].x, v[12].x, v[13].x, v[14].x, v[15].x);
    lds_rows8<fp_row2(0), fp_row2(1), fp_row2(2), fp_row2(3), fp_row2(4), fp_row2(5), fp_row2(6), fp_row2(7)>(
        m0 + FP_IM, v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y);
    lds_rows8<fp_row2(8), fp_row2(9), fp_row2(10), fp_row2(11), fp_row2(12), fp_row2(13), fp_row2(14), fp_row2(15)>(
        m0 + FP_IM, v[8].y, v[9].y, v[10].y, v[11].y, v[12].y, v[13].y, v[14].y, v[15].y);
}
// four consecutive values of both planes at byte offsets `off` / `off_im` (= fp_im(base) + the same constant) -> four complex numbers
__device__ __forceinline__ void lds_get4c(const lds_char* img, uint32_t off, uint32_t off_im, float2& c0, float2& c1, float2& c2,
                                          float2& c3) {
    const v4f_t re = *reinterpret_cast<const lds_v4f*>(img + off);
    const v4f_t im = *reinterpret_cast<const lds_v4f*>(img + off_im);
    c0 = make_float2(re.x, im.x);
    c1 = make_float2(re.y, im.y);
    c2 = make_float2(re.z, im.z);
    c3 = make_float2(re.w, im.w);
}
// twiddle-table read (LDS), volatile so that it stays one ds_read_b64: the compiler otherwise pairs these reads into
// ds_read2_b64 / ds_read2st64_b64, which cost 8 LDS cycles per pair against 2 + 2 (MI355X_MICROARCH.md, LDS table);
// A/B on one box: -1.2 % per transform
typedef const volatile __attribute__((address_space(3))) uint64_t* lds_tw_ptr;
__device__ __forceinline__ lds_tw_ptr tw_base(const float2* t, uint32_t first) {  // &t[first]: the per-thread part
    // as an opaque register: the table lies beyond the 16-bit immediate range of the DS instructions, and left to itself
    // the compiler forms every element's address with an instruction of its own (15 v_or per pass)
    uint32_t a = (uint32_t)(uintptr_t)(lds_tw_ptr)(&t[first]);
    asm volatile("" : "+v"(a));
    return (lds_tw_ptr)(uintptr_t)a;
}
// the imaginary plane's copy of a read base (FP_IM is beyond the immediate range too: one add per pass, not per read)
__device__ __forceinline__ uint32_t fp_im(uint32_t off) {
    uint32_t a = off + (uint32_t)FP_IM;
    asm volatile("" : "+v"(a));
    return a;
}
__device__ __forceinline__ float2 tw_ld(lds_tw_ptr b, int k) {  // b[k], k a compile-time constant: an immediate offset
    const uint64_t u = b[k];
    float2 r;
    __builtin_memcpy(&r, &u, 8);
    return r;
}
// Thread <-> data of the four passes (tid = lane + 64 wave):
//   pass 1: butterfly m2 = 64 b + 4 c + d with b = (lane & 3) + 4 (wave & 3), c = ((lane >> 2) & 3) + 4 (wave >> 2),
//           d = lane >> 4 -- the four b of a wave are four whole 128-byte lines of the block spectrum / template row,
//           and the low lane bits carry the digit pass 2 transforms (its 16-byte reads);
//   pass 2: wave = n1, lane = (c & 3) + 4 d + 16 (c >> 2);   pass 3: wave = n1, lane = d + 4 n2;
//   pass 4: n1 = tid & 15, n2 = (tid >> 4) & 15, n3 = (tid >> 8) + 4 i  (lanes <-> consecutive delays, as before).
// (fp_m2 / fp_tid_of themselves: caf_fft_dev.h -- the 32768-point block-spectra kernel writes this order too)
// ... and the thread whose pass-1 butterfly is m2 (the inverse of fp_m2).  The template-spectrum rows are STORED in this
// "butterfly order" -- element m of a row at (m & ~1023) + fp_tid_of(m & 1023), k_butterfly_order below -- so that the
// lanes of a wave, whose butterflies are not consecutive, still read consecutive addresses (an unshifted row: 512
// contiguous bytes per load; a row shifted by the hypothesis: a few pieces of whole quads).  In natural order the same
// loads touched 64 separate 8-byte pieces and the launch took 18.4 instead of 15.0 ms.  The block spectrum stays in
// natural order: it is read once per work item, not once per transform.
// index of element (1024 a + m2 - shift) mod 16384 of a butterfly-ordered row, as hb + 1024 a (mod 16384)
__device__ __forceinline__ uint32_t fp_hbase(uint32_t m2, int32_t shift) {
    const uint32_t t = m2 - (uint32_t)shift;
    return (t & ~1023u) + fp_tid_of(t & 1023u);
}
__device__ __forceinline__ uint32_t fp_rd2(uint32_t tid) {  // pass-2 read base: + 256 bh
    const uint32_t lane = tid & 63, wave = tid >> 6;
    return wave * FP_P1 + (lane >> 4) * 1024 + (lane & 15) * 16;
}
__device__ __forceinline__ uint32_t fp_cd2(uint32_t tid) {  // pass-2 twiddle column 4 c + d
    const uint32_t lane = tid & 63;
    return 4 * ((lane & 3) + 4 * (lane >> 4)) + ((lane >> 2) & 3);
}
__device__ __forceinline__ uint32_t fp_rd3(uint32_t tid) {  // pass-3 read base: + 64 ch
    const uint32_t lane = tid & 63, wave = tid >> 6, n2 = lane >> 2;
    return wave * FP_P1 + ((n2 >> 1) & 3) * 1088 + ((n2 & 1) + 2 * (n2 >> 3)) * 256 + (lane & 3) * 16;
}
__device__ __forceinline__ uint32_t fp_rd4(uint32_t tid) {  // pass-4 read base: + 1024 i
    return (tid & 15) * FP_P1 + (tid >> 8) * 256 + ((tid >> 4) & 15) * 16;
}

// a pointer whose value is wave-uniform, moved (back) into scalar registers
template <typename Tp>
__device__ __forceinline__ Tp* uniform_ptr(Tp* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<Tp*>(((uint64_t)hi << 32) | lo);
}
// raw buffer descriptor (gfx9 family: DATA_FORMAT = 32 in word 3) over [base, base + bytes)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_of(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// sc1 in the cache-policy operand of buffer instructions: device-scope coherent (write-through store / load that is not
// served from a stale line of this XCD's L2)
constexpr int CAF_AUX_SC1 = 16;
// nt in the same operand: a streaming hint on top of the scope bits.  The |y|^2 tiles are written once and read once, the
// surface is written once: marking the three accesses non-temporal (the scope of the tile hand-off stays sc1) keeps
// them from displacing the template-spectrum rows and block spectra in the L2s -- measured on one box, A/B/A/B:
// 14.68 -> 14.34 ms per C2 launch with all three, 14.50 with two of them, 14.64 with the surface stores alone.
constexpr int CAF_AUX_NT = 2;
// A |y|^2 tile store: uniform descriptor + uniform (SGPR) offset + one per-thread 32-bit offset.  The descriptor covers
// exactly the valid tiles of the block, so a store to a tile past the end (the last 63 of the 256 tiles of a block whose
// step is 12289 delays) is dropped by the hardware range check -- which on gfx950 includes the SGPR offset
// (scripts/ubench/buffer_bounds.hip) -- instead of being branched around 16 times per hypothesis.
template <int MODE>
__device__ __forceinline__ void tile_store(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, float val) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, val), r, (int)voff, (int)soff, MODE == 1 ? (CAF_AUX_SC1 | CAF_AUX_NT) : 0);
}

// |y|^2 tiles: vt[blk_local][s_tile][h][64]  (s_tile = delay/64 inside the block)
// FT threads per workgroup (512: 2 waves/SIMD, 256-VGPR budget; 1024: 4 waves/SIMD, 128 VGPRs);
// BPT = 1024 / FT radix-16 butterflies per thread and pass.
//
// fused_item: the whole hypothesis loop of one (rx block, hypothesis group [h0, h1)) work item, run by all FT
// threads of a workgroup.  s_d / s_tw2 / s_tw3 are the caller's LDS arrays (the tables already loaded and
// published by a barrier, or by the first barrier inside the loop).  Shared by the one-item-per-workgroup
// kernel k_fused_caf and by the work-queue kernel k_caf_persistent.
// MODE 0: |y|^2 tiles with plain stores; 1: with write-through (sc1) stores; 2: no tiles at all -- every thread
// keeps the running maximum (value + hypothesis) of its 16 delays over the item's hypotheses and writes one
// (value, hypothesis) pair per delay at the end (callers that want no surface: per-delay traces and peaks only).
// MODE 3 (no frequency scan, one hypothesis per template: config C3, TemplateCrossCorrelator-style banks): every |y|^2
// IS a per-delay result, so the FFT item normalises it and writes it where the caller wants it -- row_max (T, S) and / or
// the (T, S, 1) surface -- instead of a tile that a second role reads back, scales and writes again.
// MODE 4 (complex QF rows: caf_outputs::d_cqf, TemplateCrossCorrelator.correlate / fastXcorr(absResult=False),
// xcorrRoutines.py:352-357, :533-548): the hypothesis-major complex plane [T*F][num_shifts] is exactly what an FFT item
// produces -- one transform = one row segment -- so the item writes y * sqrt(1/energy) * sqrt(1/||t||^2) as complex64
// itself; no tiles, no tile role, any number of frequencies.
// MODE 5 (hypothesis-major QF^2 surface: caf_outputs::d_surface_t [T][F][num_shifts]): MODE 2 and MODE 3 together -- every
// transform's 12288 delays are one contiguous row segment of that layout, so the item normalises and writes them itself
// (no tiles, no transposition) and keeps the running per-delay maxima of the NORMALISED values it has just formed, so that
// row_arg is the first maximum of the written values bit for bit, inside a group and -- reduce_wave_nosurf -- across groups.
struct F1Direct {
    float* out0;          // row_max or surface, [T][num_shifts]; MODE 4: the complex plane (2 floats per value)
    float* out1;          // the other one of the two when both are wanted, else nullptr
    const float* inv_e;   // [num_shifts] 1 / window energy
    const float* tscale;  // [T]
    int64_t num_shifts;
    int32_t step, blk_abs;
    // PK: one peak record per (template, block, wave) from the finished values themselves (no pass over the rows)
    PeakRec* partial;      // [T][ppt]; this item's records at [h][16 blk_abs + wave]
    int64_t ppt, shift_start;
};
// NV4: output quarters n4 < NV4 of the last (radix-4) pass are computed -- the others hold no valid delay (tiles >= 64 NV4)
// PK (MODE 3): the item also leaves the peak of every row segment it writes -- value and first delay of the maximum
template <int FT, int MODE = 0, int NV4 = 4, bool PK = false>
__device__ __forceinline__ void fused_item(float2* __restrict__ s_d, const float2* __restrict__ s_tw2,
                                           const float2* __restrict__ s_tw3,
                                           const float2* __restrict__ xb,       // [blocks][FB] spectra
                                           const float2* __restrict__ hc,       // [T][FB] or [T*F][FB]
                                           const int32_t* __restrict__ shifts,  // [F] (shift modes)
                                           const float2* __restrict__ tw1,      // [16][1024]
                                           int32_t table_mode, int32_t nfreq, int32_t nhyp, int blk, int h0, int h1,
                                           int32_t tiles_per_blk, float* __restrict__ vt,
                                           int32_t* __restrict__ imax = nullptr, const F1Direct* f1 = nullptr) {
    static_assert(FT == 1024, "one radix-16 butterfly per thread and pass: 1024 threads (wave w owns plane n1 = w)");
    constexpr int BPT = 1;
    const int tid = threadIdx.x;
    const lds_char* img = (const lds_char*)s_d;
    // wave-uniform LDS byte addresses for the row writes (M0): this wave's row of exchange 1, its region for 2 and 3
    const uint32_t img0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)img);
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane((uint32_t)tid >> 6);
    const uint32_t m0_x1 = img0 + wave_u * 256u, m0_w = img0 + wave_u * (uint32_t)FP_P1;
    const uint32_t m2 = fp_m2((uint32_t)tid);
    // outputs n = n' + 4096 n4 lie in tiles 64 n4 ...: quarters of the last (radix-4) pass that hold no valid delay at
    // all are not computed (N = 4096 with the step rounded to 12288: n4 = 3; N = 8192: n4 = 2 and 3; -3.3 % per transform
    // at NV4 = 3)
    constexpr int nv4 = NV4;  // (compile-time: as a run-time scalar the 16 branches per transform ate the 3 % it saves)
    float bv[16];     // MODE 2 / 5: running maxima of this thread's 16 delays ...
    uint32_t bi[4];   // ... and the item-local hypothesis (8 bits each) that produced them
    float ge[16];     // MODE 3 / 4 / 5: 1 / window energy of this thread's 16 delays
    if (MODE == 2 || MODE == 5) {
#pragma unroll
        for (int o = 0; o < 16; ++o) bv[o] = -1.f;
#pragma unroll
        for (int o = 0; o < 4; ++o) bi[o] = 0u;
    }
    // MODE 3+: 1 / window energy of this thread's 16 delays, the valid extent of the block
    uint32_t f1_bytes = 0;
    int64_t f1_rel0 = 0;
    if (MODE >= 3) {
        f1_rel0 = (int64_t)f1->blk_abs * f1->step;
        int64_t nv = f1->num_shifts - f1_rel0;
        if (nv > f1->step) nv = f1->step;
        f1_bytes = (uint32_t)nv * 4u;
        const __amdgpu_buffer_rsrc_t rie = buf_of(uniform_ptr(f1->inv_e + f1_rel0), f1_bytes);  // delays past the end: 0
        const int n1 = tid & 15, n2 = (tid >> 4) & 15, q = tid >> 8;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4)
                ge[4 * i + n4] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(rie, (((n2 >> 2) + 4 * q) * 64 + n1 + 16 * (n2 & 3)) * 4,
                                                                (16 * i + 64 * n4) * 256, 0));
        if (MODE == 4) {  // amplitudes, not powers: sqrt(1 / energy) per delay
#pragma unroll
            for (int o = 0; o < 16; ++o) ge[o] = __builtin_sqrtf(ge[o]);
        }
    }

    // PK: this thread's part of the block-local delay index of its outputs (the rest is 64 (16 i + 64 n4)), valid extent
    uint32_t pk_d0 = 0, pk_nv = 0;
    if (PK) {
        const int n1 = tid & 15, n2 = (tid >> 4) & 15, q = tid >> 8;
        pk_d0 = (uint32_t)(((n2 >> 2) + 4 * q) * 64 + n1 + 16 * (n2 & 3));
        pk_nv = f1_bytes >> 2;
    }
    // hypothesis-independent per-thread state: the pass-1 twiddle base e^{+j 2 pi m2 / 16384}
    float2 w[BPT];
    const float2* xp = xb + (int64_t)blk * FB;  // uniform base; per-thread offsets stay 32-bit (saddr loads)
#pragma unroll
    for (int j = 0; j < BPT; ++j) w[j] = ld2(tw1, 1024u + m2);
    float* vt_blk = vt + (int64_t)blk * tiles_per_blk * nhyp * 64;
    const __amdgpu_buffer_rsrc_t rvt = buf_of(uniform_ptr(vt_blk), (uint32_t)tiles_per_blk * (uint32_t)nhyp * 256u);

    // row of the template-spectrum table used by hypothesis h (uniform) and its circular shift
    const float2* hrow_cur;
    uint32_t hb_cur;  // this thread's element of the (butterfly-ordered, shifted) row: hb_cur + 1024 a (mod FB)
    auto row_of = [&](int h) {
        if (table_mode) {
            hrow_cur = hc + (int64_t)h * FB;
            hb_cur = (uint32_t)tid;
        } else {
            const int t = h / nfreq;
            const int32_t sh = *((const CAF_AS4 int32_t*)shifts + (h - t * nfreq));
            hrow_cur = hc + (int64_t)t * FB;
            hb_cur = fp_hbase(m2, sh);
        }
    };
    // pr[j][a] = X[1024 a + m2] * Hc_h[1024 a + m2]: the input of pass 1, produced one hypothesis ahead
    // (during pass 4 of the previous one) from the persistent X registers and a freshly loaded Hc row.
    float2 pr[BPT][16];
    float2 xr[BPT][16];  // block spectrum, persistent
#pragma unroll
    for (int j = 0; j < BPT; ++j)
#pragma unroll
        for (int a = 0; a < 16; ++a) xr[j][a] = ld2(xp, 1024u * a + m2);
    row_of(h0);
#pragma unroll
    for (int j = 0; j < BPT; ++j)
#pragma unroll
        for (int a = 0; a < 16; ++a)
            pr[j][a] = cmul(xr[j][a],
                            ld2(hrow_cur, (1024u * a + hb_cur) & (FB - 1)));

    for (int h = h0; h < h1; ++h) {
        const bool more = h + 1 < h1;
        // h*64 as an opaque scalar: otherwise loop-strength-reduction turns the 32 store addresses of pass 4
        // into 32 64-bit induction variables (64 VGPRs + 32 adds per hypothesis)
        uint32_t hoff = (uint32_t)h * 256u;  // bytes
        asm volatile("" : "+s"(hoff));
        int lz = 0;  // an opaque zero added to loop-invariant LDS table indices (see pass 1)
        asm volatile("" : "+v"(lz));
        // MODE 3: this hypothesis' (= template's) output rows and scale
        __amdgpu_buffer_rsrc_t f1_r0 = rvt, f1_r1 = rvt;
        float f1_ts = 0.f;
        if (MODE == 3) {
            f1_ts = *((const CAF_AS4 float*)f1->tscale + h);
            f1_r0 = buf_of(uniform_ptr(f1->out0 + (int64_t)h * f1->num_shifts + f1_rel0), f1->out0 ? f1_bytes : 0u);  // past the block's
            if (f1->out1)                                                                            // delays: dropped
                f1_r1 = buf_of(uniform_ptr(f1->out1 + (int64_t)h * f1->num_shifts + f1_rel0), f1_bytes);
        }
        float pk_v[4] = {-1.f, -1.f, -1.f, -1.f};
        uint32_t pk_i[4] = {0u, 0u, 0u, 0u};
        if (MODE == 4) {
            f1_ts = __builtin_sqrtf(*((const CAF_AS4 float*)f1->tscale + h / nfreq));
            f1_r0 = buf_of(uniform_ptr(f1->out0 + 2 * ((int64_t)h * f1->num_shifts + f1_rel0)), 2u * f1_bytes);
        }
        if (MODE == 5) {  // row h = t * nfreq + f of the hypothesis-major surface
            f1_ts = *((const CAF_AS4 float*)f1->tscale + h / nfreq);
            f1_r0 = buf_of(uniform_ptr(f1->out0 + (int64_t)h * f1->num_shifts + f1_rel0), f1_bytes);
        }
        // ---- pass 1: P = X * Hc_h ; DFT16 over a ; twiddle w^n1 ; write A[n1][m2] ----
        // One butterfly at a time (32 live data registers).  pr is dead after this pass and is refilled
        // with the next hypothesis' products during pass 4, the low-pressure phase.
        // The register-only part (DFT16 + twiddles) runs BEFORE the barrier that protects the LDS image, so it
        // overlaps with slower waves still finishing pass 4 of the previous hypothesis.
        float2 v1[BPT][16];
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
#pragma unroll
            for (int a = 0; a < 16; ++a) v1[j][a] = pr[j][a];
            idft16(v1[j]);
            float2 p = w[j];
            // opaque to the optimiser: otherwise the 15 powers (and the LDS twiddles below) are hoisted out
            // of the hypothesis loop as loop invariants and cost ~180 persistent registers
            asm volatile("" : "+v"(p.x), "+v"(p.y));
            const float2 wj = p;
            v1[j][1] = cmul(v1[j][1], p);
#pragma unroll
            for (int n1 = 2; n1 < 16; ++n1) {
                p = cmul(p, wj);
                v1[j][n1] = cmul(v1[j][n1], p);
            }
        }
        __syncthreads();  // previous hypothesis' pass-4 reads are done (and the LDS tables are in place)
        lds_rows16c<FP_P1>(m0_x1, v1[0]);  // value n1 -> region n1, row = this wave, position = lane
        __syncthreads();
        row_of(more ? h + 1 : h);  // unconditional refill (the last one is redundant): no select keeps pr alive
        // Modes 2+ (no |y|^2 tiles: running maxima, finished rows, complex rows, hypothesis-major rows) have 20 registers
        // less than the tile modes, so the next template-spectrum row cannot be fetched whole before pass 3 as there.
        // EH positions of it are fetched at the start of pass 2 and EH more at the start of pass 3, each batch multiplied at
        // the end of its pass (the products wait in pr, dead since pass 1); the rest after pass 3 as before.  C2 without the
        // surface by EH, one box (profiles/r04/ab_early_template_row.log): 0 -> 9.32 ms, 4 -> 8.99, 5 -> 8.84, 6 -> 8.98,
        // 8 -> 9.59 (every early arrival beyond the free registers is parked in scratch); hypothesis-major surface
        // (20 more registers: the normalisation factors): 0 -> 11.04, 2 -> 10.78, 3 -> 10.69 with 11 spilled dwords per hypothesis -- 2 is kept.
        constexpr int EH = MODE == 5 ? 2 : MODE >= 2 ? 5 : 0;
        float2 he[2 * EH + 1];
        // Tile modes: the next row is fetched whole before pass 3 (below) -- except its first HP2 positions, which go out
        // here, a pass earlier: the same registers in flight at the peak, and the first products of pass 4 find their
        // operands there.  C2 with the surface by HP2, one box (profiles/r04/ab_template_row_before_pass2.log): 0 -> 13.80
        // ms, 2 -> 13.65, 3 -> 13.58, 4 -> 13.47, 5 -> 13.36, 6 -> 13.5, 8 -> 13.67, 16 -> 14.1.
        constexpr int HP2 = 5;
        float2 hn[BPT][16];
        if (MODE < 2) {
#pragma unroll
            for (int a = 0; a < HP2; ++a) hn[0][a] = ld2(hrow_cur, ((1024u * a + hb_cur) & (FB - 1)) + lz);
        }
#pragma unroll
        for (int a = 0; a < EH; ++a) he[a] = ld2(hrow_cur, ((1024u * a + hb_cur) & (FB - 1)) + lz);
        // ---- pass 2: DFT16 over b, in place (n1 = idx >> 6, col = idx & 63) ----
        // all of a thread's butterflies are read first, so the LDS reads of butterfly j+1 fly under the
        // arithmetic of butterfly j (they touch disjoint addresses)
        {
            float2 v[16];
            const uint32_t rd2 = fp_rd2((uint32_t)(tid + lz)), rd2i = fp_im(rd2);
            const lds_tw_ptr t2 = tw_base(s_tw2, fp_cd2((uint32_t)(tid + lz)));
#pragma unroll
            for (int bh = 0; bh < 4; ++bh) lds_get4c(img, rd2 + 256u * bh, rd2i + 256u * bh, v[4 * bh], v[4 * bh + 1], v[4 * bh + 2], v[4 * bh + 3]);
            idft16(v);
#pragma unroll
            for (int n2 = 1; n2 < 16; ++n2) v[n2] = cmul(v[n2], tw_ld(t2, n2 * 64));
            lds_rows16c_x2(m0_w, v);  // value n2 -> row n2 of this wave's region, position = lane
        }
#pragma unroll
        for (int a = 0; a < EH; ++a) pr[0][a] = cmul(xr[0][a], he[a]);
        // No workgroup barrier here: plane n1 = idx >> 6 is written in pass 2 and read in pass 3 by the SAME
        // wave (wave w owns planes w, w + FT/64, ...), and a wave's LDS operations complete in order.
        __builtin_amdgcn_wave_barrier();
        // next hypothesis' template-spectrum row: issued before pass 3 so that the L2 latency is covered by
        // the pass-3 butterfly and the pass-4 work
        if (MODE < 2) {
#pragma unroll
            for (int j = 0; j < BPT; ++j)
#pragma unroll
                for (int a = HP2; a < 16; ++a)
                    hn[j][a] = ld2(hrow_cur, (1024u * a + hb_cur) & (FB - 1));  // (hb_cur changes per hypothesis: nothing to hoist)
        }
#pragma unroll
        for (int a = EH; a < 2 * EH; ++a) he[a] = ld2(hrow_cur, ((1024u * a + hb_cur) & (FB - 1)) + lz);
        // ---- pass 3: DFT16 over c, in place (n1 = idx >> 6, n2 = (idx >> 2) & 15, d = idx & 3) ----
        {
            float2 v[16];
            const uint32_t rd3 = fp_rd3((uint32_t)(tid + lz)), rd3i = fp_im(rd3);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) lds_get4c(img, rd3 + 64u * ch, rd3i + 64u * ch, v[4 * ch], v[4 * ch + 1], v[4 * ch + 2], v[4 * ch + 3]);
            idft16(v);
            const lds_tw_ptr t3 = tw_base(s_tw3, (uint32_t)((tid & 3) + lz));
#pragma unroll
            for (int n3 = 1; n3 < 16; ++n3) v[n3] = cmul(v[n3], tw_ld(t3, n3 * 4));
            lds_rows16c<256>(m0_w, v);  // value n3 -> row n3 of this wave's region, position = lane = d + 4 n2
        }
#pragma unroll
        for (int a = EH; a < 2 * EH; ++a) pr[0][a] = cmul(xr[0][a], he[a]);
        if (MODE >= 2) {
            // (the 20 registers of the running maxima leave no room for the row during pass 3: it is fetched here,
            // under the barrier and the pass-4 work.  MODE 5, measured: a quarter / half / three quarters of the row
            // fetched before pass 3 instead cost 0.9 ms of 10.9 at C2 -- profiles/r04/ab_surface_t_early_row.log)
#pragma unroll
            for (int j = 0; j < BPT; ++j)
#pragma unroll
                for (int a = 2 * EH; a < 16; ++a)
                    hn[j][a] = ld2(hrow_cur, (1024u * a + hb_cur) & (FB - 1));  // (hb_cur changes per hypothesis: nothing to hoist)
        }
        __syncthreads();
        // ---- pass 4: DFT4 over d ; |y|^2 -> vt tiles (lanes <-> consecutive delays) ----
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            const int idx = tid + j * FT;
            const int n1 = idx & 15, n2 = (idx >> 4) & 15, q = idx >> 8;
            const uint32_t rd4 = fp_rd4((uint32_t)idx), rd4i = fp_im(rd4);
            // The next hypothesis' products X * Hc (its row arrived during pass 3) fill the wait for the first LDS reads of
            // this pass -- right after the barrier every wave of the SIMD would otherwise be waiting for data.  (Modes 2+
            // fetch the row after pass 3: there the products stay at the end of the pass.)
            __builtin_amdgcn_sched_barrier(0);
            float2 pf0, pf1, pf2, pf3;
            if (MODE < 2) {
                lds_get4c(img, rd4, rd4i, pf0, pf1, pf2, pf3);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 16; ++a) pr[j][a] = cmul(xr[j][a], hn[j][a]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t lzi = 0;
                asm volatile("" : "+v"(lzi));
                float2 a0, a1, a2, a3;
                if (MODE < 2 && i == 0)
                    a0 = pf0, a1 = pf1, a2 = pf2, a3 = pf3;
                else
                    lds_get4c(img, rd4 + 1024u * i + lzi, rd4i + 1024u * i + lzi, a0, a1, a2, a3);  // n3 = q + 4 i: the four d of (n1, n2, n3)
                // inverse DFT4 over d, one output quarter n4 at a time (idft4 spelled out: unused quarters are skipped)
                const float2 s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = mulj(csub(a1, a3));
                // n = n1 + 16 n2 + 256 n3 + 4096 n4  ->  tile = n >> 6 = (n2 >> 2) + 4 n3 + 64 n4, lane = n & 63.
                // Uniform (scalar) part of the address + one 32-bit per-thread offset, so that no per-store
                // 64-bit address is kept alive across the hypothesis loop.
#pragma unroll
                for (int n4 = 0; n4 < 4; ++n4) {
                    if (n4 >= nv4) continue;  // (scalar branch)
                    const float2 yq = n4 == 0 ? cadd(s02, s13) : n4 == 1 ? cadd(d02, d13) : n4 == 2 ? csub(s02, s13) : csub(d02, d13);
                    const int tile_u = 16 * i + 64 * n4;                      // uniform part of the tile index
                    const int tile_t = (n2 >> 2) + 4 * q;                     // per-thread part
                    const uint32_t soff = (uint32_t)tile_u * (uint32_t)nhyp * 256u + hoff;  // uniform (scalar) offset
                    const uint32_t voff = ((uint32_t)tile_t * (uint32_t)nhyp * 64u + (uint32_t)(n1 + 16 * (n2 & 3))) << 2;
                    const float val = __builtin_fmaf(yq.x, yq.x, yq.y * yq.y);  // (explicit: every mode must round it the same way)
                    if (MODE == 2) {
                        // hypotheses come in increasing order: the first maximum stays (NaN never enters)
                        constexpr int o = 0;  // (placeholder, see below)
                        (void)o;
                        const int oo = 4 * i + n4;
                        const bool up = val > bv[oo];
                        bv[oo] = up ? val : bv[oo];
                        const uint32_t sh8 = 8u * (uint32_t)(oo & 3);
                        const uint32_t repl = (bi[oo >> 2] & ~(0xffu << sh8)) | ((uint32_t)(h - h0) << sh8);
                        bi[oo >> 2] = up ? repl : bi[oo >> 2];
                    } else if (MODE == 5) {
                        // the finished value, rounded as every surface path rounds it; its running maximum is therefore
                        // the maximum of the WRITTEN values (hypotheses in increasing order: the first one stays; a NaN
                        // -- zero-energy window -- never enters)
                        const int oo = 4 * i + n4;
                        const float outv = val * (ge[oo] * f1_ts);
                        const bool up = outv > bv[oo];
                        bv[oo] = up ? outv : bv[oo];
                        const uint32_t sh8 = 8u * (uint32_t)(oo & 3);
                        const uint32_t repl = (bi[oo >> 2] & ~(0xffu << sh8)) | ((uint32_t)(h - h0) << sh8);
                        bi[oo >> 2] = up ? repl : bi[oo >> 2];
                        const uint32_t v3 = (uint32_t)(tile_t * 64 + n1 + 16 * (n2 & 3)) << 2, s3 = (uint32_t)tile_u * 256u;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, outv), f1_r0, (int)v3, (int)s3, CAF_AUX_NT);
                    } else if (MODE == 4) {
                        const float gq = ge[4 * i + n4] * f1_ts;
                        const v2f_t ov = {yq.x * gq, yq.y * gq};
                        const uint32_t v3 = (uint32_t)(tile_t * 64 + n1 + 16 * (n2 & 3)) << 3, s3 = (uint32_t)tile_u * 512u;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i_t, ov), f1_r0, (int)v3, (int)s3, CAF_AUX_NT);
                    } else if (MODE == 3) {
                        // the finished per-delay value, rounded as the tile roles round it: value * (1/energy * 1/||t||^2)
                        const float outv = val * (ge[4 * i + n4] * f1_ts);
                        if (PK) {
                            // delays of one output quarter n4 are visited in increasing order (i): a strict comparison keeps
                            // the first maximum; NaN (zero-energy window) and delays past the block's valid extent never enter
                            const bool up = ((uint32_t)(pk_d0 + (16 * i + 64 * n4) * 64) < pk_nv) & (outv > pk_v[n4]);
                            pk_v[n4] = up ? outv : pk_v[n4];
                            pk_i[n4] = up ? (uint32_t)i : pk_i[n4];
                        }
                        const uint32_t v3 = (uint32_t)(tile_t * 64 + n1 + 16 * (n2 & 3)) << 2, s3 = (uint32_t)tile_u * 256u;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, outv), f1_r0, (int)v3, (int)s3, CAF_AUX_NT);
                        if (f1->out1)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, outv), f1_r1, (int)v3, (int)s3, CAF_AUX_NT);
                    } else {
                        tile_store<MODE>(rvt, voff, soff, val);  // tiles >= tiles_per_blk: dropped by the range check
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the four sub-steps from being co-scheduled (registers)
            }
            if (MODE >= 2) {
#pragma unroll
                for (int a = 2 * EH; a < 16; ++a) pr[j][a] = cmul(xr[j][a], hn[j][a]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (PK) {
            // quarters in increasing delay order (strict: the lower quarter keeps a tie), then one unsigned maximum of
            // (value bits, ~delay) over the wave: values are >= +0, whose bit patterns order like the numbers
            float bvv = pk_v[0];
            uint32_t bq = 0, bii = pk_i[0];
#pragma unroll
            for (int n4 = 1; n4 < NV4; ++n4) {
                const bool up = pk_v[n4] > bvv;
                bvv = up ? pk_v[n4] : bvv;
                bq = up ? (uint32_t)n4 : bq;
                bii = up ? pk_i[n4] : bii;
            }
            const uint32_t dloc = pk_d0 + (16u * bii + 64u * bq) * 64u;
            unsigned long long key = bvv < 0.f ? 0ull : (((unsigned long long)__float_as_uint(bvv) << 32) | (uint32_t)~dloc);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned long long ok = __shfl_xor(key, o, 64);
                key = ok > key ? ok : key;
            }
            if ((tid & 63) == 0) {
                PeakRec r;
                r.v = key ? __uint_as_float((uint32_t)(key >> 32)) : -1.f;
                r.delay = key ? (int32_t)(f1->shift_start + f1_rel0 + (int64_t)(uint32_t)~(uint32_t)key) : 0x7fffffff;
                r.f = 0;
                f1->partial[(int64_t)h * f1->ppt + (int64_t)f1->blk_abs * 16 + (tid >> 6)] = r;
            }
        }
    }
    if (MODE == 2 || MODE == 5) {
        // one (value, hypothesis) pair per delay of the block, in 64-delay tiles: vt[tile][64], imax[tile][64]
        const int n1 = tid & 15, n2 = (tid >> 4) & 15, q = tid >> 8;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4) {
                const int tile = 16 * i + 64 * n4 + (n2 >> 2) + 4 * q;
                const int oo = 4 * i + n4;
                if (tile < tiles_per_blk) {
                    const uint32_t off = ((uint32_t)tile * 64u + (uint32_t)(n1 + 16 * (n2 & 3))) << 2;
                    gst1_wt(vt, off, bv[oo]);
                    __hip_atomic_store(reinterpret_cast<CAF_AS1 int32_t*>((CAF_AS1 char*)imax + off),
                                       (int32_t)(h0 + ((bi[oo >> 2] >> (8 * (oo & 3))) & 0xffu)), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
    }
}

// ----------------------------------------------------------------------------------------
// Templates of 8193 .. 32768 samples: the CHAINED role -- a 32768-point inverse transform as TWO 16384-point transforms in
// the same LDS image.
//   y[n] = sum_m P[m] W^{mn},  W = e^{+j 2 pi / 32768},  m = 2 m' + c:
//   y[n'] = E[n'] + W^{n'} O[n'],   y[n' + 16384] = E[n'] - W^{n'} O[n'],   E / O = IFFT_16384 of the even / odd samples
// E's sixteen outputs per thread wait in registers while O runs through the same four passes; both transforms leave a
// thread the same sixteen positions n' = n1 + 16 n2 + 256 n3 + 4096 n4, so the radix-2 combination is register-local and
// nothing but |y|^2 leaves the CU.  The combination twiddle W^{n'} is a product of one factor per output digit, and each
// pass of the O half applies the factor of the digit it produces: pass 1 through its recurrence base (2 m2 + 1 instead
// of 2 m2 in 32768ths), passes 2 and 3 through the second halves of the twiddle tables, pass 4 as the constants W_8^{n4}.
// Tiles: n -> tile n >> 6 (up to 384 tiles for N = 8193), write-through stores (MODE 1) or plain (0).
// NVH: quarters n4 < NVH of the UPPER half y[n' + 16384] hold valid delays (N = 16384: none -- the block's valid delays
// are exactly the lower half; N = 8193: two); the others are not combined, squared or stored.
//
// FOLD = false, templates of 8193 .. 16384 samples (32768-point blocks): the block spectra and the template-spectrum
// rows arrive PARITY-MAJOR ([c][16384] per row, each half in butterfly order: k_block_spectra32 / k_parity_major), so a
// half-transform reads what fused_item reads for a 16384-point block, with the (even) shift of an on-grid hypothesis
// halved.  A point = an 8-byte load of X and one of Hc (4 registers in flight), their product 2.
//
// FOLD = true, templates of 16385 .. 32768 samples: 65536-point blocks whose first 32768 outputs are the only valid delays
// (step 32768), split by the PARITY OF THE OUTPUT,
//   y[2 k + r] = sum_{m < 32768} G_r[m] W_32768^{m k},   G_r[m] = (Z[m] + (-1)^r Z[m + 32768]) W_65536^{m r},   Z = X . Hc
// -- one decimation-in-frequency step on the products -- and each residue r is a 32768-point transform of which only the
// lower half k < 16384 is wanted (NVH = 0): two chained 16384-point transforms per residue, FOUR per block and hypothesis.
// (Round 4's first form accumulated one output QUARTER from the four spectral residues m mod 4 and ran every block twice:
// eight sub-transforms for the same 32768 delays, 43 ms at the C2 shape against 24 for this one.)
// G_r[2 m' + c] needs P_c[m'] and P_c[m' + 16384] (P_c = the samples of parity c) of the block spectrum and of the template
// row: both arrive as PAIRS, [c][16384] x (P_c[m'], P_c[m' + 16384]) with every 1024-chunk of m' in butterfly order
// (k_parity_pairs), so a point is two 16-byte loads (8 registers in flight, folded 2).  A hypothesis shift s (in
// parity-major elements, mod 32768) moves the template pair to (m' - s) mod 16384 and swaps its halves where bit 14 of
// (m' - s) mod 32768 is set.  Input twiddle (r = 1): W_65536^{2 m' + c} with m' = 1024 a + m2 is a per-thread base
// e^{j 2 pi (2 m2 + c) / 65536} -- common to the sixteen inputs of the thread's pass-1 butterfly, so it moves behind the
// butterfly into the start of the pass-1 recurrence -- times the compile-time constants W_32^a, applied at the fold.
// Work item = (block, hypothesis group, r): the host doubles ngroups, group = 2 * (hypothesis group) + r.  Tiles: [r][256]
// per block, tile 256 r + (k >> 6), column k & 63 <-> delay 2 k + r: the tile roles run with a delay stride of 2 (DS).
//
// Registers (128) decide the schedule of the loads.  Neither the block spectrum nor a template row can stay resident, so
// every sub-transform's inputs are fetched while the one before it runs.  The E half of a hypothesis has room (its
// transform's 32 registers + addresses), the O half does not (E's 32 outputs wait beside the transform), so the 32 points
// a thread needs per hypothesis -- the O half's 16, then the next E half's 16 -- are fetched in quotas over the eight passes:
// each batch is issued at the start of its pass and multiplied / folded at its end, and nothing is in flight across a
// hypothesis boundary.  (Round 3's schedule -- O's inputs in one batch before pass 3 of E, the next E's inside pass 4 of O --
// left the second batch half a pass to arrive: N = 16384 at the C2 shape 22.2 -> 20.3 ms with the quotas.)
template <int N_> struct caf_ic { static constexpr int value = N_; };
template <int MODE, int NVH, bool FOLD, int R, bool PART = false>
__device__ __forceinline__ void fused_item2q(float2* __restrict__ s_d, const float2* __restrict__ s_tw2,
                                             const float2* __restrict__ s_tw3, const float2* __restrict__ xb,  // FOLD: [blocks][2][16384] pairs; else [blocks][2][16384]
                                             const float2* __restrict__ hc,       // rows of the same shape, per template or per hypothesis (PART: npart consecutive rows each)
                                             const int32_t* __restrict__ shifts, const float2* __restrict__ tw1,
                                             int32_t table_mode, int32_t nfreq, int32_t nhyp, int blk, int h0, int h1,
                                             int32_t tiles_per_blk, float* __restrict__ vt, int32_t npart = 1,
                                             const F1Direct* __restrict__ f1 = nullptr) {  // MODE 4: the complex-QF rows (no tiles)
    static_assert(FOLD || R == 0, "the output residue exists in the folded form only");
    static_assert(MODE == 1 || MODE == 4, "|y|^2 tiles (write-through) or complex-QF rows");
    static_assert(!FOLD || NVH == 0, "folded form: only the lower half of each residue's transform holds valid delays");
    static_assert(!PART || FOLD, "partitioned templates ride on the folded form");
    constexpr int ROW = FOLD ? 4 * FB : 2 * FB;  // float2 per row (block spectrum, template row)
    using pt_t = typename std::conditional<FOLD, float4, float2>::type;
    const int tid = threadIdx.x;
    const lds_char* img = (const lds_char*)s_d;
    const uint32_t img0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)img);
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane((uint32_t)tid >> 6);
    const uint32_t m0_x1 = img0 + wave_u * 256u, m0_w = img0 + wave_u * (uint32_t)FP_P1;
    const uint32_t m2 = fp_m2((uint32_t)tid);
    const float2 w = ld2(tw1, 1024u + m2);  // pass-1 twiddle base e^{+j 2 pi m2 / 16384}
    float2 base0 = make_float2(1.f, 0.f);   // r = 1: e^{+j 2 pi (2 m2) / 65536}
    if (R) {
        float sn, cs;
        sincospif((float)m2 * (1.0f / 16384.0f), &sn, &cs);
        base0 = make_float2(cs, sn);
    }
    const float* xp = (const float*)(xb + (int64_t)blk * ROW);
    float* vt_blk = vt + (int64_t)blk * tiles_per_blk * nhyp * 64;
    const __amdgpu_buffer_rsrc_t rvt = buf_of(uniform_ptr(vt_blk), (uint32_t)tiles_per_blk * (uint32_t)nhyp * 256u);
    const int n1o = tid & 15, n2o = (tid >> 4) & 15, qo = tid >> 8;  // this thread's pass-4 position
    // MODE 4 (TemplateCrossCorrelator.correlate, fastXcorr(absResult=False) beyond 8192 samples): the O half writes
    // y sqrt(1 / E) sqrt(1 / ||t||^2) as complex64 into row h of the plane itself.  The energies of the block's delays do not fit
    // the registers of this role: they are fetched per output (range-checked descriptor: 0 past the block's valid delays, where
    // the row descriptor drops the store), from the L2 after the item's first hypothesis.
    uint32_t f1_bytes = 0;
    int64_t f1_rel0 = 0;
    __amdgpu_buffer_rsrc_t rie = rvt, f1_r0 = rvt;
    float f1_ts = 0.f;
    if constexpr (MODE == 4) {
        f1_rel0 = (int64_t)f1->blk_abs * f1->step;
        int64_t nv = f1->num_shifts - f1_rel0;
        if (nv > f1->step) nv = f1->step;
        f1_bytes = (uint32_t)nv * 4u;
        rie = buf_of(uniform_ptr(f1->inv_e + f1_rel0), f1_bytes);
    }
    // template row of a hypothesis and this thread's base index into it: the shift in parity-major elements (= half the
    // block's shift); FOLD: mod 32768 -- bits 0..13 of 1024 a + hb select the pair, bit 14 swaps its halves
    const float* hrow_o;  // the row of the hypothesis in progress (its O half's inputs)
    uint32_t hb_o;
    const float* hrow_n;  // the next hypothesis' row (its E half's inputs)
    uint32_t hb_n;
    auto row_of = [&](int h, const float*& hrow, uint32_t& hb) {
        const int rows_per = PART ? npart : 1;
        if (table_mode) {
            hrow = (const float*)(hc + (int64_t)h * rows_per * ROW);
            hb = (uint32_t)tid;
        } else {
            const int t = h / nfreq;
            const int32_t sh = *((const CAF_AS4 int32_t*)shifts + (h - t * nfreq)) >> 1;
            hrow = (const float*)(hc + (int64_t)t * rows_per * ROW);
            hb = fp_hbase(m2, sh);
        }
    };
    // a batch: NP points a = A0 .. A0 + NP - 1 of parity CC, from row hrow / base hb, into bx / bh (lz: an opaque zero that
    // pins the loads behind their point of issue)
    auto issue = [&](pt_t* bx, pt_t* bh, const float* hrow, uint32_t hb, auto cc_, auto a0_, auto np_, uint32_t lz) __attribute__((always_inline)) {
        constexpr int CC = decltype(cc_)::value, A0 = decltype(a0_)::value, NP = decltype(np_)::value;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const uint32_t a = (uint32_t)(A0 + k);
            const uint32_t hi = (1024u * a + hb) & 16383u;
            if constexpr (FOLD) {
                bx[k] = gld4(xp, ((uint32_t)CC * 16384u + 1024u * a + (uint32_t)tid + lz) << 4);
                bh[k] = gld4(hrow, ((uint32_t)CC * 16384u + hi + lz) << 4);
            } else {
                bx[k] = ld2((const float2*)xp, (uint32_t)CC * 16384u + 1024u * a + (uint32_t)tid + lz);
                bh[k] = ld2((const float2*)hrow, (uint32_t)CC * 16384u + hi + lz);
            }
        }
    };
    // ... folded into pr[A0 ..]: FOLD: G = (X1 H1 + X2 H2) or (X1 H1 - X2 H2) W_32^a; else X H
    // (PART: the sum over the template's partitions is formed before the twiddle W_32^a, which then waits for pass 1)
    auto fold = [&](float2* pr, const pt_t* bx, const pt_t* bh, uint32_t hb, auto a0_, auto np_) __attribute__((always_inline)) {
        constexpr int A0 = decltype(a0_)::value, NP = decltype(np_)::value;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int a = A0 + k;
            if constexpr (FOLD) {
                constexpr double TWO_PI = 6.283185307179586476925;
                const bool sw = ((1024u * (uint32_t)a + hb) & 16384u) != 0;
                const float2 h1 = make_float2(sw ? bh[k].z : bh[k].x, sw ? bh[k].w : bh[k].y);
                const float2 h2 = make_float2(sw ? bh[k].x : bh[k].z, sw ? bh[k].y : bh[k].w);
                const float2 z1 = cmul(make_float2(bx[k].x, bx[k].y), h1);
                const float2 z2 = cmul(make_float2(bx[k].z, bx[k].w), h2);
                if constexpr (PART) {
                    pr[a] = R == 0 ? cadd(z1, z2) : csub(z1, z2);
                } else if (R == 0)
                    pr[a] = cadd(z1, z2);
                else if (a == 0)
                    pr[a] = csub(z1, z2);
                else if (a == 8)
                    pr[a] = mulj(csub(z1, z2));
                else
                    pr[a] = cmul(csub(z1, z2), make_float2((float)__builtin_cos(TWO_PI * a / 32.0), (float)__builtin_sin(TWO_PI * a / 32.0)));
            } else {
                pr[a] = cmul(bx[k], bh[k]);
            }
        }
    };
    // points fetched per pass: in the E half first the O half's (QEO), then the next E half's (QEN); in the O half QON
    // (measured at the C2 shape, profiles/r04/ab_quota_schedules.log: fetching earlier -- 16 points in the first pass -- or
    //  later -- batches in the O half's last pass -- is slower in both forms)
    // (PART: the next E half's points wait for the O half, so that the partitions fetched in front of the O half find registers)
#ifndef PQ_EO
#define PQ_EO 4, 4, 4, 4
#define PQ_EN 0, 0, 0, 0
#define PQ_ON 4, 4, 4, 4
#define PQ_PBE 4  // (8 points in flight in front of the E half spill seven loop invariants, whose reloads inside the passes wait for the
#define PQ_PBO 4  //  prefetches issued before them: 37.3 against 35.5 ms at two partitions; batches of 2 are as fast as 4: ab_partition_front_batches.log)
#endif
#ifndef FQ_EO  // (the folded role without partitions / the plain chained role: build-time switches for A/B libraries)
#define FQ_EO 8, 6, 2, 0
#define FQ_EN 0, 0, 4, 4
#define FQ_ON 4, 2, 2, 0
#endif
#ifndef CQ_EO
#define CQ_EO 8, 4, 4, 0  // (round 4: {8,8,0,0} / {0,0,6,2} / {4,2,2,0}; A/B in profiles/r05/ab_quota_schedules.log: -2 % at 12000 and 16384 samples)
#define CQ_EN 0, 0, 0, 4
#define CQ_ON 4, 4, 4, 0
#endif
    constexpr int QEO_P[4] = {PQ_EO}, QEN_P[4] = {PQ_EN}, QON_P[4] = {PQ_ON};
    constexpr int QEO_F[4] = {FQ_EO}, QEN_F[4] = {FQ_EN}, QON_F[4] = {FQ_ON};
    constexpr int QEO_C[4] = {CQ_EO}, QEN_C[4] = {CQ_EN}, QON_C[4] = {CQ_ON};
#define CAF_Q(T, i) (PART ? T##_P[i] : FOLD ? T##_F[i] : T##_C[i])
    constexpr int QEO[4] = {CAF_Q(QEO, 0), CAF_Q(QEO, 1), CAF_Q(QEO, 2), CAF_Q(QEO, 3)};
    constexpr int QEN[4] = {CAF_Q(QEN, 0), CAF_Q(QEN, 1), CAF_Q(QEN, 2), CAF_Q(QEN, 3)};
    constexpr int QON[4] = {CAF_Q(QON, 0), CAF_Q(QON, 1), CAF_Q(QON, 2), CAF_Q(QON, 3)};
#undef CAF_Q
    static_assert(QEO[0] + QEO[1] + QEO[2] + QEO[3] == 16 && QEN[0] + QEN[1] + QEN[2] + QEN[3] + QON[0] + QON[1] + QON[2] + QON[3] == 16, "quotas");
    float2 pro[16], prn[16];  // folded inputs: of the O half of the hypothesis in progress / of the next E half
    row_of(h0, hrow_o, hb_o);
    {
        pt_t bx[8], bh[8];
        issue(bx, bh, hrow_o, hb_o, caf_ic<0>{}, caf_ic<0>{}, caf_ic<8>{}, 0u);
        fold(prn, bx, bh, hb_o, caf_ic<0>{}, caf_ic<8>{});
        issue(bx, bh, hrow_o, hb_o, caf_ic<0>{}, caf_ic<8>{}, caf_ic<8>{}, 0u);
        fold(prn, bx, bh, hb_o, caf_ic<8>{}, caf_ic<8>{});
    }
    float2 e[16];  // E's outputs (register 4 i + n4 <-> n3 = q + 4 i, n4)
    uint32_t hoff = 0;

    // one sub-transform: c = 0 the E half (even samples), 1 the O half; PS = pass number for the quota tables
    auto batch_issue = [&](pt_t* bx, pt_t* bh, auto c_, auto ps_, uint32_t lz) __attribute__((always_inline)) {
        constexpr int c = decltype(c_)::value, PS = decltype(ps_)::value;
        if constexpr (c == 0) {
            constexpr int O0 = (PS > 0 ? QEO[0] : 0) + (PS > 1 ? QEO[1] : 0) + (PS > 2 ? QEO[2] : 0);
            constexpr int N0 = (PS > 0 ? QEN[0] : 0) + (PS > 1 ? QEN[1] : 0) + (PS > 2 ? QEN[2] : 0);
            issue(bx, bh, hrow_o, hb_o, caf_ic<1>{}, caf_ic<O0>{}, caf_ic<QEO[PS]>{}, lz);
            issue(bx + QEO[PS], bh + QEO[PS], hrow_n, hb_n, caf_ic<0>{}, caf_ic<N0>{}, caf_ic<QEN[PS]>{}, lz);
        } else {
            constexpr int N0 = QEN[0] + QEN[1] + QEN[2] + QEN[3] + (PS > 0 ? QON[0] : 0) + (PS > 1 ? QON[1] : 0) + (PS > 2 ? QON[2] : 0);
            issue(bx, bh, hrow_n, hb_n, caf_ic<0>{}, caf_ic<N0>{}, caf_ic<QON[PS]>{}, lz);
        }
    };
    auto batch_fold = [&](const pt_t* bx, const pt_t* bh, auto c_, auto ps_) __attribute__((always_inline)) {
        constexpr int c = decltype(c_)::value, PS = decltype(ps_)::value;
        if constexpr (c == 0) {
            constexpr int O0 = (PS > 0 ? QEO[0] : 0) + (PS > 1 ? QEO[1] : 0) + (PS > 2 ? QEO[2] : 0);
            constexpr int N0 = (PS > 0 ? QEN[0] : 0) + (PS > 1 ? QEN[1] : 0) + (PS > 2 ? QEN[2] : 0);
            fold(pro, bx, bh, hb_o, caf_ic<O0>{}, caf_ic<QEO[PS]>{});
            fold(prn, bx + QEO[PS], bh + QEO[PS], hb_n, caf_ic<N0>{}, caf_ic<QEN[PS]>{});
        } else {
            constexpr int N0 = QEN[0] + QEN[1] + QEN[2] + QEN[3] + (PS > 0 ? QON[0] : 0) + (PS > 1 ? QON[1] : 0) + (PS > 2 ? QON[2] : 0);
            fold(prn, bx, bh, hb_n, caf_ic<N0>{}, caf_ic<QON[PS]>{});
        }
    };
    // PART: partitions 1 .. npart - 1 of the template (Z = sum_p X[blk + p] . Hc_p: a template of up to npart * 32768 samples as
    // npart spectra of 32768 samples each against the block spectra of the blocks that follow), added to the inputs of the
    // sub-transform about to start, which hold partition 0.  Fetched here with their latency exposed, in batches of what the
    // registers leave WITHOUT pushing loop invariants out of them: four points at a time (the batch size does not matter to the time --
    // 2, 4: the same -- so the role is not waiting on these round trips but on what the L2 delivers).  (Letting partition 1's
    // O-half points ride in the E half's passes, four more points in flight per pass, spills inside the passes, and a reload's
    // s_waitcnt vmcnt waits for every prefetch issued before it: 54.3 against 36.9 ms at two partitions,
    // profiles/r05/ab_partition_ride.log.)
    auto parts_ahead = [&](auto c_) __attribute__((always_inline)) {
        constexpr int cc = decltype(c_)::value;
        constexpr int PB = cc == 0 ? PQ_PBE : PQ_PBO;
        static_assert(16 % PB == 0, "whole batches");
        float2* pr = cc == 0 ? prn : pro;
        for (int p = 1; p < npart; ++p) {
            const float* xq = xp + (int64_t)p * ROW * 2;
            const float* hq = hrow_o + (int64_t)p * ROW * 2;
#pragma unroll
            for (int a0 = 0; a0 < 16; a0 += PB) {
                float4 bx[PB], bh[PB];
                uint32_t lq = 0;  // (an opaque zero: the batch's loads stay behind the fold of the batch before)
                asm volatile("" : "+v"(lq));
#pragma unroll
                for (int k = 0; k < PB; ++k) {
                    const uint32_t a = (uint32_t)(a0 + k);
                    bx[k] = gld4(xq, ((uint32_t)cc * 16384u + 1024u * a + (uint32_t)tid + lq) << 4);
                    bh[k] = gld4(hq, ((uint32_t)cc * 16384u + ((1024u * a + hb_o) & 16383u) + lq) << 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < PB; ++k) {
                    const int a = a0 + k;
                    const bool sw = ((1024u * (uint32_t)a + hb_o) & 16384u) != 0;
                    const float2 h1 = make_float2(sw ? bh[k].z : bh[k].x, sw ? bh[k].w : bh[k].y);
                    const float2 h2 = make_float2(sw ? bh[k].x : bh[k].z, sw ? bh[k].y : bh[k].w);
                    const float2 z1 = cmul(make_float2(bx[k].x, bx[k].y), h1);
                    const float2 z2 = cmul(make_float2(bx[k].z, bx[k].w), h2);
                    pr[a] = cadd(pr[a], R == 0 ? cadd(z1, z2) : csub(z1, z2));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto sub = [&](auto c_) __attribute__((always_inline)) {
        constexpr int c = decltype(c_)::value;
        uint32_t lz = 0;
        asm volatile("" : "+v"(lz));
        if constexpr (PART) parts_ahead(c_);
        // ---- pass 1: DFT16 over a of the (folded) products, twiddle (r = 1: the recurrence starts at the input base), write ----
        {
            float2 v1[16];
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                v1[a] = c == 0 ? prn[a] : pro[a];
                if constexpr (PART && R != 0) {  // the fold's twiddle W_32^a, after the sum over the partitions
                    constexpr double TWO_PI = 6.283185307179586476925;
                    if (a == 8)
                        v1[a] = mulj(v1[a]);
                    else if (a != 0)
                        v1[a] = cmul(v1[a], make_float2((float)__builtin_cos(TWO_PI * a / 32.0), (float)__builtin_sin(TWO_PI * a / 32.0)));
                }
            }
            pt_t bx[8], bh[8];
            batch_issue(bx, bh, c_, caf_ic<0>{}, lz);
            idft16(v1);
            // e^{+j 2 pi / 32768} on the odd half: the n1 digit of the combination twiddle
            float2 wj = c == 0 ? w : cmul(w, make_float2(0.99999998161642933f, 1.9174759731070330e-4f));
            asm volatile("" : "+v"(wj.x), "+v"(wj.y));
            float2 p;
            if (R) {
                p = c == 0 ? base0 : cmul(base0, make_float2(0.99999999540410733f, 9.5873799095977345e-5f));  // * e^{+j 2 pi / 65536}
                v1[0] = cmul(v1[0], p);
                p = cmul(p, wj);
            } else {
                p = wj;
            }
            v1[1] = cmul(v1[1], p);
#pragma unroll
            for (int n1 = 2; n1 < 16; ++n1) {
                p = cmul(p, wj);
                v1[n1] = cmul(v1[n1], p);
            }
            __syncthreads();  // the previous sub-transform's pass-4 reads are done
            lds_rows16c<FP_P1>(m0_x1, v1);
            batch_fold(bx, bh, c_, caf_ic<0>{});
        }
        __syncthreads();
        // ---- pass 2 ----
        {
            float2 v[16];
            const uint32_t rd2 = fp_rd2((uint32_t)(tid + lz)), rd2i = fp_im(rd2);
            const lds_tw_ptr t2 = tw_base(s_tw2, fp_cd2((uint32_t)(tid + lz)));
            pt_t bx[8], bh[8];
            batch_issue(bx, bh, c_, caf_ic<1>{}, lz);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) lds_get4c(img, rd2 + 256u * q4, rd2i + 256u * q4, v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
            idft16(v);
#pragma unroll
            for (int n2 = 1; n2 < 16; ++n2) v[n2] = cmul(v[n2], tw_ld(t2, 1024 * c + n2 * 64));
            lds_rows16c_x2(m0_w, v);
            batch_fold(bx, bh, c_, caf_ic<1>{});
        }
        __builtin_amdgcn_wave_barrier();
        // ---- pass 3 ----
        {
            float2 v[16];
            const uint32_t rd3 = fp_rd3((uint32_t)(tid + lz)), rd3i = fp_im(rd3);
            pt_t bx[8], bh[8];
            batch_issue(bx, bh, c_, caf_ic<2>{}, lz);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) lds_get4c(img, rd3 + 64u * ch, rd3i + 64u * ch, v[4 * ch], v[4 * ch + 1], v[4 * ch + 2], v[4 * ch + 3]);
            idft16(v);
#pragma unroll
            for (int n3 = 1; n3 < 16; ++n3) v[n3] = cmul(v[n3], tw_ld(tw_base(s_tw3, (uint32_t)((tid & 3) + lz)), 64 * c + n3 * 4));
            lds_rows16c<256>(m0_w, v);
            batch_fold(bx, bh, c_, caf_ic<2>{});
        }
        __syncthreads();
        // ---- pass 4: DFT4 over d; the E half keeps its outputs, the O half combines and stores ----
        const uint32_t rd4 = fp_rd4((uint32_t)tid), rd4i = fp_im(rd4);
        pt_t bx4[8], bh4[8];
        batch_issue(bx4, bh4, c_, caf_ic<3>{}, lz);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t lzi = 0;
            asm volatile("" : "+v"(lzi));
            float2 a0, a1, a2, a3;
            lds_get4c(img, rd4 + 1024u * i + lzi, rd4i + 1024u * i + lzi, a0, a1, a2, a3);
            idft4(a0, a1, a2, a3);
            const float2 y[4] = {a0, a1, a2, a3};
            if (c == 0) {
#pragma unroll
                for (int n4 = 0; n4 < 4; ++n4) e[4 * i + n4] = y[n4];
            } else {
#pragma unroll
                for (int n4 = 0; n4 < 4; ++n4) {
                    // O' = O * W_8^{n4} (the last factor of the combination twiddle of the 32768-point transform)
                    constexpr float R2 = 0.70710678118654752f;
                    const float2 t = n4 == 0   ? y[0]
                                     : n4 == 1 ? make_float2((y[1].x - y[1].y) * R2, (y[1].x + y[1].y) * R2)
                                     : n4 == 2 ? mulj(y[2])
                                               : make_float2((-y[3].x - y[3].y) * R2, (y[3].x - y[3].y) * R2);
                    const float2 ylo = cadd(e[4 * i + n4], t);
                    if constexpr (MODE == 4) {
                        // delay of the output inside the block: k = tid + 1024 i + 4096 n4 (plain: + 16384 for the upper half; folded: 2 k + r)
                        const uint32_t ku = 1024u * (uint32_t)i + 4096u * (uint32_t)n4;
                        const uint32_t dv = FOLD ? 2u * (uint32_t)tid : (uint32_t)tid, du = FOLD ? 2u * ku + (uint32_t)R : ku;
                        const float gq = __builtin_sqrtf(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rie, (int)(dv << 2), (int)(du << 2), 0))) * f1_ts;
                        const v2f_t ov = {ylo.x * gq, ylo.y * gq};
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i_t, ov), f1_r0, (int)(dv << 3), (int)(du << 3), CAF_AUX_NT);
                        if (n4 < NVH) {
                            const float2 yhi = csub(e[4 * i + n4], t);
                            const float gh = __builtin_sqrtf(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rie, (int)(dv << 2), (int)((du + 16384u) << 2), 0))) * f1_ts;
                            const v2f_t oh = {yhi.x * gh, yhi.y * gh};
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i_t, oh), f1_r0, (int)(dv << 3), (int)((du + 16384u) << 3), CAF_AUX_NT);
                        }
                    } else {
                    const int tile_u = 16 * i + 64 * n4 + 256 * R;
                    const int tile_t = (n2o >> 2) + 4 * qo;
                    const uint32_t soff = (uint32_t)tile_u * (uint32_t)nhyp * 256u + hoff;  // uniform (scalar) offset
                    const uint32_t voff = ((uint32_t)tile_t * (uint32_t)nhyp * 64u + (uint32_t)(n1o + 16 * (n2o & 3))) << 2;
                    tile_store<MODE>(rvt, voff, soff, __builtin_fmaf(ylo.x, ylo.x, ylo.y * ylo.y));
                    if (n4 < NVH) {  // (plain form: the upper half y[n' + 16384] = E - O', tile + 256)
                        const uint32_t voff_hi = voff + (((uint32_t)256 * (uint32_t)nhyp * 64u) << 2);
                        const float2 yhi = csub(e[4 * i + n4], t);
                        tile_store<MODE>(rvt, voff_hi, soff, __builtin_fmaf(yhi.x, yhi.x, yhi.y * yhi.y));
                    }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        batch_fold(bx4, bh4, c_, caf_ic<3>{});
    };

    for (int h = h0; h < h1; ++h) {
        hoff = (uint32_t)h * 256u;  // bytes
        asm volatile("" : "+s"(hoff));
        if constexpr (MODE == 4) {  // this hypothesis' row of the plane (2 floats per value) and its template's scale
            f1_ts = __builtin_sqrtf(*((const CAF_AS4 float*)f1->tscale + h / nfreq));
            f1_r0 = buf_of(uniform_ptr(f1->out0 + 2 * ((int64_t)h * f1->num_shifts + f1_rel0)), 2u * f1_bytes);
        }
        row_of(h + 1 < h1 ? h + 1 : h, hrow_n, hb_n);
        sub(caf_ic<0>{});
        sub(caf_ic<1>{});
        hrow_o = hrow_n;
        hb_o = hb_n;
    }
}

// natural order -> butterfly order, chunk by chunk of 1024 elements (see fp_tid_of)
__global__ __launch_bounds__(256) void k_butterfly_order(const float2* __restrict__ in, float2* __restrict__ out, int64_t nchunks) {
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x)
        for (int j = threadIdx.x; j < 1024; j += 256) out[c * 1024 + j] = in[c * 1024 + fp_m2((uint32_t)j)];
}
void launch_butterfly_order(const float2* in, float2* out, int64_t nchunks, hipStream_t st) {
    hipLaunchKernelGGL(k_butterfly_order, dim3((unsigned)std::min<int64_t>(nchunks, 65535)), dim3(256), 0, st, in, out, nchunks);
}

// rows of B = 2 * half complex samples -> parity-major: out[r][c][m'] = in[r][2 m' + c]; BFLY: each parity half also in
// butterfly order (see fp_tid_of), the order in which fused_item2q reads its block spectra once per half-transform
template <bool BFLY>
__global__ __launch_bounds__(256) void k_parity_major(const float2* __restrict__ in, float2* __restrict__ out, int32_t half) {
    const float2* ir = in + (int64_t)blockIdx.y * 2 * half;
    float2* orow = out + (int64_t)blockIdx.y * 2 * half;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < half; j += gridDim.x * 256) {
        // (BFLY: output position j holds source element m: consecutive threads write consecutive addresses)
        const int m = BFLY ? (int)((j & ~1023) + fp_m2((uint32_t)j & 1023u)) : j;
        const float4 v = *reinterpret_cast<const float4*>(&ir[2 * m]);
        orow[j] = make_float2(v.x, v.y);
        orow[half + j] = make_float2(v.z, v.w);
    }
}
// rows of B = 4 * quarter complex samples -> PAIRS of the two halves of each parity (fused_item2q<FOLD>):
// out[r][c][j] = (in[r][2 m + c], in[r][2 (m + quarter) + c]) as one 16-byte element, m = the butterfly-order source of j
__global__ __launch_bounds__(256) void k_parity_pairs(const float2* __restrict__ in, float4* __restrict__ out, int32_t quarter) {
    const float2* ir = in + (int64_t)blockIdx.y * 4 * quarter;
    float4* orow = out + (int64_t)blockIdx.y * 2 * quarter;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < quarter; j += gridDim.x * 256) {
        const int m = (int)((j & ~1023) + fp_m2((uint32_t)j & 1023u));
        const float4 lo = *reinterpret_cast<const float4*>(&ir[2 * m]), hi = *reinterpret_cast<const float4*>(&ir[2 * (m + quarter)]);
        orow[j] = make_float4(lo.x, lo.y, hi.x, hi.y);
        orow[quarter + j] = make_float4(lo.z, lo.w, hi.z, hi.w);
    }
}
void launch_parity_pairs(const float2* in, float2* out, int64_t rows, int32_t quarter, hipStream_t st) {
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = std::min<int64_t>(65535, rows - r0);
        hipLaunchKernelGGL(k_parity_pairs, dim3(16, (unsigned)nr), dim3(256), 0, st, in + r0 * 4 * quarter,
                           reinterpret_cast<float4*>(out + r0 * 4 * quarter), quarter);
    }
}
void launch_parity_major(const float2* in, float2* out, int64_t rows, int32_t half, hipStream_t st, bool butterfly) {
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = std::min<int64_t>(65535, rows - r0);
        if (butterfly)
            hipLaunchKernelGGL(k_parity_major<true>, dim3(16, (unsigned)nr), dim3(256), 0, st, in + r0 * 2 * half, out + r0 * 2 * half, half);
        else
            hipLaunchKernelGGL(k_parity_major<false>, dim3(16, (unsigned)nr), dim3(256), 0, st, in + r0 * 2 * half, out + r0 * 2 * half, half);
    }
}

// One workgroup per (rx block, group of hyp_per_wg hypotheses).
template <int FT>
__global__ __launch_bounds__(FT) void k_fused_caf(const float2* __restrict__ xb, const float2* __restrict__ hc,
                                                  const int32_t* __restrict__ shifts, const float2* __restrict__ tw1,
                                                  const float2* __restrict__ tw23,  // [16][64] then [16][4]
                                                  int32_t table_mode, int32_t nfreq, int32_t nhyp, int32_t hyp_per_wg,
                                                  int32_t nblk, int32_t tiles_per_blk, float* __restrict__ vt) {
    __shared__ __attribute__((aligned(16))) float2 s_d[F_LDS_DATA];
    __shared__ float2 s_tw2[16 * 64];
    __shared__ float2 s_tw3[16 * 4];
    const int tid = threadIdx.x;
    // XCD-aware mapping (speed only): workgroups are dealt round-robin over the 8 XCDs, so linear id L runs
    // on XCD L % 8.  All hypothesis groups of one rx block are given to the SAME XCD (block b -> XCD b % 8,
    // its groups on consecutive slots of that XCD), so the block spectrum and the template-spectrum table
    // they share are served by one 4 MiB L2 instead of being replicated in several.
    const int ngroups = (nhyp + hyp_per_wg - 1) / hyp_per_wg;
    const int lin = blockIdx.x;
    const int q = lin >> 3;
    const int blk = (q / ngroups) * 8 + (lin & 7);
    const int grp = q - (q / ngroups) * ngroups;
    if (blk >= nblk) return;
    const int h0 = grp * hyp_per_wg;
    const int h1 = min(h0 + hyp_per_wg, nhyp);

    for (int i = tid; i < 1024; i += FT) s_tw2[i] = tw23[i];
    if (tid < 64) s_tw3[tid] = tw23[1024 + tid];
    if (tiles_per_blk <= 192)  // (see fused_item: quarters of the last pass without a valid delay are skipped)
        fused_item<FT, 0, 3>(s_d, s_tw2, s_tw3, xb, hc, shifts, tw1, table_mode, nfreq, nhyp, blk, h0, h1, tiles_per_blk, vt);
    else
        fused_item<FT, 0, 4>(s_d, s_tw2, s_tw3, xb, hc, shifts, tw1, table_mode, nfreq, nhyp, blk, h0, h1, tiles_per_blk, vt);
}

// ----------------------------------------------------------------------------------------
// |y|^2 tiles -> delay-major QF^2 surface + per-delay argmax + tile peak record.
//   in : vt[blk][s_tile][t*F+f][64] float32   (contiguous 64*F*4 bytes per (blk, s_tile, t))
//   out: as k_magsq_norm_argmax.
// One workgroup per (s_tile, template, block): streams its contiguous tile in chunks of TR_F
// hypotheses (32 KiB, 16-byte loads, 8 in flight per lane) through an LDS transpose; surface rows
// leave as 512-byte segments.
// ----------------------------------------------------------------------------------------
// TR_F = hypotheses per LDS chunk: 256 (66 KB tile, whole 1-KiB surface rows at F = 256) or 128 (33 KB).
template <int TR_F, bool NT_STORE>
__global__ __launch_bounds__(256) void k_transpose_norm_argmax(
    const float* __restrict__ vt, int32_t ntmpl, int32_t nfreq, const float* __restrict__ tscale,
    const float* __restrict__ inv_e, int64_t num_shifts, int64_t shift_start, int32_t step, int32_t blk0,
    int32_t tiles_per_blk, float* __restrict__ surface, float* __restrict__ row_max, int32_t* __restrict__ row_arg,
    PeakRec* __restrict__ partial, int64_t partial_per_tmpl) {
    __shared__ float s_tile[64][TR_F + 1];
    __shared__ float s_rowv[64];
    __shared__ int32_t s_rowi[64];
    const int z = blockIdx.z, t = blockIdx.y, tile = blockIdx.x;
    const int blk = blk0 + z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl0 = tile * 64;
    const int64_t rel0 = (int64_t)blk * step + sl0;
    int64_t nv = num_shifts - (int64_t)blk * step;
    if (nv > step) nv = step;
    const int64_t pidx = (int64_t)blk * tiles_per_blk + tile;
    if (sl0 >= nv) {
        if (threadIdx.x == 0 && partial) {
            PeakRec r;
            r.v = -1.f;
            r.delay = 0x7fffffff;
            r.f = 0;
            partial[(int64_t)t * partial_per_tmpl + pidx] = r;
        }
        return;
    }
    const int nrows = (int)min((int64_t)64, nv - sl0);
    const int nhyp = ntmpl * nfreq;
    const float* vin = vt + (((int64_t)z * tiles_per_blk + tile) * nhyp + (int64_t)t * nfreq) * 64;
    // load mapping: float4 number (i*256 + tid) of the chunk -> hypothesis fl = i*16 + (tid >> 4),
    // delays s4 .. s4+3 with s4 = 4*(tid & 15) (the same four delays for every i)
    const int s4 = 4 * (threadIdx.x & 15);
    const float ts = tscale[t];
    float g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = (s4 + k < nrows) ? inv_e[rel0 + s4 + k] * ts : -1.f;

    float bv[16];
    int32_t bi[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bv[r] = -1.f;
        bi[r] = 0;
    }
    // Per-lane running best over the columns this lane visits (lane + 64 c of every chunk); the
    // cross-lane reduction is done once per row after the last chunk.  The next chunk's loads are issued
    // before the store phase of the current one, so HBM reads stay in flight while rows are written.
    float4 q[TR_F / 16];
    auto load_chunk = [&](int f0) {
        const int nf = min(TR_F, nfreq - f0);
#pragma unroll
        for (int i = 0; i < TR_F / 16; ++i) {
            const int fl = i * 16 + (threadIdx.x >> 4);
            q[i] = (fl < nf) ? *reinterpret_cast<const float4*>(vin + (int64_t)(f0 + fl) * 64 + s4)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_chunk(0);
    for (int f0 = 0; f0 < nfreq; f0 += TR_F) {
        const int nf = min(TR_F, nfreq - f0);
#pragma unroll
        for (int i = 0; i < TR_F / 16; ++i) {
            const int fl = i * 16 + (threadIdx.x >> 4);
            // invalid delays (g < 0) are stored as -1 so that they never win the argmax
            s_tile[s4 + 0][fl] = g[0] < 0.f ? -1.f : q[i].x * g[0];
            s_tile[s4 + 1][fl] = g[1] < 0.f ? -1.f : q[i].y * g[1];
            s_tile[s4 + 2][fl] = g[2] < 0.f ? -1.f : q[i].z * g[2];
            s_tile[s4 + 3][fl] = g[3] < 0.f ? -1.f : q[i].w * g[3];
        }
        __syncthreads();
        if (f0 + TR_F < nfreq) load_chunk(f0 + TR_F);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave + 4 * r;
            if (row < nrows) {
                float* srow = surface ? surface + ((int64_t)t * num_shifts + rel0 + row) * nfreq + f0 : nullptr;
#pragma unroll
                for (int c = 0; c < TR_F / 64; ++c) {
                    const int fl = lane + 64 * c;
                    if (fl < nf) {
                        const float v = s_tile[row][fl];
                        if (srow) {
                            if (NT_STORE)
                                __builtin_nontemporal_store(v, &srow[fl]);  // write-once stream: keep it out of L2
                            else
                                srow[fl] = v;
                        }
                        if (v > bv[r]) {  // columns are visited in increasing order: first maximum wins
                            bv[r] = v;
                            bi[r] = f0 + fl;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // one cross-lane reduction per row: highest value, lowest frequency index on ties
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv[r], o, 64);
            const int32_t oi = __shfl_xor(bi[r], o, 64);
            if (ov > bv[r] || (ov == bv[r] && oi < bi[r])) {
                bv[r] = ov;
                bi[r] = oi;
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s_rowv[wave + 4 * r] = bv[r];
            s_rowi[wave + 4 * r] = bi[r];
        }
    }
    __syncthreads();
    if (wave == 0) {
        float v = -1.f;
        if (lane < nrows) {
            v = s_rowv[lane];
            const int64_t o = (int64_t)t * num_shifts + rel0 + lane;
            if (row_max) row_max[o] = v < 0.f ? __builtin_nanf("") : v;  // (nothing beat the initial value: a zero-energy window, all NaN)
            if (row_arg) row_arg[o] = s_rowi[lane];
        }
        if (partial) {
            float b = v;
            int32_t bidx = lane;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(b, o, 64);
                const int32_t oi = __shfl_xor(bidx, o, 64);
                if (ov > b || (ov == b && oi < bidx)) {
                    b = ov;
                    bidx = oi;
                }
            }
            if (lane == 0) {
                PeakRec r;
                r.v = b;
                r.delay = (int32_t)(shift_start + rel0 + bidx);
                r.f = s_rowi[bidx];
                partial[(int64_t)t * partial_per_tmpl + pidx] = r;
            }
        }
    }
}


// ----------------------------------------------------------------------------------------
// Work-queue kernel: BOTH stages in one launch, overlapped across CUs.
//
// k_fused_caf is bound by the vector ALUs and the LDS (HBM ~1.6 TB/s), k_transpose_norm_argmax by HBM
// (ALUs idle); run back to back they add up.  A k_fused_caf workgroup owns every VGPR and nearly all LDS
// of its CU, so the two cannot share a CU as separate launches either.  Here one workgroup per CU stays
// resident and pulls work items from two device-side queues:
//   * FFT items  (rx block, hypothesis group)  -> fused_item() with write-through |y|^2 stores; each finished
//     group is published with done[block]++;
//   * tile items (rx block, 16 delay tiles)    -> one tile per wave through a wave-private LDS patch, no
//     barrier (transpose_wave); runs once done[block] == ngroups.
// Workgroups in the last `tr_slots` of every 32 slots of an XCD look at the tile queue first, so about
// tr_slots*8 CUs stream HBM while the others compute; a workgroup that finds the head of the tile queue not
// ready takes an FFT item, and once the FFT queue is empty everybody drains the tile queue.
// Both queues are claimed with one fetch-add (no compare-and-swap loops), by wave 0 as scalar control flow.
// Termination: FFT items never wait.  A tile item waits for the FFT items of its block: those are either
// running on resident workgroups (which finish) or still queued, and the queue is drained by the workgroups
// that do not prefer tiles -- they take tiles only when no FFT item is left -- of which every resident prefix
// of the grid has some (workgroups 0..7 are slot 0).  So every wait ends and every workgroup reaches the exit,
// whatever number of workgroups is resident; a polling watchdog bounds the wait regardless.
// Registers: each role is a noinline function with a register allocation of its own (the hypothesis loop needs
// all 128 VGPRs; allocated together with the other role it reloads spilled values every hypothesis), and the
// arguments live in device memory (PersistParams), read with scalar loads at the start of each item.
// ----------------------------------------------------------------------------------------
// What the publish / consume pair below relies on (measured on gfx950 / ROCm 7.2: MI355X_MICROARCH.md, "Workgroup
// dispatch, XCD placement & inter-workgroup visibility", valid forms + the table of measured hand-offs; NOT
// guarantees of the HSA memory model -- which is why tests/test_gpu_persistent_protocol.py runs the residency
// matrix in every suite run):
//   producer (persistent_fft_item):
//     P1 every |y|^2 / (value, hypothesis) store of the item is an sc1 (write-through) store: the bytes leave the
//        XCD's L2 towards memory instead of staying dirty in it (gst1_wt / __hip_atomic_store relaxed, agent scope);
//     P2 each storing wave executes s_waitcnt vmcnt(0) after its stores (the workgroup-scope release fence lowers
//        to exactly that): vmcnt is decremented for an sc1 store only when the write has been acknowledged by the
//        memory side, so after the wait the wave's bytes are visible to every XCD;
//     P3 the workgroup barrier that follows orders ALL waves' waits before the one lane that signals;
//     P4 that lane's agent-scope atomic add on done[block] executes at the memory side (not in an L2), after P3.
//   consumer (pq_wait_block + the tile roles):
//     C1 the poll is a relaxed agent-scope atomic load (sc1: served from memory, never from this CU's L1 or a stale
//        L2 line) issued by wave 0 as SCALAR control flow: the branch that leaves the loop is resolved before any
//        instruction after it issues (in-order issue per wave), so no tile load can be issued ahead of the poll;
//     C2 the other waves start the item only after the workgroup barrier wave 0 joins after its poll (s_cmd hand-off);
//     C3 every load of the handed-off bytes is a buffer_load ... sc1 to registers (CAF_AUX_SC1): it bypasses the CU's
//        L1 and is not served from an L2 line older than the write-through, because P1's stores dropped / updated
//        the line on their way through.
// This is the guide's row "ONE lane of each storing workgroup signals with an agent-scope atomic add; consumer polls
// with an sc1 load; workgroup barrier between poll and loads; hipMalloc memory; one workgroup per CU; stores
// 4-byte sc1; loads 4- or 16-byte sc1".  An agent-scope release fence per item (buffer_wbl2) would make the
// hand-off model-conformant and was measured at +30 % per FFT item (it writes back the L2 shared by 32 CUs).
// int32 slots of the queue block: [0] FFT items claimed so far, [1] next tile item, [2..3] watchdog marks, [4..11] next FFT
// item of each XCD's own list (see the claim), [12 + b] finished hypothesis groups of block b
constexpr int PQ_FFT_NEXT = 0, PQ_TR_NEXT = 1, PQ_FFT_XCD = 4, PQ_DONE = 12;
constexpr int PQ_TILES = 16;  // a tile item = 16 delay tiles of one block, one per wave (~2 MB of HBM traffic at F = 256)

__device__ __forceinline__ int32_t pq_load(const int32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the parameter block through the constant address space (scalar loads); the empty asm makes the pointer
// opaque so that the loads are redone per item instead of being hoisted out of the work loop and kept alive
__device__ __forceinline__ const CAF_AS4 PersistParams* params_of(const PersistParams* p) {
    uint64_t v = reinterpret_cast<uint64_t>(p);
    asm volatile("" : "+s"(v));
    return reinterpret_cast<const CAF_AS4 PersistParams*>(v);
}
// Cache policy of the |y|^2 tile reads: sc1 = device-scope coherent load.  The tiles were written during the SAME
// launch by workgroups on other XCDs (each XCD has its own L2) with write-through stores; the readers must not be
// served from a line their own L2 may still hold.  (Measured: no cost against plain loads.)
// 16-byte vector stores to addresses that are only 4-byte aligned (delay offsets are arbitrary)
typedef float v4f_u_t __attribute__((ext_vector_type(4), aligned(4)));
typedef int v4i_u_t __attribute__((ext_vector_type(4), aligned(4)));

// Barrier-free tile role: every WAVE transposes its own (delay tile, template): 64 delays x nfreq hypotheses
// in steps of 32 hypotheses through a private 64 x 33 float LDS patch.  No workgroup barrier, no cross-wave
// traffic.  Two register sets of loads (2 x 8 KB per wave) alternate; surface rows leave as 128-byte segments,
// two rows per store instruction.  The per-delay argmax is tracked where the values are produced
// (4 delays per lane, hypotheses in increasing order) and finished with wave shuffles.
// Bounds: the tile, surface-row and energy buffers are raw buffer descriptors sized to the valid extent, so
// out-of-range loads return 0 and out-of-range row stores are dropped by the hardware; ragged tiles (fewer
// than 64 delays, nfreq not a multiple of 32) need no separate path.
constexpr int TW_H = 32;              // hypotheses per step
constexpr int TW_PITCH = TW_H + 1;    // LDS row pitch (floats): conflict-free transposed writes
constexpr int TW_LDS = 64 * TW_PITCH;  // floats per wave

// DS: delay stride of a tile (1; 2 for the folded 65536-point role, whose tile 256 r + u holds the delays 2 (64 u + j) + r)
template <bool SURF, int DS>
__device__ __forceinline__ void transpose_wave(float* __restrict__ lds, const PersistParams* pp, int z, int tile0) {
    const CAF_AS4 PersistParams* P = params_of(pp);
    const int32_t ntmpl = P->ntmpl, nfreq = P->nfreq, step = P->step, tiles_per_blk = P->tiles_per_blk;
    const int64_t num_shifts = P->num_shifts, shift_start = P->shift_start;
    const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    float* s_w = lds + wave_id * TW_LDS;
    const int tile = tile0 + wave_id;
    if (tile >= tiles_per_blk) return;  // (wave-uniform; no barriers in this role)
    const int blk = P->blk0 + z;
    const int sl0 = DS == 2 ? 128 * (tile & 255) + (tile >> 8) : tile * 64;  // the tile's first delay inside the block
    const int64_t rel0 = (int64_t)blk * step + sl0;
    int64_t nv = num_shifts - (int64_t)blk * step;
    if (nv > step) nv = step;
    const int nrows = sl0 < nv ? (int)min((int64_t)64, (nv - sl0 + DS - 1) / DS) : 0;
    const int64_t pidx = (int64_t)blk * tiles_per_blk + tile;
    PeakRec* partial = P->partial;
    const int64_t ppt = P->partial_per_tmpl;
    const int s4 = 4 * (lane & 15), fq = lane >> 4;  // this lane's four delays / hypothesis within a group of 4
    const int nsteps = (nfreq + TW_H - 1) / TW_H;
    for (int t = 0; t < ntmpl; ++t) {
        if (nrows == 0) {
            if (lane == 0 && partial) {
                PeakRec r;
                r.v = -1.f;
                r.delay = 0x7fffffff;
                r.f = 0;
                partial[(int64_t)t * ppt + pidx] = r;
            }
            continue;
        }
        // 64-bit products of uniform values are evaluated on the vector ALU (no scalar 64-bit multiply);
        // v_readfirstlane brings the bases back into SGPRs
        const float* vin = uniform_ptr(P->vt + (((int64_t)z * tiles_per_blk + tile) * ntmpl + t) * (int64_t)nfreq * 64);
        float* srow0 = SURF ? uniform_ptr(P->surface + ((int64_t)t * num_shifts + rel0) * nfreq) : nullptr;
        const float ts = P->tscale[t];
        const __amdgpu_buffer_rsrc_t rin = buf_of(vin, (uint32_t)nfreq * 256u);
        const __amdgpu_buffer_rsrc_t rout = buf_of(srow0, (uint32_t)(((nrows - 1) * DS + 1) * nfreq) * 4u);  // rows >= nrows: dropped
        const __amdgpu_buffer_rsrc_t rie = buf_of(uniform_ptr(P->inv_e + rel0), (uint32_t)((nrows - 1) * DS + 1) * 4u);
        float g[4], bv[4];
        int32_t bi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            g[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rie, (s4 + k) * 4 * DS, 0, 0)) * ts;
            bv[k] = -1.f;
            bi[k] = 0;
        }
        // load mapping of a step: float4 number (i*64 + lane) -> hypothesis 4*i + fq, delays s4 .. s4+3.
        // Steps past the end are loaded too (the descriptor bounds make them zeros without memory traffic), so
        // that the loop body is straight-line code and the outstanding-load bookkeeping stays exact.
        v4f_t qa[TW_H / 4], qb[TW_H / 4];
        auto load_step = [&](v4f_t(&q)[TW_H / 4], int s) {
#pragma unroll
            for (int i = 0; i < TW_H / 4; ++i)
                q[i] = __builtin_bit_cast(v4f_t, __builtin_amdgcn_raw_buffer_load_b128(rin, (fq * 64 + s4) * 4,
                                                                                       (s * TW_H + 4 * i) * 256, CAF_AUX_SC1 | CAF_AUX_NT));
        };
        // lanes 0..31 -> row r, lanes 32..63 -> row r+1: one per-lane offset, the row in the scalar offset
        const int half = lane >> 5, col = lane & 31;
        const float* srd = s_w + half * TW_PITCH + col;
        float* swr = s_w + s4 * TW_PITCH + fq;
        const int voff = (half * DS * nfreq + col) * 4;
        auto do_step = [&](v4f_t(&q)[TW_H / 4], int s) {
            const int f0 = s * TW_H;
#pragma unroll
            for (int i = 0; i < TW_H / 4; ++i) {
                const int hyp = f0 + 4 * i + fq;  // increasing with i and s: first maximum wins
                const bool hv = hyp < nfreq;
                const float x[4] = {q[i].x * g[0], q[i].y * g[1], q[i].z * g[2], q[i].w * g[3]};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (SURF) swr[k * TW_PITCH + 4 * i] = x[k];
                    const bool up = hv & (x[k] > bv[k]);  // selects, not branches: the step stays one basic block
                    bv[k] = up ? x[k] : bv[k];
                    bi[k] = up ? hyp : bi[k];
                }
            }
            load_step(q, s + 2);  // refill this register set: two steps stay in flight
            if (SURF) {
                __builtin_amdgcn_wave_barrier();  // LDS operations of a wave execute in order
                // columns past nfreq (ragged last step): an offset beyond the descriptor, dropped by the bounds check
                const int vo = (f0 + col < nfreq) ? voff : 0x7ffffff0;
#pragma unroll
                for (int r = 0; r < 64; r += 2)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, srd[r * TW_PITCH]), rout, vo,
                                                          (r * DS * nfreq + f0) * 4, CAF_AUX_NT);
                __builtin_amdgcn_wave_barrier();
            }
        };
        load_step(qa, 0);
        load_step(qb, 1);
        for (int s = 0; s < nsteps; s += 2) {
            do_step(qa, s);
            do_step(qb, s + 1);
        }
        // lanes l, l^16, l^32, l^48 hold the same four delays: highest value, lowest hypothesis on ties
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float ov = __shfl_xor(bv[k], o, 64);
                const int32_t oi = __shfl_xor(bi[k], o, 64);
                if (ov > bv[k] || (ov == bv[k] && oi < bi[k])) {
                    bv[k] = ov;
                    bi[k] = oi;
                }
            }
        }
        // lanes 0..15 (fq == 0) now hold delays 4*lane .. 4*lane+3
        float* row_max = P->row_max;
        int32_t* row_arg = P->row_arg;
        float best = -1.f;
        int32_t bdel = 0x7fffffff, bfrq = 0;
        if (fq == 0) {
            const int64_t o = (int64_t)t * num_shifts + rel0 + s4 * DS;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (s4 + k < nrows) {
                    if (row_max) row_max[o + k * DS] = bv[k] < 0.f ? __builtin_nanf("") : bv[k];  // (zero-energy window: all NaN, as the reference's 0 / 0)
                    if (row_arg) row_arg[o + k * DS] = bi[k];
                    if (bv[k] > best) {  // increasing delay: first maximum wins
                        best = bv[k];
                        bdel = s4 + k;
                        bfrq = bi[k];
                    }
                }
            }
        }
        if (partial) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                const float ov = __shfl_xor(best, o, 64);
                const int32_t od = __shfl_xor(bdel, o, 64);
                const int32_t of = __shfl_xor(bfrq, o, 64);
                if (ov > best || (ov == best && od < bdel)) {
                    best = ov;
                    bdel = od;
                    bfrq = of;
                }
            }
            if (lane == 0) {
                PeakRec r;
                r.v = best;
                r.delay = (int32_t)(shift_start + rel0 + bdel * DS);
                r.f = bfrq;
                partial[(int64_t)t * ppt + pidx] = r;
            }
        }
    }
}

// Tile role without a frequency scan (nfreq == 1: config C3, TemplateCrossCorrelator-style banks, fastXcorr
// branch A through the engine).  The hypothesis axis of a tile IS the template axis and every value is its own
// per-delay result, so nothing is transposed: a wave streams 32 templates x 64 delays per step and writes each
// template's 64 results as 16-byte pieces straight to row_max (and the (T, S, 1) surface), row_arg = 0, and a
// (delay, value) record per (tile, template).  One step replaces 32 passes of the general path.
// (out of line: keeps the general path's register allocation unchanged)
template <int DS>
__device__ __attribute__((noinline)) void transpose_wave_f1(const PersistParams* pp_in, int z_in, int tile0_in) {
    const PersistParams* pp = uniform_ptr(pp_in);
    const int z = __builtin_amdgcn_readfirstlane(z_in), tile0 = __builtin_amdgcn_readfirstlane(tile0_in);
    const CAF_AS4 PersistParams* P = params_of(pp);
    const int32_t nhyp = P->ntmpl, step = P->step, tiles_per_blk = P->tiles_per_blk;
    const int64_t num_shifts = P->num_shifts, shift_start = P->shift_start;
    const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int tile = tile0 + wave_id;
    if (tile >= tiles_per_blk) return;
    const int blk = P->blk0 + z;
    const int sl0 = DS == 2 ? 128 * (tile & 255) + (tile >> 8) : tile * 64;  // the tile's first delay inside the block
    const int64_t rel0 = (int64_t)blk * step + sl0;
    int64_t nv = num_shifts - (int64_t)blk * step;
    if (nv > step) nv = step;
    const int nrows = sl0 < nv ? (int)min((int64_t)64, (nv - sl0 + DS - 1) / DS) : 0;
    const int64_t pidx = (int64_t)blk * tiles_per_blk + tile;
    PeakRec* partial = P->partial;
    const int64_t ppt = P->partial_per_tmpl;
    float* row_max = P->row_max;
    int32_t* row_arg = P->row_arg;
    float* surface = P->surface;
    const int s4 = 4 * (lane & 15), fq = lane >> 4;
    const float* vin = uniform_ptr(P->vt + ((int64_t)z * tiles_per_blk + tile) * (int64_t)nhyp * 64);
    const __amdgpu_buffer_rsrc_t rin = buf_of(vin, (uint32_t)nhyp * 256u);
    const __amdgpu_buffer_rsrc_t rts = buf_of(P->tscale, (uint32_t)nhyp * 4u);
    const __amdgpu_buffer_rsrc_t rie = buf_of(uniform_ptr(P->inv_e + rel0), nrows > 0 ? (uint32_t)((nrows - 1) * DS + 1) * 4u : 0u);
    float g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rie, (s4 + k) * 4 * DS, 0, 0));
    const bool all4 = DS == 1 && s4 + 3 < nrows;  // (16-byte pieces: consecutive delays only)
    for (int h0 = 0; h0 < nhyp; h0 += TW_H) {
        v4f_t q[TW_H / 4];
        float ts[TW_H / 4];
#pragma unroll
        for (int i = 0; i < TW_H / 4; ++i) {
            q[i] = __builtin_bit_cast(v4f_t, __builtin_amdgcn_raw_buffer_load_b128(rin, (fq * 64 + s4) * 4, (h0 + 4 * i) * 256,
                                                                                   CAF_AUX_SC1));
            ts[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rts, fq * 4, (h0 + 4 * i) * 4, 0));
        }
#pragma unroll
        for (int i = 0; i < TW_H / 4; ++i) {
            const int h = h0 + 4 * i + fq;
            const bool hv = h < nhyp;
            // (value * (1/energy * 1/||t||^2): the same rounding as the other paths)
            const float x[4] = {q[i].x * (g[0] * ts[i]), q[i].y * (g[1] * ts[i]), q[i].z * (g[2] * ts[i]),
                                q[i].w * (g[3] * ts[i])};
            if (hv && nrows > 0) {
                const int64_t o = (int64_t)h * num_shifts + rel0 + s4 * DS;
                if (all4) {
                    const v4f_u_t xv = {x[0], x[1], x[2], x[3]};
                    const v4i_u_t zv = {0, 0, 0, 0};
                    if (row_max) *reinterpret_cast<v4f_u_t*>(row_max + o) = xv;
                    if (surface) *reinterpret_cast<v4f_u_t*>(surface + o) = xv;
                    if (row_arg) *reinterpret_cast<v4i_u_t*>(row_arg + o) = zv;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (s4 + k < nrows) {
                            if (row_max) row_max[o + k * DS] = x[k];
                            if (surface) surface[o + k * DS] = x[k];
                            if (row_arg) row_arg[o + k * DS] = 0;
                        }
                }
            }
            if (partial) {
                // the template's maximum over the tile's delays: first maximum wins, NaN never does
                float best = -1.f;
                int32_t bdel = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (s4 + k < nrows && x[k] > best) {
                        best = x[k];
                        bdel = s4 + k;
                    }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    const float ov = __shfl_xor(best, o, 64);
                    const int32_t od = __shfl_xor(bdel, o, 64);
                    if (ov > best || (ov == best && od < bdel)) {
                        best = ov;
                        bdel = od;
                    }
                }
                if (hv && (lane & 15) == 0) {
                    PeakRec r;
                    r.v = best;
                    r.delay = nrows > 0 && bdel != 0x7fffffff ? (int32_t)(shift_start + rel0 + bdel * DS) : 0x7fffffff;
                    r.f = 0;
                    partial[(int64_t)h * ppt + pidx] = r;
                }
            }
        }
    }
}

// Tile role when no surface is wanted (PersistParams::nosurf): the FFT items have left one (maximum |y|^2,
// hypothesis) pair per delay and hypothesis group; a wave takes a 64-delay tile (lane = delay), combines the
// groups of each template (increasing hypothesis order: the first maximum wins, as everywhere), normalises, and
// writes the per-delay trace and the tile's peak record.  Groups never straddle templates (PersistParams::gpt
// groups per template in this mode), and value * (1/energy * 1/||t||^2) rounds exactly as in the surface paths.
__device__ __attribute__((noinline)) void reduce_wave_nosurf(const PersistParams* pp_in, int z_in, int tile0_in) {
    const PersistParams* pp = uniform_ptr(pp_in);
    const int z = __builtin_amdgcn_readfirstlane(z_in), tile0 = __builtin_amdgcn_readfirstlane(tile0_in);
    const CAF_AS4 PersistParams* P = params_of(pp);
    const int32_t ntmpl = P->ntmpl, nfreq = P->nfreq, step = P->step, tiles_per_blk = P->tiles_per_blk;
    const int32_t ngroups = P->ngroups, gpt = P->gpt;
    const int64_t num_shifts = P->num_shifts, shift_start = P->shift_start;
    const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int tile = tile0 + wave_id;
    if (tile >= tiles_per_blk) return;
    const int blk = P->blk0 + z;
    const int sl0 = tile * 64;
    const int64_t rel0 = (int64_t)blk * step + sl0;
    int64_t nv = num_shifts - (int64_t)blk * step;
    if (nv > step) nv = step;
    const int nrows = sl0 < nv ? (int)min((int64_t)64, nv - sl0) : 0;
    const int64_t pidx = (int64_t)blk * tiles_per_blk + tile;
    PeakRec* partial = P->partial;
    const int64_t ppt = P->partial_per_tmpl;
    float* row_max = P->row_max;
    int32_t* row_arg = P->row_arg;
    const bool live = lane < nrows;
    const float ie = live ? P->inv_e[rel0 + lane] : 0.f;
    const uint32_t group_bytes = (uint32_t)tiles_per_blk * 256u;  // one (block, group): tiles x 64 floats
    for (int t = 0; t < ntmpl; ++t) {
        const int64_t g0 = (int64_t)z * ngroups + (int64_t)t * gpt;
        const __amdgpu_buffer_rsrc_t rv = buf_of(uniform_ptr(P->vmax + g0 * tiles_per_blk * 64), group_bytes * (uint32_t)gpt);
        const __amdgpu_buffer_rsrc_t ri = buf_of(uniform_ptr(P->imax + g0 * tiles_per_blk * 64), group_bytes * (uint32_t)gpt);
        // Across the groups the rule of the surface paths applies: the maximum of the NORMALISED float32 values, first
        // hypothesis on ties (groups come in increasing hypothesis order).  Inside a group the FFT role compared raw
        // |y|^2 (it does not have the delay's normalisation at hand), so the two modes can name different hypotheses
        // only when two hypotheses of ONE group are within a float32 ulp of each other -- and both then hold the
        // reported maximum.
        // (nosurf == 2, the hypothesis-major surface mode: the items compared and stored the normalised values themselves)
        const float gn = P->nosurf == 2 ? 1.f : ie * P->tscale[t];
        float bv = -1.f, bx = -1.f;
        int32_t bh = 0;
        for (int g = 0; g < gpt; ++g) {
            const int voff = (tile * 64 + lane) * 4;
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff, g * (int)group_bytes, CAF_AUX_SC1));
            const int32_t hh = __builtin_amdgcn_raw_buffer_load_b32(ri, voff, g * (int)group_bytes, CAF_AUX_SC1);
            const float xg = v * gn;
            if (v >= 0.f && (xg > bx || bv < 0.f)) {
                bx = xg;
                bv = v;
                bh = hh;
            }
        }
        const float x = bv * gn;
        const int32_t f = bh - t * nfreq;
        float best = -1.f;
        int32_t bdel = 0x7fffffff, bfrq = 0;
        if (live) {
            const int64_t o = (int64_t)t * num_shifts + rel0 + lane;
            const float xv = bv < 0.f ? -1.f : x;  // (no hypothesis beat the initial value: all NaN)
            // zero-energy window (the factor is NaN) or all-NaN inputs: NaN and hypothesis 0, as every other engine reports it
            const bool dead = bv < 0.f || x != x;
            if (row_max) row_max[o] = dead ? __builtin_nanf("") : x;
            if (row_arg) row_arg[o] = dead ? 0 : f;
            if (xv > best) {
                best = xv;
                bdel = lane;
                bfrq = bv < 0.f ? 0 : f;
            }
        }
        if (partial) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(best, o, 64);
                const int32_t od = __shfl_xor(bdel, o, 64);
                const int32_t of = __shfl_xor(bfrq, o, 64);
                if (ov > best || (ov == best && od < bdel)) {
                    best = ov;
                    bdel = od;
                    bfrq = of;
                }
            }
            if (lane == 0) {
                PeakRec r;
                r.v = best;
                r.delay = bdel != 0x7fffffff ? (int32_t)(shift_start + rel0 + bdel) : 0x7fffffff;
                r.f = bfrq;
                partial[(int64_t)t * ppt + pidx] = r;
            }
        }
    }
}

// waits (bounded) until every hypothesis group of the block of tile item `item` is published
__device__ __forceinline__ void pq_wait_block(int32_t* pq, int item, int ipb, int ngroups, bool lane0) {
    // The block's FFT items are either running on resident workgroups or still in the FFT queue, which the
    // workgroups that do not prefer tiles keep draining (they never wait while FFT items are left, and workgroup
    // 0 is always one of them).  Watchdog (~10 s of polling): unreachable unless the protocol is broken; it leaves
    // a mark in pq[2..3] and TRAPS, so that the launch fails loudly (the next synchronisation reports an error)
    // instead of hanging the GPU or transposing unfinished data.
    int spins = 0;
    while (__builtin_amdgcn_readfirstlane(pq_load(&pq[PQ_DONE + item / ipb])) < ngroups) {
        __builtin_amdgcn_s_sleep(64);
        if (++spins > (1 << 22)) {
            if (lane0) {
                atomicExch(&pq[2], 1 + item);
                atomicExch(&pq[3], pq_load(&pq[PQ_FFT_NEXT]));
            }
            __builtin_trap();
        }
    }
}

// The tile role of k_caf_persistent, out of line for the same reason as the FFT role (its two register sets of
// loads in flight need an allocation of their own).  Processes tile item `item_in`, then keeps taking tile items
// as long as the head of the tile queue is ready (or no FFT item is left), so that the call overhead is paid per
// run of items; returns when the workgroup should look at the FFT queue again.
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) int32_t lds_int;
__device__ __attribute__((noinline)) void persistent_tile_run(lds_float* lds_in, lds_int* s_next_in,
                                                              const PersistParams* pp_in, int item_in) {
    const PersistParams* pp = uniform_ptr(pp_in);
    int item = __builtin_amdgcn_readfirstlane(item_in);
    float* lds = (float*)lds_in;
    int32_t* s_next = (int32_t*)s_next_in;
    const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool lane0 = (threadIdx.x & 63) == 0;
    for (;;) {
        const CAF_AS4 PersistParams* P = params_of(pp);
        const int ipb = P->ipb;
        const int z = item / ipb;
        if (P->nosurf)
            reduce_wave_nosurf(pp, z, (item - z * ipb) * PQ_TILES);
        else if (P->dstride == 2) {  // folded 65536-point role: tiles [r][256] of every second delay
            if (P->nfreq == 1)
                transpose_wave_f1<2>(pp, z, (item - z * ipb) * PQ_TILES);
            else if (P->surface)
                transpose_wave<true, 2>(lds, pp, z, (item - z * ipb) * PQ_TILES);
            else
                transpose_wave<false, 2>(lds, pp, z, (item - z * ipb) * PQ_TILES);
        } else if (P->nfreq == 1)
            transpose_wave_f1<1>(pp, z, (item - z * ipb) * PQ_TILES);
        else if (P->surface)
            transpose_wave<true, 1>(lds, pp, z, (item - z * ipb) * PQ_TILES);
        else
            transpose_wave<false, 1>(lds, pp, z, (item - z * ipb) * PQ_TILES);
        if (wave_id == 0) {
            int32_t* pq = P->pq;
            const int n_fft = P->n_fft, n_tr = P->n_tr, ngroups = P->ngroups;
            int next = -1;
            const int t = __builtin_amdgcn_readfirstlane(pq_load(&pq[PQ_TR_NEXT]));
            if (t < n_tr) {
                const bool fft_left = __builtin_amdgcn_readfirstlane(pq_load(&pq[PQ_FFT_NEXT])) < n_fft;
                if (!fft_left || __builtin_amdgcn_readfirstlane(pq_load(&pq[PQ_DONE + t / ipb])) >= ngroups) {
                    int my = 0;
                    if (lane0) my = atomicAdd(&pq[PQ_TR_NEXT], 1);
                    my = __builtin_amdgcn_readfirstlane(my);
                    if (my < n_tr) {
                        pq_wait_block(pq, my, ipb, ngroups, lane0);
                        next = my;
                    }
                }
            }
            if (lane0) *s_next = next;
        }
        __syncthreads();  // every wave has finished its tile; the next item id is published
        item = __builtin_amdgcn_readfirstlane(*s_next);
        __syncthreads();  // (s_next may be rewritten only after everybody has read it)
        if (item < 0) return;
    }
}

__device__ __forceinline__ void pq_mark(const PersistParams* pp, int slot, int v) {
    int32_t* d = params_of(pp)->dbg;
    if (d) __hip_atomic_store(&d[blockIdx.x * 8 + slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The FFT role of k_caf_persistent, called out of line so that the hypothesis loop keeps a register allocation
// of its own (it needs all 128 VGPRs; allocated together with the queue logic and the tile role it reloads
// spilled values every hypothesis).  Called ~21 times per workgroup: the call and the callee-saved register
// traffic are negligible.  Arguments arrive in VGPRs; v_readfirstlane makes them scalar again.
typedef __attribute__((address_space(3))) float2 lds_float2;
template <int KIND, int NV4>  // 0: |y|^2 tiles, 1: running maxima (no surface), 2 / 4: finished rows (no frequency scan; 4: + peak records), 3: complex rows, 5: hypothesis-major surface rows + running maxima
__device__ __attribute__((noinline)) void persistent_fft_item(lds_float2* s_d, const lds_float2* s_tw2,
                                                              const lds_float2* s_tw3, const PersistParams* pp_in,
                                                              int item_in) {
    const PersistParams* pp = uniform_ptr(pp_in);
    const int item = __builtin_amdgcn_readfirstlane(item_in);
    const CAF_AS4 PersistParams* P = params_of(pp);
    const int ngroups = P->ngroups, hyp_per_wg = P->hyp_per_wg, nhyp = P->nhyp;
    const int blk = item / ngroups;
    const int grp = item - blk * ngroups;
    int h0 = grp * hyp_per_wg;
    int h1 = min(h0 + hyp_per_wg, nhyp);
    if (const int gpt = P->gpt) {  // groups formed per template (they never straddle two templates)
        const int t = grp / gpt;
        h0 = t * P->nfreq + (grp - t * gpt) * hyp_per_wg;
        h1 = min(h0 + hyp_per_wg, (t + 1) * P->nfreq);
    }
    if (KIND == 3) {
        F1Direct f1;
        f1.out0 = P->cqf;
        f1.out1 = nullptr;
        f1.partial = nullptr;
        f1.ppt = 0;
        f1.shift_start = 0;
        f1.inv_e = P->inv_e;
        f1.tscale = P->tscale;
        f1.num_shifts = P->num_shifts;
        f1.step = P->step;
        f1.blk_abs = P->blk0 + blk;
        fused_item<1024, 4, NV4>((float2*)s_d, (const float2*)s_tw2, (const float2*)s_tw3, P->xb, P->hc, P->shifts, P->tw1,
                                 P->table_mode, P->nfreq, nhyp, blk, h0, h1, P->tiles_per_blk, nullptr, nullptr, &f1);
    } else if (KIND == 2 || KIND == 4) {
        F1Direct f1;
        f1.out0 = P->row_max ? P->row_max : P->surface;  // (neither: peak records only -- the row stores are dropped)
        f1.out1 = (P->row_max && P->surface) ? P->surface : nullptr;
        f1.inv_e = P->inv_e;
        f1.tscale = P->tscale;
        f1.num_shifts = P->num_shifts;
        f1.step = P->step;
        f1.blk_abs = P->blk0 + blk;
        f1.partial = P->partial;
        f1.ppt = P->partial_per_tmpl;
        f1.shift_start = P->shift_start;
        fused_item<1024, 3, NV4, KIND == 4>((float2*)s_d, (const float2*)s_tw2, (const float2*)s_tw3, P->xb, P->hc, P->shifts, P->tw1,
                                            P->table_mode, P->nfreq, nhyp, blk, h0, h1, P->tiles_per_blk, nullptr, nullptr, &f1);
    } else if (KIND == 5) {
        // hypothesis-major surface rows + one (maximum of the written values, hypothesis) pair per delay and item
        F1Direct f1;
        f1.out0 = P->surface_t;
        f1.out1 = nullptr;
        f1.partial = nullptr;
        f1.ppt = 0;
        f1.shift_start = 0;
        f1.inv_e = P->inv_e;
        f1.tscale = P->tscale;
        f1.num_shifts = P->num_shifts;
        f1.step = P->step;
        f1.blk_abs = P->blk0 + blk;
        const int64_t o = ((int64_t)blk * ngroups + grp) * P->tiles_per_blk * 64;
        fused_item<1024, 5, NV4>((float2*)s_d, (const float2*)s_tw2, (const float2*)s_tw3, P->xb, P->hc, P->shifts, P->tw1,
                                 P->table_mode, P->nfreq, nhyp, blk, h0, h1, P->tiles_per_blk, P->vmax + o, P->imax + o, &f1);
    } else if (KIND == 1) {
        // no surface wanted: one (maximum, hypothesis) pair per delay and item instead of the |y|^2 tiles
        const int64_t o = ((int64_t)blk * ngroups + grp) * P->tiles_per_blk * 64;
        fused_item<1024, 2, NV4>((float2*)s_d, (const float2*)s_tw2, (const float2*)s_tw3, P->xb, P->hc, P->shifts, P->tw1,
                            P->table_mode, P->nfreq, nhyp, blk, h0, h1, P->tiles_per_blk, P->vmax + o, P->imax + o);
    } else {
        // the |y|^2 tiles are stored write-through (sc1): device-visible once the store has completed
        fused_item<1024, 1, NV4>((float2*)s_d, (const float2*)s_tw2, (const float2*)s_tw3, P->xb, P->hc, P->shifts, P->tw1,
                            P->table_mode, P->nfreq, nhyp, blk, h0, h1, P->tiles_per_blk, P->vt);
    }
    // publish: every wave waits for its own stores (workgroup-scope release = s_waitcnt vmcnt(0); an
    // agent-scope release would add an L2 write-back per item, which stalls the 32 CUs sharing that L2:
    // measured +30 % on every FFT item), then one thread counts the group in
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add(&params_of(pp)->pq[PQ_DONE + blk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// FFT role for 32768-point blocks (templates of 8193 .. 16384 samples): fused_item2q, same publish sequence
// (CQ: the complex-QF rows instead of the |y|^2 tiles -- PersistParams::cqf, no tile items)
__device__ __forceinline__ void cq_params(const CAF_AS4 PersistParams* P, int blk, F1Direct& f1) {
    f1.out0 = P->cqf;
    f1.out1 = nullptr;
    f1.partial = nullptr;
    f1.ppt = 0;
    f1.shift_start = 0;
    f1.inv_e = P->inv_e;
    f1.tscale = P->tscale;
    f1.num_shifts = P->num_shifts;
    f1.step = P->step;
    f1.blk_abs = P->blk0 + blk;
}
template <int NVH, bool CQ>
__device__ __attribute__((noinline)) void persistent_fft_item2(lds_float2* s_d, const lds_float2* s_tw2, const lds_float2* s_tw3,
                                                               const PersistParams* pp_in, int item_in) {
    const PersistParams* pp = uniform_ptr(pp_in);
    const int item = __builtin_amdgcn_readfirstlane(item_in);
    const CAF_AS4 PersistParams* P = params_of(pp);
    const int ngroups = P->ngroups, hyp_per_wg = P->hyp_per_wg, nhyp = P->nhyp;
    const int blk = item / ngroups;
    const int grp = item - blk * ngroups;
    const int h0 = grp * hyp_per_wg;
    const int h1 = min(h0 + hyp_per_wg, nhyp);
    if constexpr (CQ) {
        F1Direct f1;
        cq_params(P, blk, f1);
        fused_item2q<4, NVH, false, 0>((float2*)s_d, (const float2*)s_tw2, (const float2*)s_tw3, P->xb, P->hc, P->shifts, P->tw1,
                                       P->table_mode, P->nfreq, nhyp, blk, h0, h1, P->tiles_per_blk, P->vt, 1, &f1);
    } else
    fused_item2q<1, NVH, false, 0>((float2*)s_d, (const float2*)s_tw2, (const float2*)s_tw3, P->xb, P->hc, P->shifts, P->tw1, P->table_mode,
                                   P->nfreq, nhyp, blk, h0, h1, P->tiles_per_blk, P->vt);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add(&params_of(pp)->pq[PQ_DONE + blk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// FFT role for 65536-point blocks (templates of 16385 .. 32768 samples; fused_item2q<FOLD>): work item = (block, hypothesis group, output residue r), group
// number = 2 * (hypothesis group) + r, same publish sequence.  PART: templates of 32769 .. npart * 32768 samples as npart partitions
// (template-spectrum rows [spectrum][npart], block spectra of the npart blocks from the item's own on)
template <int R, bool PART, bool CQ>
__device__ __attribute__((noinline)) void persistent_fft_item2f(lds_float2* s_d, const lds_float2* s_tw2, const lds_float2* s_tw3,
                                                                const PersistParams* pp_in, int item_in) {
    const PersistParams* pp = uniform_ptr(pp_in);
    const int item = __builtin_amdgcn_readfirstlane(item_in);
    const CAF_AS4 PersistParams* P = params_of(pp);
    const int ngroups = P->ngroups, hyp_per_wg = P->hyp_per_wg, nhyp = P->nhyp;
    const int blk = item / ngroups;
    const int grp = (item - blk * ngroups) >> 1;
    const int h0 = grp * hyp_per_wg;
    const int h1 = min(h0 + hyp_per_wg, nhyp);
    if constexpr (CQ) {
        F1Direct f1;
        cq_params(P, blk, f1);
        fused_item2q<4, 0, true, R, PART>((float2*)s_d, (const float2*)s_tw2, (const float2*)s_tw3, P->xb, P->hc, P->shifts, P->tw1,
                                          P->table_mode, P->nfreq, nhyp, blk, h0, h1, P->tiles_per_blk, P->vt, PART ? P->npart : 1, &f1);
    } else
    fused_item2q<1, 0, true, R, PART>((float2*)s_d, (const float2*)s_tw2, (const float2*)s_tw3, P->xb, P->hc, P->shifts, P->tw1, P->table_mode,
                                      P->nfreq, nhyp, blk, h0, h1, P->tiles_per_blk, P->vt, PART ? P->npart : 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add(&params_of(pp)->pq[PQ_DONE + blk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool STATS>
__global__ __launch_bounds__(1024) void k_caf_persistent(const PersistParams* __restrict__ pp) {
    static_assert(16 * TW_LDS * 4 <= F_LDS_DATA * 8, "transposer patches must fit the FFT image");
    __shared__ __attribute__((aligned(16))) float2 s_d[F_LDS_DATA];
    __shared__ float2 s_tw2[2 * 16 * 64];  // [0]: every 16384-point transform; [1]: the odd half of a 32768-point block
    __shared__ float2 s_tw3[2 * 16 * 4];
    __shared__ int32_t s_cmd[2];
    const int tid = threadIdx.x;
    {
        const CAF_AS4 PersistParams* P = params_of(pp);
        const float2* tw23 = P->tw23;
        s_tw2[tid] = tw23[tid];
        s_tw2[1024 + tid] = tw23[1088 + tid];
        if (tid < 64) {
            s_tw3[tid] = tw23[1024 + tid];
            s_tw3[64 + tid] = tw23[1088 + 1024 + tid];
        }
    }
    __syncthreads();
    // The claim runs on wave 0 as SCALAR control flow (uniform values, scalar branches); only the atomic
    // read-modify-writes are predicated on lane 0.  A divergent `if (tid == 0)` around the claim loop lets the
    // compiler restructure the work loop so that the other lanes of wave 0 run ahead to the barrier.
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool lane0 = (tid & 63) == 0;
    // per-workgroup role statistics for CAF_PERSIST_DEBUG (100 MHz wall clock; wave 0 only)
    uint64_t tmark = STATS ? wall_clock64() : 0;
    const uint32_t t_start = (uint32_t)tmark;
    uint32_t t_claim = 0, t_fft = 0, t_tile = 0, c_fft = 0, c_tile = 0;
    for (;;) {
        if (wave_id == 0) {
            const CAF_AS4 PersistParams* P = params_of(pp);
            int32_t* pq = P->pq;
            const int n_fft = P->n_fft, n_tr = P->n_tr, ipb = P->ipb, ngroups = P->ngroups;
            // workgroups are dealt round-robin over the 8 XCDs: slot = index within the XCD.  The LAST tr_slots of
            // every 32 slots prefer tiles, so any resident prefix of the grid contains workgroups that do not.
            const bool prefer_tr = (int)((blockIdx.x >> 3) & 31) >= 32 - P->tr_slots;
            int kind = 0, item = 0, spins = 0;
            for (;;) {
                const bool fft_left = __builtin_amdgcn_readfirstlane(pq_load(&pq[PQ_FFT_NEXT])) < n_fft;
                if (prefer_tr || !fft_left) {
                    const int t = __builtin_amdgcn_readfirstlane(pq_load(&pq[PQ_TR_NEXT]));
                    if (t < n_tr) {
                        // take a tile item when the head of the queue is ready -- or when there is nothing else
                        // to do.  fetch-add, not compare-and-swap: with 256 claimers a CAS loop costs O(claimers)
                        // failed atomics per item (measured: 8x slower end to end).  The item obtained may lie a few
                        // blocks past the head; its block is waited for below.
                        if (!fft_left || __builtin_amdgcn_readfirstlane(pq_load(&pq[PQ_DONE + t / ipb])) >= ngroups) {
                            int my = 0;
                            if (lane0) my = atomicAdd(&pq[PQ_TR_NEXT], 1);
                            my = __builtin_amdgcn_readfirstlane(my);
                            if (my < n_tr) {
                                kind = 2;
                                item = my;
                                break;
                            }
                            continue;
                        }
                    } else if (!fft_left) {
                        break;  // both queues fully claimed: exit
                    }
                }
                if (fft_left) {
                    // FFT items are dealt per XCD: workgroup L runs on XCD L % 8, and XCD x owns the blocks b = x (mod 8),
                    // so the hypothesis groups of one block run on CUs that share an L2 -- its block spectrum (which the
                    // 32768-point role re-reads for every hypothesis) and the template-spectrum rows are served from
                    // there instead of the fabric.  A workgroup whose own list is exhausted takes from the others'.
                    // (Only where a role re-reads the spectrum per hypothesis: with the spectrum register-resident the
                    // plain order is 0.5 % faster at C2.)
                    const int nblk = P->nblk;
                    const int x0 = (int)(blockIdx.x & 7);
                    int got = -1;
                    if (P->block_log2 < 15) {
                        int i = 0;
                        if (lane0) i = atomicAdd(&pq[PQ_FFT_XCD], 1);
                        i = __builtin_amdgcn_readfirstlane(i);
                        if (i < n_fft) got = i;
                    }
                    for (int k = 0; k < 8 && got < 0 && P->block_log2 >= 15; ++k) {
                        const int x = (x0 + k) & 7;
                        const int items_x = ((nblk - x + 7) >> 3) * ngroups;  // blocks x, x + 8, ... < nblk
                        if (__builtin_amdgcn_readfirstlane(pq_load(&pq[PQ_FFT_XCD + x])) >= items_x) continue;
                        int j = 0;
                        if (lane0) j = atomicAdd(&pq[PQ_FFT_XCD + x], 1);
                        j = __builtin_amdgcn_readfirstlane(j);
                        if (j < items_x) got = ((j / ngroups) * 8 + x) * ngroups + (j - (j / ngroups) * ngroups);
                    }
                    if (got >= 0) {
                        if (lane0) atomicAdd(&pq[PQ_FFT_NEXT], 1);  // (the count that "FFT items left?" reads)
                        kind = 1;
                        item = got;
                        break;
                    }
                    continue;
                }
                __builtin_amdgcn_s_sleep(64);
                if (++spins > (1 << 22)) break;  // cannot happen (see the wait below); kind 0 = leave
            }
            if (kind == 2) pq_wait_block(pq, item, ipb, ngroups, lane0);  // a claimed tile item waits for its block
            if (lane0) {
                s_cmd[0] = kind;
                s_cmd[1] = item;
            }
            if (STATS) {
                const uint64_t now = wall_clock64();
                t_claim += (uint32_t)(now - tmark);
                tmark = now;
            }
        }
        __syncthreads();
        // uniform by construction; as scalars, everything derived from the item id stays in SGPRs
        const int kind = __builtin_amdgcn_readfirstlane(s_cmd[0]), item = __builtin_amdgcn_readfirstlane(s_cmd[1]);
        if (kind == 0) {
            if (STATS && tid == 0) {
                pq_mark(pp, 0, (int)t_claim);
                pq_mark(pp, 1, (int)t_fft);
                pq_mark(pp, 2, (int)t_tile);
                pq_mark(pp, 3, (int)c_fft);
                pq_mark(pp, 4, (int)c_tile);
                pq_mark(pp, 5, (int)t_start);
                pq_mark(pp, 6, (int)(uint32_t)wall_clock64());
            }
            break;
        }
        if (kind == 1) {
            if (__builtin_amdgcn_readfirstlane(params_of(pp)->block_log2) == 16) {
                const bool odd_r = ((item - (item / params_of(pp)->ngroups) * params_of(pp)->ngroups) & 1) != 0;
                const bool part = __builtin_amdgcn_readfirstlane(params_of(pp)->npart) > 1, cq = params_of(pp)->cqf != nullptr;
#define CAF_FFT_ROLE2F(RR, PP, CC) persistent_fft_item2f<RR, PP, CC>((lds_float2*)s_d, (const lds_float2*)s_tw2, (const lds_float2*)s_tw3, pp, item)
                if (cq) {
                    if (part) { if (odd_r) CAF_FFT_ROLE2F(1, true, true); else CAF_FFT_ROLE2F(0, true, true); }
                    else { if (odd_r) CAF_FFT_ROLE2F(1, false, true); else CAF_FFT_ROLE2F(0, false, true); }
                } else if (part) {
                    if (odd_r) CAF_FFT_ROLE2F(1, true, false); else CAF_FFT_ROLE2F(0, true, false);
                } else {
                    if (odd_r) CAF_FFT_ROLE2F(1, false, false); else CAF_FFT_ROLE2F(0, false, false);
                }
#undef CAF_FFT_ROLE2F
            } else if (__builtin_amdgcn_readfirstlane(params_of(pp)->block_log2) == 15) {
                const int tpb2 = __builtin_amdgcn_readfirstlane(params_of(pp)->tiles_per_blk);  // tiles >= 256: upper half
                const bool cq = params_of(pp)->cqf != nullptr;
#define CAF_FFT_ROLE2(NN, CC) persistent_fft_item2<NN, CC>((lds_float2*)s_d, (const lds_float2*)s_tw2, (const lds_float2*)s_tw3, pp, item)
                if (tpb2 <= 256) { if (cq) CAF_FFT_ROLE2(0, true); else CAF_FFT_ROLE2(0, false); }
                else if (tpb2 <= 384) { if (cq) CAF_FFT_ROLE2(2, true); else CAF_FFT_ROLE2(2, false); }
                else { if (cq) CAF_FFT_ROLE2(4, true); else CAF_FFT_ROLE2(4, false); }
#undef CAF_FFT_ROLE2
            }
            else {
                // (kind of output) x (valid quarters of the block: tiles <= 128 / 192 / 256) -> one out-of-line role each
                const int kind3 = params_of(pp)->cqf ? 3
                                  : params_of(pp)->surface_t ? 5
                                  : __builtin_amdgcn_readfirstlane(params_of(pp)->f1_direct) ? (params_of(pp)->partial ? 4 : 2)
                                  : __builtin_amdgcn_readfirstlane(params_of(pp)->nosurf) ? 1 : 0;
                const int tpb = __builtin_amdgcn_readfirstlane(params_of(pp)->tiles_per_blk);
#define CAF_FFT_ROLE(K, Q) persistent_fft_item<K, Q>((lds_float2*)s_d, (const lds_float2*)s_tw2, (const lds_float2*)s_tw3, pp, item)
                if (tpb <= 128) {
                    if (kind3 == 5) CAF_FFT_ROLE(5, 2); else if (kind3 == 4) CAF_FFT_ROLE(4, 2); else if (kind3 == 3) CAF_FFT_ROLE(3, 2); else if (kind3 == 2) CAF_FFT_ROLE(2, 2); else if (kind3 == 1) CAF_FFT_ROLE(1, 2); else CAF_FFT_ROLE(0, 2);
                } else if (tpb <= 192) {
                    if (kind3 == 5) CAF_FFT_ROLE(5, 3); else if (kind3 == 4) CAF_FFT_ROLE(4, 3); else if (kind3 == 3) CAF_FFT_ROLE(3, 3); else if (kind3 == 2) CAF_FFT_ROLE(2, 3); else if (kind3 == 1) CAF_FFT_ROLE(1, 3); else CAF_FFT_ROLE(0, 3);
                } else {
                    if (kind3 == 5) CAF_FFT_ROLE(5, 4); else if (kind3 == 4) CAF_FFT_ROLE(4, 4); else if (kind3 == 3) CAF_FFT_ROLE(3, 4); else if (kind3 == 2) CAF_FFT_ROLE(2, 4); else if (kind3 == 1) CAF_FFT_ROLE(1, 4); else CAF_FFT_ROLE(0, 4);
                }
#undef CAF_FFT_ROLE
            }
            if (STATS) {
                const uint64_t now = wall_clock64();
                t_fft += (uint32_t)(now - tmark);
                tmark = now;
                ++c_fft;
            }
        } else {
            persistent_tile_run((lds_float*)s_d, (lds_int*)&s_cmd[1], pp, item);
            if (STATS) {
                const uint64_t now = wall_clock64();
                t_tile += (uint32_t)(now - tmark);
                tmark = now;
                ++c_tile;
            }
        }
    }
}

// writes the parameter block (captured by value at launch time) and clears the queues
__global__ void k_persist_setup(PersistParams h, PersistParams* d) {
    int32_t* pq = h.pq;
    const int n = PQ_DONE + h.nblk;
    for (int i = threadIdx.x; i < n; i += blockDim.x) pq[i] = 0;
    if (threadIdx.x == 0) *d = h;
}

void launch_caf_persistent(const PersistParams* h, PersistParams* d_params, int32_t n_wgs, hipStream_t st) {
    hipLaunchKernelGGL(k_persist_setup, dim3(1), dim3(256), 0, st, *h, d_params);
    const dim3 grid((unsigned)n_wgs), block(1024);
    const PersistParams* dp = d_params;
    if (h->dbg)  // CAF_PERSIST_DEBUG: the variant that also keeps per-role clocks
        hipLaunchKernelGGL((k_caf_persistent<true>), grid, block, 0, st, dp);
    else
        hipLaunchKernelGGL((k_caf_persistent<false>), grid, block, 0, st, dp);
}

void launch_fused_caf(const float2* xb, const float2* hc, const int32_t* shifts, const float2* tw1,
                      const float2* tw23, int32_t table_mode, int32_t nfreq, int32_t nhyp, int32_t hyp_per_wg,
                      int32_t nblk, int32_t tiles_per_blk, float* vt, hipStream_t st) {
    const int ngroups = (nhyp + hyp_per_wg - 1) / hyp_per_wg;
    const dim3 grid((unsigned)(ngroups * 8 * ((nblk + 7) / 8)));  // 1-D, XCD-aware mapping inside the kernel
    hipLaunchKernelGGL(k_fused_caf<1024>, grid, dim3(1024), 0, st, xb, hc, shifts, tw1, tw23, table_mode, nfreq, nhyp,
                       hyp_per_wg, nblk, tiles_per_blk, vt);
}

void launch_transpose_norm_argmax(const float* vt, int32_t ntmpl, int32_t nfreq, const float* tscale,
                                  const float* inv_e, int64_t num_shifts, int64_t shift_start, int32_t step,
                                  int32_t blk0, int32_t nblk, int32_t tiles_per_blk, float* surface, float* row_max,
                                  int32_t* row_arg, PeakRec* partial, int64_t partial_per_tmpl, hipStream_t st) {
    // CAF_TR_F=64|256 selects another chunk width (A/B switch; 128 measured fastest at F = 256:
    // 256 halves the occupancy, 64 halves the store segments)
    static const int trf = [] {
        const char* e = getenv("CAF_TR_F");
        return e ? atoi(e) : 128;
    }();
    static const int nts = [] {
        const char* e = getenv("CAF_NT_STORE");
        return e ? atoi(e) : 0;
    }();
    const dim3 grid(tiles_per_blk, ntmpl, nblk);
    if (trf == 256 && nfreq > 128)
        hipLaunchKernelGGL((k_transpose_norm_argmax<256, false>), grid, dim3(256), 0, st, vt, ntmpl, nfreq, tscale,
                           inv_e, num_shifts, shift_start, step, blk0, tiles_per_blk, surface, row_max, row_arg,
                           partial, partial_per_tmpl);
    else if (trf == 64)
        hipLaunchKernelGGL((k_transpose_norm_argmax<64, false>), grid, dim3(256), 0, st, vt, ntmpl, nfreq, tscale,
                           inv_e, num_shifts, shift_start, step, blk0, tiles_per_blk, surface, row_max, row_arg,
                           partial, partial_per_tmpl);
    else if (nts)
        hipLaunchKernelGGL((k_transpose_norm_argmax<128, true>), grid, dim3(256), 0, st, vt, ntmpl, nfreq, tscale,
                           inv_e, num_shifts, shift_start, step, blk0, tiles_per_blk, surface, row_max, row_arg,
                           partial, partial_per_tmpl);
    else
        hipLaunchKernelGGL((k_transpose_norm_argmax<128, false>), grid, dim3(256), 0, st, vt, ntmpl, nfreq, tscale,
                           inv_e, num_shifts, shift_start, step, blk0, tiles_per_blk, surface, row_max, row_arg,
                           partial, partial_per_tmpl);
}

}  // namespace caf
