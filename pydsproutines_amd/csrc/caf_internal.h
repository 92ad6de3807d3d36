// Internal declarations shared by the kernel and plan translation units of libcaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "caf.h"
#include "caf_energy.h"

#include <map>
#include <mutex>
#include <string>
#include <utility>

namespace caf {

constexpr int MUL_THREADS = 256;  // spectral multiply: 256 threads x 2 points
constexpr int MAG_THREADS = 256;  // |.|^2/normalise/argmax: 4 waves
constexpr int MAG_S = 64;         // delays per tile (one wave-row of 8-byte loads = 512 B)
constexpr int MAG_F = 128;        // frequency hypotheses per LDS chunk (512 B store rows)

// Window energies.  The engines take the energy of a window as a difference of two entries of a float64 prefix of |rx|^2;
// a difference resolves nothing below ~2^-50 of the prefix it is taken from.  Where it comes out below 2^-30 of the upper
// entry (a window more than ~50 dB under everything in front of it) the energy is summed again directly, in float64, with
// no subtraction: the <= 63 samples before the first and after the last 64-sample boundary inside the window, and the
// whole 64-sample chunks between them from an array of chunk energies that the prefix pass leaves behind the prefix (each a
// plain sum of 64 squares) -- <= 126 + N / 64 additions, so a record FULL of such windows costs a bounded amount.  NaN
// therefore means an energy of exactly zero on every engine, as in the reference's 0 / 0 (xcorrRoutines.py:527-528,
// IppXcorrFFT.cpp:174); rounds 1-4 reported anything below 2^-44 of the prefix as zero energy, with the floor in different
// places on different engines.  (caf_energy.h; tests/test_gpu_engine.py::test_quiet_windows_*)
// a prefix buffer for m samples: m + 1 prefix entries (padded to an even count), then the chunk energies
inline int64_t energy_prefix_doubles(int64_t m) { return ((m + 2) & ~(int64_t)1) + (m + 63) / 64 + 1; }

struct PeakRec {
    float v;
    int32_t delay;
    int32_t f;
};

// error plumbing (thread-local message behind caf_last_error)
void set_error(const std::string& msg);
#define CAF_HIP_TRY(expr)                                                                          \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            caf::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                     \
            return (_e == hipErrorOutOfMemory) ? CAF_ERR_NOMEM : CAF_ERR_HIP;                      \
        }                                                                                          \
    } while (0)

// MaxDynamicSharedMemorySize is an attribute of a kernel PER DEVICE (a process may drive several): raised once per
// (kernel, device), remembered under a lock.
inline int allow_dynamic_lds(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> granted;
    int dev = 0;
    CAF_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    size_t& have = granted[std::make_pair(kernel, dev)];
    if (have >= bytes) return CAF_OK;
    CAF_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
    return CAF_OK;
}

// pageable host memory <-> device through the library's own pinned staging lanes (caf_host.cpp); complete on return
int host_h2d(void* d_dst, const void* h_src, int64_t bytes, hipStream_t st);
int host_d2h(void* h_dst, const void* d_src, int64_t bytes, hipStream_t st);
int host_d2h_f64(double* h_dst, const float* d_src, int64_t count, hipStream_t st);  // float32 on the device -> float64 on the host
int host_d2h_transposed(void* h_dst, bool dst_f64, const float* d_src, int64_t rows, int64_t pitch, int64_t col0, int64_t ncols,
                        hipStream_t st);
// blocking upload of host memory of any size (a caller's array, a std::vector about to be freed) on the null stream
#define CAF_H2D(dst, src, bytes)                                                        \
    do {                                                                                \
        const int _rc = caf::host_h2d((dst), (src), (int64_t)(bytes), nullptr);         \
        if (_rc) return _rc;                                                            \
    } while (0)

// caching device allocator (caf_pool.hip)
int pool_alloc(void** out, int64_t bytes);
int pool_free(void* p);
void pool_trim();
void pool_stats(int64_t* cached, int64_t* in_use, int64_t* hits, int64_t* misses);

int64_t prefix_num_tiles(int64_t m);
// prefix: energy_prefix_doubles(m) entries (the prefix itself, then the 64-sample chunk energies: caf_energy.h)
void launch_energy_prefix(const float2* rx, int64_t m, double* tile_sums, double* prefix, hipStream_t st);
void launch_inv_energy(const float2* rx, int64_t rx_len, const double* prefix, int64_t shift_start, int64_t num_shifts,
                       const int32_t* gstart, const int32_t* glen, int32_t ngroups, float* inv_e, hipStream_t st);
// peak records from finished (T, S) rows: rows_peak_chunks(S) records per template
int64_t rows_peak_chunks(int64_t num_shifts);
void launch_rows_peak(const float* rows, int32_t ntmpl, int64_t num_shifts, int64_t shift_start, PeakRec* partial,
                      int64_t partial_per_tmpl, hipStream_t st);
// direct (time-domain) engine over nk <= 64 non-zero template positions (caf_direct.hip)
void launch_direct_caf(const float2* rx, int64_t shift_start, int64_t num_shifts, int32_t ntmpl, int32_t nfreq, int32_t nk,
                       const int32_t* pos, const float2* w, const float* tscale, float* surface, float* row_max,
                       int32_t* row_arg, PeakRec* partial, int64_t partial_per_tmpl, hipStream_t st);
void launch_gather_blocks(const float2* rx, int64_t rx_len, int64_t src0, int32_t step, int32_t bsz, int32_t nblk,
                          float2* xb, hipStream_t st);
void launch_conj_scale(float2* h, int64_t n, float scale, hipStream_t st);
void launch_build_hyp_time(const float2* tm, const double* nu, int32_t n_tmpl, int32_t bsz, int32_t nfreq, int32_t ntmpl,
                           int32_t conj_u, float2* hc, hipStream_t st, int32_t npart = 1, int32_t plen = 0);
void launch_spectral_mul(int mode, const float2* xb, const float2* hc, const int32_t* shifts, int32_t bsz,
                         int32_t pitch, int32_t nfreq, int32_t nhyp, int32_t hyp_per_wg, int32_t nblk, float2* pbuf,
                         hipStream_t st);
void launch_magsq(const float2* pbuf, int32_t pitch, int32_t ntmpl, int32_t nfreq, const float* tscale,
                  const float* inv_e, int64_t num_shifts, int64_t shift_start, int32_t step, int32_t blk0,
                  int32_t nblk, int32_t tiles_per_blk, float* surface, float* row_max, int32_t* row_arg,
                  PeakRec* partial, int64_t partial_per_tmpl, hipStream_t st);
constexpr int PEAK_PARTS = 64;  // first-stage slices per template of the two-stage peak reduction
void launch_peak_reduce(const PeakRec* partial, int64_t count, int64_t stride, int32_t ntmpl, PeakRec* scratch,
                        float* pv, int32_t* pd, int32_t* pf, hipStream_t st);
void scan_tiles(double* tile_sums, int64_t ntiles, hipStream_t st);

// caf_rows.hip
void launch_sliding_multiply(const float2* x, int32_t xlen, const float2* y, int64_t ylen, const double* prefix,
                             int64_t start, int64_t step, int64_t rows, double coef, int32_t zero_oor, float2* z,
                             hipStream_t st, const double* d_coef = nullptr);  // d_coef: device scalar multiplied into coef (launch_cutout_norm's result)
// part: optional scratch of rows * rows_argmax_chunks(rows, len) 64-bit words -- long rows are then cut into chunks
void launch_rows_argmax(const float2* z, int64_t rows, int64_t len, int32_t use_normsq, float scale, uint32_t* argmax,
                        float* maxv, float* plane, hipStream_t st, unsigned long long* part = nullptr,
                        int32_t nan_empty = 0);  // nan_empty: an all-NaN row is (NaN, 0), not the CUDA workspace's (0, 0)
int rows_argmax_chunks(int64_t rows, int64_t len);
void launch_magnsq(const void* x, int64_t n, int in_c128, void* out, int out_f64, hipStream_t st);
int64_t moving_num_tiles(int64_t n);
void launch_moving_average(const float* x, int64_t n, int32_t L, int32_t sum_instead, double* tile_sums,
                           double* prefix, float* out, hipStream_t st);
int moving_tile_max_window();
void launch_moving_tile(const float* x, int64_t rows, int64_t n, int32_t L, int32_t sum_instead, float* out,
                        hipStream_t st);
void launch_complex_moving_sum(const float2* x, int64_t n, int32_t L, float* out, hipStream_t st);
void launch_multi_template_dot(const float2* tm, const float* te, int32_t ntm, int32_t L, const float2* x, int64_t xlen,
                               const double* prefix, int64_t start, int64_t nslides, int32_t* tidx, float* qf2,
                               hipStream_t st);
void launch_multiply_indexed_rows(const float2* x, int64_t xlen, const float2* rows, int32_t row_len,
                                  const int32_t* slice_start, const int32_t* slice_lens, const int32_t* row_idx,
                                  int32_t slice_len, int64_t nslices, float2* out, hipStream_t st);
void launch_copy_slices(const float2* x, int64_t xlen, const int32_t* starts, int32_t starts_stride, int64_t start0,
                        int64_t inc, int32_t len, int64_t rows, float2* out, hipStream_t st);
void launch_copy_groups(const float2* x, float2* y, const int32_t* xs, const int32_t* ys, const int32_t* lens,
                        int32_t ngroups, hipStream_t st);
int64_t local_maxima_scratch_ints(int64_t n);
void launch_find_local_maxima(const float* x, int64_t n, float min_height, int32_t* tile_scratch, int32_t max_out,
                              int32_t* idx, int32_t* count, hipStream_t st);
void launch_gather_b32(const void* x, int64_t xlen, const int32_t* idx, int64_t n, void* out, hipStream_t st);
void launch_gather_f32_f64(const float* x, int64_t xlen, const int32_t* idx, int64_t n, double* out, hipStream_t st);
void launch_fir(const float2* x, int64_t n, const float* taps, int32_t ntaps, const float2* delay, int32_t dlen,
                int32_t dsr, int32_t phase, float2* out, int64_t nout, hipStream_t st);
bool fir_decim_ok(int32_t ntaps, int32_t dsr);
bool fir_poly_fits(int32_t ntaps, int32_t dsr);  // ... and the register-tiled polyphase kernel takes it (small decimation factors)
void launch_iq16_fir(const int16_t* iq, int64_t n, float scale, const float* taps, int32_t ntaps, const int16_t* delay,
                     int32_t dlen, int32_t dsr, int32_t phase, float2* out, int64_t nout, hipStream_t st);
void launch_upfirdn(const float2* x, int64_t rows, int64_t n, const float* taps, int32_t ntaps, int32_t up, int32_t down,
                    int64_t nout, float2* out, float* out_abs, hipStream_t st);
void launch_rows_mul_vec(const float2* x, int64_t in_pitch, int64_t in_off, const float2* v, int64_t len, float2* y,
                         int64_t out_pitch, int64_t pad_to, int64_t rows, float scale, hipStream_t st);
void launch_complex_norm(const float2* pbuf, int32_t pitch, int32_t nfreq, const float* tscale, const float* inv_e,
                         int64_t num_shifts, int32_t step, int32_t blk0, int32_t nblk, int32_t nhyp, float2* cqf,
                         hipStream_t st);
void launch_scale(float2* y, int64_t n, float scale, hipStream_t st);
void launch_iq16_to_c64(const short* in, int64_t nsamp, float scale, float2* out, hipStream_t st);
void launch_argmax3d_u32(const uint32_t* x, int64_t items, int32_t d1, int32_t d2, int32_t d3, uint32_t* argmax,
                         uint32_t* maxv, hipStream_t st);

// caf_perdelay.hip: fused per-delay correlator (product -> LDS FFT -> |.|^2 -> argmax), power-of-two n in [64, 16384]
bool perdelay_fused_ok(int32_t n);
// cutouts of 100 / 1000 / 10000 samples: radix-10 in-LDS transform (energies from the caller's prefix, ||x|| from launch_cutout_norm)
bool perdelay_decimal_ok(int32_t n);
int launch_perdelay_decimal(const float2* x, int32_t n, const float2* y, int64_t ylen, const double* prefix, const double* xnorm,
                            int64_t start, int64_t step, int64_t num, int32_t zero_oor, float* qf2, uint32_t* fidx, float* plane,
                            float2* cplane, hipStream_t st);
int launch_perdelay_fused(const float2* x, int32_t n, const float2* y, int64_t ylen, int64_t start, int64_t step,
                          int64_t num, int32_t zero_oor, float* qf2, uint32_t* fidx, float* plane, float2* cplane,
                          hipStream_t st);
// cutouts of 2^a 3^b 5^c 7^d samples that are neither of the above (32 <= n <= 16200, a plan exists): mixed-radix in-LDS transform (caf_perdelay_mr.hip)
bool perdelay_mixed_ok(int32_t n);
int launch_perdelay_mixed(const float2* x, int32_t n, const float2* y, int64_t ylen, const double* prefix, const double* xnorm,
                          int64_t start, int64_t step, int64_t num, int32_t zero_oor, float* qf2, uint32_t* fidx, float* plane,
                          float2* cplane, hipStream_t st);
// the same lengths (and the powers of two and of ten) with a kernel compiled at run time for the length (caf_jit.hip, caf_perdelay_jit.h):
// false when CAF_JIT=0, hiprtc is missing or the length has no plan -- the callers then use the kernels above
bool perdelay_jit_ok(int32_t n);
void perdelay_jit_failed(int32_t n);  // (a length whose compilation failed: not tried again in this process)
int launch_perdelay_jit(const float2* x, int32_t n, const float2* y, int64_t ylen, const double* prefix, const double* xnorm,
                        int64_t start, int64_t step, int64_t num, int32_t zero_oor, float* qf2, uint32_t* fidx, float* plane,
                        float2* cplane, hipStream_t st);
int perdelay_jit_describe(int32_t n, const char* arch, const char* dump_path, std::string* text);  // plan + layout as text (no GPU needed)
int cutout_norm_scratch_doubles();
const double* launch_cutout_norm(const float2* x, int64_t n, double* parts, hipStream_t st);  // -> device address of ||x||
// e^{+j 2 pi q / 16384}, q < 16384: the twiddle table of the in-LDS transforms (caf_ldsfft.h), built once per device
int lds_fft_twiddles(int device, const float2** out);
// forward spectra of nblk overlap-save blocks of 16384 points (gather + in-LDS transform in one launch)
// inv_e != NULL: the same launch also writes 1 / (window energy) of the blocks' delays (num_shifts of them in all)
int launch_block_spectra(const float2* rx, int64_t rx_len, int64_t src0, int32_t step, int64_t nblk, float2* xb,
                         hipStream_t st, float* inv_e = nullptr, int64_t num_shifts = 0, const int32_t* gstart = nullptr,
                         const int32_t* glen = nullptr, int32_t ngroups = 0);
// the same for the 32768-point blocks of the chained role, written parity-major and butterfly-ordered ([block][2][16384])
int launch_block_spectra32(const float2* rx, int64_t rx_len, int64_t src0, int32_t step, int64_t nblk, float2* xb2, hipStream_t st);
// ... 65536-point blocks in the pair layout of the folded role (caf_perdelay.hip: k_block_spectra64)
int launch_block_spectra64(const float2* rx, int64_t rx_len, int64_t src0, int32_t step, int64_t nblk, float2* xb2, hipStream_t st);

// what caf_zoom_czt needs from a plan (caf_plan.hip)
struct PlanZoomView {
    int T, N, F, G, device;
    const float2* d_uconj;  // [T][N] conj(u): product row = rx[d + n] * d_uconj[n]
    const double* d_nu;     // [F] coarse frequency per hypothesis index, cycles per sample
    const float* d_tscale;  // [T] 1 / ||template||^2
};
int plan_zoom_view(caf_plan p, PlanZoomView* v);
// caf_zoom.hip
void launch_zoom_topk(const float* trace, const int32_t* cand, const int32_t* cand_count, int32_t max_cand, int32_t k,
                      float* vals_scratch, int32_t* sel, int32_t* sel_count, hipStream_t st);
void launch_zoom_rows(const float2* rx, const float2* uconj, int32_t n, const float* tscale_t, const int32_t* row_arg,
                      const double* nu, const float2* aa, const int32_t* sel, const int32_t* sel_count, int32_t k,
                      int64_t shift_start, int32_t nfft, float2* rows, hipStream_t st);
void launch_zoom_finish(const float* trace, const int32_t* row_arg, const double* nu, const int32_t* sel,
                        const int32_t* sel_count, int32_t k, int64_t shift_start, double span, double step,
                        const uint32_t* fine_arg, const float* fine_max, int32_t cand_overflow_cap, const int32_t* cand_count,
                        int32_t* o_count, int32_t* o_delay, int32_t* o_cidx, float* o_cqf2, int32_t* o_fidx, double* o_ffreq,
                        float* o_fqf2, hipStream_t st);

// caf_firos.hip: overlap-save FIR (fused in-LDS form for <= 8192 taps; gather / scatter kernels for the rocFFT rows)
int fir_os_fused_block(int32_t ntaps);
// rows > 1: independent signals x + r x_row_stride -> out + r out_row_stride in one launch (no carried-in history)
int launch_fir_os_fused(const float2* x, int64_t n, const float* taps, int32_t ntaps, const float2* delay, int32_t dlen,
                        int32_t dsr, int32_t phase, float2* out, int64_t nout, float2* ht, hipStream_t st, int64_t rows = 1,
                        int64_t x_row_stride = 0, int64_t out_row_stride = 0);
int launch_iq16_fir_os_fused(const int16_t* iq, int64_t n, float scale, const float* taps, int32_t ntaps, const int16_t* delay,
                             int32_t dlen, int32_t dsr, int32_t phase, float2* out, int64_t nout, float2* ht, hipStream_t st);
void launch_fos_gather(const float2* x, int64_t n, const float2* delay, int32_t dlen, int64_t b0, int64_t nb, int64_t L,
                       int64_t B, int32_t ntaps, float2* rows, hipStream_t st);
void launch_fos_gather_iq16(const int16_t* x, int64_t n, float scale, const int16_t* delay, int32_t dlen, int64_t b0,
                            int64_t nb, int64_t L, int64_t B, int32_t ntaps, float2* rows, hipStream_t st);
void launch_fos_taps_pad(const float* taps, int32_t ntaps, int64_t B, float2* row, hipStream_t st);
void launch_fos_scatter(const float2* rows, int64_t b0, int64_t nb, int64_t L, int64_t B, int32_t ntaps, int32_t dsr,
                        int32_t phase, float2* out, int64_t nout, hipStream_t st);

// caf_fused.hip
void launch_parity_major(const float2* in, float2* out, int64_t rows, int32_t half, hipStream_t st, bool butterfly = false);
// rows of 4 * quarter samples -> [c][quarter] pairs (sample 2 m + c, sample 2 (m + quarter) + c), m in butterfly order (fused_item2q<FOLD>)
void launch_parity_pairs(const float2* in, float2* out, int64_t rows, int32_t quarter, hipStream_t st);
void launch_fused_caf(const float2* xb, const float2* hc, const int32_t* shifts, const float2* tw1,
                      const float2* tw23, int32_t table_mode, int32_t nfreq, int32_t nhyp, int32_t hyp_per_wg,
                      int32_t nblk, int32_t tiles_per_blk, float* vt, hipStream_t st);
void launch_transpose_norm_argmax(const float* vt, int32_t ntmpl, int32_t nfreq, const float* tscale,
                                  const float* inv_e, int64_t num_shifts, int64_t shift_start, int32_t step,
                                  int32_t blk0, int32_t nblk, int32_t tiles_per_blk, float* surface, float* row_max,
                                  int32_t* row_arg, PeakRec* partial, int64_t partial_per_tmpl, hipStream_t st);
void launch_colmax_abs(const float2* z, int32_t rows, int64_t n, float* maxv, void* arg, int32_t arg64, hipStream_t st);
void launch_colmax_sqrt(const float* q, int32_t rows, int64_t n, float* maxv, int64_t* arg, hipStream_t st);
void launch_dot_tones(double f0, double fstep, int32_t num_freqs, int64_t len, const float2* src, float2* out,
                      hipStream_t st);
void launch_mul_conj(const float2* a, const float2* b, int64_t n, float2* out, hipStream_t st);
void launch_steer_dot(const float2* vec, const double2* steer, int64_t rows, int64_t n, double scale, double2* out,
                      hipStream_t st);
void launch_sum_planes_qf2(const float2* planes, int64_t plane_elems, int32_t cols, const int32_t* h_idx, int32_t nsel,
                           const double* row_norm, double ynormsq, double* out, hipStream_t st);
void launch_sum_groups_qf2(const float2* planes, int32_t ngroups, int64_t plane_elems, int32_t cols, const float2* phase,
                           const double* row_norm, double ynormsq, double* out, hipStream_t st);

// Arguments of the work-queue kernel k_caf_persistent, kept in device memory: each role reads the fields it
// needs at the start of a work item (scalar loads), so the other role's arguments do not occupy SGPRs.
struct PersistParams {
    // FFT items (same meaning as launch_fused_caf)
    const float2* xb;
    const float2* hc;
    const int32_t* shifts;
    const float2* tw1;
    const float2* tw23;
    float* vt;
    int32_t table_mode, nfreq, nhyp, hyp_per_wg, nblk, tiles_per_blk;
    int32_t block_log2;  // 14: one 16384-point transform per hypothesis; 15: 32768 points as two chained halves (fused_item2q);
                         // 16: 65536 points in the folded form: two chained halves per output residue r (fused_item2q<FOLD>:
                         // ngroups is then 2 x the hypothesis groups, group = 2 * hypothesis group + r)
    int32_t dstride;     // delay stride of a tile: 1; 2 with block_log2 == 16 (tile 256 r + u = the delays 2 (64 u + j) + r)
    int32_t npart;       // block_log2 == 16: partitions of 32768 samples per template (templates of 32769 .. npart * 32768 samples:
                         // hc holds npart consecutive rows per spectrum, xb the npart - 1 block spectra beyond the launch's last block)
    // tile items (same meaning as launch_transpose_norm_argmax)
    int32_t ntmpl, step, blk0;
    int32_t gpt;  // > 0: hypothesis groups are formed per template, gpt per template (group g of template t covers
                  // hypotheses t*nfreq + g*hyp_per_wg ...); 0: flat groups of hyp_per_wg over all T*F hypotheses
    const float* tscale;
    const float* inv_e;
    int64_t num_shifts, shift_start;
    float* surface;
    float* row_max;
    int32_t* row_arg;
    PeakRec* partial;
    int64_t partial_per_tmpl;
    // queues: pq[0] next FFT item, pq[1] next tile item, pq[4 + b] finished hypothesis groups of block b
    int32_t* pq;
    int32_t tr_slots, ngroups, n_fft, ipb, n_tr;
    int32_t nosurf;  // 1: no |y|^2 tiles; per item one (maximum, hypothesis) pair per delay in vmax / imax
    int32_t f1_direct;  // 1 (nfreq == 1): the FFT items write the finished per-delay values to row_max / surface; no tile items
    float* cqf;         // != nullptr: the FFT items write complex QF rows [T*F][num_shifts] (2 floats per value); no tile items
    float* surface_t;   // != nullptr: the FFT items write the hypothesis-major QF^2 surface [T*F][num_shifts] themselves and
                        // keep running maxima of the written values (nosurf == 2: the pairs hold normalised values)
    float* vmax;     // [block][group][tile][64]
    int32_t* imax;
    int32_t* dbg;  // optional host-mapped progress marks (CAF_PERSIST_DEBUG), 4 ints per workgroup
};
// natural order -> the order in which the fused engines read their template-spectrum rows (caf_fused.hip, fp_tid_of)
void launch_butterfly_order(const float2* in, float2* out, int64_t nchunks, hipStream_t st);
// copies *h to d_params, clears the queue block and launches n_wgs resident workgroups
void launch_caf_persistent(const PersistParams* h, PersistParams* d_params, int32_t n_wgs, hipStream_t st);

// rocFFT wrapper shared by the plan and the ops (caf_fft.hip)
struct FftPlan {
    void* plan = nullptr;  // rocfft_plan
    void* info = nullptr;  // rocfft_execution_info
    void* work = nullptr;
    size_t work_bytes = 0;
    int create(bool inverse, size_t len, size_t batch, size_t dist, bool inplace = true);
    int exec(void* in, void* out, hipStream_t st);
    void destroy();
    // identity for the checkout cache (set by fft_plan_acquire; key_len == 0: not cacheable)
    int key_dev = 0;
    bool key_inverse = false, key_inplace = true;
    size_t key_len = 0, key_batch = 0, key_dist = 0;
};
// checkout cache: acquire hands out a parked plan of the same shape or creates one; release parks it again
int fft_plan_acquire(FftPlan* out, bool inverse, size_t len, size_t batch, size_t dist, bool inplace = true);
void fft_plan_release(FftPlan* p);

#define CAF_REQUIRE(cond, msg)      \
    do {                            \
        if (!(cond)) {              \
            caf::set_error(msg);    \
            return CAF_ERR_INVALID; \
        }                           \
    } while (0)

}  // namespace caf
