/*
 * libcaf -- C-ABI of the MI355X (gfx950) cross-ambiguity-function / matched-filter engine.
 *
 * This header is the drop-in boundary.  It follows the reference's own C-DLL idiom
 * (icyveins7/pydsproutines cpuWola.py:38-70 + cpuWolaDll.c:107, cpuTone.py:28-47 +
 * cpuToneDll.c:33, cython_ext/compareIntPreambles/compareIntPreambles.py:7-52):
 *   - extern "C", plain pointers and explicit sizes, no C++/torch types in signatures;
 *   - every function returns int32 status, 0 == success (cpuWolaDll.c:175, cpuToneDll.c:56);
 *   - the CALLER allocates every output (cpuWola.py:60-65); the callee never returns heap memory;
 *   - complex64 is interleaved {float re, float im} (Ipp32fc, cpuWolaDll.c:23);
 *   - stateful objects are create -> use many -> destroy (CyIppXcorrFFT.pyx:25-38).
 * Argument validation with the reference's exception types (TypeError / ValueError /
 * MemoryError) is done by the Python host before the call (cupyHelpers.py:50-82); the
 * library still range-checks everything that could fault the GPU and reports
 * CAF_ERR_INVALID instead of launching.
 *
 * Pointer conventions: names starting with h_ are HOST pointers, d_ are DEVICE (HBM)
 * pointers on the current HIP device.  `stream` is a hipStream_t passed as void*
 * (NULL = the default stream).  Calls with d_ pointers are asynchronous on `stream`
 * unless stated otherwise; *_host entry points are blocking.
 */
#ifndef CAF_H_
#define CAF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAF_EXPORT __attribute__((visibility("default")))

/* ---- status codes ------------------------------------------------------------------ */
#define CAF_OK 0
#define CAF_ERR_INVALID 1   /* bad argument / shape / range (Python raises ValueError)      */
#define CAF_ERR_HIP 2       /* a HIP runtime call failed (RuntimeError)                      */
#define CAF_ERR_ROCFFT 3    /* a rocFFT call failed (RuntimeError)                           */
#define CAF_ERR_NOMEM 4     /* device or host allocation failed (MemoryError)                */
#define CAF_ERR_NODEVICE 5  /* no usable gfx950 device (RuntimeError)                        */

/* Copies the calling thread's last error message (NUL-terminated) into buf. */
CAF_EXPORT int32_t caf_last_error(char* buf, int32_t len);
/* ABI version of this header: (major << 16) | minor. */
CAF_EXPORT int32_t caf_abi_version(void);

/* ---- device + memory plumbing (replaces cp.asarray / cp.asnumpy / cp.empty at the
 *      cupy-signature entry points, xcorrRoutines.py:81-82,155-156) -------------------- */
CAF_EXPORT int32_t caf_device_count(int32_t* count);
CAF_EXPORT int32_t caf_set_device(int32_t device);
CAF_EXPORT int32_t caf_device_info(int32_t device, char* name, int32_t name_len, int64_t* total_mem,
                                   int32_t* compute_units);
/* caf_malloc / caf_free go through a caching allocator (the counterpart of cupy's default memory pool, which
 * every cp.empty / cp.zeros of the reference's wrappers hits): freed blocks are kept and reused in stream order;
 * CAF_POOL_MB bounds the cached bytes (default 8192, 0 = plain hipMalloc / hipFree).  caf_pool_trim returns the
 * cache to the driver (cp.get_default_memory_pool().free_all_blocks()). */
CAF_EXPORT int32_t caf_malloc(void** d_ptr, int64_t bytes);
CAF_EXPORT int32_t caf_free(void* d_ptr);
CAF_EXPORT int32_t caf_pool_trim(void);
CAF_EXPORT int32_t caf_pool_stats(int64_t* cached_bytes, int64_t* in_use_bytes, int64_t* hits, int64_t* misses);
CAF_EXPORT int32_t caf_memset(void* d_ptr, int32_t value, int64_t bytes, void* stream);
CAF_EXPORT int32_t caf_h2d(void* d_dst, const void* h_src, int64_t bytes, void* stream);
CAF_EXPORT int32_t caf_d2h(void* h_dst, const void* d_src, int64_t bytes, void* stream);
CAF_EXPORT int32_t caf_d2d(void* d_dst, const void* d_src, int64_t bytes, void* stream);
/* caf_h2d / caf_d2h / caf_d2h_transposed move pageable host memory through the library's own pinned staging buffers
 * (several DMA lanes + host threads for large arrays) and return when the data has arrived: the caller's pages are
 * never registered with the driver, so what the caller does with its arrays afterwards (NumPy returning them to the
 * operating system) cannot stall the process's GPU queues (csrc/caf_host.cpp).
 * caf_d2h_transposed: d_src is [rows][pitch] float32; columns col0 .. col0 + ncols - 1 arrive as the C-ordered host array
 * [ncols][rows] of float32 (dst_f64 = 0) or float64 (1) -- a hypothesis-major surface (caf_outputs2.d_surface_t, rows = F,
 * pitch = S) leaves the device as the (delays, frequencies) array the reference returns (xcorrRoutines.py:553-566,
 * 1028-1039).  rows <= 65536.  (ABI 1.8) */
CAF_EXPORT int32_t caf_d2h_transposed(void* h_dst, int32_t dst_f64, const float* d_src, int64_t rows, int64_t pitch,
                                      int64_t col0, int64_t ncols, void* stream);
/* caf_d2h_f64: count float32 values on the device arrive as float64 on the host (widened by the staging lanes' host threads:
 * the CPU signatures of the reference return float64 surfaces, xcorrRoutines.py:553-566, and NumPy's astype of a 17 GB download
 * is a second pass over it).  (ABI 1.9) */
CAF_EXPORT int32_t caf_d2h_f64(double* h_dst, const float* d_src, int64_t count, void* stream);
CAF_EXPORT int32_t caf_stream_sync(void* stream);
/* a non-blocking HIP stream of the current device for the `stream` arguments below (cupy.cuda.Stream(non_blocking=True)
 * of a cupy caller); NULL = the default stream everywhere */
CAF_EXPORT int32_t caf_stream_create(void** stream);
CAF_EXPORT int32_t caf_stream_destroy(void* stream);

/* ---- the hypothesis engine (frequency-domain overlap-save CAF) -----------------------
 *
 * Computes, for every template t < T, delay s in [shift_start, shift_start+num_shifts) and
 * frequency hypothesis f < F,
 *
 *   QF2[t][s][f] = | sum_n rx[s+n] * conj(tmpl_t[n]) * exp(-j 2 pi nu_f n) |^2
 *                  / ( ||tmpl_t||^2 * sum_{n in supp} |rx[s+n]|^2 )
 *
 * which is the quantity every reference flavour evaluates per delay:
 *   fastXcorr branches A/B/C          xcorrRoutines.py:483-566  (nu_f = k/N, k = FFT bin)
 *   GroupXcorr.xcorr                  xcorrRoutines.py:917-954  (nu_f = freqs[f]/fs, groups)
 *   GroupXcorrFFT.xcorr               xcorrRoutines.py:1137-1189 (nu_f = makeFreq(fftlen)/fs)
 *   cztXcorr / GroupXcorrCZT          xcorrRoutines.py:413-457, 996-1039 (uniform grid)
 *   TemplateCrossCorrelator.correlate xcorrRoutines.py:312-371  (F = 1, T templates)
 *   IppXcorrFFT_32fc::xcorr_thread    cython_ext/CyIppXcorrFFT/IppXcorrFFT.cpp:94-178
 * Instead of one length-N DFT per delay it evaluates one overlap-save correlation per
 * hypothesis: IFFT_B( FFT_B(rx block) * conj(FFT_B(tmpl_t * exp(+j 2 pi nu_f n))) ).
 */
typedef struct caf_plan_t* caf_plan;

/* Frequency-hypothesis specification. */
#define CAF_FREQ_BINS 0  /* nu_f = bins[f] / grid, integer bins (may be negative), grid | block   */
#define CAF_FREQ_NORM 1  /* nu_f = freqs_norm[f] (cycles per sample), arbitrary                  */

typedef struct caf_plan_desc {
    int32_t num_templates;     /* T >= 1                                                          */
    int32_t template_len;      /* N: span of every template in samples (gaps between groups = 0)  */
    const float* h_templates;  /* host, interleaved complex64 [T][N], NOT conjugated              */
    int32_t auto_conj;         /* 1: correlate against conj(template) (reference autoConj=True);  */
                               /* 0: the caller already conjugated                                */
    int32_t num_groups;        /* G >= 1: support = union of [group_start[g], +group_len[g])      */
    const int32_t* h_group_start; /* host [G], relative to the template start; NULL => {0}        */
    const int32_t* h_group_len;   /* host [G]; NULL => {N}                                        */
    int32_t freq_mode;         /* CAF_FREQ_BINS or CAF_FREQ_NORM                                  */
    int32_t num_freqs;         /* F >= 1                                                          */
    const int32_t* h_bins;     /* host [F]  (CAF_FREQ_BINS)                                       */
    int32_t grid;              /* DFT grid size the bins refer to (CAF_FREQ_BINS), e.g. N         */
    const double* h_freqs_norm;/* host [F]  (CAF_FREQ_NORM)                                       */
    int64_t max_rx_len;        /* largest rx length (samples) execute() will be given             */
    int32_t log2_block;        /* overlap-save FFT size B = 2^log2_block; 0 => library default    */
    int32_t blocks_per_batch;  /* rx blocks processed per launch group; 0 => library default      */
    int32_t engine;            /* CAF_ENGINE_AUTO / _ROCFFT / _FUSED / _PERSISTENT / _DIRECT      */
    int32_t reserved;          /* must be 0                                                       */
} caf_plan_desc;

/* Inverse-transform engine of the hypothesis plan.
 *   ROCFFT: spectral multiply kernel -> batched rocFFT inverse -> |.|^2 kernel (any block size);
 *   FUSED : one hand-written kernel does multiply + 16384-point inverse FFT in LDS + |.|^2, a second
 *           one transposes/normalises/argmaxes; needs template_len <= 8192, (bins mode) every
 *           bins[f] * 16384 / grid a whole number (any grid that divides 16384; bin 0 on any grid),
 *           and cannot produce d_cqf.
 *   PERSISTENT: the two FUSED stages as ONE work-queue launch (one resident workgroup per CU): the
 *           HBM-bound transpose runs on some CUs while the others compute FFTs.  Same results as
 *           FUSED up to 8192 samples; templates up to 16384 samples on 32768-point blocks, up to
 *           32768 on 65536-point blocks, up to 262144 on 65536-point blocks with the template cut into
 *           partitions of 32768 samples (bins mode: bins[f] * block / grid whole and even).  d_cqf:
 *           supported as the ONLY output of a call (the FFT work items write the complex rows
 *           themselves, on every block size of this engine).
 *   AUTO  : PERSISTENT when its conditions hold and log2_block is 0 or the engine's own, else ROCFFT. */
#define CAF_ENGINE_AUTO 0
#define CAF_ENGINE_ROCFFT 1
#define CAF_ENGINE_FUSED 2
#define CAF_ENGINE_PERSISTENT 3
/*   DIRECT: the definition itself, product by product in the time domain, for templates with at most 64 non-zero
 *           samples (composite templates whose groups cover only a few samples of a long span): no overlap-save
 *           blocks, so the error does not follow the block's energy; float64 window energies over the samples used.
 *           AUTO picks it for composite templates (num_groups >= 1) whose groups cover fewer than 64 samples.
 *           No d_cqf. */
#define CAF_ENGINE_DIRECT 4

CAF_EXPORT int32_t caf_plan_create(caf_plan* plan, const caf_plan_desc* desc);
CAF_EXPORT int32_t caf_plan_destroy(caf_plan plan);

/* Geometry chosen by the plan: block size B, valid outputs per block, batch, workspace bytes. */
CAF_EXPORT int32_t caf_plan_info(caf_plan plan, int32_t* block, int32_t* step, int32_t* blocks_per_batch,
                                 int64_t* workspace_bytes);

/* Engine actually selected by the plan: CAF_ENGINE_ROCFFT, _FUSED, _PERSISTENT or _DIRECT. */
CAF_EXPORT int32_t caf_plan_engine(caf_plan plan, int32_t* engine);

/* Outputs of one execute (any pointer may be NULL = not wanted).
 * A delay whose rx window holds no energy (a stretch of exact zeros: the reference's 0 / 0) is reported as NaN on the
 * surface and in d_row_max, with d_row_arg = 0, and never becomes a peak. */
typedef struct caf_outputs {
    float* d_surface;      /* [T][num_shifts][F] float32 QF2 (reference CAF layout: delay-major)  */
    float* d_row_max;      /* [T][num_shifts] float32: max over f of QF2                           */
    int32_t* d_row_arg;    /* [T][num_shifts] int32: argmax over f (lowest f on ties)              */
    float* d_peak_val;     /* [T] float32: global maximum of QF2 for template t                    */
    int32_t* d_peak_delay; /* [T] int32: its delay (absolute sample index; lowest on ties)         */
    int32_t* d_peak_freq;  /* [T] int32: its frequency-hypothesis index f                          */
    float* d_cqf;          /* [T][F][num_shifts] complex64 QF = r / (||tmpl|| * ||rx window||), i.e.   */
                           /* hypothesis-major like TemplateCrossCorrelator.correlate (:352-357) and   */
                           /* fastXcorr(absResult=False) (:533-548)                                    */
} caf_outputs;

/* d_rx: device complex64 [rx_len].  Requires shift_start >= 0 and
 * shift_start + num_shifts - 1 + N <= rx_len <= max_rx_len.  Asynchronous on `stream`.
 * The plan belongs to the device that was current at caf_plan_create (the same must be current here) and owns
 * its workspace: executions of ONE plan must be ordered (same stream, or synchronised by the caller); different
 * plans may run concurrently from different threads / streams (create -> use many -> destroy, like the
 * reference's stateful correlator objects, IppXcorrFFT.h, xcorrRoutines.py:277-371). */
CAF_EXPORT int32_t caf_plan_execute(caf_plan plan, const float* d_rx, int64_t rx_len, int64_t shift_start,
                                    int64_t num_shifts, const caf_outputs* out, void* stream);

/* The same call with the outputs added after ABI 1.4 (caf_outputs keeps its seven pointers: clients built against it --
 * INTEGRATION.md's ctypes stub, examples/c_client -- stay valid).  `base` as above (may be NULL); zero the struct first. */
typedef struct caf_outputs2 {
    caf_outputs base;
    float* d_surface_t;    /* [T][F][num_shifts] float32 QF2, HYPOTHESIS-major: the same numbers as d_surface, bit for  */
                           /* bit, transposed per template (not the reference's CAF layout, xcorrRoutines.py:553-566;  */
                           /* it is what a frequency-row consumer -- a zoom, a results store, a per-bin detector --     */
                           /* reads contiguously).  Written by the FFT work items of the PERSISTENT engine with        */
                           /* 16384-point blocks themselves: no |y|^2 tiles, no transposition (C2: 10.9 ms against     */
                           /* 13.6).  Not together with d_surface or d_cqf; F == 1: any engine (the layouts coincide). */
    void* reserved[3];     /* must be NULL */
} caf_outputs2;
CAF_EXPORT int32_t caf_plan_execute2(caf_plan plan, const float* d_rx, int64_t rx_len, int64_t shift_start,
                                     int64_t num_shifts, const caf_outputs2* out, void* stream);

/* Diagnostic of the one-launch (persistent) engine: marks[0..1] = the watchdog words of the work-queue block. Both
 * are 0 unless a tile item waited ~10 s for its block to be published -- which the protocol rules out (the launch
 * then traps) -- so a non-zero value after a successful run means a broken hand-off.  Synchronises the device. */
CAF_EXPORT int32_t caf_plan_watchdog(caf_plan plan, int32_t* marks);
/* Per-kernel device timing with HIP events recorded on the execute stream (used by bench.py
 * for the roofline figure).  enable=1 starts collecting; get() synchronises the recorded events
 * and returns accumulated milliseconds and launch counts since enable, per stage:
 *   0 energy prefix+inverse, 1 rx block gather, 2 forward FFT (rocFFT), 3 spectral conj-multiply,
 *   4 inverse FFT (rocFFT), 5 |.|^2 + normalise + argmax, 6 peak reduce.                     */
#define CAF_NUM_STAGES 7
CAF_EXPORT int32_t caf_plan_profile(caf_plan plan, int32_t enable);
CAF_EXPORT int32_t caf_plan_profile_get(caf_plan plan, double* ms /*[CAF_NUM_STAGES]*/,
                                        int64_t* launches /*[CAF_NUM_STAGES]*/);

/* Blocking host-pointer convenience in the reference DLL style (caller-allocated NumPy
 * outputs, host rx): H2D, execute, D2H.  h_* outputs follow caf_outputs shapes. */
CAF_EXPORT int32_t caf_plan_execute_host(caf_plan plan, const float* h_rx, int64_t rx_len, int64_t shift_start,
                                         int64_t num_shifts, float* h_surface, float* h_row_max,
                                         int32_t* h_row_arg, float* h_peak_val, int32_t* h_peak_delay,
                                         int32_t* h_peak_freq);

/* ---- the per-delay path (one DFT per delay; what the reference literally does) --------------
 *
 * For delays s_i = start + i*step, i < num:   z_i = FFT_N( rx[s_i : s_i+N] * cutout ) / (||cutout|| ||rx window||)
 * replaces fastXcorr branches B/B'/C/C' (xcorrRoutines.py:511-580), cp_fastXcorr (:29-167) and
 * IppXcorrFFT_32fc::xcorr_thread (IppXcorrFFT.cpp:94-178).  `d_cutout` is used AS GIVEN (the caller
 * passes conj(cutout) for the usual correlation, like ippsMul_32fc(m_cutout(conj'd), ...) :133-138).
 * Outputs (any may be NULL): d_qf2[num] = max_k |z_i[k]|^2, d_fidx[num] = first argmax bin,
 * d_caf[num][N] = |z_i|^2 (float32), d_ccaf[num][N] = z_i (complex64).
 * zero_oor != 0: delays whose window leaves [0, rx_len) give (0, 0) / zero rows (IppXcorrFFT.cpp:125-130);
 * zero_oor == 0: such a delay is an error (CAF_ERR_INVALID). Blocking (allocates and frees scratch).
 * Zero-energy windows (a stretch of exact zeros in rx under the whole cutout -- the reference's 0 / 0,
 * `pmax / cutoutNormSq / rxNormPartSq`, xcorrRoutines.py:527-528; IppXcorrFFT.cpp:174): d_qf2 = NaN, d_fidx = 0, NaN rows in
 * d_caf / d_ccaf -- the same rule as caf_outputs.d_row_max / d_row_arg above.  (Only the stand-alone
 * caf_argmax_abs_rows keeps the (0, 0) of the CUDA kernel's zero-initialised workspace, argmax.cu:108-109.) */
CAF_EXPORT int32_t caf_xcorr_perdelay(const float* d_cutout, int32_t n, const float* d_rx, int64_t rx_len,
                                      int64_t start, int64_t step, int64_t num, int32_t zero_oor, float* d_qf2,
                                      int32_t* d_fidx, float* d_caf, float* d_ccaf, int64_t batch_rows, void* stream);
/* Cutouts of 32 .. 16384 samples whose prime factors are at most 23 get a kernel compiled for the length at run time (hiprtc, as the reference
 * compiles its own through NVRTC; cached in memory and under CAF_JIT_CACHE / ~/.cache/pydsproutines_amd/jit; CAF_JIT=0 keeps the
 * prebuilt kernels).  This call describes what would run for n -- radices, threads per row, LDS strides, the bank-conflict
 * cycles its layout was chosen by -- as one line of text in buf ("" when the length has no such kernel), without a GPU.
 * arch != NULL (e.g. "gfx950"): the kernel is also compiled (not loaded); dump_path != NULL: its code object is written there.
 * (ABI 1.8) */
CAF_EXPORT int32_t caf_perdelay_jit_describe(int32_t n, const char* arch, const char* dump_path, char* buf, int32_t len);
/* 1 when caf_xcorr_perdelay serves cutouts of n samples with ONE kernel (product, N-point transform in LDS, |.|^2, argmax:
 * powers of two 64 ... 16384, 100 / 1000 / 10000, and the 2^a 3^b 5^c 7^d lengths 32 ... 16200 the mixed-radix planner
 * finds a plan for), 0 when it runs the product rows -> row FFT -> argmax chain the reference runs for every length
 * (cp_fastXcorr_v2, xcorrRoutines.py:169-274).  The results are the same either way; callers that batch the chain
 * themselves (the BATCH argument of cp_fastXcorr_v2) ask this first.  No device work. */
CAF_EXPORT int32_t caf_xcorr_perdelay_one_kernel(int32_t n);

/* ---- stand-alone kernels (device pointers), one per reference CUDA kernel wrapper ------------- */
/* batched row FFT, replaces cp.fft.fft(x, axis=1) / ifft (xcorrRoutines.py:135,246,339,352); in place when
 * d_out == d_in; inverse != 0 applies the 1/len normalisation of numpy/cupy ifft. */
CAF_EXPORT int32_t caf_fft_rows(const float* d_in, float* d_out, int64_t rows, int64_t len, int32_t inverse,
                                void* stream);
/* slidingMultiplyNormalised (multiplySlices.cu:113-216, cupyExtensions.py:491-560): d_z[idxlen][xlen] */
CAF_EXPORT int32_t caf_sliding_multiply_normalised(const float* d_x, int32_t xlen, const float* d_y, int64_t ylen,
                                                   int64_t start_idx, int64_t idxlen, double coefficient,
                                                   float* d_z, void* stream);
/* multiTemplateSlidingDotProduct (multiplySlices.cu:251-399, cupyExtensions.py:563-640) */
CAF_EXPORT int32_t caf_multi_template_sliding_dot(const float* d_templates, const float* d_energies, int32_t num_templates,
                                                  int32_t template_len, const float* d_x, int64_t xlen,
                                                  int64_t start_idx, int64_t idxlen, int32_t* d_template_idx,
                                                  float* d_qf2, void* stream);
/* multiplySlicesWithIndexedRowsOptimistic (multiplySlices.cu:25-84, cupyExtensions.py:405-488) */
/* d_out[num_slices][out_len]; slice i covers d_slice_lens[i] samples (NULL = out_len), zero beyond */
CAF_EXPORT int32_t caf_multiply_slices_indexed_rows(const float* d_x, int64_t xlen, const float* d_rows, int32_t num_rows,
                                                    int32_t row_len, const int32_t* d_slice_starts,
                                                    const int32_t* d_slice_lens, const int32_t* d_row_idx,
                                                    int32_t out_len, int64_t num_slices, float* d_out, void* stream);
/* complex_magnSq_kernel<T,U> (complex_magn.cu:8-19): in complex64 (in_c128=0) or complex128, out f32 or f64 */
CAF_EXPORT int32_t caf_complex_magnsq(const void* d_x, int64_t n, int32_t in_c128, void* d_out, int32_t out_f64,
                                      void* stream);
/* multiArgmaxAbsRows_complex64 (argmax.cu:93-153): first index of the maximum (NumPy tie-break) */
CAF_EXPORT int32_t caf_argmax_abs_rows(const float* d_x, int64_t rows, int64_t len, uint32_t* d_argmax, float* d_max,
                                       int32_t use_normsq, void* stream);
/* movingAverage (filter.cu:291-347), multiMovingAverage (:196-240): causal, zero history, per row */
CAF_EXPORT int32_t caf_moving_average(const float* d_x, int64_t rows, int64_t n, int32_t avg_length,
                                      int32_t sum_instead, float* d_out, void* stream);
/* movingComplexSum (filter.cu:374-438): d_out[n - L + 1] = |sum of L consecutive samples|^2 */
CAF_EXPORT int32_t caf_complex_moving_sum(const float* d_x, int64_t n, int32_t sum_length, float* d_out, void* stream);
/* copy*SlicesToMatrix_32fc (copying.cu:8-138): row i = x[starts[i] : +len] (starts_stride 1), the
 * [start,end) rows of an (N,2) bounds array (starts_stride 2, zero beyond end), or x[start0 + i*inc : +len]
 * when d_starts is NULL; samples outside x read as 0 */
CAF_EXPORT int32_t caf_copy_slices_to_matrix(const float* d_x, int64_t xlen, const int32_t* d_starts,
                                             int32_t starts_stride, int64_t start0, int64_t increment, int32_t len,
                                             int64_t rows, float* d_out, void* stream);
/* copy_groups_kernel32fc (cupyExtensions.py:17-38) */
CAF_EXPORT int32_t caf_copy_groups(const float* d_x, float* d_y, const int32_t* d_x_starts, const int32_t* d_y_starts,
                                   const int32_t* d_lengths, int32_t num_groups, void* stream);
/* findLocalMaxima (peakfinding.cu:14-58): indices in ASCENDING order (deterministic), *d_count = total found */
CAF_EXPORT int32_t caf_find_local_maxima(const float* d_x, int64_t n, float min_height, int32_t max_peaks,
                                         int32_t* d_peak_index, int32_t* d_count, void* stream);
/* d_out[i] = d_x[d_index[i]] for 4-byte elements (float32 / int32; 0 for indices outside [0, xlen)): the values and
 * frequency arguments at the candidate peaks without copying whole per-delay traces to the host (the fancy
 * indexing d_trace[d_idx] of a cupy caller after cupyFindLocalMaxima, cupyExtensions.py:651-686) */
CAF_EXPORT int32_t caf_gather_b32(const void* d_x, int64_t xlen, const int32_t* d_index, int64_t n, void* d_out,
                                  void* stream);
/* ---- multi-GPU step --------------------------------------------------------------------------------
 * One process per GPU; templates are block-sharded over the ranks, rx is replicated, and the only exchange of the
 * path is the all-gather of the per-template peak rows over RCCL / xGMI.  Rank 0 obtains a 128-byte id and hands it
 * to the other ranks through any host channel (MPI, a file, torch.distributed's store); every rank then creates the
 * communicator with its GPU current.  d_local = this rank's rows as [3][rows_per_rank] int32 (delays, frequency
 * indices, float32 peak values as bits: the three d_peak_* arrays of caf_outputs laid end to end, shards padded to
 * the largest); d_table = [world][3][rows_per_rank], identical on every rank.  The reference has no multi-GPU code;
 * its nearest analogue is the thread-strided split of cython_ext/CyIppXcorrFFT/IppXcorrFFT.cpp:117. */
typedef struct caf_comm_t* caf_comm;
CAF_EXPORT int32_t caf_comm_unique_id(void* id128);
CAF_EXPORT int32_t caf_comm_create(caf_comm* comm, int32_t world_size, int32_t rank, const void* id128);
CAF_EXPORT int32_t caf_comm_destroy(caf_comm comm);
CAF_EXPORT int32_t caf_peak_table_allgather(caf_comm comm, const int32_t* d_local, int32_t rows_per_rank,
                                            int32_t* d_table, void* stream);

/* d_out[i] = (double) d_x[d_index ? d_index[i] : i] (0 outside [0, xlen)): float32 per-delay traces into the float64
 * device arrays the reference's GPU entry points return (GroupXcorrFFT.xcorrGPU xc = cp.zeros(shifts.size),
 * xcorrRoutines.py:1198-1203; cp_fastXcorr's d_result, :95-101) with no host round trip */
CAF_EXPORT int32_t caf_gather_f32_f64(const float* d_x, int64_t xlen, const int32_t* d_index, int64_t n, double* d_out,
                                      void* stream);
/* ---- config C5 in one call: coarse CAF -> top-k local maxima -> chirp-Z fine zoom ------------------------------
 * Replaces the reference's two-stage workflow of a coarse xcorr followed by a CZT over a narrow span at the delays of
 * interest (benchmarks/benchmark_czts.py:31-82 with pbIppCZT32fc.runMany; cztXcorr xcorrRoutines.py:413-457;
 * pybinds/ippGroupXcorrCZT/GroupXcorrCZT.cpp:202-329).  Input: the per-delay trace of ONE template of a finished
 * caf_plan_execute (d_row_max / d_row_arg offset to that template, num_shifts entries starting at shift_start).
 * On the device: local maxima above min_height (peakfinding.cu:52 predicate) -> the k strongest (value descending,
 * delay ascending) -> per peak the normalised product row rx[d:d+N] * conj(template) -> |CZT|^2 on the grid
 * nu0 - span ... nu0 + span in steps of `step` (all in cycles per sample; nu0 = the peak's coarse frequency).
 * Every output is a caller-allocated DEVICE array with k entries (NULL to skip); unused entries have delay -1.
 * d_count[0] = peaks returned (<= k), or -1 when more than 2^20 local maxima exceed min_height (raise it). */
typedef struct caf_zoom_outputs {
    int32_t* d_count;              /* [1] */
    int32_t* d_delay;              /* [k] absolute delay index */
    int32_t* d_coarse_freq_index;  /* [k] hypothesis index of the coarse maximum at that delay */
    float* d_coarse_qf2;           /* [k] */
    int32_t* d_fine_index;         /* [k] argmax on the fine grid (first index) */
    double* d_fine_freq;           /* [k] its frequency, cycles per sample: nu0 - span + index * step */
    float* d_fine_qf2;             /* [k] */
    float* d_planes;               /* [k][num_bins] fine |CZT|^2 (num_bins from caf_zoom_num_bins) */
} caf_zoom_outputs;
CAF_EXPORT int32_t caf_zoom_num_bins(double span, double step, int32_t* num_bins);
CAF_EXPORT int32_t caf_zoom_czt(caf_plan plan, int32_t template_index, const float* d_rx, int64_t rx_len,
                                const float* d_row_max, const int32_t* d_row_arg, int64_t shift_start, int64_t num_shifts,
                                int32_t k, float min_height, double span, double step, const caf_zoom_outputs* out,
                                void* stream);

/* filter_smtaps* (filter.cu:9-181) == scipy.signal.lfilter(taps, 1, x)[ds_phase::dsr] with carried-in history */
CAF_EXPORT int32_t caf_fir_lfilter(const float* d_x, int64_t n, const float* d_taps, int32_t num_taps,
                                   const float* d_delay, int32_t delay_len, int32_t dsr, int32_t ds_phase, float* d_out,
                                   int64_t out_len, void* stream);
/* upfirdn_naive / upfirdn_sm (upfirdn.cu:6-182) == scipy.signal.upfirdn(taps, x, up, down) per row */
CAF_EXPORT int32_t caf_upfirdn(const float* d_x, int64_t rows, int64_t n, const float* d_taps, int32_t num_taps,
                               int32_t up, int32_t down, float* d_out, float* d_out_abs, int64_t out_len, void* stream);
/* CZTCachedGPU.runMany (spectralRoutines.py:370-391) / IppCZT32fc::runMany (CZT.cpp): rows of length m ->
 * k CZT bins; aa[m], fv[nfft], ww_slice[k] are the cached complex64 constants (host computes them in f64). */
CAF_EXPORT int32_t caf_czt_run_many(const float* d_x, int64_t rows, int32_t m, int32_t k, int32_t nfft,
                                    const float* d_aa, const float* d_fv, const float* d_ww, float* d_out,
                                    void* stream);
/* multiArgmax3d_uint32 (argmax.cu:11-81, cupyExtensions.py:225-265): per item of a (items, d1, d2, d3) uint32
 * array, the three indices of its maximum (first flat index on ties) -> d_argmax[items][3], d_max[items] */
CAF_EXPORT int32_t caf_argmax3d_u32(const uint32_t* d_x, int64_t num_items, int32_t dim1, int32_t dim2, int32_t dim3,
                                    uint32_t* d_argmax, uint32_t* d_max, void* stream);
/* IQ ingest: interleaved int16 I/Q -> complex64 * scale on the device, i.e. the
 * np.fromfile(int16).astype(float32).view(complex64) of usrpRoutines.simpleBinRead (usrpRoutines.py:51-67)
 * done after the (half-size) H2D copy as in benchmarks/benchmark_cupyCopyAndConvert.py:17-25 */
CAF_EXPORT int32_t caf_iq16_to_c64(const int16_t* d_iq, int64_t num_samples, float scale, float* d_out, void* stream);
/* Front-end filter/decimate fused into the rx load: one kernel that is
 *   caf_iq16_to_c64 -> filter_smtaps(dsr, dsPhase)   (usrpRoutines.py:51-67 + filter.cu:9-58, filterRoutines.py:417-501)
 * i.e. d_out[o] = lfilter(taps, 1, scale * iq)[ds_phase + o*dsr] as complex64, reading the raw int16 pairs (4 B per
 * input sample) and computing only the kept outputs.  d_delay = the delay_len int16 IQ pairs that precede d_iq
 * (streaming state: the tail of the previous chunk), NULL/0 for zeros.  num_taps <= 2048, dsr <= 16. */
CAF_EXPORT int32_t caf_iq16_fir_decimate(const int16_t* d_iq, int64_t num_samples, float scale, const float* d_taps,
                                         int32_t num_taps, const int16_t* d_delay, int32_t delay_len, int32_t dsr,
                                         int32_t ds_phase, float* d_out, int64_t out_len, void* stream);
/* per column of a complex64 (rows, n) matrix: max_r |z| and the first row attaining it
 * (TemplateCrossCorrelator.correlate(returnMax=True), xcorrRoutines.py:361-371) */
CAF_EXPORT int32_t caf_colmax_abs(const float* d_z, int32_t rows, int64_t n, float* d_max, void* d_arg,
                                  int32_t arg_int64 /* d_arg is int64[n] (cp.argmax's dtype) instead of int32[n] */, void* stream);
/* the same reduction on per-template QF^2 traces (float32 (rows, n), e.g. caf_outputs.d_row_max of an F = 1 plan):
 * d_max[i] = max_r sqrt(q2[r][i]), d_arg[i] = its first row as int64 (the dtype cp.argmax returns) */
CAF_EXPORT int32_t caf_colmax_sqrt(const float* d_q2, int32_t rows, int64_t n, float* d_max, int64_t* d_arg, void* stream);
/* Tone-dot zoom, dotTonesScaling_32f (genTones.cu:165-283, cupyDotTonesScaling spectralRoutines.py:580-630):
 * d_out[b][k] (complex64, ceil(len/64) x num_freqs) = sum over the 64-sample block b of
 * d_src[i] * exp(j 2 pi (f0 + k fstep) i), f0 / fstep normalised (cycles per sample); summing over b gives the
 * CZT of d_src at the normalised frequencies -(f0 + k fstep) */
CAF_EXPORT int32_t caf_dot_tones(const float* d_src, int64_t len, double f0, double fstep, int32_t num_freqs,
                                 float* d_out, void* stream);
/* Sub-sample refinement after the peak (fineFreqTimeSearch / GenXcorr.xcorr, xcorrRoutines.py:583-719):
 * d_out[i] = d_a[i] * conj(d_b[i]) (complex64; `x_fft * y_fft.conj()` :648, `y_aligned.conj() * x_aligned` :622) */
CAF_EXPORT int32_t caf_mul_conj(const float* d_a, const float* d_b, int64_t n, float* d_out, void* stream);
/* d_out[r] (complex128) = scale * sum_k d_vec[k] (complex64) * conj(d_steer[r][k]) (complex128 [rows][n]):
 * `np.dot(rx_vec, steeringvec.conj().T) / norms` :661-665, `np.vdot(precomputed, fineshifts[j])` :630 */
CAF_EXPORT int32_t caf_steer_dot(const float* d_vec, const double* d_steer, int64_t rows, int64_t n, double scale,
                                 double* d_out, void* stream);
/* GroupXcorrCZT_Permutations.getCAF / getCAF_GPU (xcorrRoutines.py:1454-1484, 1549-1585): sum the selected
 * per-template complex64 planes d_planes[num_planes][rows][cols] (h_sel: host array of num_sel <= 64 plane
 * numbers), then d_out[rows][cols] (float64) = |sum|^2 / d_row_norm[row] (float64) / ynormsq */
CAF_EXPORT int32_t caf_sum_planes_qf2(const float* d_planes, int32_t num_planes, int64_t rows, int32_t cols,
                                      const int32_t* h_sel, int32_t num_sel, const double* d_row_norm, double ynormsq,
                                      double* d_out, void* stream);
/* The coherent sum over the groups of ONE composite template on the per-delay path (GroupXcorrCZT.xcorr,
 * xcorrRoutines.py:996-1039; GroupXcorrCZT.cpp:106-329): d_planes[num_groups][rows][cols] complex64 = the chirp-Z rows of
 * every group's products as if the group began at sample 0, d_phase[num_groups][cols] complex64 = e^{-j 2 pi f_col start_g / fs}
 * (NULL: all ones); d_out[rows][cols] (float64) = |sum_g phase planes|^2 / d_row_norm[row] / ynormsq.  Any number of groups. */
CAF_EXPORT int32_t caf_sum_groups_qf2(const float* d_planes, int32_t num_groups, int64_t rows, int32_t cols,
                                      const float* d_phase, const double* d_row_norm, double ynormsq, double* d_out,
                                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CAF_H_ */
