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
CAF_EXPORT int32_t caf_malloc(void** d_ptr, int64_t bytes);
CAF_EXPORT int32_t caf_free(void* d_ptr);
CAF_EXPORT int32_t caf_memset(void* d_ptr, int32_t value, int64_t bytes, void* stream);
CAF_EXPORT int32_t caf_h2d(void* d_dst, const void* h_src, int64_t bytes, void* stream);
CAF_EXPORT int32_t caf_d2h(void* h_dst, const void* d_src, int64_t bytes, void* stream);
CAF_EXPORT int32_t caf_d2d(void* d_dst, const void* d_src, int64_t bytes, void* stream);
CAF_EXPORT int32_t caf_stream_sync(void* stream);

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
} caf_plan_desc;

CAF_EXPORT int32_t caf_plan_create(caf_plan* plan, const caf_plan_desc* desc);
CAF_EXPORT int32_t caf_plan_destroy(caf_plan plan);

/* Geometry chosen by the plan: block size B, valid outputs per block, batch, workspace bytes. */
CAF_EXPORT int32_t caf_plan_info(caf_plan plan, int32_t* block, int32_t* step, int32_t* blocks_per_batch,
                                 int64_t* workspace_bytes);

/* Outputs of one execute (any pointer may be NULL = not wanted). */
typedef struct caf_outputs {
    float* d_surface;      /* [T][num_shifts][F] float32 QF2 (reference CAF layout: delay-major)  */
    float* d_row_max;      /* [T][num_shifts] float32: max over f of QF2                           */
    int32_t* d_row_arg;    /* [T][num_shifts] int32: argmax over f (lowest f on ties)              */
    float* d_peak_val;     /* [T] float32: global maximum of QF2 for template t                    */
    int32_t* d_peak_delay; /* [T] int32: its delay (absolute sample index; lowest on ties)         */
    int32_t* d_peak_freq;  /* [T] int32: its frequency-hypothesis index f                          */
} caf_outputs;

/* d_rx: device complex64 [rx_len].  Requires shift_start >= 0 and
 * shift_start + num_shifts - 1 + N <= rx_len <= max_rx_len.  Asynchronous on `stream`. */
CAF_EXPORT int32_t caf_plan_execute(caf_plan plan, const float* d_rx, int64_t rx_len, int64_t shift_start,
                                    int64_t num_shifts, const caf_outputs* out, void* stream);

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

#ifdef __cplusplus
}
#endif
#endif /* CAF_H_ */
