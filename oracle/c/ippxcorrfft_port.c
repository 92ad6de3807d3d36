/* TEST INFRASTRUCTURE ONLY -- CPU restatement, in plain C with pthreads, of the reference's threaded
 * native branch-B correlator
 *
 *     IppXcorrFFT_32fc::xcorr / xcorr_thread      cython_ext/CyIppXcorrFFT/IppXcorrFFT.cpp:13-52, 94-178
 *     IppXcorrFFT_32fc::IppXcorrFFT_32fc          cython_ext/CyIppXcorrFFT/IppXcorrFFT.cpp:181-196
 *     getOutputLength                              cython_ext/CyIppXcorrFFT/IppXcorrFFT.cpp:3-11
 *
 * per output t (threads strided over t): i = start + t*step; out of range -> (0, 0); otherwise
 * cutout (conjugated) * src[i:i+N] -> forward DFT (no scaling) -> power spectrum -> first maximum ->
 * peak / (float)||cutout||^2 / (float)||slice||^2 with the norms accumulated in double.
 *
 * The reference calls Intel IPP (ippsMul_32fc, ippsDFTFwd_CToC_32fc, ippsPowerSpectr_32fc, ippsMaxIndx_32f,
 * ippsNorm_L2_32fc64f), which is not in this image; the DFT here is a radix-4/2 Stockham FFT for powers of two
 * and a direct O(N^2) DFT (double accumulation) otherwise.  Pinned by tests/test_oracle_c_port.py against the
 * KAT-2 golden vector and the NumPy oracle.  Used by tests and by bench.py's threaded cpu baseline, never by the
 * product path.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float re, im;
} c32;

typedef struct {
    const c32* cutout; /* already conjugated when autoConj */
    int n;
    float cutout_normsq;
    const c32* tw; /* exp(-2 pi i k / n), k < n */
    int pow2;
    const c32* src;
    int srclen, start, step, outlen, nthreads, tid;
    float* peaks;
    int* inds;
} job_t;

static int output_length(int start, int end, int step) {
    int length = (end - start) / step;
    if ((end - start) % step != 0) length += 1;
    return length;
}

/* Stockham autosort FFT, radix 4 with one radix-2 stage when log2(n) is odd; returns the buffer holding the result */
static c32* fft_pow2(int n, c32* x, c32* y, const c32* tw) {
    int len = n, s = 1;
    while (len > 1) {
        if ((len & 3) == 0) {
            const int n1 = len >> 2;
            for (int p = 0; p < n1; ++p) {
                const c32 w1 = tw[p * s], w2 = tw[2 * p * s], w3 = tw[3 * p * s];
                for (int q = 0; q < s; ++q) {
                    const c32 a = x[q + s * p], b = x[q + s * (p + n1)], c = x[q + s * (p + 2 * n1)],
                              d = x[q + s * (p + 3 * n1)];
                    const float apc_r = a.re + c.re, apc_i = a.im + c.im, amc_r = a.re - c.re, amc_i = a.im - c.im;
                    const float bpd_r = b.re + d.re, bpd_i = b.im + d.im;
                    /* j * (b - d) */
                    const float jbmd_r = -(b.im - d.im), jbmd_i = b.re - d.re;
                    c32* o = y + q + s * 4 * p;
                    o[0].re = apc_r + bpd_r;
                    o[0].im = apc_i + bpd_i;
                    const float t1r = amc_r - jbmd_r, t1i = amc_i - jbmd_i;
                    o[s].re = w1.re * t1r - w1.im * t1i;
                    o[s].im = w1.re * t1i + w1.im * t1r;
                    const float t2r = apc_r - bpd_r, t2i = apc_i - bpd_i;
                    o[2 * s].re = w2.re * t2r - w2.im * t2i;
                    o[2 * s].im = w2.re * t2i + w2.im * t2r;
                    const float t3r = amc_r + jbmd_r, t3i = amc_i + jbmd_i;
                    o[3 * s].re = w3.re * t3r - w3.im * t3i;
                    o[3 * s].im = w3.re * t3i + w3.im * t3r;
                }
            }
            len >>= 2;
            s <<= 2;
        } else {
            const int m = len >> 1;
            for (int p = 0; p < m; ++p) {
                const c32 w = tw[p * s];
                for (int q = 0; q < s; ++q) {
                    const c32 a = x[q + s * p], b = x[q + s * (p + m)];
                    c32* o = y + q + s * 2 * p;
                    o[0].re = a.re + b.re;
                    o[0].im = a.im + b.im;
                    const float tr = a.re - b.re, ti = a.im - b.im;
                    o[s].re = w.re * tr - w.im * ti;
                    o[s].im = w.re * ti + w.im * tr;
                }
            }
            len >>= 1;
            s <<= 1;
        }
        c32* t = x;
        x = y;
        y = t;
    }
    return x;
}

static void dft_direct(int n, const c32* x, c32* y, const c32* tw) {
    for (int k = 0; k < n; ++k) {
        double ar = 0.0, ai = 0.0;
        int idx = 0;
        for (int j = 0; j < n; ++j) {
            const c32 w = tw[idx];
            ar += (double)x[j].re * w.re - (double)x[j].im * w.im;
            ai += (double)x[j].re * w.im + (double)x[j].im * w.re;
            idx += k;
            if (idx >= n) idx -= n;
        }
        y[k].re = (float)ar;
        y[k].im = (float)ai;
    }
}

static void* worker(void* arg) {
    const job_t* jb = (const job_t*)arg;
    const int n = jb->n;
    c32* w1 = (c32*)malloc(sizeof(c32) * (size_t)n);
    c32* w2 = (c32*)malloc(sizeof(c32) * (size_t)n);
    if (!w1 || !w2) {
        free(w1);
        free(w2);
        return (void*)1;
    }
    for (int t = jb->tid; t < jb->outlen; t += jb->nthreads) {
        const int64_t i = (int64_t)jb->start + (int64_t)t * jb->step;
        if (i < 0 || i + n > jb->srclen) { /* IppXcorrFFT.cpp:125-130 */
            jb->peaks[t] = 0.0f;
            jb->inds[t] = 0;
            continue;
        }
        const c32* s = jb->src + i;
        double e = 0.0;
        for (int k = 0; k < n; ++k) {
            const c32 a = jb->cutout[k], b = s[k];
            w1[k].re = a.re * b.re - a.im * b.im;
            w1[k].im = a.re * b.im + a.im * b.re;
            e += (double)b.re * b.re + (double)b.im * b.im;
        }
        const c32* spec;
        if (jb->pow2) {
            spec = fft_pow2(n, w1, w2, jb->tw);
        } else {
            dft_direct(n, w1, w2, jb->tw);
            spec = w2;
        }
        float maxval = -1.0f;
        int maxind = 0;
        for (int k = 0; k < n; ++k) {
            const float p = spec[k].re * spec[k].re + spec[k].im * spec[k].im;
            if (p > maxval) {
                maxval = p;
                maxind = k;
            }
        }
        const double slicenorm = sqrt(e);
        jb->peaks[t] = maxval / jb->cutout_normsq / (float)(slicenorm * slicenorm); /* :174 */
        jb->inds[t] = maxind;
    }
    free(w1);
    free(w2);
    return NULL;
}

/* cutout, src: interleaved complex64.  Returns 0, 1 (bad arguments / wrong outlen, the reference's
 * std::runtime_error :63-66) or 2 (allocation / thread failure). */
int ippxcorrfft_port(const float* cutout, int n, int auto_conj, const float* src, int srclen, int start, int end, int step,
                     int nthreads, float* peaks, int* inds, int outlen) {
    if (!cutout || !src || n < 1 || step < 1 || nthreads < 1 || outlen < 0) return 1;
    if (output_length(start, end, step) != outlen) return 1;
    c32* c = (c32*)malloc(sizeof(c32) * (size_t)n);
    c32* tw = (c32*)malloc(sizeof(c32) * (size_t)n);
    if (!c || !tw) {
        free(c);
        free(tw);
        return 2;
    }
    double e = 0.0;
    for (int k = 0; k < n; ++k) {
        c[k].re = cutout[2 * k];
        c[k].im = auto_conj ? -cutout[2 * k + 1] : cutout[2 * k + 1];
        e += (double)c[k].re * c[k].re + (double)c[k].im * c[k].im;
        const double ang = -2.0 * M_PI * (double)k / (double)n;
        tw[k].re = (float)cos(ang);
        tw[k].im = (float)sin(ang);
    }
    const double norm2 = sqrt(e);
    const float normsq = (float)(norm2 * norm2); /* :193 */
    if (nthreads > 256) nthreads = 256;
    job_t jobs[256];
    pthread_t th[256];
    int created[256];
    int rc = 0;
    for (int t = 0; t < nthreads; ++t) {
        job_t j = {c, n, normsq, tw, (n & (n - 1)) == 0, (const c32*)src, srclen, start, step, outlen, nthreads, t, peaks, inds};
        jobs[t] = j;
    }
    for (int t = 0; t < nthreads; ++t) {
        created[t] = pthread_create(&th[t], NULL, worker, &jobs[t]) == 0;
        if (!created[t] && worker(&jobs[t]) != NULL) rc = 2; /* no thread: do its share here */
    }
    for (int t = 0; t < nthreads; ++t) {
        void* r = NULL;
        if (created[t]) {
            pthread_join(th[t], &r);
            if (r != NULL) rc = 2;
        }
    }
    free(c);
    free(tw);
    return rc;
}
