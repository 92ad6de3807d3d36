/*
 * A client of libcaf.so written in plain C (C99): what a maintainer of the reference's native twins
 * (cython_ext/CyIppXcorrFFT, pybinds/ippGroupXcorrCZT) would link instead of Intel IPP.  It does what
 * benchmarks/benchmark_xcorrs.py:28-59 does with fastXcorr -- one template, a band of frequency bins, every delay of
 * an rx that holds one planted echo -- through the C-ABI only: caf_plan_create -> caf_plan_execute (device buffers)
 * -> caf_zoom_czt around the peak, and the blocking host-pointer call caf_plan_execute_host.
 *
 * Build:  gcc -std=c99 -O2 -I../../include caf_client.c -L../../pydsproutines_amd -lcaf -lm -Wl,-rpath,$PWD/../../pydsproutines_amd
 * Exit status 0 and "caf_client: ok" when every planted value is recovered.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "caf.h"

#define CHECK(call)                                                            \
    do {                                                                       \
        const int32_t rc_ = (call);                                            \
        if (rc_ != CAF_OK) {                                                   \
            char msg_[512];                                                    \
            caf_last_error(msg_, (int32_t)sizeof msg_);                        \
            fprintf(stderr, "%s failed (%d): %s\n", #call, (int)rc_, msg_);    \
            return 1;                                                          \
        }                                                                      \
    } while (0)

static uint32_t lcg_state = 12345u;
static float uniform_pm1(void) {
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)((lcg_state >> 8) & 0xffff) / 32768.0f - 1.0f;
}

int main(void) {
    enum { N = 1024, M = 200000, F = 33, D0 = 123457, K0 = -7 };
    const double two_pi = 6.283185307179586;
    int32_t ndev = 0;
    CHECK(caf_device_count(&ndev));
    if (ndev < 1) {
        fprintf(stderr, "no gfx950 device\n");
        return 2;
    }
    CHECK(caf_set_device(0));
    printf("caf_client: ABI %d.%d, %d device(s)\n", caf_abi_version() >> 16, caf_abi_version() & 0xffff, (int)ndev);

    /* QPSK-like template, noise rx with the template planted at delay D0 and bin K0 of an N-point grid */
    float* tmpl = (float*)malloc(sizeof(float) * 2 * N);
    float* rx = (float*)malloc(sizeof(float) * 2 * M);
    int32_t bins[F];
    if (!tmpl || !rx) return 3;
    for (int n = 0; n < N; ++n) {
        tmpl[2 * n] = uniform_pm1() > 0 ? 0.70710678f : -0.70710678f;
        tmpl[2 * n + 1] = uniform_pm1() > 0 ? 0.70710678f : -0.70710678f;
    }
    for (int i = 0; i < 2 * M; ++i) rx[i] = 0.25f * uniform_pm1();
    for (int n = 0; n < N; ++n) {
        const double ph = two_pi * (double)K0 * n / N;
        const float c = (float)cos(ph), s = (float)sin(ph);
        rx[2 * (D0 + n)] += tmpl[2 * n] * c - tmpl[2 * n + 1] * s;
        rx[2 * (D0 + n) + 1] += tmpl[2 * n] * s + tmpl[2 * n + 1] * c;
    }
    for (int f = 0; f < F; ++f) bins[f] = f - F / 2;

    caf_plan_desc desc;
    memset(&desc, 0, sizeof desc);
    desc.num_templates = 1;
    desc.template_len = N;
    desc.h_templates = tmpl;
    desc.auto_conj = 1;
    desc.num_groups = 0;
    desc.freq_mode = CAF_FREQ_BINS;
    desc.num_freqs = F;
    desc.h_bins = bins;
    desc.grid = N;
    desc.max_rx_len = M;
    desc.engine = CAF_ENGINE_AUTO;
    caf_plan plan = NULL;
    CHECK(caf_plan_create(&plan, &desc));
    int32_t block = 0, step = 0, nb = 0, engine = 0;
    int64_t ws = 0;
    CHECK(caf_plan_info(plan, &block, &step, &nb, &ws));
    CHECK(caf_plan_engine(plan, &engine));
    printf("caf_client: block %d, %d delays per block, engine %d, workspace %.1f MB\n", (int)block, (int)step, (int)engine,
           (double)ws / 1e6);

    /* device-resident call: rx in HBM, per-delay trace + peak record out */
    const int64_t S = M - N + 1;
    void *d_rx = NULL, *d_row_max = NULL, *d_row_arg = NULL, *d_pv = NULL, *d_pd = NULL, *d_pf = NULL;
    CHECK(caf_malloc(&d_rx, (int64_t)sizeof(float) * 2 * M));
    CHECK(caf_malloc(&d_row_max, 4 * S));
    CHECK(caf_malloc(&d_row_arg, 4 * S));
    CHECK(caf_malloc(&d_pv, 4));
    CHECK(caf_malloc(&d_pd, 4));
    CHECK(caf_malloc(&d_pf, 4));
    CHECK(caf_h2d(d_rx, rx, (int64_t)sizeof(float) * 2 * M, NULL));
    caf_outputs out;
    memset(&out, 0, sizeof out);
    out.d_row_max = (float*)d_row_max;
    out.d_row_arg = (int32_t*)d_row_arg;
    out.d_peak_val = (float*)d_pv;
    out.d_peak_delay = (int32_t*)d_pd;
    out.d_peak_freq = (int32_t*)d_pf;
    CHECK(caf_plan_execute(plan, (const float*)d_rx, M, 0, S, &out, NULL));
    float pv = 0.f;
    int32_t pd = -1, pf = -1;
    CHECK(caf_d2h(&pv, d_pv, 4, NULL));
    CHECK(caf_d2h(&pd, d_pd, 4, NULL));
    CHECK(caf_d2h(&pf, d_pf, 4, NULL));
    CHECK(caf_stream_sync(NULL));
    printf("caf_client: device call  -> peak QF^2 %.4f at delay %d, bin %d\n", pv, (int)pd, (int)bins[pf]);
    int bad = !(pd == D0 && bins[pf] == K0 && pv > 0.8f && pv < 1.0f);
    /* the per-delay maxima around the peak as float64 on the host (what the reference's CPU signatures return): widened by the
     * download itself */
    double trace[5];
    CHECK(caf_d2h_f64(trace, (const float*)d_row_max + (D0 - 2), 5, NULL));
    bad |= !(trace[2] == (double)pv && trace[1] < trace[2] && trace[3] < trace[2]);

    /* fine frequency around the peak: one call, +-1 bin in steps of 1/16 bin */
    int32_t nfine = 0;
    CHECK(caf_zoom_num_bins(1.0 / N, 1.0 / (16.0 * N), &nfine));
    void *z_cnt = NULL, *z_delay = NULL, *z_ci = NULL, *z_cq = NULL, *z_fi = NULL, *z_ff = NULL, *z_fq = NULL;
    CHECK(caf_malloc(&z_cnt, 4));
    CHECK(caf_malloc(&z_delay, 4 * 4));
    CHECK(caf_malloc(&z_ci, 4 * 4));
    CHECK(caf_malloc(&z_cq, 4 * 4));
    CHECK(caf_malloc(&z_fi, 4 * 4));
    CHECK(caf_malloc(&z_ff, 8 * 4));
    CHECK(caf_malloc(&z_fq, 4 * 4));
    caf_zoom_outputs zo;
    memset(&zo, 0, sizeof zo);
    zo.d_count = (int32_t*)z_cnt;
    zo.d_delay = (int32_t*)z_delay;
    zo.d_coarse_freq_index = (int32_t*)z_ci;
    zo.d_coarse_qf2 = (float*)z_cq;
    zo.d_fine_index = (int32_t*)z_fi;
    zo.d_fine_freq = (double*)z_ff;
    zo.d_fine_qf2 = (float*)z_fq;
    CHECK(caf_zoom_czt(plan, 0, (const float*)d_rx, M, (const float*)d_row_max, (const int32_t*)d_row_arg, 0, S, 4, 0.5f,
                       1.0 / N, 1.0 / (16.0 * N), &zo, NULL));
    int32_t zc = 0, zd = -1;
    double zf = 0.0;
    float zq = 0.f;
    CHECK(caf_d2h(&zc, z_cnt, 4, NULL));
    CHECK(caf_d2h(&zd, z_delay, 4, NULL));
    CHECK(caf_d2h(&zf, z_ff, 8, NULL));
    CHECK(caf_d2h(&zq, z_fq, 4, NULL));
    CHECK(caf_stream_sync(NULL));
    printf("caf_client: zoom (%d fine bins) -> %d peak(s); best at delay %d, %.4f bins, QF^2 %.4f\n", (int)nfine, (int)zc,
           (int)zd, zf * N, zq);
    bad |= !(zc >= 1 && zd == D0 && fabs(zf * N - (double)K0) < 0.07 && zq >= pv - 1e-3f);

    /* the reference-DLL style: host pointers in, caller-allocated host outputs, blocking */
    float hv = 0.f;
    int32_t hd = -1, hf = -1;
    CHECK(caf_plan_execute_host(plan, rx, M, 0, S, NULL, NULL, NULL, &hv, &hd, &hf));
    printf("caf_client: host call    -> peak QF^2 %.4f at delay %d, bin %d\n", hv, (int)hd, (int)bins[hf]);
    bad |= !(hd == D0 && bins[hf] == K0 && hv == pv);

    CHECK(caf_free(z_cnt));
    CHECK(caf_free(z_delay));
    CHECK(caf_free(z_ci));
    CHECK(caf_free(z_cq));
    CHECK(caf_free(z_fi));
    CHECK(caf_free(z_ff));
    CHECK(caf_free(z_fq));
    CHECK(caf_free(d_rx));
    CHECK(caf_free(d_row_max));
    CHECK(caf_free(d_row_arg));
    CHECK(caf_free(d_pv));
    CHECK(caf_free(d_pd));
    CHECK(caf_free(d_pf));
    CHECK(caf_plan_destroy(plan));
    free(tmpl);
    free(rx);
    printf(bad ? "caf_client: MISMATCH\n" : "caf_client: ok\n");
    return bad ? 4 : 0;
}
