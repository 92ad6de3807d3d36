#!/usr/bin/env python3
"""DESIGN 4.5.1's table from profiles/<round>/kernels_summary.json (labels here, numbers from the summary)."""
import json
import sys

ROWS = [  # (workload, kernel prefix of the manifest entry, label)
    ("c2_surface", "k_caf_persistent", "`k_caf_persistent` (C2 surface)"),
    ("c2_surface_t", "k_caf_persistent", "`k_caf_persistent` (C2, hypothesis-major surface written by the FFT items)"),
    ("c2_nosurface", "k_caf_persistent", "`k_caf_persistent` (C2 no surface)"),
    ("c4_share", "k_caf_persistent", "`k_caf_persistent` (C4 share: 64 templates × 512 bins)"),
    ("c2_long_template", "k_caf_persistent", "`k_caf_persistent` (N = 16384, 2 × 16384)"),
    ("c2_lb16", "k_caf_persistent", "`k_caf_persistent` (N = 32768, 65536-point blocks folded: 2 × 16384 per output residue)"),
    ("c3", "k_caf_persistent", "`k_caf_persistent` (C3: rows written by the FFT items)"),
    ("c3_complex_rows", "k_caf_persistent", "`k_caf_persistent` (C3 complex-QF rows, `TemplateCrossCorrelator.correlate`)"),
    ("c2_fused", "k_fused_caf", "`k_fused_caf`"),
    ("c2_fused", "k_transpose_norm_argmax", "`k_transpose_norm_argmax`"),
    ("c2_rocfft", "k_spectral_mul", "`k_spectral_mul`"),
    ("c2_rocfft", "k_magsq_norm_argmax", "`k_magsq_norm_argmax`"),
    ("c3", "k_rows_peak", "`k_rows_peak` (64 × 2²⁴ rows → peak records)"),
    ("c2_surface", "k_block_spectra", "`k_block_spectra` (1366 blocks of 16384)"),
    ("c2_long_template", "k_block_spectra32", "`k_block_spectra32` (683 blocks of 32768, parity-major + butterfly order)"),
    ("perdelay_fused_4096", "k_perdelay_fused", "`k_perdelay_fused<12>` (4096 × 10⁶)"),
    ("perdelay_fused_256", "k_perdelay_fused", "`k_perdelay_fused<8>` (256 × 10⁶)"),
    ("direct_small_support", "k_direct_caf", "`k_direct_caf` (16 samples of support, 64 frequencies, 2²² delays, surface)"),
    ("cp_fastxcorr_1e7", "k_sliding_multiply", "`k_sliding_multiply` (128 rows × 10⁷)"),
    ("perdelay_decimal_1000", "k_perdelay_r10", "`k_perdelay_r10<3>` (1000 × 10⁶, radix 10)"),
    ("perdelay_mixed_1200", "k_pdj", "`k_pdj` (1200 × 10⁵, compiled for the length: radices 5·16·15)"),
    ("perdelay_mixed_5000", "k_pdj", "`k_pdj` (5000 × 10⁵: radices 10·25·20)"),
    ("perdelay_mixed_1400", "k_pdj", "`k_pdj` (1400 × 10⁵: radices 4·25·14)"),
    ("perdelay_bluestein_1450", "k_pdj", "`k_pdj` (1450 × 10⁵: Bluestein, 3072-point convolution)"),
    ("perdelay_split_65536", "k_pdj", "`k_pdj` (65536 × 2·10⁴: 4 residues × 16384 points; bytes = L2 reads)"),
    ("perdelay_rows_10007", "k_sliding_multiply", "`k_sliding_multiply` (2·10⁴ rows × 10007)"),
    ("cp_fastxcorr_1e7", "k_rows_argmax", "`k_rows_argmax` (chunked, 128 rows × 10⁷)"),
    ("kernels_misc", "k_magnsq", "`k_magnsq`"),
    ("kernels_misc", "k_iq16_to_c64", "`k_iq16_to_c64`"),
    ("kernels_misc", "k_fir_os", "`k_fir_os<12>` (1024 taps)"),
    ("fir_direct", "k_fir_fast", "`k_fir_fast` (64 / 128 taps)"),
    ("kernels_misc", "k_fir_poly", "`k_fir_poly` (int16 → 64 taps → ÷4)"),
    ("kernels_misc", "k_multi_template_dot", "`k_multi_template_dot_rt` (20 × 100, 10⁷ slides)"),
    ("kernels_misc", "k_moving_tile", "`k_moving_tile` (L = 100, 2²⁴ samples)"),
    ("kernels_misc", "k_upfirdn_poly", "`k_upfirdn_poly` (64 × 2¹⁸, 128 taps, up 5 down 2)"),
    ("kernels_misc", "k_local_max", "`k_local_max_flags` + `k_local_max_write` (2²⁴ trace, read once)"),
    ("kernels_misc", "k_prefix_tiles", "`k_prefix_tiles` × 2 + `k_scan_tile_sums` (energy prefix of 10⁷ samples)"),
    ("c2_rocfft", "k_inv_energy", "`k_inv_energy`"),
    ("c2_rocfft", "k_gather_blocks", "`k_gather_blocks`"),
]


def t(us):
    return "%.2f ms" % (us / 1e3) if us >= 1000 else "%.0f µs" % us


def main(path):
    j = json.load(open(path))
    print("| kernel (workload) | per call | achieved | of peak | meas/alg |")
    print("|---|---|---|---|---|")
    for w, pref, label in ROWS:
        ks = [k for k in j["workloads"][w]["kernels"] if k["kernel"].startswith(pref)]
        if not ks:
            continue
        k = ks[0]
        # (kernels whose bytes do not scale with their work -- the fused per-delay correlators -- are priced on f32 only)
        hbm = k["alg_bytes_per_call"] and not (k["alg_flops_per_call"] and k["frac_hbm_peak"] < 0.02 and k["kernel"].startswith("k_perdelay"))
        ach = "%.2f TB/s" % (k["achieved_GBs"] / 1e3) if hbm else ""
        of = "%.2f HBM" % k["frac_hbm_peak"] if hbm else ""
        if k["alg_flops_per_call"]:
            ach += (" + " if ach else "") + "%.1f TFLOP/s" % k["achieved_TFLOPs"]
            of += (" / " if of else "") + "%.2f f32" % k["frac_f32_peak"]
        r = k["measured_over_algorithmic"]
        print("| %s | %s (median launch %s, min %s; %d dispatches) | %s | %s | %s |" % (
            label, t(k["per_call_us"]), t(k["median_launch_us"]), t(k["min_launch_us"]), k["dispatches"], ach, of,
            "%.2f" % r if r else "–"))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "profiles/r04/kernels_summary.json")
