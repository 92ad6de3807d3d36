// Internal declarations shared by the kernel and plan translation units of libcaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "caf.h"

#include <string>

namespace caf {

constexpr int MUL_THREADS = 256;  // spectral multiply: 256 threads x 2 points
constexpr int MAG_THREADS = 256;  // |.|^2/normalise/argmax: 4 waves
constexpr int MAG_S = 64;         // delays per tile (one wave-row of 8-byte loads = 512 B)
constexpr int MAG_F = 128;        // frequency hypotheses per LDS chunk (512 B store rows)

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

int64_t prefix_num_tiles(int64_t m);
void launch_energy_prefix(const float2* rx, int64_t m, double* tile_sums, double* prefix, hipStream_t st);
void launch_inv_energy(const double* prefix, int64_t shift_start, int64_t num_shifts, const int32_t* gstart,
                       const int32_t* glen, int32_t ngroups, float* inv_e, hipStream_t st);
void launch_gather_blocks(const float2* rx, int64_t rx_len, int64_t src0, int32_t step, int32_t bsz, int32_t nblk,
                          float2* xb, hipStream_t st);
void launch_conj_scale(float2* h, int64_t n, float scale, hipStream_t st);
void launch_spectral_mul(int mode, const float2* xb, const float2* hc, const int32_t* shifts, int32_t bsz,
                         int32_t pitch, int32_t nfreq, int32_t nhyp, int32_t hyp_per_wg, int32_t nblk, float2* pbuf,
                         hipStream_t st);
void launch_magsq(const float2* pbuf, int32_t pitch, int32_t ntmpl, int32_t nfreq, const float* tscale,
                  const float* inv_e, int64_t num_shifts, int64_t shift_start, int32_t step, int32_t blk0,
                  int32_t nblk, int32_t tiles_per_blk, float* surface, float* row_max, int32_t* row_arg,
                  PeakRec* partial, int64_t partial_per_tmpl, hipStream_t st);
void launch_peak_reduce(const PeakRec* partial, int64_t count, int64_t stride, int32_t ntmpl, float* pv, int32_t* pd,
                        int32_t* pf, hipStream_t st);

}  // namespace caf
