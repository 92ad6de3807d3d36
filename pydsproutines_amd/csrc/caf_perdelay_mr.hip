// Per-delay correlator for cutouts of N = 2^a 3^b 5^c samples (32 <= N <= 16200, N neither a power of two nor of ten: those
// have kernels of their own in caf_perdelay.hip): rx window x conj(cutout) -> N-point FFT in LDS -> |.|^2 -> (max, first
// argmax) per delay, in ONE kernel -- the reference's literal per-delay algorithm (fastXcorr branches B / C,
// xcorrRoutines.py:511-566; cp_fastXcorr :29-167, whose cutout length is a free argument, benchmark_xcorrs.py:62-71;
// IppXcorrFFT.cpp:94-178) without the (rows, N) product matrix that the product kernel -> rocFFT rows -> argmax chain
// moves through HBM four times.
//
// Mixed-radix Stockham autosort transform, radices from {16, 10, 8, 5, 4, 3, 2}, chosen on the host (fewest passes, the
// largest first).  The lengths are too many to instantiate one kernel each, so the passes after the first are driven by
// a small plan in the kernel arguments: a uniform switch picks the butterfly, the trip counts are compile-time bounds with
// a guard (a thread owns up to MR_PT = 20 points: floor(20 / R) butterflies of radix R), divisions by the pass stride are
// multiplications by a host-prepared reciprocal.  Only the FIRST radix is a template parameter: that pass is the one with
// the global loads.  One LDS image per row (padded by one element per sixteen), inputs
// read before a barrier and outputs written after it; ceil(N / 16) threads per row, several rows per workgroup for short
// cutouts.  Window energies from the caller's float64 prefix and ||cutout|| from launch_cutout_norm, exactly as the
// three-kernel form and the radix-10 kernel normalise; same out-of-range and zero-energy rules (include/caf.h).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#include "caf_internal.h"
#include "caf_fft_dev.h"

namespace caf {
namespace {

constexpr int MR_MAXP = 8;  // passes (2^14 as radix-2 passes would need 14: the planner never produces more than 7)
constexpr int MR_PT = 20;   // points per thread and pass

struct MrPlan {
    int32_t n, tpr, rpw, npass, img;      // length, threads per row, rows per workgroup, passes, padded row image (elements)
    int32_t radix[MR_MAXP], ns[MR_MAXP];  // pass p: radix, product of the radices before it
    uint32_t ns_rcp[MR_MAXP];             // floor(2^32 / ns) + 1: j / ns == umulhi(j, ns_rcp) for j < 2^16
};

__device__ __forceinline__ int mr_pad(int a) { return a + (a >> 4); }

template <int R>
__device__ __forceinline__ void mr_idft(float2* v) {
    if constexpr (R == 2) idft2(v[0], v[1]);
    if constexpr (R == 3) idft3(v[0], v[1], v[2]);
    if constexpr (R == 4) idft4(v[0], v[1], v[2], v[3]);
    if constexpr (R == 5) idft5(v[0], v[1], v[2], v[3], v[4]);
    if constexpr (R == 8) idft8(*reinterpret_cast<float2(*)[8]>(v));
    if constexpr (R == 10) idft10(*reinterpret_cast<float2(*)[10]>(v));
    if constexpr (R == 16) idft16(*reinterpret_cast<float2(*)[16]>(v));
}

// One Stockham pass of radix R with ns = the product of the earlier radices: image -> registers, barrier, twiddles,
// butterflies, registers -> image, barrier.  Nothing is live across a pass but the image: each of the seven bodies behind the
// uniform switch gets a register allocation of its own (with the data registers carried from pass to pass through the
// switch, and the last pass' outputs into a per-radix epilogue, every body spilled 10 .. 20 registers at 128).
// A thread's butterflies past the end (j >= N / R: floor(20 / R) tpr may exceed N / R) are CLAMPED to the last one instead
// of being branched around: they recompute it and store the same values to the same places.
// last_reg (the last pass of a row whose planes are not wanted): the outputs are not written back; |z|^2 and the thread's
// (maximum, first index) come straight from the registers -- register (c, t) <-> spectrum index j + t N / R, visited t-major,
// i.e. ascending (a clamped butterfly offers the last one's value at the last one's index again: the strict comparison
// ignores it) -- and the row saves a round trip through the image and two barriers.
template <int R, bool FIRST>
__device__ __forceinline__ void mr_pass(const MrPlan& pl, int p, float2* __restrict__ buf, const float2* __restrict__ tw, int l_in,
                                        bool active, float2 (&v)[MR_PT], bool last_reg, float inv, float& bv, uint32_t& bi) {
    constexpr int CNT = MR_PT / R;
    const int nb = pl.n / R, ns = pl.ns[p], tpr = pl.tpr;
    // (an opaque copy of the lane's index: the image addresses of a radix depend on nothing that changes from row to row,
    //  and hoisted out of the row loop -- twenty per radix, seven radices -- they were spilled: ~330 registers of scratch)
    int l = l_in;
    asm volatile("" : "+v"(l));
    if (!FIRST) {
        // the twiddle bases W_{R ns}^k = W_N^{k N / (R ns)} of all of the thread's butterflies are fetched FIRST: they do not
        // depend on the image, and their L1 / L2 latency then runs under the image reads and the barrier
        const int tws = nb / ns;
        float2 w1[CNT];
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
            const int j = min(l + c * tpr, nb - 1);
            const int k = j - ns * (int)__umulhi((uint32_t)j, pl.ns_rcp[p]);
            w1[c] = tw[k * tws];
        }
        if (active) {
#pragma unroll
            for (int c = 0; c < CNT; ++c) {
                const int j = min(l + c * tpr, nb - 1);
#pragma unroll
                for (int t = 0; t < R; ++t) v[c * R + t] = buf[mr_pad(j + t * nb)];
            }
        }
        __syncthreads();  // every butterfly has its inputs: the image may be overwritten
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
            float2 pw = w1[c];
            v[c * R + 1] = cmul(v[c * R + 1], pw);
#pragma unroll
            for (int t = 2; t < R; ++t) {
                pw = cmul(pw, w1[c]);
                v[c * R + t] = cmul(v[c * R + t], pw);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        mr_idft<R>(&v[c * R]);
        if (CNT > 1) __builtin_amdgcn_sched_barrier(0);  // (one butterfly's temporaries at a time)
    }
    if (last_reg) {
#pragma unroll
        for (int t = 0; t < R; ++t)
#pragma unroll
            for (int c = 0; c < CNT; ++c) {
                const int j = min(l + c * tpr, nb - 1);
                const float zr = v[c * R + t].x * inv, zi = v[c * R + t].y * inv;
                const float val = __builtin_fmaf(zr, zr, zi * zi);
                const bool up = val > bv;
                bv = up ? val : bv;
                bi = up ? (uint32_t)(j + t * nb) : bi;
            }
        return;
    }
    if (active) {
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
            const int j = min(l + c * tpr, nb - 1);
            const int k = FIRST ? 0 : j - ns * (int)__umulhi((uint32_t)j, pl.ns_rcp[p]);
            const int dst = (j - k) * R + k;
#pragma unroll
            for (int t = 0; t < R; ++t) buf[mr_pad(dst + t * ns)] = v[c * R + t];
        }
    }
    __syncthreads();
}

// WGMAX: the largest workgroup the instance is launched with (512 for cutouts of up to 8192 samples, 1024 beyond)
template <int R0, int WGMAX>
__global__ __launch_bounds__(WGMAX, 4) void k_perdelay_mr(MrPlan pl, const float2* __restrict__ x, const float2* __restrict__ y,
                                                          int64_t ylen, const float2* __restrict__ tw,
                                                          const double* __restrict__ prefix, const double* __restrict__ xnorm,
                                                          int64_t start, int64_t step, int64_t num, int32_t rows_per_wg,
                                                          int32_t zero_oor, float* __restrict__ qf2, uint32_t* __restrict__ fidx,
                                                          float* __restrict__ plane, float2* __restrict__ cplane) {
    extern __shared__ __attribute__((aligned(16))) float2 s_buf[];  // rpw row images, then 2 x rpw key slots
    constexpr int CNT0 = MR_PT / R0;
    const int tid = threadIdx.x, N = pl.n, tpr = pl.tpr, rpw = pl.rpw, nb0 = N / R0;
    unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_buf + (size_t)rpw * pl.img);
    const bool active = tid < rpw * tpr;  // (the last threads of the workgroup only keep the barriers company)
    const int rl = active ? tid / tpr : 0, l = active ? tid - rl * tpr : 0;
    float2* buf = s_buf + (size_t)rl * pl.img;
    const double xn = *xnorm;
    if (tid < 2 * rpw) s_key[tid] = 0ull;
    __syncthreads();
    const int64_t row0 = (int64_t)blockIdx.x * rows_per_wg * rpw;
    for (int it = 0; it < rows_per_wg; ++it) {
        const int64_t row = row0 + (int64_t)it * rpw + rl;
        const bool live = active && row < num;
        const int64_t s = start + row * step;
        const bool oor = (s < 0) || (s + N > ylen);
        const bool zero = !live || (oor && zero_oor);
        // normalisation exactly as the three-kernel form computes it (k_sliding_multiply): window energy from the prefix.
        // (Up here: the two prefix loads travel with the window loads instead of starting a round trip of their own after
        //  the last pass.)
        float inv = 0.f;
        if (!zero) {
            const int64_t a = s < 0 ? 0 : (s > ylen ? ylen : s);
            int64_t b = s + N;
            b = b < 0 ? 0 : (b > ylen ? ylen : b);
            inv = (float)(1.0 / (sqrt(prefix[b] - prefix[a]) * xn));
        }
        // rows whose planes are not wanted finish in the registers of the last pass (mr_pass, last_reg); with planes the
        // finished spectrum goes through the image once more and leaves in contiguous runs
        const bool reg_tail = !plane && !cplane;
        float bv = -1.f;
        uint32_t bi = 0;
        {
            float2 v[MR_PT];
            // (the cutout is re-read per row -- it stays in the L1 / L2 --: twenty more resident points per thread do not fit
            //  beside the transform's twenty.  Butterflies past the end are clamped, see mr_pass.)
            int lx = l;  // (opaque: otherwise the cutout loads are hoisted out of the row loop and held -- and spilled -- after all)
            asm volatile("" : "+v"(lx));
            if (!zero && !oor) {
                const float2* yrow = y + s;
#pragma unroll
                for (int c = 0; c < CNT0; ++c) {
                    const int j = min(lx + c * tpr, nb0 - 1);
#pragma unroll
                    for (int t = 0; t < R0; ++t) {
                        const float2 a = x[j + t * nb0], b = yrow[j + t * nb0];
                        v[c * R0 + t] = make_float2(a.x * b.x - a.y * b.y, -(a.x * b.y + a.y * b.x));  // conj(x y): see k_perdelay_fused
                    }
                    // (one butterfly's loads in flight at a time: all twenty points' x and y at once are 80 registers)
                    if (CNT0 > 1) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int c = 0; c < CNT0; ++c) {
                    const int j = min(lx + c * tpr, nb0 - 1);
#pragma unroll
                    for (int t = 0; t < R0; ++t) {
                        const int64_t g = s + j + t * nb0;
                        const float2 a = x[j + t * nb0];
                        const float2 b = (!zero && g >= 0 && g < ylen) ? y[g] : make_float2(0.f, 0.f);
                        v[c * R0 + t] = make_float2(a.x * b.x - a.y * b.y, -(a.x * b.y + a.y * b.x));
                    }
                }
            }
            mr_pass<R0, true>(pl, 0, buf, tw, l, active, v, reg_tail && pl.npass == 1, inv, bv, bi);
        }
        for (int p = 1; p < pl.npass; ++p) {
            float2 v[MR_PT];
            const bool lastr = reg_tail && p + 1 == pl.npass;
            switch (pl.radix[p]) {  // (uniform)
                case 2: mr_pass<2, false>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 3: mr_pass<3, false>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 4: mr_pass<4, false>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 5: mr_pass<5, false>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 8: mr_pass<8, false>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 10: mr_pass<10, false>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                default: mr_pass<16, false>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
            }
        }
        // With planes: the finished spectrum is in the image in natural order: thread l takes indices l, l + tpr, ... (ascending,
        // so a strict comparison keeps the first maximum); planes leave as contiguous runs.
        float* prow = (plane && live) ? plane + row * N : nullptr;
        float2* crow = (cplane && live) ? cplane + row * N : nullptr;
        if (active && !reg_tail) {
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int idx = l + i * tpr;
                if (idx < N) {
                    const float2 z = buf[mr_pad(idx)];
                    const float zr = z.x * inv, zi = z.y * inv;
                    const float val = __builtin_fmaf(zr, zr, zi * zi);
                    if (prow) prow[idx] = val;
                    if (crow) crow[idx] = make_float2(zr, -zi);
                    const bool up = val > bv;
                    bv = up ? val : bv;
                    bi = up ? (uint32_t)idx : bi;
                }
            }
        }
        if (qf2 || fidx) {
            // (a thread that saw only NaNs offers nothing: an all-NaN row -- a zero-energy window -- keeps key 0)
            unsigned long long* slot = &s_key[(it & 1) * rpw + rl];
            const unsigned long long key = bv < 0.f ? 0ull : (((unsigned long long)__float_as_uint(bv) << 32) | (uint32_t)~bi);
            if (live && key) atomicMax(slot, key);
            __syncthreads();
            if (live && l == 0) {
                const unsigned long long kk = *slot;
                if (qf2) qf2[row] = kk ? __uint_as_float((uint32_t)(kk >> 32)) : __builtin_nanf("");  // (zero-energy window: (NaN, 0))
                if (fidx) fidx[row] = kk ? ~(uint32_t)kk : 0u;
                *slot = 0ull;  // (next used two rows from now, behind the barriers of the row in between)
            }
        } else if (!reg_tail) {
            __syncthreads();  // the image is read to the end before the next row's first pass overwrites it
        }
    }
}

// fewest passes with radices from {16, 10, 8, 5, 4, 3, 2}; ties: the lexicographically largest sequence (big radices first)
bool mr_factor(int32_t n, std::vector<int>& best) {
    static const int radices[] = {16, 10, 8, 5, 4, 3, 2};
    std::vector<int> cur;
    best.clear();
    bool found = false;
    // (radices in non-increasing order: one representative per multiset; the first sequence with the fewest passes stays)
    std::function<void(int32_t, int)> rec = [&](int32_t rem, int max_r) {
        if (rem == 1) {
            if (!found || cur.size() < best.size()) best = cur, found = true;
            return;
        }
        if ((int)cur.size() >= MR_MAXP || (found && cur.size() + 1 >= best.size())) return;
        for (int r : radices) {
            if (r > max_r || rem % r) continue;
            cur.push_back(r);
            rec(rem / r, r);
            cur.pop_back();
        }
    };
    rec(n, 16);
    return found;
}

// e^{+j 2 pi q / n}, q < n (built once per device and length; at most 64 tables are kept)
int mr_twiddles(int device, int32_t n, const float2** out) {
    static std::mutex mu;
    static std::vector<std::pair<std::pair<int, int32_t>, float2*>> tabs;
    std::lock_guard<std::mutex> lk(mu);
    for (auto& e : tabs)
        if (e.first.first == device && e.first.second == n) {
            *out = e.second;
            return CAF_OK;
        }
    if (tabs.size() >= 64) {  // a caller sweeping lengths: drop the oldest table once nothing can be reading it any more
        CAF_HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(tabs.front().second);
        tabs.erase(tabs.begin());
    }
    std::vector<std::complex<float>> t(n);
    for (int q = 0; q < n; ++q) {
        const double ph = 2.0 * M_PI * (double)q / (double)n;
        t[q] = std::complex<float>((float)std::cos(ph), (float)std::sin(ph));
    }
    float2* d = nullptr;
    CAF_HIP_TRY(hipMalloc((void**)&d, (size_t)n * 8));
    const hipError_t e = hipMemcpy(d, t.data(), (size_t)n * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        CAF_HIP_TRY(e);
    }
    tabs.push_back({{device, n}, d});
    *out = d;
    return CAF_OK;
}

template <int R0, int WGMAX>
int mr_launch(const MrPlan& pl, size_t lds, int dev, const float2* x, const float2* y, int64_t ylen, const float2* tw,
              const double* prefix, const double* xnorm, int64_t start, int64_t step, int64_t num, int32_t zero_oor, float* qf2,
              uint32_t* fidx, float* plane, float2* cplane, hipStream_t st) {
    // the LDS limit of the function is raised per device (a process may drive several), remembered under a lock
    static std::mutex mu;
    static std::vector<size_t> attr_bytes;
    {
        std::lock_guard<std::mutex> lk(mu);
        if ((int)attr_bytes.size() <= dev) attr_bytes.resize(dev + 1, 0);
        if (lds > 65536 && attr_bytes[dev] < lds) {
            CAF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_perdelay_mr<R0, WGMAX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            163840));
            attr_bytes[dev] = 163840;
        }
    }
    const int wg = (pl.rpw * pl.tpr + 63) / 64 * 64;
    const int64_t groups = (num + pl.rpw - 1) / pl.rpw;
    const int32_t rows_per_wg = (int32_t)std::max<int64_t>(1, std::min<int64_t>(16, groups / 4096));
    const int64_t nwg = (groups + rows_per_wg - 1) / rows_per_wg;
    CAF_REQUIRE(nwg <= 0x7fffffff, "caf_xcorr_perdelay: too many delays for one launch");
    hipLaunchKernelGGL((k_perdelay_mr<R0, WGMAX>), dim3((unsigned)nwg), dim3((unsigned)wg), lds, st, pl, x, y, ylen, tw, prefix, xnorm, start,
                       step, num, rows_per_wg, zero_oor, qf2, fidx, plane, cplane);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

}  // namespace

bool perdelay_mixed_ok(int32_t n) {
    if (n < 32 || n > 16200) return false;
    int32_t r = n;
    for (int p : {2, 3, 5})
        while (r % p == 0) r /= p;
    return r == 1;
}

int launch_perdelay_mixed(const float2* x, int32_t n, const float2* y, int64_t ylen, const double* prefix, const double* xnorm,
                          int64_t start, int64_t step, int64_t num, int32_t zero_oor, float* qf2, uint32_t* fidx, float* plane,
                          float2* cplane, hipStream_t st) {
    std::vector<int> rad;
    if (!perdelay_mixed_ok(n) || !mr_factor(n, rad)) {
        set_error("launch_perdelay_mixed: unsupported length");
        return CAF_ERR_INVALID;
    }
    int dev = 0;
    CAF_HIP_TRY(hipGetDevice(&dev));
    const float2* tw = nullptr;
    int rc = mr_twiddles(dev, n, &tw);
    if (rc) return rc;
    MrPlan pl;
    std::memset(&pl, 0, sizeof(pl));
    pl.n = n;
    pl.tpr = (n + 15) / 16;
    pl.rpw = std::max(1, 256 / pl.tpr);
    pl.npass = (int)rad.size();
    pl.img = n + (n >> 4) + 1;
    int32_t ns = 1;
    for (int p = 0; p < pl.npass; ++p) {
        pl.radix[p] = rad[p];
        pl.ns[p] = ns;
        pl.ns_rcp[p] = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)ns + 1);  // exact for j * ns < 2^32 (the first pass, ns = 1, never divides)
        ns *= rad[p];
    }
    const size_t lds = (size_t)pl.rpw * pl.img * sizeof(float2) + (size_t)2 * pl.rpw * sizeof(unsigned long long);
    CAF_REQUIRE(lds <= 163840, "launch_perdelay_mixed: row image does not fit the LDS");
#define CAF_MR_GO(R)                                                                                                                   \
    return pl.rpw * pl.tpr <= 512                                                                                                       \
               ? mr_launch<R, 512>(pl, lds, dev, x, y, ylen, tw, prefix, xnorm, start, step, num, zero_oor, qf2, fidx, plane, cplane, st) \
               : mr_launch<R, 1024>(pl, lds, dev, x, y, ylen, tw, prefix, xnorm, start, step, num, zero_oor, qf2, fidx, plane, cplane, st)
    switch (rad[0]) {
        case 2: CAF_MR_GO(2);
        case 3: CAF_MR_GO(3);
        case 4: CAF_MR_GO(4);
        case 5: CAF_MR_GO(5);
        case 8: CAF_MR_GO(8);
        case 10: CAF_MR_GO(10);
        default: CAF_MR_GO(16);
    }
#undef CAF_MR_GO
}

}  // namespace caf
