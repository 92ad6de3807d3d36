// Per-delay correlator for cutouts of N = 2^a 3^b 5^c 7^d samples (32 <= N <= 16200, N neither a power of two nor of ten:
// those have kernels of their own in caf_perdelay.hip): rx window x conj(cutout) -> N-point FFT in LDS -> |.|^2 -> (max, first
// argmax) per delay, in ONE kernel -- the reference's literal per-delay algorithm (fastXcorr branches B / C,
// xcorrRoutines.py:511-566; cp_fastXcorr :29-167, whose cutout length is a free argument, benchmark_xcorrs.py:62-71;
// IppXcorrFFT.cpp:94-178) without the (rows, N) product matrix that the product kernel -> rocFFT rows -> argmax chain
// moves through HBM four times.
//
// Mixed-radix Stockham autosort transform, radices from {2 .. 10, 12, 14, 15, 16, 18, 20} (the composite ones as
// Cooley-Tukey butterflies in registers: 1200 = 15 x 10 x 8 is three exchanges where 16 x 5 x 5 x 3 was four), chosen on
// the host together with the threads per row by a cost model (mr_plan: butterfly points per thread + a charge per pass,
// times the threads a row occupies).  The lengths are too many to instantiate one kernel each, so the passes after the first are driven by
// a small plan in the kernel arguments: a uniform switch picks the butterfly, the trip counts are compile-time bounds with
// a guard (a thread owns up to MR_PT = 20 points: at most floor(20 / R) butterflies of radix R, ceil(N / R / tpr) of them
// used), divisions by the pass stride are multiplications by a host-prepared reciprocal.  Only the FIRST radix is a
// template parameter: that pass is the one with the global loads.  One LDS image per row (padded by one element per
// sixteen), inputs read before a barrier and outputs written after it; tpr threads per row (N / 16 ... N / 12), several
// rows per workgroup for short cutouts.  Window energies from the caller's float64 prefix and ||cutout|| from launch_cutout_norm, exactly as the
// three-kernel form and the radix-10 kernel normalise; same out-of-range and zero-energy rules (include/caf.h).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "caf_internal.h"
#include "caf_energy.h"
#include "caf_mr_dev.h"

namespace caf {
namespace {

constexpr int MR_MAXP = 8;  // passes (2^14 as radix-2 passes would need 14: the planner never produces more than 7)
constexpr int MR_PT = 20;   // points per thread and pass

struct MrPlan {
    int32_t n, tpr, rpw, npass, img;      // length, threads per row, rows per workgroup, passes, padded row image (elements)
    int32_t radix[MR_MAXP], ns[MR_MAXP];  // pass p: radix, product of the radices before it
    int32_t cnt[MR_MAXP];                 // butterflies per thread of pass p: ceil(n / radix / tpr) <= floor(MR_PT / radix)
    uint32_t ns_rcp[MR_MAXP];             // floor(2^32 / ns) + 1: j / ns == umulhi(j, ns_rcp) for j < 2^16
};

__device__ __forceinline__ int mr_pad(int a) { return a + (a >> 4); }

// One Stockham pass (any but the first) of radix R with ns = the product of the earlier radices: image -> registers, barrier, twiddles,
// butterflies, registers -> image, barrier.  Nothing is live across a pass but the image: each of the seven bodies behind the
// uniform switch gets a register allocation of its own (with the data registers carried from pass to pass through the
// switch, and the last pass' outputs into a per-radix epilogue, every body spilled 10 .. 20 registers at 128).
// A thread's butterflies past the end (j >= N / R: floor(20 / R) tpr may exceed N / R) are CLAMPED to the last one instead
// of being branched around: they recompute it and store the same values to the same places.
// last_reg (the last pass of a row whose planes are not wanted): the outputs are not written back; |z|^2 and the thread's
// (maximum, first index) come straight from the registers -- register (c, t) <-> spectrum index j + t N / R, visited t-major,
// i.e. ascending (a clamped butterfly offers the last one's value at the last one's index again: the strict comparison
// ignores it) -- and the row saves a round trip through the image and two barriers.
template <int R>
__device__ __forceinline__ void mr_pass(const MrPlan& pl, int p, float2* __restrict__ buf, const float2* __restrict__ tw, int l_in,
                                        bool active, float2 (&v)[MR_PT], bool last_reg, float inv, float& bv, uint32_t& bi) {
    constexpr int CNT = MR_PT / R;
    const int nb = pl.n / R, ns = pl.ns[p], tpr = pl.tpr;
    const int cnt = pl.cnt[p];  // butterflies per thread that the pass needs (uniform, <= CNT): ceil(nb / tpr)
    // (an opaque copy of the lane's index: the image addresses of a radix depend on nothing that changes from row to row,
    //  and hoisted out of the row loop -- twenty per radix, seven radices -- they were spilled: ~330 registers of scratch)
    int l = l_in;
    asm volatile("" : "+v"(l));
    // the twiddle bases W_{R ns}^k = W_N^{k N / (R ns)} of all of the thread's butterflies are fetched FIRST: they do not
    // depend on the image, and their L1 / L2 latency then runs under the image reads and the barrier
    const int tws = nb / ns;
    float2 w1[CNT];
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        if (c >= cnt) continue;
        const int j = min(l + c * tpr, nb - 1);
        const int k = j - ns * (int)__umulhi((uint32_t)j, pl.ns_rcp[p]);
        w1[c] = tw[k * tws];
    }
    if (active) {
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
            if (c >= cnt) continue;
            const int j = min(l + c * tpr, nb - 1);
#pragma unroll
            for (int t = 0; t < R; ++t) v[c * R + t] = buf[mr_pad(j + t * nb)];
        }
    }
    __syncthreads();  // every butterfly has its inputs: the image may be overwritten
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        if (c >= cnt) continue;
        float2 pw = w1[c];
        v[c * R + 1] = cmul(v[c * R + 1], pw);
#pragma unroll
        for (int t = 2; t < R; ++t) {
            pw = cmul(pw, w1[c]);
            v[c * R + t] = cmul(v[c * R + t], pw);
        }
    }
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        if (c >= cnt) continue;
        mr_idft<R>(&v[c * R]);
        if (CNT > 1) __builtin_amdgcn_sched_barrier(0);  // (one butterfly's temporaries at a time)
    }
    if (last_reg) {
#pragma unroll
        for (int t = 0; t < R; ++t)
#pragma unroll
            for (int c = 0; c < CNT; ++c) {
                if (c >= cnt) continue;
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
            if (c >= cnt) continue;
            const int j = min(l + c * tpr, nb - 1);
            const int k = j - ns * (int)__umulhi((uint32_t)j, pl.ns_rcp[p]);
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
            // 1 / (sqrt(E) ||x||) as rsq(E ||x||^2) + one Newton step (2^-45 or better before the rounding to float32), as in
            // k_perdelay_fused: a float64 square root and a float64 division per row and thread are ~40 half-rate
            // instructions.  E = 0: rsq = inf, 0 * inf = NaN -> NaN row, as before.
            const double en = window_energy(prefix, y, ylen, a, b) * (xn * xn);  // (exact where the difference is not: caf_energy.h)
            const double y0 = __builtin_amdgcn_rsq(en);
            inv = (float)__builtin_fma(__builtin_fma(-(en * y0), 0.5 * y0, 0.5), y0, y0);
        }
        // rows whose planes are not wanted finish in the registers of the last pass (mr_pass, last_reg); with planes the
        // finished spectrum goes through the image once more and leaves in contiguous runs
        const bool reg_tail = !plane && !cplane;
        float bv = -1.f;
        uint32_t bi = 0;
        {
            // First pass, one butterfly at a time from the loads to the image: product with the cutout (re-read per row -- it
            // stays in the L1 / L2 --), butterfly, R0 stores.  Only R0 points are live at once (all MR_PT of them, their x and y
            // in flight beside them, spilled up to 50 registers in the radix-10 and radix-20 instances).  A cutout has more than
            // MR_PT samples, so this is never the last pass.  Butterflies past the end are clamped, see mr_pass.
            int lx = l;  // (opaque: otherwise the cutout loads are hoisted out of the row loop and held -- and spilled -- after all)
            asm volatile("" : "+v"(lx));
            const int cnt0 = pl.cnt[0];
            const bool inside = !zero && !oor;
#pragma unroll
            for (int c = 0; c < CNT0; ++c) {
                if (c >= cnt0) continue;
                const int j = min(lx + c * tpr, nb0 - 1);
                float2 v[R0];
                if (inside) {
                    const float2* yrow = y + s;
#pragma unroll
                    for (int t = 0; t < R0; ++t) {
                        const float2 a = x[j + t * nb0], b = yrow[j + t * nb0];
                        v[t] = make_float2(a.x * b.x - a.y * b.y, -(a.x * b.y + a.y * b.x));  // conj(x y): see k_perdelay_fused
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < R0; ++t) {
                        const int64_t g = s + j + t * nb0;
                        const float2 a = x[j + t * nb0];
                        const float2 b = (!zero && g >= 0 && g < ylen) ? y[g] : make_float2(0.f, 0.f);
                        v[t] = make_float2(a.x * b.x - a.y * b.y, -(a.x * b.y + a.y * b.x));
                    }
                }
                mr_idft<R0>(v);
                if (active) {
#pragma unroll
                    for (int t = 0; t < R0; ++t) buf[mr_pad(j * R0 + t)] = v[t];
                }
                if (CNT0 > 1) __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
        for (int p = 1; p < pl.npass; ++p) {
            float2 v[MR_PT];
            const bool lastr = reg_tail && p + 1 == pl.npass;
            switch (pl.radix[p]) {  // (uniform)
                case 2: mr_pass<2>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 3: mr_pass<3>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 4: mr_pass<4>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 5: mr_pass<5>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 6: mr_pass<6>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 7: mr_pass<7>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 8: mr_pass<8>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 9: mr_pass<9>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 10: mr_pass<10>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 15: mr_pass<15>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                case 20: mr_pass<20>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
                default: mr_pass<16>(pl, p, buf, tw, l, active, v, lastr, inv, bv, bi); break;
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

// The plan of a length: radices (non-increasing; the first from MR_FIRST -- the kernel instances --, the others from
// MR_LATER) and threads per row, by the smallest modelled cost per row:
//   threads a row occupies x ( sum over passes of butterflies per thread x points x weight(radix)  +  MR_PASS_COST per pass )
// The weights are a least-squares fit to the timings of 807 plans of 19 lengths (scripts/sweep_mr_plans.py,
// scripts/fit_mr_model.py, profiles/r04/mr_plan_fit.log: the model's pick is within 0.5 % of the fastest plan on average, 3 % at
// worst): a point of a later pass costs 22 .. 40 whatever the radix (the twiddle product, the LDS write and read, the index
// arithmetic outweigh the butterfly), a point of the first pass -- no twiddles, no image reads -- 7 .. 36, a pass 232 (two
// barriers and the exchange latency), and rows that do not fill whole waves (tpr not a multiple of 64) 1.4 more per point.
// A thread holds at most MR_PT points of a pass, so tpr >= N / (floor(20 / R) R) for every radix R of the plan, and >= N / 16
// (the natural-order epilogue of rows with planes takes 16 points per thread).
// CAF_MR_PLAN="15,10,8/80" overrides the search (radices / threads per row; checked for validity) -- for measurements.
constexpr int MR_FIRST[] = {20, 18, 16, 15, 14, 12, 10, 9, 8, 7, 5};
constexpr int MR_LATER[] = {20, 16, 15, 10, 9, 8, 7, 6, 5, 4, 3, 2};
constexpr double MR_PASS_COST = 232.2, MR_UNALIGNED = 1.4;
static double mr_weight(int r, bool first) {
    if (first) {
        switch (r) {
            case 8: return 13.1;
            case 9: return 14.8;
            case 10: return 19.4;
            case 12: return 6.7;
            case 14: return 6.0;  // (the fit says less than nothing: too few plans to tell it from the pass charge)
            case 15: return 28.0;
            case 16: return 16.9;
            case 18: return 20.4;
            case 20: return 35.8;
            default: return 20.0;  // (5 and 7 first: short lengths only, not in the fit)
        }
    }
    switch (r) {
        case 2: return 29.9;
        case 3: return 27.7;
        case 4: return 30.8;
        case 5: return 28.6;
        case 6: return 28.6;
        case 8: return 31.8;
        case 9: return 22.2;
        case 10: return 31.1;
        case 15: return 27.6;
        case 16: return 40.0;
        case 20: return 30.1;
        default: return 28.0;  // (7: too few plans in the fit to tell it from its neighbours)
    }
}

struct MrChoice {
    std::vector<int> rad;
    int tpr = 0;
    double cost = 0.0;
};

static bool mr_valid(int32_t n, const std::vector<int>& rad, int tpr) {
    if (rad.empty() || (int)rad.size() > MR_MAXP || tpr < 1 || tpr > 1024 || (int64_t)tpr * 16 < n) return false;
    int64_t prod = 1;
    for (size_t i = 0; i < rad.size(); ++i) {
        const int r = rad[i];
        const int* lo = i ? MR_LATER : MR_FIRST;
        const int* hi = i ? MR_LATER + sizeof(MR_LATER) / sizeof(int) : MR_FIRST + sizeof(MR_FIRST) / sizeof(int);
        if (std::find(lo, hi, r) == hi || (i && r > rad[i - 1])) return false;
        if (((n / r + tpr - 1) / tpr) > MR_PT / r) return false;
        prod *= r;
    }
    return prod == n;
}

static double mr_cost(int32_t n, const std::vector<int>& rad, int tpr) {
    const int rpw = std::max(1, 256 / tpr);
    const double threads = (double)((rpw * tpr + 63) / 64 * 64) / rpw;
    double per_thread = 0.0;
    for (size_t i = 0; i < rad.size(); ++i) {
        const int r = rad[i];
        per_thread += (double)((n / r + tpr - 1) / tpr) * r * (mr_weight(r, i == 0) + (tpr % 64 ? MR_UNALIGNED : 0.0)) + MR_PASS_COST;
    }
    return threads * per_thread;
}

bool mr_plan(int32_t n, MrChoice& best) {
    if (const char* ov = std::getenv("CAF_MR_PLAN")) {
        MrChoice c;
        const char* q = ov;
        while (*q && *q != '/') {
            c.rad.push_back((int)std::strtol(q, const_cast<char**>(&q), 10));
            if (*q == ',') ++q;
        }
        if (*q == '/') c.tpr = (int)std::strtol(q + 1, nullptr, 10);
        if (mr_valid(n, c.rad, c.tpr)) {
            c.cost = mr_cost(n, c.rad, c.tpr);
            best = c;
            return true;
        }
    }
    // (the search costs 10 .. 50 us: remembered per length -- a call of a few hundred microseconds asks twice)
    static std::mutex mu;
    static std::unordered_map<int32_t, MrChoice> memo;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = memo.find(n);
        if (it != memo.end()) {
            best = it->second;
            return best.tpr > 0;
        }
    }
    bool found = false;
    std::vector<int> cur;
    std::function<void(int32_t, int)> rec = [&](int32_t rem, int max_r) {
        if (rem == 1) {
            int cap = 16;
            for (int r : cur) cap = std::min(cap, MR_PT / r * r);
            const int t0 = (n + cap - 1) / cap;
            // (the fewest threads, or rows rounded up to quarter, half and whole waves)
            for (int t : {t0, (t0 + 15) / 16 * 16, (t0 + 31) / 32 * 32, (t0 + 63) / 64 * 64}) {
                if (!mr_valid(n, cur, t)) continue;
                const double c = mr_cost(n, cur, t);
                if (!found || c < best.cost) best.rad = cur, best.tpr = t, best.cost = c, found = true;
            }
            return;
        }
        if ((int)cur.size() >= MR_MAXP) return;
        const int* lo = cur.empty() ? MR_FIRST : MR_LATER;
        const int cnt = cur.empty() ? (int)(sizeof(MR_FIRST) / sizeof(int)) : (int)(sizeof(MR_LATER) / sizeof(int));
        for (int i = 0; i < cnt; ++i) {
            const int r = lo[i];
            if (r > max_r || rem % r) continue;
            cur.push_back(r);
            rec(rem / r, r);
            cur.pop_back();
        }
    };
    rec(n, 20);
    if (!found) best = MrChoice();
    {
        std::lock_guard<std::mutex> lk(mu);
        memo[n] = best;  // (tpr == 0: no plan)
    }
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
    for (int p : {2, 3, 5, 7})
        while (r % p == 0) r /= p;
    if (r != 1) return false;
    MrChoice c;
    return mr_plan(n, c);
}

int launch_perdelay_mixed(const float2* x, int32_t n, const float2* y, int64_t ylen, const double* prefix, const double* xnorm,
                          int64_t start, int64_t step, int64_t num, int32_t zero_oor, float* qf2, uint32_t* fidx, float* plane,
                          float2* cplane, hipStream_t st) {
    MrChoice ch;
    if (n < 32 || n > 16200 || !mr_plan(n, ch)) {
        set_error("launch_perdelay_mixed: unsupported length");
        return CAF_ERR_INVALID;
    }
    const std::vector<int>& rad = ch.rad;
    int dev = 0;
    CAF_HIP_TRY(hipGetDevice(&dev));
    const float2* tw = nullptr;
    int rc = mr_twiddles(dev, n, &tw);
    if (rc) return rc;
    MrPlan pl;
    std::memset(&pl, 0, sizeof(pl));
    pl.n = n;
    pl.tpr = ch.tpr;
    pl.rpw = std::max(1, 256 / pl.tpr);
    pl.npass = (int)rad.size();
    pl.img = n + (n >> 4) + 1;
    int32_t ns = 1;
    for (int p = 0; p < pl.npass; ++p) {
        pl.radix[p] = rad[p];
        pl.ns[p] = ns;
        pl.ns_rcp[p] = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)ns + 1);  // exact for j * ns < 2^32 (the first pass, ns = 1, never divides)
        pl.cnt[p] = (n / rad[p] + pl.tpr - 1) / pl.tpr;
        ns *= rad[p];
    }
    if (std::getenv("CAF_MR_DEBUG")) {
        std::string t;
        for (int r : rad) t += (t.empty() ? "" : ",") + std::to_string(r);
        std::fprintf(stderr, "[caf mr] n=%d plan=%s/%d rows_per_workgroup=%d cost=%.0f\n", n, t.c_str(), pl.tpr, pl.rpw, ch.cost);
    }
    const size_t lds = (size_t)pl.rpw * pl.img * sizeof(float2) + (size_t)2 * pl.rpw * sizeof(unsigned long long);
    CAF_REQUIRE(lds <= 163840, "launch_perdelay_mixed: row image does not fit the LDS");
#define CAF_MR_GO(R)                                                                                                                   \
    return pl.rpw * pl.tpr <= 512                                                                                                       \
               ? mr_launch<R, 512>(pl, lds, dev, x, y, ylen, tw, prefix, xnorm, start, step, num, zero_oor, qf2, fidx, plane, cplane, st) \
               : mr_launch<R, 1024>(pl, lds, dev, x, y, ylen, tw, prefix, xnorm, start, step, num, zero_oor, qf2, fidx, plane, cplane, st)
    switch (rad[0]) {
        case 5: CAF_MR_GO(5);
        case 7: CAF_MR_GO(7);
        case 8: CAF_MR_GO(8);
        case 9: CAF_MR_GO(9);
        case 10: CAF_MR_GO(10);
        case 12: CAF_MR_GO(12);
        case 14: CAF_MR_GO(14);
        case 15: CAF_MR_GO(15);
        case 18: CAF_MR_GO(18);
        case 20: CAF_MR_GO(20);
        default: CAF_MR_GO(16);
    }
#undef CAF_MR_GO
}

}  // namespace caf
