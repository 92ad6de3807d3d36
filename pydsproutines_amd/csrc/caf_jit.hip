// Run-time specialisation of the per-delay correlator (caf_perdelay_jit.h) per cutout length: plan (radices, threads per row),
// LDS layout by bank simulation, hiprtc compilation, code-object cache (memory + disk), launch.
//
// The reference compiles its GPU kernels at run time as well (cupy RawModule over NVRTC, cupyExtensions.py / cupyHelpers.py);
// on ROCm the counterpart is hiprtc, which rocFFT itself uses for the lengths it has no prebuilt kernel for.  hiprtc is loaded
// with dlopen on first use: libcaf.so does not depend on it, and a box without it keeps the plan-driven kernel of
// caf_perdelay_mr.hip (CAF_JIT=0 selects that one too: the A/B switch and the tests' cross-check).
//
//   CAF_JIT=0            never compile at run time
//   CAF_JIT_CACHE=<dir>  where code objects are kept between processes (default ~/.cache/pydsproutines_amd/jit; "off": memory only)
//   CAF_JIT_DEBUG=1      print plan, layout, simulated bank-conflict cycles and compile time per length
//   CAF_PDJ_PLAN="16,15,5/80"   force radices / threads per row (validated), for measurements
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "caf_internal.h"

namespace caf {
namespace {

#include "build/jit_sources.inc"  // JITSRC_<file>: the device headers as string literals (Makefile)

// ---- hiprtc by dlopen ----------------------------------------------------------------------------------------------
typedef struct _hiprtcProgram* rtcProgram;
struct Rtc {
    void* lib = nullptr;
    int (*CreateProgram)(rtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
    int (*CompileProgram)(rtcProgram, int, const char**) = nullptr;
    int (*GetProgramLogSize)(rtcProgram, size_t*) = nullptr;
    int (*GetProgramLog)(rtcProgram, char*) = nullptr;
    int (*GetCodeSize)(rtcProgram, size_t*) = nullptr;
    int (*GetCode)(rtcProgram, char*) = nullptr;
    int (*DestroyProgram)(rtcProgram*) = nullptr;
    int (*Version)(int*, int*) = nullptr;
    std::string why;  // why it is unusable
};
Rtc* rtc() {
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"libhiprtc.so.7", "libhiprtc.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.why = std::string("hiprtc not found: ") + (dlerror() ? dlerror() : "?");
            return;
        }
        auto sym = [&](const char* n) { return dlsym(r.lib, n); };
        r.CreateProgram = (decltype(r.CreateProgram))sym("hiprtcCreateProgram");
        r.CompileProgram = (decltype(r.CompileProgram))sym("hiprtcCompileProgram");
        r.GetProgramLogSize = (decltype(r.GetProgramLogSize))sym("hiprtcGetProgramLogSize");
        r.GetProgramLog = (decltype(r.GetProgramLog))sym("hiprtcGetProgramLog");
        r.GetCodeSize = (decltype(r.GetCodeSize))sym("hiprtcGetCodeSize");
        r.GetCode = (decltype(r.GetCode))sym("hiprtcGetCode");
        r.DestroyProgram = (decltype(r.DestroyProgram))sym("hiprtcDestroyProgram");
        r.Version = (decltype(r.Version))sym("hiprtcVersion");
        if (!r.CreateProgram || !r.CompileProgram || !r.GetCodeSize || !r.GetCode || !r.DestroyProgram) {
            r.why = "hiprtc lacks entry points";
            r.lib = nullptr;
        }
    });
    return &r;
}

// (read on every call: a getenv is nothing beside a launch, and tests switch these per case)
bool jit_enabled() {
    const char* e = std::getenv("CAF_JIT");
    return !e || std::atoi(e) != 0;
}
bool jit_debug() {
    const char* e = std::getenv("CAF_JIT_DEBUG");
    return e && std::atoi(e) != 0;
}

uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) h = (h ^ c) * 1099511628211ull;
    return h;
}

std::string cache_dir() {
    static const std::string dir = [] {
        std::string d;
        if (const char* e = std::getenv("CAF_JIT_CACHE")) {
            d = e;
            if (d == "off" || d == "0") return std::string();
        } else {
            const char* home = std::getenv("XDG_CACHE_HOME");
            if (home && *home)
                d = std::string(home) + "/pydsproutines_amd/jit";
            else if ((home = std::getenv("HOME")) && *home)
                d = std::string(home) + "/.cache/pydsproutines_amd/jit";
            else
                return std::string();
        }
        // mkdir -p
        for (size_t i = 1; i <= d.size(); ++i)
            if (i == d.size() || d[i] == '/') (void)mkdir(d.substr(0, i).c_str(), 0755);
        return access(d.c_str(), W_OK) == 0 ? d : std::string();
    }();
    return dir;
}

struct Loaded {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
};
std::mutex g_jit_mu;
std::map<std::pair<int, uint64_t>, Loaded> g_loaded;  // (device, hash of source + options) -> module

// `src` -> code object for `arch` ("--offload-arch=..."), the device headers above visible under their file names
int jit_compile(const std::string& arch, const std::string& src, const std::vector<std::string>& opts, const char* what,
                std::vector<char>& code) {
    Rtc* r = rtc();
    if (!r->lib) {
        set_error(r->why);
        return CAF_ERR_HIP;
    }
    rtcProgram prog = nullptr;
    const char* hdr_src[] = {JITSRC_caf_fft_dev_h, JITSRC_caf_mr_dev_h, JITSRC_caf_energy_h, JITSRC_caf_perdelay_jit_h};
    const char* hdr_name[] = {"caf_fft_dev.h", "caf_mr_dev.h", "caf_energy.h", "caf_perdelay_jit.h"};
    if (r->CreateProgram(&prog, src.c_str(), "caf_jit.hip", 4, hdr_src, hdr_name) != 0) {
        set_error("hiprtcCreateProgram failed");
        return CAF_ERR_HIP;
    }
    std::vector<const char*> o = {arch.c_str(), "-O3", "-fno-slp-vectorize", "-std=c++17"};
    for (auto& s : opts) o.push_back(s.c_str());
    const int rc = r->CompileProgram(prog, (int)o.size(), o.data());
    if (rc != 0) {
        size_t ls = 0;
        std::string log;
        if (r->GetProgramLogSize && r->GetProgramLogSize(prog, &ls) == 0 && ls > 1) {
            log.resize(ls);
            (void)r->GetProgramLog(prog, &log[0]);
        }
        (void)r->DestroyProgram(&prog);
        set_error(std::string("hiprtc could not compile ") + what + ": " + log.substr(0, 3000));
        return CAF_ERR_HIP;
    }
    size_t cs = 0;
    (void)r->GetCodeSize(prog, &cs);
    code.resize(cs);
    (void)r->GetCode(prog, code.data());
    (void)r->DestroyProgram(&prog);
    return CAF_OK;
}

// an unsigned entry of the code object's metadata (msgpack in the AMDGPU note: the key as a string, then the value), e.g.
// ".vgpr_spill_count"; -1 when absent
long code_object_meta(const std::vector<char>& code, const char* key) {
    const size_t kl = std::strlen(key);
    for (size_t i = 0; i + kl + 1 < code.size(); ++i)
        if (code[i] == key[0] && std::memcmp(&code[i], key, kl) == 0) {
            const unsigned char* v = reinterpret_cast<const unsigned char*>(&code[i + kl]);
            const size_t left = code.size() - (i + kl);
            if (v[0] < 0x80) return v[0];
            if (v[0] == 0xcc && left >= 2) return v[1];
            if (v[0] == 0xcd && left >= 3) return ((long)v[1] << 8) | v[2];
            if (v[0] == 0xce && left >= 5) return ((long)v[1] << 24) | ((long)v[2] << 16) | ((long)v[3] << 8) | v[4];
            return -1;
        }
    return -1;
}

// compile (or fetch) `src` and load it on device `dev`
// max_spills >= 0: a code object whose kernel spills more vector registers than that is not loaded (*out = nullptr, CAF_OK)
int jit_function(int dev, const std::string& src, const std::vector<std::string>& opts, const char* entry, const char* what,
                 hipFunction_t* out, long max_spills = -1) {
    hipDeviceProp_t prop;
    CAF_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    const std::string arch = std::string("--offload-arch=") + prop.gcnArchName;
    int vmaj = 0, vmin = 0;
    Rtc* r = rtc();
    if (!r->lib) {
        set_error(r->why);
        return CAF_ERR_HIP;
    }
    if (r->Version) (void)r->Version(&vmaj, &vmin);
    std::string keystr = arch + "|" + std::to_string(vmaj) + "." + std::to_string(vmin) + "|" + entry;
    for (auto& o : opts) keystr += "|" + o;
    uint64_t h = fnv1a(keystr);
    for (const char* s : {JITSRC_caf_fft_dev_h, JITSRC_caf_mr_dev_h, JITSRC_caf_energy_h, JITSRC_caf_perdelay_jit_h}) h = fnv1a(s, h);
    h = fnv1a(src, h);
    std::lock_guard<std::mutex> lk(g_jit_mu);
    auto it = g_loaded.find({dev, h});
    if (it != g_loaded.end()) {
        *out = it->second.fn;  // (nullptr: a variant rejected for its spills)
        return CAF_OK;
    }
    std::vector<char> code;
    char hex[32];
    std::snprintf(hex, sizeof(hex), "%016llx", (unsigned long long)h);
    const std::string dir = cache_dir();
    const std::string path = dir.empty() ? std::string() : dir + "/" + hex + ".hsaco";
    if (!path.empty()) {
        if (FILE* f = std::fopen(path.c_str(), "rb")) {
            std::fseek(f, 0, SEEK_END);
            const long n = std::ftell(f);
            std::fseek(f, 0, SEEK_SET);
            if (n > 0) {
                code.resize((size_t)n);
                if (std::fread(code.data(), 1, (size_t)n, f) != (size_t)n) code.clear();
            }
            std::fclose(f);
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    bool compiled = false;
    if (code.empty()) {
        const int rc = jit_compile(arch, src, opts, what, code);
        if (rc) return rc;
        compiled = true;
        if (!path.empty() && !code.empty()) {  // (written under a temporary name: another process may be doing the same)
            const std::string tmp = path + "." + std::to_string((long)getpid());
            if (FILE* f = std::fopen(tmp.c_str(), "wb")) {
                const bool ok = std::fwrite(code.data(), 1, code.size(), f) == code.size();
                std::fclose(f);
                if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str());
            }
        }
    }
    if (max_spills >= 0) {
        const long sp = code_object_meta(code, ".vgpr_spill_count");
        if (sp > max_spills) {
            if (jit_debug()) std::fprintf(stderr, "[caf jit] %s: variant spills %ld vector registers, not used\n", what, sp);
            g_loaded[{dev, h}] = Loaded();  // (remembered as rejected)
            *out = nullptr;
            return CAF_OK;
        }
    }
    Loaded l;
    hipError_t e = hipModuleLoadData(&l.mod, code.data());
    if (e != hipSuccess && !compiled && !path.empty()) {  // a stale / truncated cache file: drop it, the next call compiles
        (void)std::remove(path.c_str());
    }
    CAF_HIP_TRY(e);
    CAF_HIP_TRY(hipModuleGetFunction(&l.fn, l.mod, entry));
    if (jit_debug())
        std::fprintf(stderr, "[caf jit] %s: %s in %.0f ms (%zu bytes, key %s)\n", what, compiled ? "compiled" : "loaded from the cache",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), code.size(), hex);
    g_loaded[{dev, h}] = l;
    *out = l.fn;
    return CAF_OK;
}

// ---- the plan of a length -------------------------------------------------------------------------------------------
constexpr int PDJ_MAXP = 5;
constexpr int PDJ_PT = 20;  // points a thread holds in a pass (radix 25 alone: 25)
struct PdjPlan {
    int n = 0, np = 0, tpr = 0, rpw = 0, wg = 0, img = 0, xreg = 0, p0_linear = 0;
    int blu = 0, nx = 0;  // blu: n is the length of Bluestein's circular convolution for a cutout of nx samples (pdj_plan_total)
    int q = 1;  // cutout = q x n samples: q residues of an n-point transform (caf_perdelay_jit.h, PDJ_Q); set by pdj_plan_total
    int rad[PDJ_MAXP] = {1, 1, 1, 1, 1}, str[PDJ_MAXP] = {0, 0, 0, 0, 0};
    int ord[PDJ_MAXP][4];
    long conflict_cycles = 0, lds_ops = 0;  // simulated: extra LDS cycles per workgroup and row group / conflict-free cycles
    double cost = 0.0;
};

constexpr int PDJ_RADICES[] = {25, 23, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2};

bool pdj_valid(int n, const std::vector<int>& rad, int tpr) {
    if (rad.size() < 2 || (int)rad.size() > PDJ_MAXP || tpr < 1 || tpr > 1024) return false;
    int64_t prod = 1;
    if (rad[0] > 16 && !std::getenv("CAF_PDJ_PLAN")) return false;  // (a first pass of 18 / 20 / 25 points spills; forced plans may try)
    for (int r : rad) {
        if (std::find(std::begin(PDJ_RADICES), std::end(PDJ_RADICES), r) == std::end(PDJ_RADICES)) return false;
        const int cnt = (n / r + tpr - 1) / tpr;
        if (cnt * r > std::max(PDJ_PT, r)) return false;
        prod *= r;
    }
    return prod == n;
}

// Modelled cost of a row (arbitrary units): threads the row occupies x ( sum over passes of points per thread x w(position, radix)
// + a charge per butterfly of the thread + a charge per pass ) + a charge per row.  w = a + b log2(radix) + c [radix not a power
// of two], separately for the first pass (global loads, products), the middle passes (image in and out, twiddles) and the last
// one (image in, |.|^2 and maxima): a least-squares fit to 1227 timed plans of 48 lengths with a first radix of at most 16
// (scripts/sweep_pdj_plans.py -> profiles/r05/pdj_plan_sweep*.csv, scripts/fit_pdj_model.py); median error 8.5 %, the model's pick within 10 % of the fastest measured plan for three lengths in four (profiles/r05/pdj_plan_fit.log).  What the fit
// cannot see is kept out by rule (pdj_valid): a first radix above 16 spills (x and y of 20 / 25 points in flight).
double pdj_cost(int n, const std::vector<int>& rad, int tpr) {
    static const double A[3] = {0.8, 18.6, 51.9}, B[3] = {24.3, 6.0, -0.2}, C[3] = {2.5, 4.1, 7.6};
    static const double PASS[3] = {-338.3, 190.2, -338.3}, BFLY[3] = {115.6, 27.3, -20.8};
    const int rpw = std::max(1, 256 / tpr);
    const double threads = (double)((rpw * tpr + 63) / 64 * 64) / rpw;
    double per_thread = 0.0;
    for (size_t i = 0; i < rad.size(); ++i) {
        const int r = rad[i], pos = i == 0 ? 0 : (i + 1 == rad.size() ? 2 : 1);
        const int cnt = (n / r + tpr - 1) / tpr;
        const bool pow2 = (r & (r - 1)) == 0;
        per_thread += (double)cnt * r * (A[pos] + B[pos] * std::log2((double)r) + (pow2 ? 0.0 : C[pos])) + BFLY[pos] * cnt + PASS[pos];
    }
    return threads * per_thread + 64.0 * 98.1;
}

// Plans measured fastest on an MI355X for the lengths of the sweep (the same file): consulted before the model.
struct PdjTuned {
    int n;
    const char* plan;
};
const PdjTuned PDJ_TUNED[] = {
#include "caf_pdj_tuned.inc"
};

// Bank model (MI355X_MICROARCH.md, LDS): ds_write_b64 is served in four groups of 16 consecutive lanes, bank = 8-byte element
// index mod 16; ds_read_b64 in two groups of 32 lanes, bank = element index mod 32; each extra distinct address on a bank
// costs one cycle.  Returns the extra cycles of all LDS accesses of one row group (every wave of the workgroup).
long pdj_conflicts(const PdjPlan& pl, long* base_cycles, int max_waves = 1 << 20) {
    const int P = pl.np, N = pl.n;
    int M[PDJ_MAXP];
    for (int d = 0; d < P; ++d) {
        M[d] = 1;
        for (int q = d + 1; q < P; ++q) M[d] *= pl.rad[q];
    }
    auto base_of = [&](int p, int b) {
        int r = b, base = 0;
        for (int i = 0; i < P - 1; ++i) {
            const int d = pl.ord[p][i];
            const int dig = (i == P - 2) ? r : r % pl.rad[d];
            r /= pl.rad[d];
            base += dig * pl.str[d];
        }
        return base;
    };
    long extra = 0, basec = 0;
    const int nwaves = pl.wg / 64;
    const int wstep = std::max(1, (nwaves + max_waves - 1) / max_waves);  // (the search looks at a sample of the waves: they repeat)
    std::vector<int> pos(64);
    for (int p = 0; p < P; ++p) {
        const int R = pl.rad[p], NB = N / R, cnt = (NB + pl.tpr - 1) / pl.tpr;
        const bool reads = p > 0, writes = p < P - 1;
        for (int c = 0; c < cnt; ++c)
            for (int w = 0; w < nwaves; w += wstep) {
                // positions of t = 0 for the wave's lanes (-1: lane inactive); the other t shift every lane by t STR_p
                for (int lane = 0; lane < 64; ++lane) {
                    const int tid = w * 64 + lane;
                    pos[lane] = -1;
                    if (tid >= pl.rpw * pl.tpr) continue;
                    const int rl = tid / pl.tpr, l = tid % pl.tpr, b = l + c * pl.tpr;
                    if (b >= NB) continue;
                    pos[lane] = rl * pl.img + base_of(p, b);
                }
                auto group_cycles = [&](int lo, int hi, int banks, int shift) {
                    int mult[32] = {0};
                    int seen[64];
                    int ns = 0, mx = 0;
                    for (int lane = lo; lane < hi; ++lane) {
                        if (pos[lane] < 0) continue;
                        const int a = pos[lane] + shift;
                        bool dup = false;
                        for (int k = 0; k < ns; ++k) dup |= seen[k] == a;
                        if (dup) continue;
                        seen[ns++] = a;
                        mx = std::max(mx, ++mult[a % banks]);
                    }
                    return mx;
                };
                for (int t = 0; t < R; ++t) {
                    const int sh = t * pl.str[p];
                    if (reads)
                        for (int g = 0; g < 2; ++g) {
                            const int cyc = group_cycles(32 * g, 32 * g + 32, 32, sh);
                            if (cyc) basec += 1, extra += cyc - 1;
                        }
                    if (writes)
                        for (int g = 0; g < 4; ++g) {
                            const int cyc = group_cycles(16 * g, 16 * g + 16, 16, sh);
                            if (cyc) basec += 1, extra += cyc - 1;
                        }
                }
            }
    }
    if (base_cycles) *base_cycles = basec;
    return extra;
}

void pdj_strides(PdjPlan& pl, const int* pads) {
    pl.str[pl.np - 1] = 1;
    for (int d = pl.np - 2; d >= 0; --d) pl.str[d] = pl.rad[d + 1] * pl.str[d + 1] + pads[d];
}

// layout of a plan: per-dimension pads, lane orders of the middle passes and the row-image pitch, by coordinate descent on the
// simulated conflict cycles (a few thousand evaluations of a few thousand accesses: milliseconds, once per length and process)
void pdj_layout(PdjPlan& pl) {
    const int P = pl.np;
    int pads[PDJ_MAXP] = {0, 0, 0, 0, 0};
    for (int p = 0; p < PDJ_MAXP; ++p)
        for (int i = 0; i < 4; ++i) pl.ord[p][i] = -1;
    for (int p = 0; p < P; ++p) {
        int k = 0;
        if (p == 0) {
            for (int d = P - 1; d >= 1; --d) pl.ord[p][k++] = d;  // n_{P-1} fastest: butterfly index == m_0 (coalesced loads)
        } else {
            for (int d = 0; d < P; ++d)
                if (d != p) pl.ord[p][k++] = d;  // k_0 fastest (the last pass must: natural spectrum order)
        }
    }
    int img_pad = 0;
    auto apply = [&]() {
        pdj_strides(pl, pads);
        pl.img = pl.rad[0] * pl.str[0] + img_pad;
    };
    constexpr long TOO_BIG = 1L << 40;  // (a layout whose row images do not fit the LDS beside the key slots)
    auto eval = [&]() {
        apply();
        if ((size_t)pl.rpw * pl.img * 8 + 16 * pl.rpw + 64 > 160 * 1024) return TOO_BIG;
        return pdj_conflicts(pl, nullptr, 4);
    };
    long best = eval();
    // stage 1: the first dimension's pad x the lane orders of the middle passes, jointly (what one pass wants of STR_0 depends
    // on the order it walks its butterflies in); the image may grow by a quarter at most
    if (best > 0) {
        const int nat0 = pl.n / pl.rad[0];
        const int max_pad0 = std::min(32, nat0 / 4 + 8);
        std::vector<std::vector<std::vector<int>>> cands(P);  // per middle pass: candidate orders
        for (int p = 1; p + 1 < P; ++p) {
            std::vector<int> dims;
            for (int d = 0; d < P; ++d)
                if (d != p) dims.push_back(d);
            if (P <= 4) {
                std::vector<int> perm = dims;
                do cands[p].push_back(perm);
                while (std::next_permutation(perm.begin(), perm.end()));
            } else {  // five passes: the transformed dimensions first / the untransformed ones first, each in natural order
                cands[p].push_back(dims);
                std::vector<int> alt;
                for (int d = P - 1; d > p; --d) alt.push_back(d);
                for (int d = 0; d < p; ++d) alt.push_back(d);
                cands[p].push_back(alt);
            }
        }
        std::vector<int> pick(P, 0), arg_pick(P, 0);
        int arg_pad0 = 0;
        std::function<void(int)> rec = [&](int p) {
            if (p + 1 >= P) {
                for (int v = 0; v <= max_pad0; ++v) {
                    pads[0] = v;
                    const long c = eval();
                    if (c < best) best = c, arg_pad0 = v, arg_pick = pick;
                }
                return;
            }
            for (size_t k = 0; k < cands[p].size(); ++k) {
                pick[p] = (int)k;
                for (int i = 0; i < P - 1; ++i) pl.ord[p][i] = cands[p][k][i];
                rec(p + 1);
            }
        };
        const long before = best;
        std::vector<std::vector<int>> ord0(P);
        for (int p = 1; p + 1 < P; ++p) ord0[p].assign(pl.ord[p], pl.ord[p] + (P - 1));
        rec(1);
        pads[0] = best < before ? arg_pad0 : 0;
        for (int p = 1; p + 1 < P; ++p)
            for (int i = 0; i < P - 1; ++i) pl.ord[p][i] = best < before ? cands[p][arg_pick[p]][i] : ord0[p][i];
    }
    // stage 2: the inner pads and the phase of the row images, one at a time
    for (int sweep = 0; sweep < 2 && best > 0; ++sweep) {
        bool improved = false;
        for (int d = 1; d + 1 < P; ++d) {
            int arg = pads[d];
            for (int v = 0; v <= 8; ++v) {
                pads[d] = v;
                const long c = eval();
                if (c < best) best = c, arg = v, improved = true;
            }
            pads[d] = arg;
        }
        if (pl.rpw > 1) {  // rows of one wave start in different bank phases
            int arg = img_pad;
            for (int v = 0; v < 32; ++v) {
                img_pad = v;
                const long c = eval();
                if (c < best) best = c, arg = v, improved = true;
            }
            img_pad = arg;
        }
        if (!improved) break;
    }
    apply();
    pl.conflict_cycles = pdj_conflicts(pl, &pl.lds_ops);
    pl.p0_linear = 1;
    for (int d = 0; d + 2 < P; ++d)
        if (pads[d + 1]) pl.p0_linear = 0;  // (pads of the dimensions after the first break position == butterfly index)
}

// "pdjplan <version> n np tpr rpw wg img p0_linear conflict base | rad.. | str.. | ord.." in <cache>/plan_v<V>_<n>.txt; the version
// changes with the planner (model constants, layout search), so that a new library does not pick up an old library's choice
constexpr int PDJ_PLAN_FILE_VERSION = 5;
std::string plan_file_path(int n) {
    const std::string dir = cache_dir();
    return dir.empty() ? std::string() : dir + "/plan_v" + std::to_string(PDJ_PLAN_FILE_VERSION) + "_" + std::to_string(n) + ".txt";
}
void plan_file_store(const PdjPlan& pl) {
    const std::string path = plan_file_path(pl.n);
    if (path.empty()) return;
    const std::string tmp = path + "." + std::to_string((long)getpid());
    FILE* f = std::fopen(tmp.c_str(), "w");
    if (!f) return;
    std::fprintf(f, "pdjplan %d %d %d %d %d %d %d %d %ld %ld\n", PDJ_PLAN_FILE_VERSION, pl.n, pl.np, pl.tpr, pl.rpw, pl.wg, pl.img, pl.p0_linear,
                 pl.conflict_cycles, pl.lds_ops);
    for (int p = 0; p < PDJ_MAXP; ++p) std::fprintf(f, "%d %d %d %d %d %d\n", pl.rad[p], pl.str[p], pl.ord[p][0], pl.ord[p][1], pl.ord[p][2], pl.ord[p][3]);
    const bool ok = std::fclose(f) == 0;
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str());
}
bool plan_file_load(int n, PdjPlan& pl) {
    const std::string path = plan_file_path(n);
    if (path.empty()) return false;
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    int ver = 0;
    PdjPlan q;
    bool ok = std::fscanf(f, "pdjplan %d %d %d %d %d %d %d %d %ld %ld", &ver, &q.n, &q.np, &q.tpr, &q.rpw, &q.wg, &q.img, &q.p0_linear, &q.conflict_cycles,
                          &q.lds_ops) == 10;
    for (int p = 0; ok && p < PDJ_MAXP; ++p)
        ok = std::fscanf(f, "%d %d %d %d %d %d", &q.rad[p], &q.str[p], &q.ord[p][0], &q.ord[p][1], &q.ord[p][2], &q.ord[p][3]) == 6;
    std::fclose(f);
    // (a file is only trusted as far as it can be checked: the radices multiply to n, the images fit the LDS)
    if (!ok || ver != PDJ_PLAN_FILE_VERSION || q.n != n || q.np < 2 || q.np > PDJ_MAXP || q.tpr < 1 || q.tpr > 1024 || q.rpw < 1) return false;
    int64_t prod = 1;
    std::vector<int> rad;
    for (int p = 0; p < q.np; ++p) prod *= q.rad[p], rad.push_back(q.rad[p]);
    if (prod != n || q.wg != (q.rpw * q.tpr + 63) / 64 * 64 || q.wg > 1024 || (size_t)q.rpw * q.img * 8 + 16 * q.rpw + 64 > 160 * 1024) return false;
    for (int p = 0; p < q.np; ++p)
        if (std::find(std::begin(PDJ_RADICES), std::end(PDJ_RADICES), q.rad[p]) == std::end(PDJ_RADICES) ||
            ((n / q.rad[p] + q.tpr - 1) / q.tpr) * q.rad[p] > std::max(PDJ_PT, q.rad[p]))
            return false;
    // strides must address disjoint positions inside the image: recompute the natural nesting and compare
    if (q.str[q.np - 1] != 1) return false;
    for (int d = q.np - 2; d >= 0; --d)
        if (q.str[d] < q.rad[d + 1] * q.str[d + 1]) return false;
    if (q.img < q.rad[0] * q.str[0]) return false;
    for (int p = 0; p < q.np; ++p) {  // every lane order a permutation of the other dimensions
        int seen = 0;
        for (int i = 0; i < q.np - 1; ++i) {
            const int d = q.ord[p][i];
            if (d < 0 || d >= q.np || d == p || (seen >> d & 1)) return false;
            seen |= 1 << d;
        }
    }
    if (q.np >= 2) {  // pass 0 must walk its butterflies in natural order (coalesced loads), the last pass in natural spectrum order
        for (int i = 0; i < q.np - 1; ++i)
            if (q.ord[0][i] != q.np - 1 - i || q.ord[q.np - 1][i] != i) return false;
    }
    q.cost = pdj_cost(n, rad, q.tpr);
    q.xreg = 0;
    pl = q;
    return true;
}

// every valid (radix order, threads per row) of a length with its modelled cost
template <class F>
void pdj_enumerate(int n, F&& emit) {
    std::vector<int> cur;
    std::function<void(int)> rec = [&](int rem) {
        if (rem == 1) {
            if (cur.size() < 2) return;
            int cap = PDJ_PT;
            for (int r : cur) cap = std::min(cap, std::max(PDJ_PT, r) / r * r);
            const int t0 = (n + cap - 1) / cap;
            // (rows of 32 threads and more in whole quarter / half / full waves: 250 or 500 threads per row measured 15-20 %
            //  slower than 256 / 512 with the same radices)
            for (int t : {t0 < 32 ? t0 : 0, (t0 + 15) / 16 * 16, (t0 + 31) / 32 * 32, (t0 + 63) / 64 * 64})
                if (t && pdj_valid(n, cur, t)) emit(cur, t, pdj_cost(n, cur, t));
            return;
        }
        if ((int)cur.size() >= PDJ_MAXP) return;
        for (int r : PDJ_RADICES) {  // (every ORDER of the radices is a plan of its own: the passes cost differently by position)
            if (rem % r) continue;
            cur.push_back(r);
            rec(rem / r);
            cur.pop_back();
        }
    };
    rec(n);
}

bool pdj_plan(int n, PdjPlan& out) {
    static std::mutex mu;
    static std::map<std::pair<int, std::string>, PdjPlan> memo;  // (length, forced plan or "") -> plan + layout
    const char* forced_env = std::getenv("CAF_PDJ_PLAN");
    const std::pair<int, std::string> memo_key(n, forced_env ? forced_env : "");
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = memo.find(memo_key);
        if (it != memo.end()) {
            out = it->second;
            return out.tpr > 0;
        }
    }
    // plan + layout of earlier processes (the search is ~0.1-1 s for a length outside the tuned table: kept beside the code objects)
    if (!forced_env) {
        PdjPlan q;
        if (plan_file_load(n, q)) {
            out = q;
            std::lock_guard<std::mutex> lk(mu);
            memo[memo_key] = q;
            return true;
        }
    }
    std::vector<int> best_rad;
    int best_tpr = 0;
    double best_cost = 0.0;
    bool forced = false;
    // "r0,r1,../threads per row" with an optional "xR" = rows per workgroup (default: what fills 256 threads)
    auto parse = [&](const char* txt, std::vector<int>& rad, int& tpr, int& rpw) {
        rad.clear();
        const char* q = txt;
        while (*q && *q != '/') {
            rad.push_back((int)std::strtol(q, const_cast<char**>(&q), 10));
            if (*q == ',') ++q;
        }
        tpr = *q == '/' ? (int)std::strtol(q + 1, const_cast<char**>(&q), 10) : 0;
        rpw = *q == 'x' ? (int)std::strtol(q + 1, nullptr, 10) : 0;
        if (rpw < 0 || (int64_t)rpw * tpr > 1024) rpw = 0;
    };
    int best_rpw = 0;
    if (forced_env) {
        std::vector<int> rad;
        int tpr = 0, rpw = 0;
        parse(forced_env, rad, tpr, rpw);
        if (pdj_valid(n, rad, tpr)) best_rad = rad, best_tpr = tpr, best_rpw = rpw, best_cost = pdj_cost(n, rad, tpr), forced = true;
    }
    if (!forced && !std::getenv("CAF_PDJ_MODEL_ONLY"))
        for (const PdjTuned& t : PDJ_TUNED)
            if (t.n == n) {
                std::vector<int> rad;
                int tpr = 0, rpw = 0;
                parse(t.plan, rad, tpr, rpw);
                if (pdj_valid(n, rad, tpr)) best_rad = rad, best_tpr = tpr, best_rpw = rpw, best_cost = pdj_cost(n, rad, tpr), forced = true;
            }
    struct Cand {
        std::vector<int> rad;
        int tpr;
        double cost;
        int rpw;  // 0: the default
    };
    std::vector<Cand> cands;
    if (forced) {
        cands.push_back({best_rad, best_tpr, best_cost, best_rpw});
    } else {
        pdj_enumerate(n, [&](const std::vector<int>& rad, int t, double cost) { cands.push_back({rad, t, cost, 0}); });
        std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.cost < b.cost; });
    }
    // Among the plans the model prices within 4 % of the cheapest (four at most), the one whose layout leaves the fewest
    // simulated bank-conflict cycles per conflict-free one: conflicts hardly show in the time of these VALU-bound kernels
    // (the fit gives a conflict cycle 1/7 of the weight of a conflict-free one), so the tie is broken in favour of the LDS.
    auto build = [&](const Cand& c, PdjPlan& pl) {
        pl = PdjPlan();
        pl.n = n;
        pl.np = (int)c.rad.size();
        for (int p = 0; p < pl.np; ++p) pl.rad[p] = c.rad[p];
        pl.tpr = c.tpr;
        pl.rpw = c.rpw ? c.rpw : std::max(1, 256 / pl.tpr);
        pl.wg = (pl.rpw * pl.tpr + 63) / 64 * 64;
        pl.cost = c.cost;
        pdj_layout(pl);
        // the row images must fit the LDS beside the key slots; fewer rows per workgroup if they do not
        while (pl.rpw > 1 && (size_t)pl.rpw * pl.img * 8 + 16 * pl.rpw + 64 > 160 * 1024) {
            --pl.rpw;
            pl.wg = (pl.rpw * pl.tpr + 63) / 64 * 64;
            pdj_layout(pl);
        }
        return (size_t)pl.rpw * pl.img * 8 + 16 * pl.rpw + 64 <= 160 * 1024;
    };
    PdjPlan pl;
    double best_ratio = 0.0;
    int tried = 0;
    for (const Cand& c : cands) {
        if (tried >= 4 || (tried && c.cost > 1.04 * cands.front().cost)) break;
        PdjPlan q;
        if (!build(c, q)) continue;
        const double ratio = q.lds_ops ? (double)q.conflict_cycles / (double)q.lds_ops : 0.0;
        if (!tried || ratio < best_ratio - 1e-9) pl = q, best_ratio = ratio;
        ++tried;
    }
    out = pl;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (memo.size() > 4096) memo.clear();
        memo[memo_key] = pl;
    }
    if (!forced_env && pl.tpr > 0) plan_file_store(pl);
    return pl.tpr > 0;
}

// Plan of a cutout of nt samples: the transform itself where one LDS image holds it, otherwise q residues of an (nt / q)-point
// transform, 2 <= q <= PDJ_QMAX.  Every residue re-reads the cutout and the window from the L2 -- 16 q bytes per sample, delivered
// at ~90 GB/s per CU, which is what bounds the split form -- so its lead over the rows path's trip through HBM shrinks with q:
// 32768 samples (q = 2) 2.8 x, 40000 3.6 x, 65536 (4) 2.2 x, 100000 (5) 3.3 x, 131072 (8) 1.9 x, 262144 (16) 1.2 x, 320000 (16)
// 1.06 x (profiles/r05/timing_perdelay_long.log).  CAF_PDJ_Q=q forces a split (the tests run small lengths through it).
constexpr int PDJ_QMAX = 16;
constexpr int PDJ_NMAX = 20000;  // (160 KB of LDS hold a row image of ~20000 points; the plan decides)
//
// Any other length of at most (PDJ_NMAX + 1) / 2 samples -- a prime factor above 23, which is most lengths a burst happens to
// have -- runs as Bluestein's chirp transform in the same image: n k = (n^2 + k^2 - (k - n)^2) / 2 turns the nx-point transform
// into a circular convolution of any length m >= 2 nx - 1, i.e. a forward m-point transform, a product with the transformed chirp
// and the transform back (caf_perdelay_jit.h, PDJ_BLU).  m is the smooth length in [2 nx - 1, 4 nx] whose cheapest plan the cost
// model prices lowest.  CAF_PDJ_BLUESTEIN=1 sends smooth lengths that way too, CAF_PDJ_BLU_M=m fixes m (tests).
bool pdj_plan_total(int nt, PdjPlan& out) {
    int qforced = 0;
    if (const char* e = std::getenv("CAF_PDJ_Q")) qforced = std::atoi(e);
    const char* be = std::getenv("CAF_PDJ_BLUESTEIN");
    const bool blu_forced = be && std::atoi(be) == 1, blu_off = be && std::atoi(be) == 0;
    if (!blu_forced)
        for (int q = qforced > 0 ? qforced : 1; q <= (qforced > 0 ? qforced : PDJ_QMAX); ++q) {
            if (nt % q || nt / q > PDJ_NMAX || nt / q < 32) continue;
            if (pdj_plan(nt / q, out)) {
                out.q = q;
                return true;
            }
        }
    if (blu_off || qforced > 0 || nt < 16 || 2 * (int64_t)nt - 1 > PDJ_NMAX) return false;
    static std::mutex mu;
    static std::map<std::pair<int, int>, std::vector<int>> memo;  // (cutout length, forced m) -> convolution lengths, cheapest first
    std::vector<int> ms;
    const char* me = std::getenv("CAF_PDJ_BLU_M");
    const std::pair<int, int> memo_key(nt, me ? std::atoi(me) : 0);
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = memo.find(memo_key);
        if (it != memo.end()) ms = it->second;
    }
    if (ms.empty()) {
        if (me) {
            if (std::atoi(me) >= 2 * nt - 1) ms.push_back(std::atoi(me));
        } else {
            std::vector<std::pair<double, int>> priced;
            for (int m = std::max(32, 2 * nt - 1); m <= std::min(PDJ_NMAX, 4 * nt); ++m) {
                int r = m;
                for (int p : {2, 3, 5, 7})
                    while (r % p == 0) r /= p;
                if (r != 1) continue;
                double best = 0.0;
                pdj_enumerate(m, [&](const std::vector<int>&, int, double cost) { best = best == 0.0 || cost < best ? cost : best; });
                if (best > 0.0) priced.push_back({best, m});
            }
            std::sort(priced.begin(), priced.end());
            for (size_t i = 0; i < priced.size() && i < 4; ++i) ms.push_back(priced[i].second);
        }
        std::lock_guard<std::mutex> lk(mu);
        if (memo.size() > 4096) memo.clear();
        memo[memo_key] = ms;
    }
    for (int m : ms)
        if (pdj_plan(m, out)) {
            out.blu = 1, out.nx = nt;
            return true;
        }
    return false;
}

// the chirp c[i] = e^{+j pi i^2 / nx}, i < nx, and the transformed, scaled chirp of the convolution, bhat[k] = (1 / m) sum_i b[i]
// W_m^{i k} with b[i] = b[m - i] = conj(c[i]) for i < nx and zero between -- in float64 on the host (a mixed-radix transform over
// the plan's own radices), rounded to float32 once
int pdj_bluestein_tables(const PdjPlan& pl, float2** chirp_out, float2** bhat_out) {
    using cd = std::complex<double>;
    const int nx = pl.nx, m = pl.n;
    std::vector<std::complex<float>> chirp(nx), bhat(m);
    std::vector<cd> b(m, cd(0.0, 0.0)), wm(m);
    for (int i = 0; i < m; ++i) wm[i] = std::polar(1.0, 2.0 * M_PI * (double)i / (double)m);
    for (int i = 0; i < nx; ++i) {
        const double ph = M_PI * (double)(((int64_t)i * i) % (2 * (int64_t)nx)) / (double)nx;
        const cd c = std::polar(1.0, ph);
        chirp[i] = std::complex<float>((float)c.real(), (float)c.imag());
        b[i] = std::conj(c);
        if (i) b[m - i] = std::conj(c);
    }
    std::vector<cd> out(m), tmp(32);
    std::function<void(const cd*, cd*, int, int, int)> rec = [&](const cd* in, cd* o, int n, int stride, int level) {
        if (n == 1) {
            o[0] = in[0];
            return;
        }
        const int r = pl.rad[level], sub = n / r;
        for (int q = 0; q < r; ++q) rec(in + (int64_t)q * stride, o + (int64_t)q * sub, sub, stride * r, level + 1);
        std::vector<cd> t(r), u(r);
        for (int k = 0; k < sub; ++k) {
            for (int q = 0; q < r; ++q) t[q] = o[(int64_t)q * sub + k] * wm[(int64_t)q * k * (m / n) % m];
            for (int j = 0; j < r; ++j) {
                cd acc(0.0, 0.0);
                for (int q = 0; q < r; ++q) acc += t[q] * wm[(int64_t)q * j * (m / r) % m];
                u[j] = acc;
            }
            for (int j = 0; j < r; ++j) o[k + (int64_t)sub * j] = u[j];
        }
    };
    rec(b.data(), out.data(), m, 1, 0);
    for (int k = 0; k < m; ++k) bhat[k] = std::complex<float>((float)(out[k].real() / m), (float)(out[k].imag() / m));
    float2 *dc = nullptr, *db = nullptr;
    int rc = pool_alloc((void**)&dc, (int64_t)nx * 8);
    if (rc) return rc;
    if ((rc = pool_alloc((void**)&db, (int64_t)m * 8))) {
        (void)pool_free(dc);
        return rc;
    }
    if ((rc = host_h2d(dc, chirp.data(), (int64_t)nx * 8, nullptr)) || (rc = host_h2d(db, bhat.data(), (int64_t)m * 8, nullptr))) {
        (void)pool_free(dc);
        (void)pool_free(db);
        return rc;
    }
    *chirp_out = dc, *bhat_out = db;
    return CAF_OK;
}

// e^{+j 2 pi q / n}, q < n: one table per loaded kernel (owned by its entry in launch_perdelay_jit's cache)
int pdj_twiddles(int32_t n, float2** out) {
    std::vector<std::complex<float>> t(n);
    for (int q = 0; q < n; ++q) {
        const double ph = 2.0 * M_PI * (double)q / (double)n;
        t[q] = std::complex<float>((float)std::cos(ph), (float)std::sin(ph));
    }
    float2* d = nullptr;
    int rc = pool_alloc((void**)&d, (int64_t)n * 8);
    if (rc) return rc;
    rc = host_h2d(d, t.data(), (int64_t)n * 8, nullptr);
    if (rc) {
        (void)pool_free(d);
        return rc;
    }
    *out = d;
    return CAF_OK;
}

std::string brace_list(const int* v, int n) {
    std::string s = "{";
    for (int i = 0; i < n; ++i) s += (i ? "," : "") + std::to_string(v[i]);
    return s + "}";
}

std::vector<std::string> pdj_options(const PdjPlan& pl) {
    std::vector<std::string> opts = {
        "-DPDJ_N=" + std::to_string(pl.n),       "-DPDJ_NP=" + std::to_string(pl.np),   "-DPDJ_TPR=" + std::to_string(pl.tpr),
        "-DPDJ_RPW=" + std::to_string(pl.rpw),   "-DPDJ_WG=" + std::to_string(pl.wg),   "-DPDJ_IMG=" + std::to_string(pl.img),
        "-DPDJ_XREG=" + std::to_string(pl.xreg), "-DPDJ_P0_LINEAR=" + std::to_string(pl.p0_linear), "-DPDJ_Q=" + std::to_string(pl.q),
        "-DPDJ_BLU=" + std::to_string(pl.blu),   "-DPDJ_NX=" + std::to_string(pl.blu ? pl.nx : pl.n * pl.q)};
    for (int p = 0; p < PDJ_MAXP; ++p) {
        opts.push_back("-DPDJ_R" + std::to_string(p) + "=" + std::to_string(pl.rad[p]));
        opts.push_back("-DPDJ_S" + std::to_string(p) + "=" + std::to_string(pl.str[p]));
        opts.push_back("-DPDJ_ORD" + std::to_string(p) + "=" + brace_list(pl.ord[p], 4));
    }
    return opts;
}

}  // namespace

namespace {
std::mutex g_failed_mu;
std::map<int, bool> g_failed;  // lengths whose compilation failed once are not tried again
}  // namespace
void perdelay_jit_failed(int32_t n) {
    std::lock_guard<std::mutex> lk(g_failed_mu);
    g_failed[n] = true;
}
bool perdelay_jit_ok(int32_t n) {
    if (!jit_enabled() || n < 32 || n > PDJ_NMAX * PDJ_QMAX) return false;
    if (!rtc()->lib) return false;
    {
        std::lock_guard<std::mutex> lk(g_failed_mu);
        if (g_failed.count(n)) return false;
    }
    PdjPlan pl;
    return pdj_plan_total(n, pl);
}

int launch_perdelay_jit(const float2* x, int32_t n, const float2* y, int64_t ylen, const double* prefix, const double* xnorm,
                        int64_t start, int64_t step, int64_t num, int32_t zero_oor, float* qf2, uint32_t* fidx, float* plane,
                        float2* cplane, hipStream_t st) {
    // what a call needs -- plan, loaded function, twiddle table -- is looked up once per (device, length, forced plan): a call costs
    // one map lookup (the first version hashed the kernel sources and asked for the device properties on EVERY call: ~0.1 ms)
    struct Ready {
        PdjPlan pl;
        hipFunction_t fn = nullptr;
        float2* tw = nullptr;  // owned: freed when the entry is evicted
        float2* twq = nullptr;  // split form: e^{+j 2 pi i / (q n)}; Bluestein: the chirp; owned
        float2* bhat = nullptr;  // Bluestein: the transformed chirp, owned
        uint64_t used = 0;     // last call (for the eviction of the least recently used entry)
    };
    static std::mutex mu;
    static std::map<std::tuple<int, int, std::string>, Ready> ready;
    static uint64_t tick = 0;
    constexpr size_t READY_MAX = 256;  // lengths x forced plans kept per process (a table of n x 8 B each)
    int dev = 0;
    CAF_HIP_TRY(hipGetDevice(&dev));
    const char* forced_env = std::getenv("CAF_PDJ_PLAN");
    std::string variant = forced_env ? forced_env : "";
    for (const char* name : {"CAF_PDJ_Q", "CAF_PDJ_BLUESTEIN", "CAF_PDJ_BLU_M"})
        if (const char* e = std::getenv(name)) variant += std::string("|") + name + "=" + e;
    const auto key = std::make_tuple(dev, (int)n, variant);
    Ready r;
    bool have = false;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = ready.find(key);
        if (it != ready.end()) {
            it->second.used = ++tick;
            r = it->second, have = true;
        }
    }
    if (!have) {
        if (!pdj_plan_total(n, r.pl)) {
            set_error("launch_perdelay_jit: unsupported length");
            return CAF_ERR_INVALID;
        }
        const std::string what = "the per-delay correlator for " + std::to_string(n) + "-sample cutouts";
        // the cutout held in registers across a workgroup's rows where that costs no spills (half of the first pass' loads),
        // re-read per row from the L1 / L2 otherwise: decided by compiling the first variant and reading its spill count
        int rc = CAF_OK;
        const char* xe = std::getenv("CAF_PDJ_XREG");
        if (r.pl.q == 1 && !r.pl.blu && (!xe || std::atoi(xe))) {
            r.pl.xreg = 1;
            rc = jit_function(dev, "#include \"caf_perdelay_jit.h\"\n", pdj_options(r.pl), "k_pdj", what.c_str(), &r.fn, xe ? 1L << 30 : 0);
            if (rc) return rc;
        }
        if (!r.fn) {
            r.pl.xreg = 0;
            rc = jit_function(dev, "#include \"caf_perdelay_jit.h\"\n", pdj_options(r.pl), "k_pdj", what.c_str(), &r.fn);
            if (rc) return rc;
        }
        if ((rc = pdj_twiddles(r.pl.n, &r.tw))) return rc;
        if (r.pl.q > 1 && (rc = pdj_twiddles(n, &r.twq))) {
            (void)pool_free(r.tw);
            return rc;
        }
        if (r.pl.blu && (rc = pdj_bluestein_tables(r.pl, &r.twq, &r.bhat))) {
            (void)pool_free(r.tw);
            return rc;
        }
        std::lock_guard<std::mutex> lk(mu);
        if (ready.size() >= READY_MAX) {
            // the least recently used entry goes, with its twiddle table -- once nothing on the device can still be reading it
            // (the first version kept the tables in a cache of their own that dropped the oldest of 64 while the entries here
            // kept pointing at them: garbage twiddles after 64 lengths in one process)
            auto old = ready.begin();
            for (auto it = ready.begin(); it != ready.end(); ++it)
                if (it->second.used < old->second.used) old = it;
            CAF_HIP_TRY(hipDeviceSynchronize());
            (void)pool_free(old->second.tw);
            if (old->second.twq) (void)pool_free(old->second.twq);
            if (old->second.bhat) (void)pool_free(old->second.bhat);
            ready.erase(old);
        }
        r.used = ++tick;
        ready[key] = r;
    }
    const PdjPlan& pl = r.pl;
    if (jit_debug()) {
        std::string t, sd;
        for (int p = 0; p < pl.np; ++p) t += (p ? "," : "") + std::to_string(pl.rad[p]), sd += (p ? "," : "") + std::to_string(pl.str[p]);
        std::fprintf(stderr, "[caf jit] n=%d plan=%s/%d residues=%d bluestein=%d rows_per_workgroup=%d strides=%s image=%d cost=%.0f | simulated LDS: %ld conflict "
                     "cycles on %ld conflict-free ones (%.1f %%)\n", n, t.c_str(), pl.tpr, pl.q, pl.blu ? pl.n : 0, pl.rpw, sd.c_str(), pl.img, pl.cost,
                     pl.conflict_cycles, pl.lds_ops, pl.lds_ops ? 100.0 * pl.conflict_cycles / pl.lds_ops : 0.0);
    }
    const float2* tw = r.tw;
    const float2* twq = r.twq;
    const float2* bhat = r.bhat;
    const int64_t groups = (num + pl.rpw - 1) / pl.rpw;
    int32_t rows_per_wg = (int32_t)std::max<int64_t>(1, std::min<int64_t>(16, groups / 4096));
    const int64_t nwg = (groups + rows_per_wg - 1) / rows_per_wg;
    CAF_REQUIRE(nwg <= 0x7fffffff, "caf_xcorr_perdelay: too many delays for one launch");
    int32_t zo = zero_oor;
    void* args[] = {(void*)&x,     (void*)&y,    (void*)&ylen, (void*)&tw,          (void*)&twq, (void*)&bhat, (void*)&prefix, (void*)&xnorm, (void*)&start,
                    (void*)&step,  (void*)&num,  (void*)&rows_per_wg, (void*)&zo,   (void*)&qf2, (void*)&fidx,   (void*)&plane, (void*)&cplane};
    CAF_HIP_TRY(hipModuleLaunchKernel(r.fn, (unsigned)nwg, 1, 1, (unsigned)pl.wg, 1, 1, 0, st, args, nullptr));
    return CAF_OK;
}

// plan + layout of a length as text (the planner runs without a GPU); "" when there is none.  With `arch` the kernel is also
// compiled for that architecture (not loaded), and with `dump_path` its code object written there (llvm-objdump -d reads it).
int perdelay_jit_describe(int32_t n, const char* arch, const char* dump_path, std::string* text) {
    PdjPlan pl;
    text->clear();
    if (!pdj_plan_total(n, pl)) return CAF_OK;
    char buf[640];
    std::string t, s, o;
    for (int p = 0; p < pl.np; ++p) {
        t += (p ? "," : "") + std::to_string(pl.rad[p]), s += (p ? "," : "") + std::to_string(pl.str[p]);
        o += (p ? " " : "") + brace_list(pl.ord[p], pl.np - 1);
    }
    std::snprintf(buf, sizeof(buf), "n=%d q=%d bluestein=%d radices=%s tpr=%d rpw=%d wg=%d strides=%s orders=%s img=%d lds_bytes=%d xreg=%d conflict_cycles=%ld base_cycles=%ld",
                  n, pl.q, pl.blu ? pl.n : 0, t.c_str(), pl.tpr, pl.rpw, pl.wg, s.c_str(), o.c_str(), pl.img, pl.rpw * pl.img * 8, pl.xreg, pl.conflict_cycles, pl.lds_ops);
    *text = buf;
    if (arch && *arch) {
        std::vector<char> code;
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = jit_compile(std::string("--offload-arch=") + arch, "#include \"caf_perdelay_jit.h\"\n", pdj_options(pl), "the per-delay correlator",
                                   code);
        if (rc) return rc;
        std::snprintf(buf, sizeof(buf), " code_bytes=%zu compile_ms=%.0f", code.size(),
                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        *text += buf;
        if (dump_path && *dump_path) {
            FILE* f = std::fopen(dump_path, "wb");
            if (!f || std::fwrite(code.data(), 1, code.size(), f) != code.size()) {
                if (f) std::fclose(f);
                set_error(std::string("cannot write ") + dump_path);
                return CAF_ERR_INVALID;
            }
            std::fclose(f);
        }
    }
    return CAF_OK;
}

}  // namespace caf
