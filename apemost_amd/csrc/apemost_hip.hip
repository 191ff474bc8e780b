// apemost_hip.hip -- kernels and C ABI of the gfx950 parallel-tempering engine
// (declared in include/apemost_hip.h).  Written for MI355X only.
#include "pt_kernels.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <hip/hiprtc.h>
#include <dlfcn.h>

#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace apemost;

// ===========================================================================
// kernels that are not templates over the model (the others: pt_kernels.h)
// ===========================================================================
// ---- test hooks ----
__global__ void rng_raw_kernel(u64 seed, u64 subseq, u64 offset, int n, unsigned int *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, subseq, offset, &st);
    for (int i = 0; i < n; i++)
        out[i] = rocrand(&st);
}

__global__ void rng_attempts_kernel(u64 seed, u64 chain, int slot, u64 tick, u64 q0, int n, double *y,
                                    double *s, int *valid, double *log_u) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double yy, ss;
        valid[i] = gaussian_attempt(seed, chain, slot, tick, q0 + (u64)i, yy, ss) ? 1 : 0;
        y[i] = yy;
        s[i] = ss;
    }
    if (i == 0)
        *log_u = accept_log_uniform(seed, chain, slot, tick);
}

// edge records for sharded ladders: beta, prob, prob_best, params[n], params_best[n]
// adapt() of -DADAPT (src/parallel_tempering.c:282-301), called by the reference once per round
// between the n_swap steps and tempering_interaction (:404): one thread per chain.  The counters
// summed over the parameters (src/mcmc_gettersetter.c:25-41), accepts / REJECTS against the
// target, all step widths of the chain scaled by 0.99 or by the double 1 / 0.99
// (gsl_vector_scale), counters restarted past 100000 counted updates (reset_accept_rejects).
// A kernel of its own rather than a tail of the round kernels: those carry no register for it.
__global__ void pt_adapt_kernel(DevArrays d, double target) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.n)
        return;
    const int n = d.np;
    u64 *pacc = d.params_accepts() + (size_t)c * n, *prej = d.params_rejects() + (size_t)c * n;
    double *step = d.step() + (size_t)c * n;
    u64 acc = 0, rej = 0;
    for (int p = 0; p < n; p++) {
        acc += pacc[p];
        rej += prej[p];
    }
    if (acc + rej < 20000)
        return;
    const double ratio = (double)acc * 1.0 / (double)rej;
    if (ratio < target - 0.05) {
        for (int p = 0; p < n; p++)
            step[p] *= 0.99;
    } else if (ratio > target + 0.05) {
        for (int p = 0; p < n; p++)
            step[p] *= 1 / 0.99;
    }
    if (acc + rej > 100000) {
        for (int p = 0; p < n; p++)
            pacc[p] = prej[p] = 0;
        d.accept()[c] = d.reject()[c] = 0;
    }
}

// adapt() of -DRWM (src/parallel_tempering.c:268-281 + rmw_adapt_stepwidth, src/markov_chain.c:342-367) around
// ONE more step of the ordinary round kernel: `pre` keeps what that step must not be seen to have changed (the
// log-posterior before it, the best point), `post` puts the best point and n_iter back and moves the step widths.
// keep: [n][2 + n_par] = prob, prob_best, params_best.  One thread per chain; `half` = the half of the
// double-buffered state the chain currently lives in.
__global__ void pt_rwm_pre_kernel(DevArrays d, int half, double *keep) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.n)
        return;
    const int n = d.np, row = c + 1;
    double *k = keep + (size_t)c * (2 + n);
    k[0] = d.prob(half)[row];
    k[1] = d.prob_best(half)[row];
    for (int p = 0; p < n; p++)
        k[2 + p] = d.params_best(half)[(size_t)row * n + p];
}

__global__ void pt_rwm_post_kernel(DevArrays d, ChainShape sh, int half, const double *keep, double target) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.n)
        return;
    const int n = d.np, row = c + 1;
    const double *k = keep + (size_t)c * (2 + n);
    d.prob_best(half)[row] = k[1];
    for (int p = 0; p < n; p++)
        d.params_best(half)[(size_t)row * n + p] = k[2 + p];
    const u64 n_iter = d.n_iter()[c] - 1; // (no mcmc_check behind the extra step)
    d.n_iter()[c] = n_iter;
    double alpha = exp(d.prob(half)[row] - k[0]);
    if (alpha > 1)
        alpha = 1;
    const u64 tick = d.ticks()[c] - 1; // the extra step's
    const u64 g = (u64)(sh.chain_offset + c);
    for (int p = 0; p < n; p++) {
        const size_t i = (size_t)c * n + p;
        const double scale = d.pmax()[i] - d.pmin()[i];
        const double lo = 0.0000001 * scale, hi = 1000000 * scale;
        const uint4 b = philox_block(sh.seed, g * APEMOST_HIP_STREAMS_PER_CHAIN + (u64)n, (tick << kTickShift) | (u64)(1 + p / 4));
        const unsigned int w = (p & 3) == 0 ? b.x : (p & 3) == 1 ? b.y : (p & 3) == 2 ? b.z : b.w;
        double step = d.step()[i];
        step += u32_to_uniform(w) / sqrt((double)n_iter) * (alpha - target) * scale;
        if (step < lo)
            step = lo;
        if (step > hi)
            step = hi;
        d.step()[i] = step;
    }
}

// Sample rows [n_steps][n_chains][n_par+2] -> what a sink writes, for the kept steps skip, skip + thin, ...
// layout 0: the record of the C host's binary sink (apemost_amd/host/src/parallel_tempering.c): the
//   parameter vectors of chains 0..n_param_chains-1, then (prob, prob - prior) of every chain;
// layout 1: the rows themselves, thinned.
// One thread per output double: coalesced writes, gathered reads; a few KB per step against HBM.
__global__ void samples_pack_kernel(const double *rows, int n_chains, int n_par, unsigned long long n_kept,
                                    unsigned long long skip, unsigned long long thin, int n_param_chains, int layout,
                                    double *out) {
    const size_t row = (size_t)n_chains * (n_par + 2);
    const size_t record = layout == 0 ? (size_t)n_param_chains * n_par + 2 * (size_t)n_chains : row;
    const size_t total = (size_t)n_kept * record;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t k = i / record, r = i - k * record;
        const double *src = rows + (skip + k * thin) * row;
        size_t from = r;
        if (layout == 0) {
            const size_t head = (size_t)n_param_chains * n_par;
            if (r < head) {
                const size_t chain = r / n_par;
                from = chain * (n_par + 2) + (r - chain * n_par);
            } else {
                const size_t q = r - head;
                from = (q >> 1) * (n_par + 2) + n_par + (q & 1);
            }
        }
        out[i] = src[from];
    }
}

__global__ void edge_export_kernel(DevArrays d, int n_par, int cur, int row, double *buf) {
    const int t = threadIdx.x;
    if (t == 0) {
        buf[0] = d.beta()[row];
        buf[1] = d.prob(cur)[row];
        buf[2] = d.prob_best(cur)[row];
    }
    if (t < n_par) {
        buf[3 + t] = d.params(cur)[(size_t)row * n_par + t];
        buf[3 + n_par + t] = d.params_best(cur)[(size_t)row * n_par + t];
    }
}

__global__ void edge_import_kernel(DevArrays d, int n_par, int cur, int row, const double *buf) {
    const int t = threadIdx.x;
    if (t == 0) {
        d.beta()[row] = buf[0];
        d.prob(cur)[row] = buf[1];
        d.prob_best(cur)[row] = buf[2];
    }
    if (t < n_par) {
        d.params(cur)[(size_t)row * n_par + t] = buf[3 + t];
        d.params_best(cur)[(size_t)row * n_par + t] = buf[3 + n_par + t];
    }
}

// ===========================================================================
// host side of the C ABI
// ===========================================================================

// ---- run-time model -> the translation unit that holds its kernels (pt_kernels.h) ----
#define APEMOST_EXTERN_MODEL(M) extern template hipError_t apemost::model_dispatch<M>(int, const AnyOp &);
APEMOST_EXTERN_MODEL(0)
APEMOST_EXTERN_MODEL(1)
APEMOST_EXTERN_MODEL(2)
APEMOST_EXTERN_MODEL(3)
APEMOST_EXTERN_MODEL(8)
APEMOST_EXTERN_MODEL(9)
APEMOST_EXTERN_MODEL(10)
APEMOST_EXTERN_MODEL(11)

static hipError_t dispatch_any(int model, int waves, const AnyOp &op) {
    switch (model) {
    case 0:
        return model_dispatch<0>(waves, op);
    case 1:
        return model_dispatch<1>(waves, op);
    case 2:
        return model_dispatch<2>(waves, op);
    case 3:
        return model_dispatch<3>(waves, op);
    case 8:
        return model_dispatch<8>(waves, op);
    case 9:
        return model_dispatch<9>(waves, op);
    case 10:
        return model_dispatch<10>(waves, op);
    case 11:
        return model_dispatch<11>(waves, op);
    }
    return hipErrorInvalidDeviceFunction; // (APEMOST_MODEL_USER never comes here: its kernels are a run-time module)
}
static hipError_t dispatch(int model, int waves, const LaunchOp &f) {
    AnyOp op;
    op.what = AnyOp::LAUNCH;
    op.launch = f;
    return dispatch_any(model, waves, op);
}
static hipError_t dispatch(int model, int waves, const LdsAttrOp &f) {
    AnyOp op;
    op.what = AnyOp::LDS_ATTR;
    op.lds = f;
    return dispatch_any(model, waves, op);
}
static hipError_t dispatch(int model, int waves, const OccupancyOp<true> &f) {
    AnyOp op;
    op.what = AnyOp::OCCUPANCY_LDS;
    op.occ_lds = f;
    return dispatch_any(model, waves, op);
}
static hipError_t dispatch(int model, int waves, const OccupancyOp<false> &f) {
    AnyOp op;
    op.what = AnyOp::OCCUPANCY_PLAIN;
    op.occ_plain = f;
    return dispatch_any(model, waves, op);
}

static thread_local std::string g_last_error;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t err__ = (expr);                                                               \
        if (err__ != hipSuccess)                                                                 \
            return fail(APEMOST_HIP_ERR_RUNTIME, "%s failed: %s (%s:%d)", #expr,                 \
                        hipGetErrorString(err__), __FILE__, __LINE__);                           \
    } while (0)

struct apemost_hip_sampler {
    apemost_hip_config cfg;
    DevArrays d;
    ChainShape sh;
    int waves;
    bool producers; // candidate-producer wavefronts in the round / calibrate kernels
    bool lds_data;
    size_t lds_bytes, lds_fixed_bytes;
    bool resident_ok; // the whole grid of the round kernel fits the device at once ...
    bool resident_lds, resident_plain; // ... with / without the data vector staged in LDS
    int cur;
    u64 round;
    int swap_pending;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    u64 launches, launches_at_begin;
    std::vector<void *> allocations;
    // a calibration in progress (calibrate_begin .. calibrate_end): launched in segments
    struct {
        bool open;      // between begin and end
        bool in_flight; // a segment has been launched and its records not collected yet
        bool cancelled;
        int first, count, burn_in_only, capacity, progress_slot, progress_cap;
        apemost_hip_calib_config cfg;
        CalibRec *d_rec;
        double *d_orig, *d_progress;
        int *d_list;
        CalibRec *h_rec; // pinned [capacity]
        int *h_list;     // pinned [capacity]: slots still calibrating
        int n_active;
        hipEvent_t ev;
        u64 segments, launches_by_waves[9];
        double seconds_by_waves[9]; // wall time of the segments of each workgroup shape (launch to collection)
        std::chrono::steady_clock::time_point segment_start;
        int segment_waves;
    } cal;
    unsigned big_lds_set; // bit w: the LDS opt-in of the w-wave kernels has been made
    // APEMOST_MODEL_USER: the kernels of the user's likelihood, compiled by hiprtc at create time
    // (indexed by the waves per chain of the two-phase kernels: 1, 2, 4, 8 -- a user likelihood is an
    // arbitrary function of the data sum, so the one-barrier kernels' threshold form is not for it)
    struct {
        hipModule_t module;
        hipFunction_t round[9], calibrate[9], calc_model[9], loglike[9];
    } user;
    double *edge_out, *edge_in;  // edge records for in-process shard exchanges (created on first use)
    hipEvent_t ev_exported, ev_imported;
    hipStream_t copy_stream; // drains sample rows while the next launch runs (created on first use)
    hipEvent_t ev_copy;
    u64 *h_word;             // pinned copy of the launch error word, refreshed by every async read
    bool one_barrier;    // stepping launches use pt_round_ob_kernel
    bool ob_helper;      // ... in its form with a helper wavefront (see ob_wants_helper)
    int cus;             // compute units of the device
    int kmodel;          // template argument of this sampler's kernels: cfg.model, + kVariantModel when a
                         // non-default proposal law or swap schedule is asked for (pt_device.h)
    bool cooperative;    // multi-round launches through hipLaunchCooperativeKernel
    bool handoff_failed; // an in-launch hand-off timed out once: single-round launches from then on
    double user_compile_seconds; // hiprtc's time for this sampler's user model (0: taken from the process's cache)
    double *rwm_keep;            // APEMOST_HIP_FLAG_RWM: what the extra step of a round must not be seen to change
    bool in_rwm;
};

extern "C" const char *apemost_hip_last_error(void) { return g_last_error.c_str(); }
extern "C" int apemost_hip_abi_version(void) { return APEMOST_HIP_ABI_VERSION; }

extern "C" int apemost_hip_device_count(int *count) {
    if (!count)
        return fail(APEMOST_HIP_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess || n <= 0) {
        *count = 0;
        return fail(APEMOST_HIP_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(err));
    }
    *count = n;
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_device_info(int device, char *name, size_t name_len, int *compute_units,
                                       uint64_t *hbm_bytes) {
    int n = 0;
    int rc = apemost_hip_device_count(&n);
    if (rc != APEMOST_HIP_OK)
        return rc;
    if (device < 0 || device >= n)
        return fail(APEMOST_HIP_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name && name_len) {
        strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (compute_units)
        *compute_units = prop.multiProcessorCount;
    if (hbm_bytes)
        *hbm_bytes = prop.totalGlobalMem;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(APEMOST_HIP_ERR_NO_DEVICE, "device %d is %s; this engine is built for gfx950 only",
                    device, prop.gcnArchName);
    return APEMOST_HIP_OK;
}

static int enable_big_lds(apemost_hip_sampler *s, int waves);
static int max_rounds_per_launch(apemost_hip_sampler *s);
static int user_model_build(apemost_hip_sampler *s);
template <bool LDS>
static hipError_t round_occupancy(int model, int waves, bool producers, bool one_barrier, bool helper, size_t lds_bytes, int *blocks);
static size_t ob_lds_bytes(const apemost_hip_sampler *s, bool lds_data);
static bool ob_wants_helper(const apemost_hip_sampler *s, int n_chains);

template <class T>
static int dev_alloc(apemost_hip_sampler *s, T **p, size_t count) {
    void *q = nullptr;
    HIP_TRY(hipMalloc(&q, (count ? count : 1) * sizeof(T)));
    s->allocations.push_back(q); // owned from here on, whatever fails next
    HIP_TRY(hipMemsetAsync(q, 0, (count ? count : 1) * sizeof(T), s->stream));
    *p = (T *)q;
    return APEMOST_HIP_OK;
}

// Staging pays when the staged bytes are re-read (several steps per launch) and the LDS
// footprint still lets enough workgroups share a CU; otherwise the data vector is read through
// L2 (it is shared by every chain, so it stays resident there).
static bool choose_lds(const apemost_hip_config &c, size_t lds_bytes) {
    const long long wg_per_cu = (160 * 1024) / (long long)lds_bytes;
    const long long wanted = ((long long)c.n_chains + 255) / 256;
    return wg_per_cu >= (wanted < 4 ? wanted : 4);
}

static int choose_waves(const apemost_hip_config &c) {
    if (c.waves_per_chain > 0)
        return c.waves_per_chain;
    // Measured on one MI355X (simplesin, 1024 points, steps/s by waves 1/2/4/8):
    //    128 chains  -   /  -   / 1.23 / 1.33 e8      512 chains  2.07 / 2.10 / 2.59 / 1.97 e8
    //    256 chains  1.09 / 1.46 / 2.22 / 1.82 e8    1024 chains  3.98 / 3.29 / 2.84 / 2.07 e8
    // (sine3 1024 x 8192: 3.5e7 with 1 wave, 3.1e7 with 2; pulse_vrot 2048 x 65536: 4.2 / 3.9 / 4.0 /
    // 3.5 e6.)  A chain's step is a latency chain: while CUs are idle more waves per chain shorten
    // it; once every SIMD has a wave, a chain per wave without barriers does more.
    // Since the one-barrier kernel (see has_one_barrier): four likelihood waves per chain up to 512
    // chains -- two waves per SIMD with the owner and the producers, no likelihood wave waits for a
    // sibling on its SIMD -- and eight only where a step is long enough to be bound by issue rather
    // than by the chain of dependent operations (>= 8192 points at <= 128 chains).
    // The pulse likelihood with more than three modes -- a loop over the modes that reads its parameters
    // from LDS as it goes, a longer chain per point than the others -- is the exception: eight waves up
    // to 256 chains (256 x 1024: 1.51 vs 1.41e8, 128 x 1024: 9.0 vs 8.2e7 in round 2; pulse_vrot and
    // sine3 stay with four: 8.9 vs 8.4e7 and 1.05 vs 1.03e8 at 128 x 1024; tools/experiments/gpu_exp_w8.sh).  Up
    // to three modes the spectrum is taken over a common denominator since round 3 (16 instead of 46
    // instructions per point) and four waves do more: 64 / 128 / 256 / 512 x 1024 5.14 / 11.2 / 23.6 /
    // 35.5e7 against 5.01 / 10.9 / 21.5 / 4.8e7 with eight (profiles/r03_pulse_waves.txt).
    // Two waves per chain only while such a ladder is still resident (three or four two-wave
    // workgroups per CU: up to 768 chains; 700 x 1024: 2.55 vs 2.46e8); beyond that a launch would
    // hold one round and one wave per chain does more (900 x 1024: 3.11 vs 1.68e8).
    int by_chip = (c.n_chains <= 128 && c.n_data >= 8192) ? 8 : c.n_chains <= 512 ? 4 : c.n_chains <= 768 ? 2 : 1;
    if (c.model == APEMOST_MODEL_PULSE && c.n_chains <= 256 && c.n_par > 8)
        by_chip = 8;
    // never fewer than 2 data points per lane
    int by_data = 1;
    while (by_data < 8 && c.n_data >= by_data * 2 * kWave * 2)
        by_data *= 2;
    return by_chip < by_data ? by_chip : by_data;
}

// everything a sampler owns on the device; safe on a half-built sampler
static void release(apemost_hip_sampler *s) {
    if (s->stream)
        hipStreamSynchronize(s->stream);
    for (void *p : s->allocations)
        hipFree(p);
    if (s->cal.d_rec)
        hipFree(s->cal.d_rec);
    if (s->cal.d_orig)
        hipFree(s->cal.d_orig);
    if (s->cal.d_progress)
        hipFree(s->cal.d_progress);
    if (s->cal.d_list)
        hipFree(s->cal.d_list);
    if (s->cal.h_rec)
        hipHostFree(s->cal.h_rec);
    if (s->cal.h_list)
        hipHostFree(s->cal.h_list);
    if (s->cal.ev)
        hipEventDestroy(s->cal.ev);
    if (s->user.module)
        hipModuleUnload(s->user.module);
    if (s->copy_stream) {
        hipStreamSynchronize(s->copy_stream);
        hipStreamDestroy(s->copy_stream);
    }
    if (s->ev_copy)
        hipEventDestroy(s->ev_copy);
    if (s->ev_exported)
        hipEventDestroy(s->ev_exported);
    if (s->ev_imported)
        hipEventDestroy(s->ev_imported);
    if (s->h_word)
        hipHostFree(s->h_word);
    if (s->ev0)
        hipEventDestroy(s->ev0);
    if (s->ev1)
        hipEventDestroy(s->ev1);
    if (s->stream)
        hipStreamDestroy(s->stream);
    delete s;
}

// ---- APEMOST_MODEL_USER: the user's likelihood compiled into the two-phase kernels at run time ----
// hiprtc is loaded on demand (a sampler of a built-in model never needs it).  The translation unit is
// "#include pt_kernels.h" -- the same kernel templates this library was built from -- followed by the
// user's file, which defines the two functions Model<APEMOST_MODEL_USER> calls; the four kernels a
// sampler launches, for workgroups of 1, 2, 4 and 8 wavefronts per chain (round 4: the reference's
// ladders are <= 99 chains, where one wave per chain leaves most of the chip idle), are named as template
// instantiations and fetched by their lowered names.
namespace {
struct HipRtc {
    void *lib;
    hiprtcResult (*create)(hiprtcProgram *, const char *, const char *, int, const char **, const char **);
    hiprtcResult (*destroy)(hiprtcProgram *);
    hiprtcResult (*add_name)(hiprtcProgram, const char *);
    hiprtcResult (*compile)(hiprtcProgram, int, const char **);
    hiprtcResult (*log_size)(hiprtcProgram, size_t *);
    hiprtcResult (*log)(hiprtcProgram, char *);
    hiprtcResult (*lowered)(hiprtcProgram, const char *, const char **);
    hiprtcResult (*code_size)(hiprtcProgram, size_t *);
    hiprtcResult (*code)(hiprtcProgram, char *);
};
} // namespace

static int hiprtc_load(HipRtc &r) {
    static HipRtc cached = {};
    static std::mutex once;
    std::lock_guard<std::mutex> hold(once); // (samplers may be created from several host threads)
    if (!cached.lib) {
        const char *names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
        for (const char *n : names)
            if ((cached.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
                break;
        if (!cached.lib)
            return fail(APEMOST_HIP_ERR_UNSUPPORTED, "a user-supplied device model needs libhiprtc.so: %s", dlerror());
        bool ok = true;
        auto sym = [&](const char *n) {
            void *p = dlsym(cached.lib, n);
            ok = ok && p;
            return p;
        };
        cached.create = (decltype(cached.create))sym("hiprtcCreateProgram");
        cached.destroy = (decltype(cached.destroy))sym("hiprtcDestroyProgram");
        cached.add_name = (decltype(cached.add_name))sym("hiprtcAddNameExpression");
        cached.compile = (decltype(cached.compile))sym("hiprtcCompileProgram");
        cached.log_size = (decltype(cached.log_size))sym("hiprtcGetProgramLogSize");
        cached.log = (decltype(cached.log))sym("hiprtcGetProgramLog");
        cached.lowered = (decltype(cached.lowered))sym("hiprtcGetLoweredName");
        cached.code_size = (decltype(cached.code_size))sym("hiprtcGetCodeSize");
        cached.code = (decltype(cached.code))sym("hiprtcGetCode");
        if (!ok) {
            cached.lib = nullptr;
            return fail(APEMOST_HIP_ERR_UNSUPPORTED, "libhiprtc.so lacks an expected entry point");
        }
    }
    r = cached;
    return APEMOST_HIP_OK;
}

// where this library's kernel headers are: csrc/ beside the .so, include/ one level up (in-tree
// layout), or APEMOST_HIP_SOURCE_DIR / APEMOST_HIP_INCLUDE_DIR
static void source_dirs(std::string &csrc, std::string &inc) {
    Dl_info info;
    std::string dir = ".";
    if (dladdr((const void *)&apemost_hip_abi_version, &info) && info.dli_fname) {
        dir = info.dli_fname;
        const size_t slash = dir.rfind('/');
        dir = slash == std::string::npos ? "." : dir.substr(0, slash);
    }
    const char *e1 = getenv("APEMOST_HIP_SOURCE_DIR"), *e2 = getenv("APEMOST_HIP_INCLUDE_DIR");
    csrc = e1 ? e1 : dir + "/csrc";
    inc = e2 ? e2 : dir + "/../include";
}

// a compiled user model: the code object and the lowered names of its four kernels.  Kept per process
// and keyed by (source text, kernel variant, device architecture): a host creates several samplers per
// phase (the one that checks the device model against the host plugin, one per shard, the one-chain
// twin of the single-chain API) and compiles once.
namespace {
constexpr int kUserShapes = 4;
constexpr int kUserWaves[kUserShapes] = {1, 2, 4, 8};
struct UserModelCode {
    std::string key;
    std::vector<char> code;
    std::string lowered[4 * kUserShapes];
    double compile_seconds;
};
std::vector<UserModelCode> g_user_models;
std::mutex g_user_models_lock;
} // namespace

static int user_model_load(apemost_hip_sampler *s, const UserModelCode &m) {
    HIP_TRY(hipModuleLoadData(&s->user.module, m.code.data()));
    {
        // the headers hiprtc compiled are the ones this library was built from
        hipDeviceptr_t sym = nullptr;
        size_t bytes = 0;
        unsigned long long theirs = 0;
        if (hipModuleGetGlobal(&sym, &bytes, s->user.module, "apemost_rtc_fingerprint") != hipSuccess || bytes != sizeof theirs)
            return fail(APEMOST_HIP_ERR_RUNTIME, "the compiled device model has no layout fingerprint: kernel headers older than this library?");
        HIP_TRY(hipMemcpyDtoH(&theirs, sym, sizeof theirs));
        if (theirs != kAbiFingerprint)
            return fail(APEMOST_HIP_ERR_RUNTIME,
                        "the kernel headers compiled for the device model (APEMOST_HIP_SOURCE_DIR, or csrc/ beside the library) "
                        "do not match this library: layout fingerprint %llx, library %llx", theirs, kAbiFingerprint);
    }
    for (int k = 0; k < kUserShapes; k++) {
        const int w = kUserWaves[k];
        HIP_TRY(hipModuleGetFunction(&s->user.round[w], s->user.module, m.lowered[4 * k + 0].c_str()));
        HIP_TRY(hipModuleGetFunction(&s->user.calibrate[w], s->user.module, m.lowered[4 * k + 1].c_str()));
        HIP_TRY(hipModuleGetFunction(&s->user.calc_model[w], s->user.module, m.lowered[4 * k + 2].c_str()));
        HIP_TRY(hipModuleGetFunction(&s->user.loglike[w], s->user.module, m.lowered[4 * k + 3].c_str()));
    }
    s->user_compile_seconds = m.compile_seconds;
    return APEMOST_HIP_OK;
}

static int user_model_build(apemost_hip_sampler *s) {
    HipRtc rtc;
    int rc = hiprtc_load(rtc);
    if (rc)
        return rc;
    FILE *f = fopen(s->cfg.device_model_source, "rb");
    if (!f)
        return fail(APEMOST_HIP_ERR_INVALID, "device model source %s: cannot be read", s->cfg.device_model_source);
    std::string user;
    char buf[4096];
    for (size_t n; (n = fread(buf, 1, sizeof buf, f)) > 0;)
        user.append(buf, n);
    fclose(f);
    std::string csrc, inc;
    source_dirs(csrc, inc);
    std::string shown; // the path as a C string literal (#line): backslashes and quotes escaped
    for (const char *q = s->cfg.device_model_source; *q; q++) {
        if (*q == '\\' || *q == '"')
            shown += '\\';
        if (*q != '\n' && *q != '\r')
            shown += *q;
    }
    const std::string src = "#define APEMOST_USER_MODEL 1\n#include \"pt_kernels.h\"\n#line 1 \"" + shown + "\"\n" + user + "\n";
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, s->cfg.device)); // (before the program exists: nothing to release on failure)
    hiprtcProgram prog = nullptr;
    if (rtc.create(&prog, src.c_str(), "apemost_user_model.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return fail(APEMOST_HIP_ERR_RUNTIME, "hiprtcCreateProgram failed");
    char names[4 * kUserShapes][128];
    const int km = s->kmodel, base = APEMOST_MODEL_USER;
    for (int k = 0; k < kUserShapes; k++) {
        const int w = kUserWaves[k];
        const char *prod = has_producer(w) ? "true" : "false";
        snprintf(names[4 * k + 0], sizeof names[0], "apemost::pt_round_kernel<%d, %d, false, %s>", km, w, prod);
        snprintf(names[4 * k + 1], sizeof names[0], "apemost::pt_calibrate_kernel<%d, %d, false, %s>", km, w, prod);
        snprintf(names[4 * k + 2], sizeof names[0], "apemost::pt_calc_model_kernel<%d, %d, false>", base, w);
        snprintf(names[4 * k + 3], sizeof names[0], "apemost::pt_loglike_kernel<%d, %d, false>", base, w);
    }
    for (auto &n : names)
        rtc.add_name(prog, n);
    const std::string key = std::to_string(s->kmodel) + "|" + prop.gcnArchName + "|" + user;
    {
        std::lock_guard<std::mutex> hold(g_user_models_lock);
        for (const UserModelCode &m : g_user_models)
            if (m.key == key) {
                rtc.destroy(&prog);
                return user_model_load(s, m);
            }
    }
    const std::string arch = std::string("--offload-arch=") + prop.gcnArchName, i1 = "-I" + csrc, i2 = "-I" + inc;
    const char *opts[] = {arch.c_str(), "-O3", "-ffp-contract=off", "-std=c++17", i1.c_str(), i2.c_str(), "-I/opt/rocm/include"};
    const auto t_compile = std::chrono::steady_clock::now();
    const hiprtcResult cr = rtc.compile(prog, (int)(sizeof opts / sizeof opts[0]), opts);
    if (cr != HIPRTC_SUCCESS) {
        size_t n = 0;
        rtc.log_size(prog, &n);
        std::string log(n + 1, 0);
        if (n)
            rtc.log(prog, &log[0]);
        rtc.destroy(&prog);
        return fail(APEMOST_HIP_ERR_INVALID, "device model %s does not compile:\n%s", s->cfg.device_model_source, log.c_str());
    }
    UserModelCode m;
    m.key = key;
    m.compile_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_compile).count();
    size_t code_bytes = 0;
    rtc.code_size(prog, &code_bytes);
    m.code.resize(code_bytes);
    rtc.code(prog, m.code.data());
    for (int i = 0; i < 4 * kUserShapes; i++) {
        const char *low = nullptr;
        if (rtc.lowered(prog, names[i], &low) != HIPRTC_SUCCESS || !low) {
            rtc.destroy(&prog);
            return fail(APEMOST_HIP_ERR_RUNTIME, "hiprtcGetLoweredName(%s) failed", names[i]);
        }
        m.lowered[i] = low;
    }
    rtc.destroy(&prog);
    rc = user_model_load(s, m);
    if (rc == APEMOST_HIP_OK) {
        std::lock_guard<std::mutex> hold(g_user_models_lock);
        g_user_models.push_back(std::move(m));
    }
    return rc;
}

// the part of apemost_hip_create that can fail after the sampler object exists
static int create_body(apemost_hip_sampler *s) {
    const apemost_hip_config *cfg = &s->cfg;
    int rc;
    // waves 1-3 produce the proposal candidates in the serial window of each step
    s->producers = has_producer(s->waves);
    if (s->waves != 1 && s->waves != 2 && s->waves != 4 && s->waves != 6 && s->waves != 8)
        return fail(APEMOST_HIP_ERR_INVALID, "waves_per_chain must be 1, 2, 4, 6 or 8");
    const size_t fixed_lds = (kFixedLdsDoubles + (size_t)cand_slots(s->waves) * 2 * kWave) * sizeof(double);
    const size_t data_lds = (size_t)2 * cfg->n_data * sizeof(double);
    const size_t fixed_max = fixed_lds > kObFixedDoubles * sizeof(double) ? fixed_lds : kObFixedDoubles * sizeof(double);
    s->lds_data = fixed_max + data_lds <= 160 * 1024 - 1024 && cfg->lds_policy != 2 && cfg->model != APEMOST_MODEL_USER &&
                  (cfg->lds_policy == 1 || choose_lds(*cfg, fixed_lds + data_lds));
    s->lds_bytes = fixed_lds + (s->lds_data ? data_lds : 0);
    s->lds_fixed_bytes = fixed_lds;
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&s->ev0));
    HIP_TRY(hipEventCreate(&s->ev1));

    DevArrays &d = s->d;
    d.n = cfg->n_chains;
    d.np = cfg->n_par;
    double *data = nullptr;
    if ((rc = dev_alloc(s, &d.f, d.f_count())) || (rc = dev_alloc(s, &d.u, d.u_count())) ||
        (rc = dev_alloc(s, &data, (size_t)cfg->n_cols * cfg->n_data)))
        return rc;
    d.data = data;

    s->sh.n_par = cfg->n_par;
    s->sh.n_data = cfg->n_data;
    s->sh.n_chains = cfg->n_chains;
    s->sh.chain_offset = cfg->chain_offset;
    s->sh.n_global = cfg->n_chains_global;
    s->sh.seed = cfg->seed;
    s->sh.consts.sigma = cfg->sigma;
    s->sh.consts.hmin = cfg->hmin;
    s->sh.circular = cfg->circular_params;
    if (cfg->flags & APEMOST_HIP_FLAG_PROPOSAL_LOGISTIC)
        s->sh.circular |= (u64)kProposalLogistic << kProposalShift;
    if (cfg->flags & APEMOST_HIP_FLAG_PROPOSAL_UNIFORM)
        s->sh.circular |= (u64)kProposalFlat << kProposalShift;
    s->sh.variant = (cfg->flags & APEMOST_HIP_FLAG_RANDOMSWAP) ? kVariantRandomSwap : 0;
    s->sh.variant |= (int)((unsigned)cfg->n_cols << 16);
    if (cfg->flags & APEMOST_HIP_FLAG_TEST_WITHHOLD_PUBLISH)
        s->sh.variant |= kVariantTestWithhold;
    s->kmodel = cfg->model + ((cfg->flags & (APEMOST_HIP_FLAG_PROPOSAL_LOGISTIC | APEMOST_HIP_FLAG_PROPOSAL_UNIFORM |
                                             APEMOST_HIP_FLAG_RANDOMSWAP))
                                  ? kVariantModel
                                  : 0);
    s->sh.x_abs_max = INFINITY; // until set_data
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (cfg->model == APEMOST_MODEL_USER && (rc = user_model_build(s)))
        return rc;
    if ((rc = enable_big_lds(s, s->waves)))
        return rc;
    {
        // Multi-round launches need every workgroup resident at once.  Blocks per CU from the
        // occupancy query, one held back (the query over-reports by one block for scalar-register-
        // heavy kernels: MI355X_MICROARCH.md "Residency and cooperative launch").
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
        // the round kernel this sampler's stepping launches use: one barrier per step where that
        // variant exists (8 likelihood waves per chain), the classic two-phase step otherwise
        s->one_barrier = has_one_barrier(s->waves) && !(cfg->flags & APEMOST_HIP_FLAG_TWO_BARRIER_STEP) && !s->user.module;
        s->cus = prop.multiProcessorCount;
        s->ob_helper = s->one_barrier && ob_wants_helper(s, cfg->n_chains);
        int b_lds = 0, b_plain = 0;
        if (s->user.module) {
            HIP_TRY(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&b_plain, s->user.round[s->waves], s->waves * kWave, s->lds_fixed_bytes));
        } else {
        if (s->lds_data)
            HIP_TRY(round_occupancy<true>(s->kmodel, s->waves, s->producers, s->one_barrier, s->ob_helper,
                                          s->one_barrier ? ob_lds_bytes(s, true) : s->lds_bytes, &b_lds));
        HIP_TRY(round_occupancy<false>(s->kmodel, s->waves, s->producers, s->one_barrier, s->ob_helper,
                                       s->one_barrier ? ob_lds_bytes(s, false) : s->lds_fixed_bytes, &b_plain));
        }
        const long long cus = prop.multiProcessorCount;
        s->resident_lds = s->lds_data && (long long)cfg->n_chains <= (long long)(b_lds - 1) * cus;
        s->resident_plain = (long long)cfg->n_chains <= (long long)(b_plain - 1) * cus;
        if (s->ob_helper) {
            // Workgroups with a helper wavefront are chosen for ladders of at most one chain per CU (ob_wants_helper)
            // and a CU admits one of them or two: a kernel that launches at all has a workgroup per CU resident,
            // whatever the occupancy query's last block is worth.
            s->resident_lds = s->lds_data && b_lds >= 1;
            s->resident_plain = b_plain >= 1;
        }
        if ((long long)cfg->n_chains * 2 <= cus) { // at most one workgroup per two CUs
            s->resident_lds = s->lds_data;
            s->resident_plain = true;
        }
        s->resident_ok = s->resident_lds || s->resident_plain;
        s->cooperative = (cfg->flags & APEMOST_HIP_FLAG_COOPERATIVE_LAUNCH) && prop.cooperativeLaunch && !s->user.module;
        // (APEMOST_COOP_ANY=1: the same for the two-phase kernels -- an experiment switch, tools/experiments/r04_session6.sh)
        if (!s->resident_ok && prop.cooperativeLaunch && (s->one_barrier || getenv("APEMOST_COOP_ANY")) && !s->user.module) {
            // (the one-barrier kernels only: their steps take a microsecond and a launch per round
            // costs a multiple of that; 2048 one-wave chains of config 5 would fit too, but a round
            // there is a 340 us launch and the in-launch hand-off was measured 3 % behind)
            // Between the cautious estimate and the occupancy figure itself (e.g. 257-512 chains of
            // eight-wave workgroups, two per CU) the runtime decides: multi-round launches go through
            // hipLaunchCooperativeKernel, which places the whole grid or refuses -- and a refusal
            // turns this sampler to one round per launch (apemost_hip_run retries, see there).
            const bool fits_lds = s->lds_data && (long long)cfg->n_chains <= (long long)b_lds * cus;
            const bool fits_plain = (long long)cfg->n_chains <= (long long)b_plain * cus;
            if (fits_lds || fits_plain) {
                s->resident_lds = fits_lds;
                s->resident_plain = fits_plain;
                s->resident_ok = true;
                s->cooperative = true;
            }
        }
        // -DADAPT: adapt() sits between a round's steps and its swap attempt and runs as a launch
        // of its own (pt_adapt_kernel), so every round is a launch
        if (cfg->flags & (APEMOST_HIP_FLAG_SINGLE_ROUND_LAUNCHES | APEMOST_HIP_FLAG_ADAPT | APEMOST_HIP_FLAG_RWM))
            s->resident_ok = false;
    }
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_create(const apemost_hip_config *cfg, apemost_hip_sampler **out) {
    if (!cfg || !out)
        return fail(APEMOST_HIP_ERR_INVALID, "cfg/out is NULL");
    *out = nullptr;
    if (cfg->abi_version != APEMOST_HIP_ABI_VERSION)
        return fail(APEMOST_HIP_ERR_INVALID, "ABI version %d, library has %d", cfg->abi_version,
                    APEMOST_HIP_ABI_VERSION);
    if (cfg->n_par < 1 || cfg->n_par > APEMOST_HIP_MAX_PAR)
        return fail(APEMOST_HIP_ERR_INVALID, "n_par %d outside [1,%d]", cfg->n_par, APEMOST_HIP_MAX_PAR);
    if (cfg->n_chains < 1 || cfg->n_data < 1 || cfg->n_cols < 2)
        return fail(APEMOST_HIP_ERR_INVALID, "n_chains %d, n_data %d, n_cols %d invalid", cfg->n_chains,
                    cfg->n_data, cfg->n_cols);
    if (cfg->n_cols > 65535)
        return fail(APEMOST_HIP_ERR_INVALID, "n_cols %d: at most 65535 data columns", cfg->n_cols);
    if (cfg->chain_offset < 0 || cfg->chain_offset + cfg->n_chains > cfg->n_chains_global)
        return fail(APEMOST_HIP_ERR_INVALID, "shard [%lld,%lld) outside ladder of %lld chains",
                    (long long)cfg->chain_offset, (long long)(cfg->chain_offset + cfg->n_chains),
                    (long long)cfg->n_chains_global);
    if (cfg->n_par < 64 && (cfg->circular_params >> cfg->n_par) != 0)
        return fail(APEMOST_HIP_ERR_INVALID, "circular_params names a parameter beyond n_par");
    if (cfg->flags & ~(APEMOST_HIP_FLAG_SINGLE_ROUND_LAUNCHES | APEMOST_HIP_FLAG_COOPERATIVE_LAUNCH |
                       APEMOST_HIP_FLAG_TWO_BARRIER_STEP | APEMOST_HIP_FLAG_PROPOSAL_LOGISTIC |
                       APEMOST_HIP_FLAG_PROPOSAL_UNIFORM | APEMOST_HIP_FLAG_RANDOMSWAP | APEMOST_HIP_FLAG_ADAPT |
                       APEMOST_HIP_FLAG_TEST_REFUSE_COOPERATIVE | APEMOST_HIP_FLAG_TEST_WITHHOLD_PUBLISH | APEMOST_HIP_FLAG_RWM))
        return fail(APEMOST_HIP_ERR_INVALID, "unknown bits in flags: 0x%x", (unsigned)cfg->flags);
    if ((cfg->flags & APEMOST_HIP_FLAG_PROPOSAL_LOGISTIC) && (cfg->flags & APEMOST_HIP_FLAG_PROPOSAL_UNIFORM))
        return fail(APEMOST_HIP_ERR_INVALID, "PROPOSAL_LOGISTIC and PROPOSAL_UNIFORM are alternatives");
    if (!(cfg->adapt_target >= 0 && cfg->adapt_target < 1e300))
        return fail(APEMOST_HIP_ERR_INVALID, "adapt_target %g invalid", cfg->adapt_target);
    if (cfg->n_chains_global > 2000000)
        return fail(APEMOST_HIP_ERR_INVALID, "n_beta*1000 must fit an int (interaction.c:92)");
    switch (cfg->model) {
    case APEMOST_MODEL_SIMPLESIN:
        if (cfg->n_par != 4)
            return fail(APEMOST_HIP_ERR_INVALID, "simplesin needs n_par = 4");
        break;
    case APEMOST_MODEL_SINE3:
        if (cfg->n_par != 10)
            return fail(APEMOST_HIP_ERR_INVALID, "sine3 needs n_par = 10");
        break;
    case APEMOST_MODEL_PULSE:
        if (cfg->n_par < 4 || (cfg->n_par - 2) % 2 != 0)
            return fail(APEMOST_HIP_ERR_INVALID, "pulse needs n_par = 2 + 2*modes");
        break;
    case APEMOST_MODEL_PULSE_VROT:
        if (cfg->n_par != 7)
            return fail(APEMOST_HIP_ERR_INVALID, "pulse_vrot needs n_par = 7");
        break;
    case APEMOST_MODEL_USER:
        if (!cfg->device_model_source || !*cfg->device_model_source)
            return fail(APEMOST_HIP_ERR_INVALID, "APEMOST_MODEL_USER needs device_model_source (include/apemost_device_model.h)");
        if (cfg->lds_policy == 1)
            return fail(APEMOST_HIP_ERR_INVALID, "a user-supplied model reads the data rows by index, through L2: they are not staged in LDS");
        if (cfg->waves_per_chain == 6)
            return fail(APEMOST_HIP_ERR_INVALID, "a user-supplied device model runs with 1, 2, 4 or 8 waves per chain");
        break;
    default:
        return fail(APEMOST_HIP_ERR_UNSUPPORTED, "unknown device model %d", cfg->model);
    }
    if (cfg->model != APEMOST_MODEL_USER && cfg->device_model_source)
        return fail(APEMOST_HIP_ERR_INVALID, "device_model_source is for APEMOST_MODEL_USER only");
    int rc = apemost_hip_device_info(cfg->device, nullptr, 0, nullptr, nullptr);
    if (rc != APEMOST_HIP_OK)
        return rc;
    HIP_TRY(hipSetDevice(cfg->device));

    apemost_hip_sampler *s = new apemost_hip_sampler();
    s->cfg = *cfg;
    s->cur = 0;
    s->round = 0;
    s->swap_pending = 0;
    s->launches = 0;
    s->launches_at_begin = 0;
    s->cal = {};
    memset(&s->user, 0, sizeof s->user);
    s->big_lds_set = 0;
    s->stream = nullptr;
    s->ev0 = s->ev1 = nullptr;
    s->copy_stream = nullptr;
    s->ev_copy = nullptr;
    s->edge_out = s->edge_in = nullptr;
    s->ev_exported = s->ev_imported = nullptr;
    s->h_word = nullptr;
    s->handoff_failed = false;
    s->waves = choose_waves(*cfg);
    s->user_compile_seconds = 0;
    s->rwm_keep = nullptr;
    s->in_rwm = false;
    if (s->waves == 6 && (cfg->flags & (APEMOST_HIP_FLAG_PROPOSAL_LOGISTIC | APEMOST_HIP_FLAG_PROPOSAL_UNIFORM |
                                        APEMOST_HIP_FLAG_RANDOMSWAP))) {
        delete s;
        return fail(APEMOST_HIP_ERR_INVALID, "the proposal / swap variants are built for 1, 2, 4 or 8 waves per chain");
    }
    rc = create_body(s);
    if (rc != APEMOST_HIP_OK) {
        release(s); // the stream, the events and every allocation made so far
        return rc;
    }
    *out = s;
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_destroy(apemost_hip_sampler *s) {
    if (!s)
        return APEMOST_HIP_OK;
    hipSetDevice(s->cfg.device);
    release(s);
    return APEMOST_HIP_OK;
}

#define CHECK_S(s)                                                                               \
    do {                                                                                         \
        if (!(s))                                                                                \
            return fail(APEMOST_HIP_ERR_INVALID, "sampler is NULL");                             \
        HIP_TRY(hipSetDevice((s)->cfg.device));                                                  \
    } while (0)

// Words the kernels raise instead of spinning for ever: 1 = an in-launch swap hand-off timed out (a
// partner workgroup was not resident), 2 = an in-launch swap picked a pair that straddles the shard,
// 3 = a proposal found no point inside [min,max] in 2^24 attempts.  The results of that launch are
// void.  The word is cleared here so that the sampler can be reloaded (set_state) and used again;
// after a hand-off timeout it only issues single-round launches, which never wait for anybody.
static int check_handoff(apemost_hip_sampler *s) {
    u64 word = 0;
    HIP_TRY(hipMemcpyAsync(&word, s->d.timeout_word(), sizeof word, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (word == 0)
        return APEMOST_HIP_OK;
    HIP_TRY(hipMemsetAsync(s->d.timeout_word(), 0, sizeof(u64), s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (word == 1)
        s->handoff_failed = true;
    return fail(APEMOST_HIP_ERR_RUNTIME, "%s (code %llu): the results of that launch are void%s",
                word == 3 ? "a proposal found no point inside its prior box"
                          : "in-launch swap hand-off failed",
                (unsigned long long)word, word == 1 ? "; falling back to one round per launch" : "");
}

extern "C" int apemost_hip_synchronize(apemost_hip_sampler *s) {
    CHECK_S(s);
    HIP_TRY(hipStreamSynchronize(s->stream));
    return check_handoff(s);
}

extern "C" int apemost_hip_stream(apemost_hip_sampler *s, void **stream) {
    CHECK_S(s);
    if (!stream)
        return fail(APEMOST_HIP_ERR_INVALID, "stream is NULL");
    *stream = (void *)s->stream;
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_waves_per_chain(apemost_hip_sampler *s, int *waves, int *data_in_lds) {
    CHECK_S(s);
    if (waves)
        *waves = s->waves;
    if (data_in_lds)
        *data_in_lds = s->lds_data ? 1 : 0;
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_user_model_compile_seconds(apemost_hip_sampler *s, double *seconds) {
    CHECK_S(s);
    if (!s->user.module || !seconds)
        return fail(APEMOST_HIP_ERR_INVALID, "not a sampler of a user-supplied device model");
    *seconds = s->user_compile_seconds;
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_launch_policy(apemost_hip_sampler *s, int32_t *one_barrier, int32_t *cooperative,
                                         int32_t *max_rounds) {
    CHECK_S(s);
    if (one_barrier)
        *one_barrier = s->one_barrier ? (s->ob_helper ? 2 : 1) : 0;
    if (cooperative)
        *cooperative = (s->cooperative && s->resident_ok && !s->handoff_failed) ? 1 : 0;
    if (max_rounds)
        *max_rounds = max_rounds_per_launch(s);
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_set_chain_offset(apemost_hip_sampler *s, int64_t chain_offset) {
    CHECK_S(s);
    if (chain_offset < 0 || chain_offset + s->cfg.n_chains > s->cfg.n_chains_global)
        return fail(APEMOST_HIP_ERR_INVALID, "chain offset %lld outside the ladder", (long long)chain_offset);
    s->cfg.chain_offset = chain_offset;
    s->sh.chain_offset = chain_offset;
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_set_data(apemost_hip_sampler *s, const double *data_rowmajor) {
    CHECK_S(s);
    if (!data_rowmajor)
        return fail(APEMOST_HIP_ERR_INVALID, "data is NULL");
    const int n = s->cfg.n_data, nc = s->cfg.n_cols;
    std::vector<double> col((size_t)n * nc);
    double x_abs_max = 0;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < nc; j++)
            col[(size_t)j * n + i] = data_rowmajor[(size_t)i * nc + j];
        const double ax = std::fabs(data_rowmajor[(size_t)i * nc]);
        if (!(ax <= x_abs_max)) // also catches NaN
            x_abs_max = std::isfinite(ax) ? ax : INFINITY;
    }
    s->sh.x_abs_max = x_abs_max;
    HIP_TRY(hipMemcpyAsync((void *)s->d.data, col.data(), col.size() * sizeof(double),
                           hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return APEMOST_HIP_OK;
}

// copy one [n_chains][width] field between host and the interior rows of a device array
template <class T>
static int xfer(apemost_hip_sampler *s, T *dev, T *host, size_t width, bool rows_layout, bool to_device) {
    if (!host)
        return APEMOST_HIP_OK;
    T *p = dev + (rows_layout ? width : 0);
    const size_t bytes = (size_t)s->cfg.n_chains * width * sizeof(T);
    if (to_device)
        HIP_TRY(hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, s->stream));
    else
        HIP_TRY(hipMemcpyAsync(host, p, bytes, hipMemcpyDeviceToHost, s->stream));
    return APEMOST_HIP_OK;
}

static int xfer_state(apemost_hip_sampler *s, const apemost_hip_state_view *v, bool up) {
    CHECK_S(s);
    if (!v)
        return fail(APEMOST_HIP_ERR_INVALID, "state view is NULL");
    const size_t np = s->cfg.n_par;
    const int h = s->cur;
    DevArrays &d = s->d;
    int rc;
    if ((rc = xfer(s, d.params(h), v->params, np, true, up)) ||
        (rc = xfer(s, d.params_best(h), v->params_best, np, true, up)) ||
        (rc = xfer(s, d.prob(h), v->prob, 1, true, up)) ||
        (rc = xfer(s, d.prob_best(h), v->prob_best, 1, true, up)) ||
        (rc = xfer(s, d.prior(h), v->prior, 1, true, up)) || (rc = xfer(s, d.beta(), v->beta, 1, true, up)) ||
        (rc = xfer(s, d.step(), v->step, np, false, up)) || (rc = xfer(s, d.pmin(), v->pmin, np, false, up)) ||
        (rc = xfer(s, d.pmax(), v->pmax, np, false, up)) ||
        (rc = xfer(s, (uint64_t *)d.params_accepts(), v->params_accepts, np, false, up)) ||
        (rc = xfer(s, (uint64_t *)d.params_rejects(), v->params_rejects, np, false, up)) ||
        (rc = xfer(s, (uint64_t *)d.accept(), v->accept, 1, false, up)) ||
        (rc = xfer(s, (uint64_t *)d.reject(), v->reject, 1, false, up)) ||
        (rc = xfer(s, (uint64_t *)d.n_iter(), v->n_iter, 1, false, up)) ||
        (rc = xfer(s, (uint64_t *)d.swapcount(), v->swapcount, 1, false, up)) ||
        (rc = xfer(s, (uint64_t *)d.ticks(), v->ticks, 1, false, up)))
        return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    return APEMOST_HIP_OK;
}

// A prior box with min > max can never be hit: the reference's redraw loop (src/markov_chain.c:235-240)
// would spin on the host, the kernel's on the GPU.  Refuse it at the door.
static int check_box(apemost_hip_sampler *s, const apemost_hip_state_view *v) {
    if (!v->pmin && !v->pmax)
        return APEMOST_HIP_OK;
    const size_t count = (size_t)s->cfg.n_chains * s->cfg.n_par;
    std::vector<double> other;
    const double *lo = v->pmin, *hi = v->pmax;
    if (!lo || !hi) { // only one side comes with this view: the other one is what the device holds
        other.resize(count);
        HIP_TRY(hipMemcpyAsync(other.data(), lo ? s->d.pmax() : s->d.pmin(), count * sizeof(double),
                               hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        (lo ? hi : lo) = other.data();
    }
    for (size_t k = 0; k < count; k++)
        if (!(lo[k] <= hi[k]))
            return fail(APEMOST_HIP_ERR_INVALID, "chain %zu parameter %zu: min %g > max %g (or NaN)",
                        k / s->cfg.n_par, k % s->cfg.n_par, lo[k], hi[k]);
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_set_state(apemost_hip_sampler *s, const apemost_hip_state_view *v) {
    CHECK_S(s);
    if (!v)
        return fail(APEMOST_HIP_ERR_INVALID, "state view is NULL");
    const int rc = check_box(s, v);
    return rc ? rc : xfer_state(s, v, true);
}
extern "C" int apemost_hip_get_state(apemost_hip_sampler *s, const apemost_hip_state_view *v) {
    int rc = xfer_state(s, v, false);
    return rc ? rc : check_handoff(s);
}

extern "C" int apemost_hip_set_round(apemost_hip_sampler *s, uint64_t round, int swap_pending) {
    CHECK_S(s);
    s->round = round;
    s->swap_pending = swap_pending ? 1 : 0;
    // hand-off words count swap indices upwards; a rewound swap stream restarts them
    HIP_TRY(hipMemsetAsync(s->d.published(), 0, (2 * (size_t)s->cfg.n_chains + 2) * sizeof(u64), s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return APEMOST_HIP_OK;
}
extern "C" int apemost_hip_get_round(apemost_hip_sampler *s, uint64_t *round, int *swap_pending) {
    CHECK_S(s);
    if (round)
        *round = s->round;
    if (swap_pending)
        *swap_pending = s->swap_pending;
    return APEMOST_HIP_OK;
}

// The one-barrier step with a helper wavefront (pt_onebarrier.h, HELPER): the models with a prior, where every
// workgroup of the launch has a CU to itself -- nine (thirteen) wavefronts; two such workgroups on a CU were not
// measured to gain (ladders of 257-512 chains keep the eight-wave form and its two workgroups per CU).
// APEMOST_OB_HELPER=0 switches it off (A/B runs, tests); it never runs on ladders of more than one chain per CU,
// whose nine-wave grids would not be resident.
static bool ob_wants_helper(const apemost_hip_sampler *s, int n_chains) {
    if (!ob_can_help(s->kmodel))
        return false;
    if (const char *env = getenv("APEMOST_OB_HELPER")) // (0 switches it off; 1 is the default where it may run at all)
        return atoi(env) != 0 && n_chains <= s->cus;
    return n_chains <= s->cus;
}

static size_t ob_lds_bytes(const apemost_hip_sampler *s, bool lds_data) {
    return (size_t)kObFixedDoubles * sizeof(double) + (lds_data ? (size_t)2 * s->cfg.n_data * sizeof(double) : 0);
}
// dynamic LDS of the two-phase kernels for workgroups of `waves` likelihood wavefronts
static size_t classic_lds_bytes(const apemost_hip_sampler *s, int waves, bool lds_data) {
    return (kFixedLdsDoubles + (size_t)cand_slots(waves) * 2 * kWave) * sizeof(double) +
           (lds_data ? (size_t)2 * s->cfg.n_data * sizeof(double) : 0);
}

// one launch with an explicit workgroup shape (the stepping launches use the sampler's own; the
// calibration picks one per segment, by the number of chains that are still calibrating)
static int launch_shape(apemost_hip_sampler *s, KernelKind kind, int grid, const void *args, int waves, bool lds_data,
                        bool coop, bool helper = false) {
    LaunchOp op;
    op.kind = kind;
    op.lds_data = lds_data;
    op.producers = has_producer(waves);
    op.helper = helper;
    op.coop = coop;
    op.grid = grid;
    op.lds = (kind == K_ROUND_OB || kind == K_CALIB_OB) ? ob_lds_bytes(s, lds_data) : classic_lds_bytes(s, waves, lds_data);
    op.st = s->stream;
    op.args = args;
    if (coop && (s->cfg.flags & APEMOST_HIP_FLAG_TEST_REFUSE_COOPERATIVE)) // test hook: see the header
        return fail(APEMOST_HIP_ERR_RUNTIME, "kernel launch failed: cooperative launch refused (test hook)");
    if (s->user.module) {
        // the kernels of a user-supplied model live in a run-time module: the two-phase step, data through L2
        hipFunction_t f = nullptr;
        if (waves >= 1 && waves <= 8)
            f = kind == K_ROUND ? s->user.round[waves] : kind == K_CALIB ? s->user.calibrate[waves]
                : kind == K_CALC ? s->user.calc_model[waves] : kind == K_EVAL ? s->user.loglike[waves] : nullptr;
        if (!f || lds_data || coop)
            return fail(APEMOST_HIP_ERR_RUNTIME, "kernel launch failed: no such kernel for a user-supplied model");
        void *params[] = {const_cast<void *>(args)};
        const hipError_t e = hipModuleLaunchKernel(f, (unsigned)grid, 1, 1, (unsigned)(waves * kWave), 1, 1, (unsigned)op.lds, s->stream, params, nullptr);
        if (e != hipSuccess)
            return fail(APEMOST_HIP_ERR_RUNTIME, "kernel launch failed: %s", hipGetErrorString(e));
        return APEMOST_HIP_OK;
    }
    const hipError_t err = dispatch(s->kmodel, waves, op);
    if (err != hipSuccess)
        return fail(APEMOST_HIP_ERR_RUNTIME, "kernel launch failed: %s", hipGetErrorString(err));
    return APEMOST_HIP_OK;
}

// stage_data: a launch that walks the data vector only a few times (n_swap < 4, single
// likelihood evaluations) reads it through L2 instead of copying it into LDS first
static int launch(apemost_hip_sampler *s, KernelKind kind, int grid, const void *args, bool stage_data = true,
                  bool coop = false) {
    return launch_shape(s, kind, grid, args, s->waves, s->lds_data && stage_data, coop, kind == K_ROUND_OB && s->ob_helper);
}

extern "C" int apemost_hip_calc_model(apemost_hip_sampler *s, int32_t first, int32_t count) {
    CHECK_S(s);
    if (count < 0)
        count = s->cfg.n_chains - first;
    if (first < 0 || count < 1 || first + count > s->cfg.n_chains)
        return fail(APEMOST_HIP_ERR_INVALID, "calc_model: chains [%d,%d) outside [0,%d)", first,
                    first + count, s->cfg.n_chains);
    RoundArgs a;
    a.d = s->d;
    a.sh = s->sh;
    a.cur = s->cur;
    a.first = first;
    a.which = -1;
    a.apply_swap = 0;
    a.n_steps = 0;
    a.n_rounds = 0;
    a.round = 0;
    a.samples = nullptr;
    return launch(s, K_CALC, count, &a, false);
}

static int loglike_on_device(apemost_hip_sampler *s, int32_t n, const double *params, const double *beta,
                             double *prob, double *prior, double *d_params, double *d_beta, double *d_prob,
                             double *d_prior) {
    const size_t np = s->cfg.n_par;
    HIP_TRY(hipMemcpyAsync(d_params, params, n * np * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(d_beta, beta, n * sizeof(double), hipMemcpyHostToDevice, s->stream));
    EvalArgs a;
    a.sh = s->sh;
    a.data = s->d.data;
    a.params = d_params;
    a.beta = d_beta;
    a.prob = d_prob;
    a.prior = d_prior;
    const int rc = launch(s, K_EVAL, n, &a, false);
    if (rc) {
        hipStreamSynchronize(s->stream);
        return rc;
    }
    HIP_TRY(hipMemcpyAsync(prob, d_prob, n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (prior)
        HIP_TRY(hipMemcpyAsync(prior, d_prior, n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_loglike(apemost_hip_sampler *s, int32_t n, const double *params,
                                   const double *beta, double *prob, double *prior) {
    CHECK_S(s);
    if (n < 1 || !params || !beta || !prob)
        return fail(APEMOST_HIP_ERR_INVALID, "loglike: bad arguments");
    const size_t np = s->cfg.n_par;
    // one scratch block: params [n][np], beta [n], prob [n], prior [n]
    double *scratch = nullptr;
    HIP_TRY(hipMalloc((void **)&scratch, (size_t)n * (np + 3) * sizeof(double)));
    double *d_params = scratch, *d_beta = scratch + (size_t)n * np, *d_prob = d_beta + n, *d_prior = d_prob + n;
    const int rc = loglike_on_device(s, n, params, beta, prob, prior, d_params, d_beta, d_prob, d_prior);
    hipFree(scratch);
    return rc;
}

template <bool LDS>
static hipError_t round_occupancy(int model, int waves, bool producers, bool one_barrier, bool helper, size_t lds_bytes, int *blocks) {
    OccupancyOp<LDS> op;
    op.producers = producers;
    op.one_barrier = one_barrier;
    op.helper = helper;
    op.lds_bytes = lds_bytes;
    op.blocks = blocks;
    return dispatch(model, waves, op);
}

// Kernels that stage more than 64 KiB must opt in once per function: every kernel of the `waves`-wave
// shape that may stage the data vector, each with its own footprint (the one-barrier kernels carve
// 1 KiB more than the two-phase ones).  Made once per shape, before its first launch.
static int enable_big_lds(apemost_hip_sampler *s, int waves) {
    if (((s->big_lds_set >> waves) & 1) || s->user.module)
        return APEMOST_HIP_OK;
    LdsAttrOp op;
    op.bytes = classic_lds_bytes(s, waves, true);
    op.ob_bytes = ob_lds_bytes(s, true);
    const size_t most = has_one_barrier(waves) && op.ob_bytes > op.bytes ? op.ob_bytes : op.bytes;
    if (s->cfg.lds_policy != 2 && most > 64 * 1024 && most <= 160 * 1024 - 1024) {
        const hipError_t e = dispatch(s->kmodel, waves, op);
        if (e != hipSuccess)
            return fail(APEMOST_HIP_ERR_RUNTIME, "hipFuncSetAttribute(LDS %zu B): %s", most, hipGetErrorString(e));
    }
    s->big_lds_set |= 1u << waves;
    return APEMOST_HIP_OK;
}

static int launch_round_impl(apemost_hip_sampler *s, uint32_t n_rounds, uint32_t n_steps, int apply_swap, int which,
                             double *d_samples);

extern "C" int apemost_hip_launch_round(apemost_hip_sampler *s, uint32_t n_steps, int apply_swap,
                                        double *d_samples) {
    return launch_round_impl(s, 1, n_steps, apply_swap, -1, d_samples);
}

extern "C" int apemost_hip_launch_rounds(apemost_hip_sampler *s, uint32_t n_rounds, uint32_t n_steps,
                                         int apply_swap, double *d_samples) {
    if (n_rounds < 1)
        return fail(APEMOST_HIP_ERR_INVALID, "launch_rounds: n_rounds must be >= 1");
    return launch_round_impl(s, n_rounds, n_steps, apply_swap, -1, d_samples);
}

extern "C" int apemost_hip_launch_round_for(apemost_hip_sampler *s, uint32_t n_steps, int32_t param,
                                            double *d_samples) {
    if (s && (param < 0 || param >= s->cfg.n_par))
        return fail(APEMOST_HIP_ERR_INVALID, "launch_round_for: parameter %d outside [0,%d)", param, s->cfg.n_par);
    return launch_round_impl(s, 1, n_steps, 0, param, d_samples);
}

// Multi-round launches hand swap records from workgroup to workgroup inside the launch, which
// is only safe when every workgroup of the grid is resident at once.
static int max_rounds_per_launch(apemost_hip_sampler *s) {
    return (s->resident_ok && !s->handoff_failed) ? 1024 : 1;
}

extern "C" int apemost_hip_max_rounds_per_launch(apemost_hip_sampler *s, int32_t *max_rounds) {
    CHECK_S(s);
    if (!max_rounds)
        return fail(APEMOST_HIP_ERR_INVALID, "max_rounds is NULL");
    *max_rounds = max_rounds_per_launch(s);
    return APEMOST_HIP_OK;
}

static int launch_round_impl(apemost_hip_sampler *s, uint32_t n_rounds, uint32_t n_steps, int apply_swap, int which,
                             double *d_samples) {
    CHECK_S(s);
    int rc;
    if ((int)n_rounds > max_rounds_per_launch(s))
        return fail(APEMOST_HIP_ERR_INVALID, "%u rounds in one launch, this sampler allows %d (grid residency)",
                    n_rounds, max_rounds_per_launch(s));
    if (n_rounds > 1 && n_steps == 0)
        return fail(APEMOST_HIP_ERR_INVALID, "multi-round launches need n_steps > 0");
    RoundArgs a;
    a.d = s->d;
    a.sh = s->sh;
    a.cur = s->cur;
    a.first = 0;
    a.which = which;
    a.apply_swap = apply_swap ? 1 : 0;
    a.n_steps = n_steps;
    a.n_rounds = n_rounds;
    a.round = s->round;
    a.samples = d_samples;
    bool stage = (u64)n_steps * n_rounds >= 4 || s->cfg.lds_policy == 1;
    if (n_rounds > 1) // residency decides when workgroups wait for each other
        stage = s->resident_lds ? stage || !s->resident_plain : false;
    const bool one_barrier = s->one_barrier && which < 0 && n_steps > 0;
    rc = launch(s, one_barrier ? K_ROUND_OB : K_ROUND, s->cfg.n_chains, &a, stage, s->cooperative && n_rounds > 1);
    if (rc && s->cooperative && n_rounds > 1) {
        // The runtime cannot place the grid at once (nothing has run): this sampler issues one
        // round per launch from now on, beginning with the rounds asked for here -- the swap
        // attempt between two of them is then the fused swap-in at the second one's start.
        s->resident_ok = false;
        const size_t row = (size_t)s->cfg.n_chains * (s->cfg.n_par + 2);
        for (uint32_t i = 0; i < n_rounds; i++) {
            rc = launch_round_impl(s, 1, n_steps, i == 0 ? apply_swap : 1, which,
                                   d_samples ? d_samples + (size_t)i * n_steps * row : nullptr);
            if (rc)
                return rc;
        }
        return APEMOST_HIP_OK;
    }
    if (rc)
        return rc;
    if ((s->cfg.flags & APEMOST_HIP_FLAG_RWM) && n_steps > 0 && which < 0 && !s->in_rwm) {
        // a round of run_sampler has stepped (n_rounds is 1 here): adapt()'s RWM block before its ADAPT block
        // and before the swap attempt.  The chain now lives in the other half of the state; the extra step is
        // an ordinary launch of one step (no rows, the pending swap attempt left pending).
        if (!s->rwm_keep && (rc = dev_alloc(s, &s->rwm_keep, (size_t)s->cfg.n_chains * (2 + s->cfg.n_par))))
            return rc;
        const dim3 grid((s->cfg.n_chains + 255) / 256), block(256);
        const double target = s->cfg.adapt_target != 0 ? s->cfg.adapt_target : 0.5;
        const u64 round_before = s->round;
        s->cur ^= 1;
        s->launches++;
        hipLaunchKernelGGL(pt_rwm_pre_kernel, grid, block, 0, s->stream, s->d, s->cur, s->rwm_keep);
        HIP_TRY(hipGetLastError());
        s->in_rwm = true;
        rc = launch_round_impl(s, 1, 1, 0, -1, nullptr);
        s->in_rwm = false;
        if (rc)
            return rc;
        hipLaunchKernelGGL(pt_rwm_post_kernel, grid, block, 0, s->stream, s->d, s->sh, s->cur, (const double *)s->rwm_keep, target);
        HIP_TRY(hipGetLastError());
        s->cur ^= 1; // (the bookkeeping below flips it back: the state is where the extra step left it)
        s->launches--;
        s->round = round_before;
    }
    if ((s->cfg.flags & APEMOST_HIP_FLAG_ADAPT) && n_steps > 0 && which < 0 && !s->in_rwm) {
        // a round of run_sampler has stepped: adapt() before its swap attempt (n_rounds is 1 here)
        const double target = s->cfg.adapt_target != 0 ? s->cfg.adapt_target : 0.5;
        hipLaunchKernelGGL(pt_adapt_kernel, dim3((s->cfg.n_chains + 255) / 256), dim3(256), 0, s->stream, s->d, target);
        HIP_TRY(hipGetLastError());
    }
    s->cur ^= 1;
    s->launches++;
    s->round += (apply_swap ? 1 : 0) + (n_rounds - 1); // swap attempts consumed by this launch
    if (apply_swap)
        s->swap_pending = 0;
    if (n_steps > 0 && which < 0)
        s->swap_pending = 1;
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_run(apemost_hip_sampler *s, uint64_t n_rounds, uint32_t n_swap,
                               double *d_samples) {
    CHECK_S(s);
    if (s->cfg.n_chains != s->cfg.n_chains_global)
        return fail(APEMOST_HIP_ERR_INVALID,
                    "apemost_hip_run needs the whole ladder on one device; sharded ladders drive "
                    "apemost_hip_launch_rounds + apemost_hip_edge_*");
    const size_t row = (size_t)s->cfg.n_chains * (s->cfg.n_par + 2);
    for (uint64_t r = 0; r < n_rounds;) {
        const uint64_t per_launch = n_swap > 0 ? (uint64_t)max_rounds_per_launch(s) : 1;
        const uint64_t k = n_rounds - r < per_launch ? n_rounds - r : per_launch;
        int rc = launch_round_impl(s, (uint32_t)k, n_swap, s->swap_pending, -1,
                                   d_samples ? d_samples + r * n_swap * row : nullptr);
        if (rc)
            return rc;
        r += k;
    }
    if (s->swap_pending)
        return apemost_hip_launch_round(s, 0, 1, nullptr);
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_samples_alloc(apemost_hip_sampler *s, uint64_t n_steps, double **d_samples) {
    CHECK_S(s);
    if (!d_samples || n_steps == 0)
        return fail(APEMOST_HIP_ERR_INVALID, "samples_alloc: bad arguments");
    const size_t bytes = (size_t)n_steps * s->cfg.n_chains * (s->cfg.n_par + 2) * sizeof(double);
    HIP_TRY(hipMalloc((void **)d_samples, bytes));
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_samples_read(apemost_hip_sampler *s, const double *d_samples, uint64_t n_steps,
                                        double *host) {
    CHECK_S(s);
    if (!d_samples || !host)
        return fail(APEMOST_HIP_ERR_INVALID, "samples_read: bad arguments");
    const size_t bytes = (size_t)n_steps * s->cfg.n_chains * (s->cfg.n_par + 2) * sizeof(double);
    HIP_TRY(hipMemcpyAsync(host, d_samples, bytes, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return check_handoff(s); // rows of a void launch are not handed to the caller as samples
}

extern "C" int apemost_hip_samples_read_async(apemost_hip_sampler *s, const double *d_samples, uint64_t n_steps,
                                              double *host_samples, uint64_t *counters) {
    CHECK_S(s);
    if (!d_samples || !host_samples)
        return fail(APEMOST_HIP_ERR_INVALID, "samples_read_async: bad arguments");
    if (!s->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_copy, hipEventDisableTiming));
        HIP_TRY(hipHostMalloc((void **)&s->h_word, sizeof(u64), hipHostMallocDefault));
        *s->h_word = 0;
    }
    const size_t n = s->cfg.n_chains;
    // small things ride on the sampler's own stream, in launch order: the counters as they stand now
    // and the error word of the launches so far
    if (counters)
        HIP_TRY(hipMemcpyAsync(counters, s->d.accept(), 2 * n * sizeof(u64), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(s->h_word, s->d.timeout_word(), sizeof(u64), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipEventRecord(s->ev_copy, s->stream));
    HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->ev_copy, 0));
    const size_t bytes = (size_t)n_steps * n * (s->cfg.n_par + 2) * sizeof(double);
    HIP_TRY(hipMemcpyAsync(host_samples, d_samples, bytes, hipMemcpyDeviceToHost, s->copy_stream));
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_samples_pack_read_async(apemost_hip_sampler *s, const double *d_samples, uint64_t n_steps,
                                                   uint64_t skip, uint64_t thin, int32_t n_param_chains, int32_t layout,
                                                   double *d_packed, double *host_packed, uint64_t *counters,
                                                   uint64_t *n_kept) {
    CHECK_S(s);
    if (!d_samples || !d_packed || !host_packed || thin < 1 || (layout != 0 && layout != 1) || n_param_chains < 0 ||
        n_param_chains > s->cfg.n_chains)
        return fail(APEMOST_HIP_ERR_INVALID, "samples_pack_read_async: bad arguments");
    if (!s->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_copy, hipEventDisableTiming));
        HIP_TRY(hipHostMalloc((void **)&s->h_word, sizeof(u64), hipHostMallocDefault));
        *s->h_word = 0;
    }
    const size_t n = s->cfg.n_chains, np = s->cfg.n_par;
    const uint64_t kept = skip < n_steps ? (n_steps - skip + thin - 1) / thin : 0;
    const size_t record = layout == 0 ? (size_t)n_param_chains * np + 2 * n : n * (np + 2);
    if (n_kept)
        *n_kept = kept;
    if (kept > 0) {
        const size_t total = (size_t)kept * record;
        size_t blocks = (total + 255) / 256;
        if (blocks > 65535)
            blocks = 65535;
        hipLaunchKernelGGL(samples_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s->stream, d_samples, (int)n, (int)np,
                           (unsigned long long)kept, (unsigned long long)skip, (unsigned long long)thin, (int)n_param_chains,
                           (int)layout, d_packed);
        HIP_TRY(hipGetLastError());
    }
    if (counters) {
        HIP_TRY(hipMemcpyAsync(counters, s->d.accept(), 2 * n * sizeof(u64), hipMemcpyDeviceToHost, s->stream));
        // ... and, behind them, chain 0's parameter vector of the batch's LAST step (n_par doubles): what a
        // progress line prints, whatever the packing kept
        if (n_steps > 0)
            HIP_TRY(hipMemcpyAsync(counters + 2 * n, d_samples + (size_t)(n_steps - 1) * n * (np + 2), np * sizeof(double),
                                   hipMemcpyDeviceToHost, s->stream));
    }
    HIP_TRY(hipMemcpyAsync(s->h_word, s->d.timeout_word(), sizeof(u64), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipEventRecord(s->ev_copy, s->stream));
    HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->ev_copy, 0));
    if (kept > 0)
        HIP_TRY(hipMemcpyAsync(host_packed, d_packed, (size_t)kept * record * sizeof(double), hipMemcpyDeviceToHost,
                               s->copy_stream));
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_samples_wait(apemost_hip_sampler *s) {
    CHECK_S(s);
    if (!s->copy_stream)
        return APEMOST_HIP_OK;
    HIP_TRY(hipStreamSynchronize(s->copy_stream));
    if (*s->h_word != 0)
        return apemost_hip_synchronize(s); // reports, clears and falls back (check_handoff)
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_host_alloc(size_t bytes, void **p) {
    if (!p || bytes == 0)
        return fail(APEMOST_HIP_ERR_INVALID, "host_alloc: bad arguments");
    HIP_TRY(hipHostMalloc(p, bytes, hipHostMallocDefault));
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_host_free(void *p) {
    if (p)
        HIP_TRY(hipHostFree(p));
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_samples_free(apemost_hip_sampler *s, double *d_samples) {
    CHECK_S(s);
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (d_samples)
        HIP_TRY(hipFree(d_samples));
    return APEMOST_HIP_OK;
}

// host Philox4x32-10, bit-identical to the device's rocRAND stream; only used to
// tell sharded hosts which pair the next swap touches
static void philox_host(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]}, k[2] = {key[0], key[1]};
    for (int i = 0; i < 10; i++) {
        const uint64_t m0 = (uint64_t)ROCRAND_PHILOX_M4x32_0 * c[0];
        const uint64_t m1 = (uint64_t)ROCRAND_PHILOX_M4x32_1 * c[2];
        const uint32_t n0 = (uint32_t)(m1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)m1;
        const uint32_t n2 = (uint32_t)(m0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)m0;
        c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
        k[0] += ROCRAND_PHILOX_W32_0;
        k[1] += ROCRAND_PHILOX_W32_1;
    }
    memcpy(out, c, sizeof c);
}

extern "C" int64_t apemost_hip_swap_pair(uint64_t seed, uint64_t round, int64_t n_chains_global) {
    if (n_chains_global <= 1)
        return -1;
    const uint64_t sub = APEMOST_HIP_SWAP_SUBSEQUENCE;
    uint32_t ctr[4] = {(uint32_t)round, (uint32_t)(round >> 32), (uint32_t)sub, (uint32_t)(sub >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, out[4];
    philox_host(ctr, key, out);
    const double u = out[0] * (1.0 / 4294967296.0);
    const int nb = (int)n_chains_global;
    return (int)(nb * 1000 * u) % (nb - 1);
}

extern "C" int64_t apemost_hip_sampler_swap_pair(const apemost_hip_sampler *s, uint64_t round) {
    if (!s || s->cfg.n_chains_global <= 1)
        return -1;
    if (!(s->cfg.flags & APEMOST_HIP_FLAG_RANDOMSWAP))
        return apemost_hip_swap_pair(s->cfg.seed, round, s->cfg.n_chains_global);
    const uint64_t sub = APEMOST_HIP_SWAP_SUBSEQUENCE;
    uint32_t ctr[4] = {(uint32_t)round, (uint32_t)(round >> 32), (uint32_t)sub, (uint32_t)(sub >> 32)};
    uint32_t key[2] = {(uint32_t)s->cfg.seed, (uint32_t)(s->cfg.seed >> 32)}, out[4];
    philox_host(ctr, key, out);
    if (!(out[0] * (1.0 / 4294967296.0) < 1.0 / 1)) // swap_probability < 1.0 / n_swap, n_swap = 1
        return -1;
    const int nb = (int)s->cfg.n_chains_global;
    return (int)(nb * 1000 * (out[1] * (1.0 / 4294967296.0))) % (nb - 1);
}

extern "C" int64_t apemost_hip_rounds_within_shard(const apemost_hip_sampler *s, uint64_t first_round,
                                                   int64_t max_rounds) {
    if (!s || max_rounds <= 0)
        return 0;
    const int64_t lo = s->cfg.chain_offset, hi = lo + s->cfg.n_chains, n = s->cfg.n_chains_global;
    if (lo == 0 && hi == n) // the whole ladder: no edge to straddle
        return max_rounds;
    int64_t k = 0;
    for (; k < max_rounds; k++) {
        const int64_t a = apemost_hip_sampler_swap_pair(s, first_round + (uint64_t)k);
        if (a >= 0 && (a == lo - 1 || (a == hi - 1 && a + 1 < n)))
            break;
    }
    return k;
}

extern "C" int32_t apemost_hip_edge_doubles(int32_t n_par) { return 3 + 2 * n_par; }

extern "C" int apemost_hip_edge_export(apemost_hip_sampler *s, int side, double *d_buf) {
    CHECK_S(s);
    if (!d_buf || (side != 0 && side != 1))
        return fail(APEMOST_HIP_ERR_INVALID, "edge_export: bad arguments");
    const int row = side == 0 ? 1 : s->cfg.n_chains;
    hipLaunchKernelGGL(edge_export_kernel, dim3(1), dim3(kWave), 0, s->stream, s->d, s->cfg.n_par, s->cur, row,
                       d_buf);
    HIP_TRY(hipGetLastError());
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_edge_import(apemost_hip_sampler *s, int side, const double *d_buf) {
    CHECK_S(s);
    if (!d_buf || (side != 0 && side != 1))
        return fail(APEMOST_HIP_ERR_INVALID, "edge_import: bad arguments");
    const int row = side == 0 ? 0 : s->cfg.n_chains + 1;
    hipLaunchKernelGGL(edge_import_kernel, dim3(1), dim3(kWave), 0, s->stream, s->d, s->cfg.n_par, s->cur, row,
                       d_buf);
    HIP_TRY(hipGetLastError());
    return APEMOST_HIP_OK;
}

static int edge_buffers(apemost_hip_sampler *s) {
    if (s->edge_out)
        return APEMOST_HIP_OK;
    HIP_TRY(hipSetDevice(s->cfg.device));
    const size_t n = (size_t)apemost_hip_edge_doubles(s->cfg.n_par);
    int rc;
    if ((rc = dev_alloc(s, &s->edge_out, n)) || (rc = dev_alloc(s, &s->edge_in, n)))
        return rc;
    HIP_TRY(hipEventCreateWithFlags(&s->ev_exported, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_imported, hipEventDisableTiming));
    HIP_TRY(hipStreamSynchronize(s->stream));
    // recorded once so that the first exchange has something to wait on
    HIP_TRY(hipEventRecord(s->ev_exported, s->stream));
    HIP_TRY(hipEventRecord(s->ev_imported, s->stream));
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_edge_exchange(apemost_hip_sampler *lower, apemost_hip_sampler *upper) {
    if (!lower || !upper || lower == upper)
        return fail(APEMOST_HIP_ERR_INVALID, "edge_exchange: two different samplers are needed");
    if (lower->cfg.n_par != upper->cfg.n_par || lower->cfg.n_chains_global != upper->cfg.n_chains_global ||
        lower->cfg.seed != upper->cfg.seed ||
        lower->cfg.chain_offset + lower->cfg.n_chains != upper->cfg.chain_offset)
        return fail(APEMOST_HIP_ERR_INVALID, "edge_exchange: the shards are not neighbours of one ladder");
    int rc;
    if ((rc = edge_buffers(lower)) || (rc = edge_buffers(upper)))
        return rc;
    const size_t bytes = (size_t)apemost_hip_edge_doubles(lower->cfg.n_par) * sizeof(double);
    apemost_hip_sampler *side[2] = {lower, upper};
    // 1. each shard packs its edge chain, once its neighbour has finished reading the previous record
    for (int k = 0; k < 2; k++) {
        apemost_hip_sampler *me = side[k], *other = side[k ^ 1];
        HIP_TRY(hipSetDevice(me->cfg.device));
        HIP_TRY(hipStreamWaitEvent(me->stream, other->ev_imported, 0));
        if ((rc = apemost_hip_edge_export(me, k == 0 ? 1 : 0, me->edge_out)))
            return rc;
        HIP_TRY(hipEventRecord(me->ev_exported, me->stream));
    }
    // 2. each shard pulls the neighbour's record into its halo row
    for (int k = 0; k < 2; k++) {
        apemost_hip_sampler *me = side[k], *other = side[k ^ 1];
        HIP_TRY(hipSetDevice(me->cfg.device));
        HIP_TRY(hipStreamWaitEvent(me->stream, other->ev_exported, 0));
        HIP_TRY(hipMemcpyPeerAsync(me->edge_in, me->cfg.device, other->edge_out, other->cfg.device, bytes, me->stream));
        HIP_TRY(hipEventRecord(me->ev_imported, me->stream));
        if ((rc = apemost_hip_edge_import(me, k == 0 ? 1 : 0, me->edge_in)))
            return rc;
    }
    return APEMOST_HIP_OK;
}

// the shard pair (j, j+1) the swap attempt `index` straddles, or -1
static int straddled_edge(apemost_hip_sampler **sh, int n_shards, u64 index) {
    const int64_t a = apemost_hip_sampler_swap_pair(sh[0], index);
    for (int j = 0; a >= 0 && j + 1 < n_shards; j++)
        if (a == sh[j]->cfg.chain_offset + sh[j]->cfg.n_chains - 1)
            return j;
    return -1;
}

extern "C" int apemost_hip_run_shards(apemost_hip_sampler **sh, int32_t n_shards, uint64_t n_rounds, uint32_t n_swap,
                                      double **d_samples) {
    if (!sh || n_shards < 1)
        return fail(APEMOST_HIP_ERR_INVALID, "run_shards: no shards");
    int64_t next = 0;
    for (int j = 0; j < n_shards; j++) {
        if (!sh[j] || sh[j]->cfg.chain_offset != next || sh[j]->cfg.n_chains_global != sh[0]->cfg.n_chains_global ||
            sh[j]->cfg.seed != sh[0]->cfg.seed || sh[j]->round != sh[0]->round ||
            sh[j]->swap_pending != sh[0]->swap_pending)
            return fail(APEMOST_HIP_ERR_INVALID, "run_shards: shard %d does not continue the ladder (offset, seed, "
                                                 "ladder size and swap position must agree)", j);
        next += sh[j]->cfg.n_chains;
    }
    if (next != sh[0]->cfg.n_chains_global)
        return fail(APEMOST_HIP_ERR_INVALID, "run_shards: the shards cover %lld of %lld chains", (long long)next,
                    (long long)sh[0]->cfg.n_chains_global);
    // Shards that share a device launch their grids side by side on separate streams: the residency
    // every multi-round launch relies on (a workgroup may wait for its swap partner inside the launch)
    // was established for one grid alone on the device, so such shards hold one round per launch.
    bool device_shared = false;
    for (int j = 0; j < n_shards; j++)
        for (int i = 0; i < j; i++)
            device_shared = device_shared || sh[i]->cfg.device == sh[j]->cfg.device;
    int rc;
    for (uint64_t r = 0; r < n_rounds || (r == n_rounds && sh[0]->swap_pending);) {
        const bool finalise = r == n_rounds; // the swap attempt that closes the last round
        const int pending = sh[0]->swap_pending;
        const u64 first_inside = sh[0]->round + (pending ? 1 : 0);
        uint64_t limit = finalise || n_swap == 0 || device_shared ? 1 : n_rounds - r;
        for (int j = 0; j < n_shards; j++)
            if ((uint64_t)max_rounds_per_launch(sh[j]) < limit)
                limit = (uint64_t)max_rounds_per_launch(sh[j]);
        uint64_t k = 1;
        while (k < limit && straddled_edge(sh, n_shards, first_inside + k - 1) < 0)
            k++;
        if (pending) {
            const int j = straddled_edge(sh, n_shards, sh[0]->round);
            if (j >= 0 && (rc = apemost_hip_edge_exchange(sh[j], sh[j + 1])))
                return rc;
        }
        for (int j = 0; j < n_shards; j++) {
            const size_t row = (size_t)sh[j]->cfg.n_chains * (sh[j]->cfg.n_par + 2);
            double *out = (d_samples && d_samples[j] && !finalise) ? d_samples[j] + r * n_swap * row : nullptr;
            if ((rc = launch_round_impl(sh[j], (uint32_t)k, finalise ? 0 : n_swap, pending, -1, out)))
                return rc;
        }
        if (finalise)
            break;
        r += k;
    }
    return APEMOST_HIP_OK;
}

extern "C" void apemost_hip_calib_defaults(apemost_hip_calib_config *c) {
    if (!c)
        return;
    c->burn_in_iterations = 10000;
    c->iter_limit = 100000;
    c->iter_readjust = 200;
    c->no_rescaling_limit = 15;
    c->rat_limit = 0.5;
    c->target_global = 0.5;
    c->max_ar_deviation = 0.01;
    c->mul = 0.85;
    c->adjust_step = 0.5;
    c->progress_chain = -1;
    c->reserved = 0;
}

// ---- markov_chain_calibrate on the device, in segments -------------------------------------
// workgroup shape of a segment: what the sampler itself would choose for a ladder of as many chains
// as are still calibrating (a forced waves_per_chain stays forced)
struct CalibShape {
    int waves;
    bool lds_data, one_barrier, helper;
    u64 budget;
};

static CalibShape calib_shape(const apemost_hip_sampler *s, int n_active) {
    CalibShape g;
    apemost_hip_config c = s->cfg;
    c.n_chains = n_active;
    g.waves = choose_waves(c);
    if (!s->user.module && !built(s->kmodel, g.waves))
        g.waves = s->waves; // (development builds hold only some shapes)
    g.one_barrier = has_one_barrier(g.waves) && s->kmodel < kVariantModel && !(s->cfg.flags & APEMOST_HIP_FLAG_TWO_BARRIER_STEP) &&
                    !s->user.module;
    g.helper = g.one_barrier && ob_wants_helper(s, n_active);
    const size_t bytes = g.one_barrier ? ob_lds_bytes(s, true) : classic_lds_bytes(s, g.waves, true);
    g.lds_data = !s->user.module && bytes <= 160 * 1024 - 1024 && c.lds_policy != 2 && (c.lds_policy == 1 || choose_lds(c, bytes));
    // A segment of about a quarter of a second: likelihood evaluations a chain gets through in that
    // time, from a coarse model of one evaluation (1.7 ns per data point and wavefront, 0.7 us at
    // least, stretched when the wavefronts outnumber the SIMDs).  Always whole blocks, at least one.
    const double waves_per_wg = g.one_barrier ? g.waves + 4 + (g.helper ? 1 : 0) : g.waves;
    double t_eval = 1.7e-9 * s->cfg.n_data / g.waves;
    if (t_eval < 0.7e-6)
        t_eval = 0.7e-6;
    const double crowd = n_active * waves_per_wg / 1024.0;
    if (crowd > 1)
        t_eval *= crowd;
    g.budget = (u64)(0.25 / t_eval);
    if (g.budget < 1)
        g.budget = 1;
    // (tests cut the calibration into many more segments: APEMOST_CALIB_SEGMENT_EVALS=1 ends every
    // launch after one block)
    if (const char *env = getenv("APEMOST_CALIB_SEGMENT_EVALS")) {
        const long long v = atoll(env);
        if (v > 0)
            g.budget = (u64)v;
    }
    return g;
}

static int calib_launch_segment(apemost_hip_sampler *s) {
    auto &k = s->cal;
    const CalibShape g = calib_shape(s, k.n_active);
    int rc = enable_big_lds(s, g.waves);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(k.d_list, k.h_list, (size_t)k.n_active * sizeof(int), hipMemcpyHostToDevice, s->stream));
    CalibArgs a;
    a.d = s->d;
    a.sh = s->sh;
    a.cur = s->cur;
    a.first = k.first;
    a.burn_in_only = k.burn_in_only;
    a.progress_slot = k.progress_slot;
    a.cfg = k.cfg;
    a.list = k.d_list;
    a.rec = k.d_rec;
    a.orig_step = k.d_orig;
    a.progress = k.d_progress;
    a.progress_cap = k.progress_cap;
    a.budget = g.budget;
    rc = launch_shape(s, g.one_barrier ? K_CALIB_OB : K_CALIB, k.n_active, &a, g.waves, g.lds_data, false, g.helper);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(k.h_rec, k.d_rec, (size_t)k.count * sizeof(CalibRec), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipEventRecord(k.ev, s->stream));
    k.in_flight = true;
    k.segments++;
    k.launches_by_waves[g.waves]++;
    k.segment_start = std::chrono::steady_clock::now();
    k.segment_waves = g.waves;
    return APEMOST_HIP_OK;
}

// the records of the segment that has just ended: who is still calibrating
static void calib_collect(apemost_hip_sampler *s) {
    auto &k = s->cal;
    int n = 0;
    for (int i = 0; i < k.n_active; i++) {
        const int slot = k.h_list[i];
        if (k.h_rec[slot].stage != CAL_DONE)
            k.h_list[n++] = slot;
    }
    k.n_active = n;
    k.in_flight = false;
    k.seconds_by_waves[k.segment_waves] +=
        std::chrono::duration<double>(std::chrono::steady_clock::now() - k.segment_start).count();
}

extern "C" int apemost_hip_calibrate_begin(apemost_hip_sampler *s, int32_t first, int32_t count,
                                           const apemost_hip_calib_config *c, int burn_in_only) {
    CHECK_S(s);
    if (!c || first < 0 || count < 1 || first + count > s->cfg.n_chains)
        return fail(APEMOST_HIP_ERR_INVALID, "calibrate_chains: chains [%d,%d) outside [0,%d)", first,
                    first + count, s->cfg.n_chains);
    if (c->iter_readjust == 0)
        return fail(APEMOST_HIP_ERR_INVALID, "iter_readjust must be > 0");
    if (c->progress_chain >= 0 && (c->progress_chain < first || c->progress_chain >= first + count))
        return fail(APEMOST_HIP_ERR_INVALID, "progress_chain %d outside the chains [%d,%d) of this calibration",
                    c->progress_chain, first, first + count);
    auto &k = s->cal;
    if (k.open)
        return fail(APEMOST_HIP_ERR_INVALID, "calibrate_begin: the previous calibration was not collected (calibrate_end)");
    const int cap = (int)(c->iter_limit / c->iter_readjust) + 2; // readjustments a chain can reach
    if (count > k.capacity || cap > k.progress_cap) {
        if (k.d_rec)
            hipFree(k.d_rec);
        if (k.d_orig)
            hipFree(k.d_orig);
        if (k.d_list)
            hipFree(k.d_list);
        if (k.d_progress)
            hipFree(k.d_progress);
        if (k.h_rec)
            hipHostFree(k.h_rec);
        if (k.h_list)
            hipHostFree(k.h_list);
        k.d_rec = nullptr, k.d_orig = nullptr, k.d_list = nullptr, k.d_progress = nullptr, k.h_rec = nullptr, k.h_list = nullptr;
        k.capacity = k.progress_cap = 0;
        const int want = count > k.capacity ? count : k.capacity, pcap = cap > k.progress_cap ? cap : k.progress_cap;
        HIP_TRY(hipMalloc((void **)&k.d_rec, (size_t)want * sizeof(CalibRec)));
        HIP_TRY(hipMalloc((void **)&k.d_orig, (size_t)want * s->cfg.n_par * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&k.d_list, (size_t)want * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&k.d_progress, (size_t)pcap * (1 + 2 * s->cfg.n_par) * sizeof(double)));
        HIP_TRY(hipHostMalloc((void **)&k.h_rec, (size_t)want * sizeof(CalibRec), hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void **)&k.h_list, (size_t)want * sizeof(int), hipHostMallocDefault));
        k.capacity = want;
        k.progress_cap = pcap;
    }
    if (!k.ev)
        HIP_TRY(hipEventCreateWithFlags(&k.ev, hipEventDisableTiming));
    k.first = first;
    k.count = count;
    k.burn_in_only = burn_in_only ? 1 : 0;
    k.cfg = *c;
    k.progress_slot = c->progress_chain >= 0 ? c->progress_chain - first : -1;
    k.cancelled = false;
    k.segments = 0;
    memset(k.launches_by_waves, 0, sizeof k.launches_by_waves);
    for (double &t : k.seconds_by_waves)
        t = 0;
    HIP_TRY(hipMemsetAsync(k.d_rec, 0, (size_t)count * sizeof(CalibRec), s->stream)); // stage = CAL_INIT
    HIP_TRY(hipMemsetAsync(k.d_orig, 0, (size_t)count * s->cfg.n_par * sizeof(double), s->stream));
    for (int i = 0; i < count; i++)
        k.h_list[i] = i;
    k.n_active = count;
    const int rc = calib_launch_segment(s);
    if (rc)
        return rc;
    k.open = true;
    return APEMOST_HIP_OK;
}

// One turn of the segment loop without blocking: if the segment in flight has ended, its records are
// collected and the chains that are not done yet are launched again.  *active = chains still
// calibrating (0: calibrate_end will not wait).
extern "C" int apemost_hip_calibrate_poll(apemost_hip_sampler *s, int32_t *active) {
    CHECK_S(s);
    auto &k = s->cal;
    if (!k.open)
        return fail(APEMOST_HIP_ERR_INVALID, "calibrate_poll without calibrate_begin");
    if (k.in_flight) {
        const hipError_t q = hipEventQuery(k.ev);
        if (q == hipErrorNotReady) {
            if (active)
                *active = k.n_active;
            return APEMOST_HIP_OK;
        }
        HIP_TRY(q);
        calib_collect(s);
        if (k.n_active > 0 && !k.cancelled) {
            const int rc = calib_launch_segment(s);
            if (rc)
                return rc;
        }
    }
    if (active)
        *active = k.in_flight ? k.n_active : 0;
    return APEMOST_HIP_OK;
}

// for hosts that calibrate on several devices from one thread: returns when a segment of one of the
// samplers has ended (and its successor has been launched), or when none of them has chains left
extern "C" int apemost_hip_calibrate_wait_any(apemost_hip_sampler **ss, int32_t n, int32_t *active_total) {
    if (!ss || n < 1)
        return fail(APEMOST_HIP_ERR_INVALID, "calibrate_wait_any: no samplers");
    for (;;) {
        int total = 0;
        bool advanced = false;
        for (int i = 0; i < n; i++) {
            if (!ss[i] || !ss[i]->cal.open)
                continue;
            const u64 before = ss[i]->cal.segments;
            const bool was_busy = ss[i]->cal.in_flight;
            int32_t active = 0;
            const int rc = apemost_hip_calibrate_poll(ss[i], &active);
            if (rc)
                return rc;
            total += active;
            advanced = advanced || ss[i]->cal.segments != before || (was_busy && !ss[i]->cal.in_flight);
        }
        if (advanced || total == 0) {
            if (active_total)
                *active_total = total;
            return APEMOST_HIP_OK;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// stop after the segment in flight: calibrate_end then returns the chains as they are (status -1 for
// those that were not done)
extern "C" int apemost_hip_calibrate_cancel(apemost_hip_sampler *s) {
    CHECK_S(s);
    if (!s->cal.open)
        return fail(APEMOST_HIP_ERR_INVALID, "calibrate_cancel without calibrate_begin");
    s->cal.cancelled = true;
    return APEMOST_HIP_OK;
}

// ... and its results: status[count] / iters[count] of the chains of the matching begin
extern "C" int apemost_hip_calibrate_end(apemost_hip_sampler *s, int32_t *status, uint64_t *iters) {
    CHECK_S(s);
    auto &k = s->cal;
    if (!k.open)
        return fail(APEMOST_HIP_ERR_INVALID, "calibrate_end without calibrate_begin");
    int rc = APEMOST_HIP_OK;
    while (k.in_flight && !rc) {
        const hipError_t w = hipEventSynchronize(k.ev);
        if (w != hipSuccess)
            rc = fail(APEMOST_HIP_ERR_RUNTIME, "hipEventSynchronize failed: %s", hipGetErrorString(w));
        else
            rc = apemost_hip_calibrate_poll(s, nullptr);
    }
    k.open = false;
    k.in_flight = false;
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    rc = check_handoff(s);
    if (rc)
        return rc;
    int worst = 0;
    for (int i = 0; i < k.count; i++) {
        const CalibRec &r = k.h_rec[i];
        const int st = r.stage == CAL_DONE ? r.status : -1;
        if (status)
            status[i] = st;
        if (iters)
            iters[i] = r.sweeps;
        if (st && !worst)
            worst = st;
    }
    if (worst)
        return fail(APEMOST_HIP_ERR_CALIBRATION, "calibration %s",
                    worst == 1 ? "failed: a step width became too large"
                               : worst == 2 ? "failed: iteration limit reached" : "cancelled");
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_calibrate_chains(apemost_hip_sampler *s, int32_t first, int32_t count,
                                            const apemost_hip_calib_config *c, int burn_in_only,
                                            int32_t *status, uint64_t *iters) {
    const int rc = apemost_hip_calibrate_begin(s, first, count, c, burn_in_only);
    return rc ? rc : apemost_hip_calibrate_end(s, status, iters);
}

extern "C" int apemost_hip_calibrate_progress(apemost_hip_sampler *s, double *rows, int32_t capacity_rows,
                                              int32_t *n_rows) {
    CHECK_S(s);
    auto &k = s->cal;
    if (k.open || !n_rows || capacity_rows < 0 || (capacity_rows > 0 && !rows))
        return fail(APEMOST_HIP_ERR_INVALID, "calibrate_progress: bad arguments, or a calibration is still open");
    *n_rows = 0;
    if (k.progress_slot < 0 || !k.h_rec || k.progress_slot >= k.count)
        return APEMOST_HIP_OK;
    long long n = (long long)(k.h_rec[k.progress_slot].sweeps / k.cfg.iter_readjust);
    const CalibRec &r = k.h_rec[k.progress_slot];
    if (r.stage == CAL_ALL) // the sweeps of the last readjustment were not followed by their steps
        n -= 1;
    if (r.stage == CAL_DONE && r.status == 1)
        n -= 1;
    if (n > k.progress_cap)
        n = k.progress_cap;
    if (n < 0)
        n = 0;
    *n_rows = (int32_t)n;
    const long long take = n < capacity_rows ? n : capacity_rows;
    if (take > 0)
        HIP_TRY(hipMemcpy(rows, k.d_progress, (size_t)take * (1 + 2 * s->cfg.n_par) * sizeof(double), hipMemcpyDeviceToHost));
    return APEMOST_HIP_OK;
}

// segments and launches per workgroup shape of the latest calibration (bench, tests)
extern "C" int apemost_hip_calibrate_stats(apemost_hip_sampler *s, uint64_t *segments, uint64_t *evaluations,
                                           uint64_t launches_by_waves[9], double seconds_by_waves[9]) {
    CHECK_S(s);
    auto &k = s->cal;
    if (segments)
        *segments = k.segments;
    if (evaluations) {
        u64 sum = 0;
        for (int i = 0; k.h_rec && i < k.count; i++)
            sum += k.h_rec[i].evals;
        *evaluations = sum;
    }
    if (launches_by_waves)
        memcpy(launches_by_waves, k.launches_by_waves, sizeof k.launches_by_waves);
    if (seconds_by_waves)
        memcpy(seconds_by_waves, k.seconds_by_waves, sizeof k.seconds_by_waves);
    return APEMOST_HIP_OK;
}

static int rng_device(int device) {
    int rc = apemost_hip_device_info(device, nullptr, 0, nullptr, nullptr);
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(device));
    return APEMOST_HIP_OK;
}

// device scratch of the two test hooks, released on every path
struct DeviceScratch {
    std::vector<void *> blocks;
    ~DeviceScratch() {
        for (void *p : blocks)
            hipFree(p);
    }
    template <class T>
    hipError_t get(T **p, size_t count) {
        void *q = nullptr;
        const hipError_t e = hipMalloc(&q, count * sizeof(T));
        if (e == hipSuccess)
            blocks.push_back(q);
        *p = (T *)q;
        return e;
    }
};

extern "C" int apemost_hip_rng_raw(int device, uint64_t seed, uint64_t subsequence, uint64_t offset,
                                   int32_t n, uint32_t *out) {
    int rc = rng_device(device);
    if (rc)
        return rc;
    if (n < 1 || !out)
        return fail(APEMOST_HIP_ERR_INVALID, "rng_raw: bad arguments");
    DeviceScratch mem;
    unsigned int *d;
    HIP_TRY(mem.get(&d, n));
    hipLaunchKernelGGL(rng_raw_kernel, dim3(1), dim3(kWave), 0, 0, seed, subsequence, offset, n, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d, n * sizeof(unsigned int), hipMemcpyDeviceToHost));
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_rng_attempts(int device, uint64_t seed, uint64_t chain, int32_t slot, uint64_t tick,
                                        uint64_t q0, int32_t n, double *y, double *s, int32_t *valid,
                                        double *accept_log_u) {
    int rc = rng_device(device);
    if (rc)
        return rc;
    if (n < 1 || !y || !s || !valid)
        return fail(APEMOST_HIP_ERR_INVALID, "rng_attempts: bad arguments");
    DeviceScratch mem;
    double *dy, *ds, *dl;
    int *dv;
    HIP_TRY(mem.get(&dy, n));
    HIP_TRY(mem.get(&ds, n));
    HIP_TRY(mem.get(&dl, 1));
    HIP_TRY(mem.get(&dv, n));
    hipLaunchKernelGGL(rng_attempts_kernel, dim3((n + 63) / 64), dim3(kWave), 0, 0, seed, chain, slot, tick, q0, n,
                       dy, ds, dv, dl);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(y, dy, n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(s, ds, n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(valid, dv, n * sizeof(int), hipMemcpyDeviceToHost));
    if (accept_log_u)
        HIP_TRY(hipMemcpy(accept_log_u, dl, sizeof(double), hipMemcpyDeviceToHost));
    return APEMOST_HIP_OK;
}

#ifdef APEMOST_STAMPS
// diagnostic build only: read and clear the per-segment cycle sums of workgroup 0
extern "C" int apemost_hip_debug_stamps(unsigned long long *out16) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), 16 * sizeof(unsigned long long)));
    unsigned long long zero[16] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof zero));
    return APEMOST_HIP_OK;
}
// steps x 16 waves x points s_memtime values of workgroup 0 (zero = stamp not reached)
extern "C" int apemost_hip_debug_timeline(unsigned long long *out, int *steps, int *points) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * kTimelineSteps * 16 * kTimelinePoints));
    *steps = kTimelineSteps;
    *points = kTimelinePoints;
    return APEMOST_HIP_OK;
}
#endif

extern "C" int apemost_hip_timer_begin(apemost_hip_sampler *s) {
    CHECK_S(s);
    s->launches_at_begin = s->launches;
    HIP_TRY(hipEventRecord(s->ev0, s->stream));
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_timer_end(apemost_hip_sampler *s, float *elapsed_ms, uint64_t *launches) {
    CHECK_S(s);
    HIP_TRY(hipEventRecord(s->ev1, s->stream));
    HIP_TRY(hipEventSynchronize(s->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    if (elapsed_ms)
        *elapsed_ms = ms;
    if (launches)
        *launches = s->launches - s->launches_at_begin;
    return APEMOST_HIP_OK;
}

// A build from this file alone (development and diagnostic builds: apemost_amd/build.py build_dev,
// build_stamps) holds every model's kernels itself.
#ifdef APEMOST_SINGLE_TU
namespace apemost {
template hipError_t model_dispatch<0>(int, const AnyOp &);
template hipError_t model_dispatch<1>(int, const AnyOp &);
template hipError_t model_dispatch<2>(int, const AnyOp &);
template hipError_t model_dispatch<3>(int, const AnyOp &);
template hipError_t model_dispatch<8>(int, const AnyOp &);
template hipError_t model_dispatch<9>(int, const AnyOp &);
template hipError_t model_dispatch<10>(int, const AnyOp &);
template hipError_t model_dispatch<11>(int, const AnyOp &);
} // namespace apemost
#endif
