// pt_onebarrier.h -- the round kernel for ladders of few chains: one barrier per Metropolis step.
//
// When a ladder has fewer chains than the chip has CUs (BASELINE config 2: 128 chains), a chain's
// step is a latency chain and the throughput is chains / step latency.  The step of pt_device.h's
// Engine is two phases separated by two barriers: every wave evaluates its slice of the data
// vector (A -> B), then wave 0 alone adds the partial sums, finishes the likelihood, runs the
// accept test, the bookkeeping and the next proposal, and hands the proposal to the other waves
// through LDS (B -> A); the second phase is ~1300 of the ~2300 cycles of a step
// (profiles/r02_c2_stamps_classic.txt).
//
// Here the serial phase is taken off the critical path:
//
//   * The accept test (src/markov_chain.c:282-311) becomes ONE comparison on the raw data sum.
//     accept <=> prob_new >= prob or ln U < prob_new - prob <=> prob_new > prob + ln U (ln U < 0),
//     and prob_new is a decreasing function of the data sum S, so accept <=> S < S_max where
//     S_max is computed from (prob, ln U, the proposal's prior) while the likelihood is still
//     being evaluated.  The two forms differ only when prob_new - prob and ln U agree to ~1 ulp.
//   * Both possible next proposals are prepared ahead: the proposal of step t+1 starts from the
//     proposal of step t if that is accepted and from the current point if not, and everything
//     else it needs (the Gaussian candidates of tick t+1, the step widths, the prior box) is known
//     while step t's likelihood runs.  Both variants are published in LDS before the barrier.
//   * After the barrier EVERY wave adds the partial sums itself, compares with S_max, selects its
//     variant of the proposal and goes straight into the next likelihood.  No second barrier, no
//     hand-over.
//
// Roles in a workgroup of LW + 4 wavefronts: waves 0..LW-1 evaluate the likelihood; wave LW (the
// owner) keeps the chain: counters, current point, best point, sample rows, thresholds and the
// speculative proposals -- all one step behind the likelihood waves, in their shadow; waves
// LW+1..LW+3 produce the Gaussian candidates, each one set every three steps, one third per step
// (Philox block + polar test | logarithm | division, square root): a whole set is a ~2300-cycle
// dependent chain, longer than a step, a third of it is not.
//
// Draws, proposals, sums (same tree order as Engine with the same number of likelihood waves) and
// recorded values are those of Engine; only the form of the accept comparison differs.
#pragma once

#include "pt_device.h"

namespace apemost {

// LDS carve (doubles); [2] = parity of the step that published the entry
constexpr int kObPart = 0;                        // [2][16] wave partials of the data sum
constexpr int kObThr = 32;                        // [2]     S_max of the step the partials belong to
constexpr int kObFlag = 34;                       // [2]     != 0: a prepared proposal needs the redraw path
constexpr int kObCtl = 36;                        // 4 control words (word 2 of the int view: Engine's fail flag slot)
constexpr int kObThx = 40;                        // [2]     models with a prior: the helper's half of S_max (kObThr then holds the owner's half)
constexpr int kObPri = 42;                        // [2]     ... and the prior of the proposal that half belongs to
constexpr int kObProp = 44;                       // [2][2][64] proposals: [parity][0 = after accept, 1 = after reject][parameter]
constexpr int kObCand = kObProp + 2 * 2 * kWave;  // [8][64] double2 candidate ring
constexpr int kObLogTab = kObCand + 8 * 2 * kWave;  // the logarithm's table (pulse models)
constexpr int kObFixedDoubles = kObLogTab + kLogTabLdsDoubles;

// S_max of a step: accept <=> data sum < S_max.  T = prob + ln U.
template <int MODEL>
struct ObThreshold;

// kSplit (the models with a prior): S_max = X - Y in two halves that different wavefronts compute side by
// side -- X from the proposal alone (its prior and additive parameter: the helper, a candidate producer), Y
// from the chain (T: the owner) -- and every wave subtracts for itself behind the barrier.  The prior's
// logarithm was 616 of the 2194 ticks the owner was busy per step at config 4, with everybody waiting for
// it (profiles/r03_ob_owner_segments.txt); it depends on nothing the owner knows that the published rows
// and the decision every wave makes anyway do not say (round 4).
template <>
struct ObThreshold<APEMOST_MODEL_SIMPLESIN> {
    double denom_over_beta; // (-2 sigma^2) / beta < 0
    __device__ __forceinline__ void init(const ModelConsts &c, double beta) {
        denom_over_beta = (-2 * c.sigma * c.sigma) / beta;
    }
    // prob_new = beta S / denom > T  <=>  S < T denom / beta
    template <class M>
    __device__ __forceinline__ double s_max(double T, const M &m, double, double) const {
        const double s = T * denom_over_beta;
        return m.nan_hi ? -__builtin_inf() : s; // an argument outside the sine's range: prob_new is NaN, never accepted
    }
};
template <>
struct ObThreshold<APEMOST_MODEL_SINE3> : ObThreshold<APEMOST_MODEL_SIMPLESIN> {};

// The models with a prior run with a HELPER wavefront where a workgroup has a CU to itself (ObEngine's HELPER
// argument, chosen by the host): S_max = X - Y, the proposal's half X from the helper, the chain's half Y from the
// owner.  History (profiles/r04_ob_helper_roles.txt, r04_pulse_helper_wave.txt): with the prior on the owner config 4's
// shard ran 2.30e8 steps/s and a helper changed nothing (2.28e8) while the likelihood waves' step still carried the
// guard's branch and three empty per-lane loops; with those gone (2.39e8) the owner is what the step waits for, and
// the ninth wavefront gives 2.48e8 and a calibration of 0.33 instead of 0.40 s.  A candidate producer as helper (its
// Philox step becomes the longest wave of the workgroup: 2.14-2.16e8) is not an option and no longer in the source.
template <>
struct ObThreshold<APEMOST_MODEL_PULSE> {
    double inv_beta;
    __device__ __forceinline__ void init(const ModelConsts &, double beta) { inv_beta = 1.0 / beta; }
    // prob_new = prior + -beta (p1 + S) > T  <=>  S < (prior - T) / beta - p1 = (prior / beta - p1) - T / beta
    template <class M>
    __device__ __forceinline__ double s_max(double T, const M &, double prior_new, double p1) const { // (no helper: the owner alone)
        return (prior_new - T) * inv_beta - p1;
    }
    __device__ __forceinline__ double x_part(double prior_new, double p1) const { return prior_new * inv_beta - p1; }
    __device__ __forceinline__ double y_part(double T) const { return T * inv_beta; }
    static __device__ __forceinline__ double limit(double x, double y) { return x - y; }
};
template <>
struct ObThreshold<APEMOST_MODEL_PULSE_VROT> : ObThreshold<APEMOST_MODEL_PULSE> {};

// Diagnostic build (-DAPEMOST_STAMPS): the owner's step in segments, s_memtime ticks between the points it
// reaches (workgroup 0; g_stamps[LW + 4 ...], tools/ob_profile.py).  A point is reached when everything
// before it has issued, waits for operands included.
#ifdef APEMOST_STAMPS
#define OB_SEG(i)                                                                                 \
    do {                                                                                          \
        const u64 now__ = __builtin_amdgcn_s_memtime();                                           \
        seg_acc[i] += now__ - seg_last;                                                           \
        seg_last = now__;                                                                         \
    } while (0)
#else
#define OB_SEG(i)                                                                                 \
    do {                                                                                          \
    } while (0)
#endif

#ifndef APEMOST_OB_WAVE_PERM
#define APEMOST_OB_WAVE_PERM 0
#endif
// The likelihood waves' step, read in the code object (round 4, profiles/r04_lik_step_microfixes.txt): measured on, kept on.
#ifndef APEMOST_OB_SKIP_LOOPS
#define APEMOST_OB_SKIP_LOOPS 1
#endif
#ifndef APEMOST_OB_PART_FIRST
#define APEMOST_OB_PART_FIRST 1
#endif
#ifndef APEMOST_OB_FLAG_LATE
#define APEMOST_OB_FLAG_LATE 0
#endif
// HELPER: a ninth (LW + 5th) wavefront that computes the proposal's half of the threshold (the prior's logarithms);
// for the models with a prior only, and only where the host finds one workgroup per CU (nine-wave workgroups do not
// fit a CU twice under the kernel's register budget: ladders of 257-512 chains keep the eight-wave form).
__host__ __device__ constexpr bool ob_can_help(int model) {
    return model % kVariantModel == APEMOST_MODEL_PULSE || model % kVariantModel == APEMOST_MODEL_PULSE_VROT;
}
// (experiment, round 4: APEMOST_OB_HELPER_SIMD = s puts s wavefronts that end at once between the producers and the
// helper, so that the helper -- wave w runs on SIMD w mod 4 -- lands on SIMD s instead of beside likelihood wave 0 and the owner)
#ifndef APEMOST_OB_HELPER_SIMD
#define APEMOST_OB_HELPER_SIMD 0
#endif
__host__ __device__ constexpr int ob_block(int lw, bool helper) { return (lw + 4 + (helper ? 1 + APEMOST_OB_HELPER_SIMD : 0)) * kWave; }

// Register budget of the one-barrier kernels: waves per SIMD the compiler must leave room for.  Eight- and twelve-wave
// workgroups share a CU two by two (4 per SIMD: 128 registers); the nine-wave ones (four likelihood waves + helper)
// have a CU to themselves, three waves on SIMD 0: 168 registers, the calibration kernel stops spilling (12-13
// registers at 128) -- config 4 2.668 -> 2.697e8 steps/s, calibration 0.328 -> 0.320 s (profiles/r04_pulse_helper_wave.txt).
#ifndef APEMOST_OB_HELPER_EU
#define APEMOST_OB_HELPER_EU 3
#endif
__host__ __device__ constexpr int ob_waves_per_eu(int lw, bool helper) {
    return helper && lw == 4 ? APEMOST_OB_HELPER_EU : APEMOST_OB_WAVES_PER_EU;
}

template <int MODEL, int LW, bool LDS_DATA, bool HELPER = false>
struct ObEngine {
#ifdef APEMOST_STAMPS
    u64 seg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64 seg_last = 0;
#endif
    static constexpr int kLikThreads = LW * kWave;
    static constexpr int kProducers = 3;
    static_assert(!HELPER || ob_can_help(MODEL), "a helper wavefront is for the models with a prior");
    static constexpr int kBlock = ob_block(LW, HELPER);
    static constexpr bool kHelperWave = HELPER; // the proposal's half of the threshold comes from a wavefront of its own
    static constexpr int kWide = LW < 8 ? 4 : 2;
    static constexpr int kShortChain = LW >= APEMOST_SHORT_CHAIN_WAVES ? 1 : LW >= APEMOST_EVEN_ODD_WAVES ? 2 : 0;
    static constexpr bool kVariants = MODEL >= kVariantModel; // see kVariantModel (pt_device.h)
    static constexpr int kBase = MODEL % kVariantModel;
    static constexpr bool kSine = kBase == APEMOST_MODEL_SIMPLESIN || kBase == APEMOST_MODEL_SINE3;
#ifndef APEMOST_PHILOX_MERGED
#define APEMOST_PHILOX_MERGED 2 // 0: never, 1: always, 2: the sine models
#endif
    static constexpr bool kMergedPhilox = APEMOST_PHILOX_MERGED == 1 || (APEMOST_PHILOX_MERGED == 2 && kSine);

    // ---- identity ----
    int lane, hw;  // lane, hardware wave index in the workgroup
    int wave;      // 0 = owner: the name and value the shared chain_load / chain_store / swap helpers test
    int tid;       // likelihood thread index (likelihood waves)
    __device__ __forceinline__ bool is_lik() const { return hw < LW; }
    __device__ __forceinline__ bool is_owner() const { return hw == LW; }
    __device__ __forceinline__ bool is_producer() const { return hw > LW && hw <= LW + kProducers; }
    __device__ __forceinline__ bool is_helper() const { return kHelperWave && hw == LW + kProducers + 1 + APEMOST_OB_HELPER_SIMD; }

    int n_par, n_data;
    int Q, grp, qidx, n_cand_lanes;
    u64 grpmask, lowmask;
    __device__ __forceinline__ bool cand() const { return lane < n_cand_lanes; }
    ModelConsts consts;
    u64 circular, seed, g;
    const double *xs, *ys;
    double *lds;
    u64 tick; // tick of the step in flight (uniform in the workgroup)
    double beta_all, x_abs_max;
    Model<kBase> m;
    ObThreshold<kBase> thr_fn;

    // ---- owner: the chain (same meaning as Engine's fields) ----
    double cur, best, stepw, lo, hi;
    u64 pacc, prej;
    double prob, prior, prob_best;
    u64 accept, reject;
    double par_val;     // the step in flight proposes this value for my parameter
    double sel_a, sel_r; // ... and the next step would, after an accept / after a reject
    double thr;         // S_max of the step in flight
    double prior_inflight; // prior of the proposal in flight (models with a prior)
    u64 fail_a, fail_r; // parameter groups whose prepared attempts all failed, per variant
    double cand_y, cand_s, next_y, next_s; // candidates of the tick in flight / of the next one

    // ---- likelihood waves: data rows kept in registers when the vector is one pass ----
    bool rows_in_regs;
    double row_x[kWide], row_y[kWide];

    __device__ __forceinline__ double *s_part(int parity) const { return lds + kObPart + parity * 16; }
    __device__ __forceinline__ double *s_thr(int parity) const { return lds + kObThr + parity; }
    __device__ __forceinline__ double *s_thx(int parity) const { return lds + kObThx + parity; }
    __device__ __forceinline__ double *s_pri(int parity) const { return lds + kObPri + parity; }
    static constexpr bool kSplit = HELPER; // S_max = X - Y (see ObThreshold)
    __device__ __forceinline__ int *s_flag(int parity) const { return (int *)(lds + kObFlag + parity); }
    // uniform by construction (every lane reads the same word): say so, the branch on it guards a barrier
    __device__ __forceinline__ bool redraw_pending(int parity) const {
        return __builtin_amdgcn_readfirstlane(*s_flag(parity)) != 0;
    }
    __device__ __forceinline__ double *s_prop(int parity, int variant) const {
        return lds + kObProp + (parity * 2 + variant) * kWave;
    }
    __device__ __forceinline__ double2 *s_cand(u64 t) const { return (double2 *)(lds + kObCand) + (int)(t & 7) * kWave; }
    __device__ __forceinline__ volatile int *fail_flag() const { return (volatile int *)(lds + kObCtl) + 2; }

    // every wave: who am I, where is the data; stages the data vector (all threads copy)
    __device__ __forceinline__ void setup_common(const DevArrays &d, const ChainShape &sh, int c, double *lds_) {
        lane = threadIdx.x & (kWave - 1);
        hw = threadIdx.x / kWave;
#if APEMOST_OB_WAVE_PERM
        // (experiment, round 4) which role a hardware wave takes, i.e. which roles share a SIMD (wave w runs on
        // SIMD w mod 4): 1 = owner + a producer on SIMD 0, two likelihood waves on SIMD 1; 2 = owner + producer,
        // likelihood waves in two pairs, two producers together.  One nibble per hardware wave, LW = 4 only.
        if (LW == 4 && !kHelperWave)
            hw = (int)(((APEMOST_OB_WAVE_PERM == 1 ? 0x76352104u : 0x73256104u) >> (4 * hw)) & 0xFu);
#endif
        wave = hw == LW ? 0 : hw + 1;
        tid = hw * kWave + lane;
        n_par = sh.n_par;
        n_data = sh.n_data;
        lds = lds_;
        tick = d.ticks()[c];
        if (threadIdx.x == 0) {
            *fail_flag() = 0;
            *s_flag(0) = 0;
            *s_flag(1) = 0;
        }
        if constexpr (Model<kBase>::kUsesLogTable) { // (published by the kernel's first barrier)
            stage_logtab(lds + kObLogTab, (int)threadIdx.x, kBlock);
            m.set_logtab(lds + kObLogTab);
        }
        double *s_data = lds + kObFixedDoubles;
        if (LDS_DATA) {
            for (int i = threadIdx.x; i < 2 * sh.n_data; i += kBlock)
                s_data[i] = d.data[i];
            xs = s_data;
            ys = s_data + sh.n_data;
        } else {
            xs = d.data;
            ys = d.data + sh.n_data;
        }
    }
    // owner and producer: the (parameter, attempt) lane layout and the RNG address
    __device__ __forceinline__ void setup_lanes(const ChainShape &sh, int c) {
        circular = sh.circular;
        seed = sh.seed;
        g = (u64)(sh.chain_offset + c);
        Q = 63 / n_par;
        grp = lane / Q;
        qidx = lane - grp * Q;
        n_cand_lanes = n_par * Q;
        lowmask = grpmask = ~0ull;
        if (cand()) {
            grpmask = ((1ull << Q) - 1ull) << (grp * Q);
            lowmask = grpmask & ((1ull << lane) - 1ull);
        } else {
            grp = 0;
            qidx = 1 << 30;
        }
    }
    __device__ __forceinline__ void setup_lik(const DevArrays &, const ChainShape &sh, int c) {
        setup_lanes(sh, c); // (the first two candidate sets are made by likelihood waves)
        consts = sh.consts;
        x_abs_max = sh.x_abs_max;
        rows_in_regs = false;
        m.init();
        m.set_consts(consts);
    }
    __device__ __forceinline__ void setup_owner(const DevArrays &d, const ChainShape &sh, int c) {
        consts = sh.consts;
        x_abs_max = sh.x_abs_max;
        fail_a = fail_r = 0;
        cand_y = cand_s = next_y = next_s = 0;
        par_val = thr = sel_a = sel_r = prior_inflight = 0;
        accepted = false;
        n_accepted = 0;
        m.init_scalar();
        if constexpr (Model<kBase>::kUsesLogTable)
            m.set_lean(false);
        m.set_consts(consts);
        m.set_box(d.pmin() + (size_t)c * sh.n_par, d.pmax() + (size_t)c * sh.n_par, sh.x_abs_max);
    }

    // ---- candidates (same arithmetic as Engine::cand_begin / cand_finish) ----
    struct Half {
        double y, v; // attempt lanes: second polar coordinate and r^2; lane 63: (unused, the uniform)
        bool ok;
    };
    // ONE Philox evaluation for the whole wavefront: the attempt lanes' blocks and the accept uniform's (lane 63)
    // differ in their address only.  (Until round 4 each had its own call under its own branch, and a wavefront runs
    // the two sides of a divergent branch one after the other: twice the ten rounds, 43 v_mad_u64_u32 and 103 v_xor
    // in the producers' step -- whose Philox phase, 1313-1344 ticks, was as long as the owner's step at config 2.)
    __device__ __forceinline__ Half cand_begin(u64 t) const {
        if constexpr (kMergedPhilox) {
            Half h;
            const bool uni = lane == 63;
            const uint4 b = philox_block(seed, g * APEMOST_HIP_STREAMS_PER_CHAIN + (u64)(uni ? n_par : grp),
                                         (t << kTickShift) | (u64)(uni ? 0 : qidx));
            const double u0 = u32_to_uniform(b.x);
            const double x = -1 + 2 * u0;
            double y = -1 + 2 * u32_to_uniform(b.y);
            double v = x * x + y * y;
            bool ok = b.x != 0 && b.y != 0 && !(v > 1.0 || v == 0);
            if (kVariants && proposal_law(circular) != kProposalGaussian) { // uniform; see Engine::cand_begin
                const bool logistic = proposal_law(circular) == kProposalLogistic;
                y = u0;
                v = logistic ? u0 / (1 - u0) : 1.0;
                ok = logistic ? b.x != 0 : true;
            }
            h.y = cand() ? y : 0.0;
            h.v = cand() ? v : uni ? u0 : 0.0;
            h.ok = cand() && ok;
            return h;
        } else {
            Half h;
            h.y = h.v = 0;
            h.ok = false;
            if (cand()) {
                const uint4 b = philox_block(seed, g * APEMOST_HIP_STREAMS_PER_CHAIN + (u64)grp, (t << kTickShift) | (u64)qidx);
                const double x = -1 + 2 * u32_to_uniform(b.x);
                h.y = -1 + 2 * u32_to_uniform(b.y);
                h.v = x * x + h.y * h.y;
                h.ok = b.x != 0 && b.y != 0 && !(h.v > 1.0 || h.v == 0);
                if (kVariants && proposal_law(circular) != kProposalGaussian) { // uniform; see Engine::cand_begin
                    const double x0 = u32_to_uniform(b.x);
                    const bool logistic = proposal_law(circular) == kProposalLogistic;
                    h.y = x0;
                    h.v = logistic ? x0 / (1 - x0) : 1.0;
                    h.ok = logistic ? b.x != 0 : true;
                }
            } else if (lane == 63) {
                const uint4 b = philox_block(seed, g * APEMOST_HIP_STREAMS_PER_CHAIN + (u64)n_par, t << kTickShift);
                h.v = u32_to_uniform(b.x);
            }
            return h;
        }
    }
    // ln r^2 for the attempt lanes, ln u for lane 63 (one field for both: a select between two
    // fields of a struct turns into a computed address and the whole engine into scratch memory)
    __device__ __forceinline__ double cand_log(const Half &h) const { return log(h.v); }
    __device__ __forceinline__ double2 cand_end(const Half &h, double lg) const {
        double2 c;
        c.x = c.y = 0;
        if (cand()) {
            const double sq = sqrt(-2.0 * lg / h.v);
            c.x = h.y;
            c.y = h.ok ? sq : __builtin_nan("");
            if (kVariants && proposal_law(circular) != kProposalGaussian) { // uniform; see Engine::cand_pair
                if (proposal_law(circular) == kProposalLogistic)
                    c.x = lg;
                c.y = h.ok ? 1.0 : __builtin_nan("");
            }
        } else if (lane == 63) {
            c.x = lg;
        }
        return c;
    }
    __device__ __forceinline__ void make_set(u64 t) {
        const Half h = cand_begin(t);
        s_cand(t)[lane] = cand_end(h, cand_log(h));
    }

    // Producer wave j works on the ticks T = t0 + 2 + j (mod 3), one third of the set per step,
    // timed so that set T is published by the barrier that closes step T - 2 (the owner fetches the
    // candidates of tick t + 1 while step t is in flight):  step T-4: Philox blocks and polar test,
    // step T-3: logarithm, step T-2: division, square root, store.
    Half pipe;
    double pipe_log;
    u64 pipe_tick;
    int pipe_phase; // what the coming step does: 0 blocks, 1 logarithm, 2 finish
    // before the kernel's first barrier (likelihood waves 0 and 1 make the sets of ticks t0 and
    // t0+1 meanwhile): set t0+2 is two thirds done, set t0+3 one third, set t0+4 not begun
    __device__ __forceinline__ void producer_prologue() {
        const int j = hw - LW - 1;
        pipe_tick = tick + 2 + (u64)j;
        pipe_phase = 2 - j;
        pipe.y = pipe.v = pipe_log = 0;
        pipe.ok = false;
        if (j <= 1)
            pipe = cand_begin(pipe_tick);
        if (j == 0)
            pipe_log = cand_log(pipe);
    }
    // The helper's duty (HELPER): decide the step that just ended like every other wave, take the prior of the
    // proposal now in flight -- the row that decision selects -- and publish the proposal's half of S_max with it.
    // `parity` as in lik_step.
    __device__ __forceinline__ void setup_helper(const DevArrays &d, const ChainShape &sh, int c) {
        consts = sh.consts;
        beta_all = d.beta()[c + 1];
        thr_fn.init(consts, beta_all);
        m.init_scalar();
        if constexpr (Model<kBase>::kUsesLogTable)
            m.set_lean(false);
        m.set_consts(consts);
    }
    // The priors of BOTH prepared rows, requested with the step's first LDS reads and taken side by side
    // (Model::prior_two), the decision beside them, a select at the end: the decision's LDS round trip, tree and
    // compare (~300 ticks) are not in front of the logarithm's chain.
    __device__ __forceinline__ void helper_step(int parity) {
        const double *ra = s_prop(parity, 0), *rr = s_prop(parity, 1);
        double part[LW];
        const double *sp = s_part(parity);
#pragma unroll
        for (int w = 0; w < LW; w++)
            part[w] = sp[w];
        const double limit = ObThreshold<kBase>::limit(*s_thx(parity), *s_thr(parity));
        const double p1a = ra[1], p1r = rr[1];
        double pa, pr;
        m.prior_two(ra, rr, n_par, consts, pa, pr);
#pragma unroll
        for (int span = 1; span < LW; span *= 2) {
#pragma unroll
            for (int w = 0; w + span < LW; w += 2 * span)
                part[w] += part[w + span];
        }
        const bool first = part[0] < limit;
        const double prior_k = first ? pa : pr;
        const double x = thr_fn.x_part(prior_k, first ? p1a : p1r);
        if (lane == 0) {
            *s_thx(parity ^ 1) = x;
            *s_pri(parity ^ 1) = prior_k;
        }
    }
    __device__ __forceinline__ void producer_step(int parity) {
        if (pipe_phase == 0) {
            pipe = cand_begin(pipe_tick);
            pipe_phase = 1;
        } else if (pipe_phase == 1) {
            pipe_log = cand_log(pipe);
            pipe_phase = 2;
        } else {
            s_cand(pipe_tick)[lane] = cand_end(pipe, pipe_log);
            pipe_tick += kProducers;
            pipe_phase = 0;
        }
    }

    // ---- the partial sums of a step, added in Engine's order ----
    __device__ __forceinline__ double tree(int parity) const {
        double part[LW];
        const double *sp = s_part(parity);
#pragma unroll
        for (int w = 0; w < LW; w++)
            part[w] = sp[w];
#pragma unroll
        for (int span = 1; span < LW; span *= 2) {
#pragma unroll
            for (int w = 0; w + span < LW; w += 2 * span)
                part[w] += part[w + span];
        }
        return part[0];
    }

    // ---- likelihood waves ----
    __device__ __forceinline__ void cache_rows() {
        rows_in_regs = n_data == kWide * kLikThreads;
#pragma unroll
        for (int j = 0; j < kWide; j++) {
            row_x[j] = rows_in_regs ? xs[tid + j * kLikThreads] : 0.0;
            row_y[j] = rows_in_regs ? ys[tid + j * kLikThreads] : 0.0;
        }
    }

    // the wave's share of sum_i term(x_i, y_i) at the loaded parameters (Engine::reduce_data's loops)
    __device__ __forceinline__ double lik_partial() {
        double acc = 0;
        typename Model<kBase>::LogAcc lp; // (kLogProduct models: the lane's prod y; unused otherwise)
        lp.init();
        int i = tid;
        if (rows_in_regs) {
            add_terms<kWide, kShortChain, false>(m, row_x, row_y, acc, lp);
            i = n_data;
        }
#if APEMOST_OB_SKIP_LOOPS
        // The whole vector is one pass held in registers (configs 2 and 4): all three loops below are empty -- but each
        // loop test is per lane (v_cmp -> s_and_saveexec -> s_cbranch_execz -> s_or exec) and sat on the step's dependent
        // chain; ONE uniform test in front of them: config 2 2.078 -> 2.141e8 steps/s (+3 %), config 4 +0.8 %.
        if (!rows_in_regs)
#endif
        {
        if (LW < 8) {
            for (; i + 3 * kLikThreads < n_data; i += 4 * kLikThreads) {
                const double x4[4] = {xs[i], xs[i + kLikThreads], xs[i + 2 * kLikThreads], xs[i + 3 * kLikThreads]};
                const double y4[4] = {ys[i], ys[i + kLikThreads], ys[i + 2 * kLikThreads], ys[i + 3 * kLikThreads]};
                add_terms<4, kShortChain, false>(m, x4, y4, acc, lp);
                if constexpr (Model<kBase>::kLogProduct)
                    lp.renorm();
            }
        }
        for (; i + kLikThreads < n_data; i += 2 * kLikThreads) {
            const double x2[2] = {xs[i], xs[i + kLikThreads]};
            const double y2[2] = {ys[i], ys[i + kLikThreads]};
            add_terms<2, kShortChain, false>(m, x2, y2, acc, lp);
            if constexpr (Model<kBase>::kLogProduct)
                lp.renorm();
        }
        for (; i < n_data; i += kLikThreads) {
            const double x1[1] = {xs[i]}, y1[1] = {ys[i]};
            if constexpr (Model<kBase>::kLogProduct) {
                add_terms<1, kShortChain, false>(m, x1, y1, acc, lp);
                lp.renorm();
            } else if constexpr (Model<kBase>::kFusedAcc)
                add_terms<1, kShortChain, false>(m, x1, y1, acc, lp);
            else
                acc += m.term(xs[i], ys[i]);
        }
        }
        if constexpr (Model<kBase>::kLogProduct) {
            acc += lp.template finish<false>(m.tab);
            // rare: a pair product next to the subnormals or a sum that is not finite -- the wave takes its
            // share again in the reference's operation order (LogProdT, Model::term_ref).  The test sits BEHIND
            // the cross-lane sum of the fast form: the ballot and its branch are then off the step's dependent
            // chain (the branch is decided while the six DPP stages run) instead of in front of it.
            const u64 suspect = __ballot(lp.bad(acc));
            double total = wave_reduce_sum_lane63(acc);
            if (suspect != 0) {
                acc = 0;
                for (int k = tid; k < n_data; k += kLikThreads)
                    acc += m.term_ref(xs[k], ys[k]);
                total = wave_reduce_sum_lane63(acc);
            }
            return total; // valid in lane 63
        }
        return wave_reduce_sum_lane63(acc); // valid in lane 63
    }

    // One step of a likelihood wave.  `parity` holds what the previous step published: its partial
    // sums and threshold, and the two prepared proposals of this step.
    __device__ __forceinline__ void lik_step(int parity) {
        // everything the decision needs is requested at once: the partial sums, the threshold, the
        // redraw flag and BOTH prepared proposals (selecting the row first and reading it afterwards
        // would put a second LDS round trip on the critical path)
        double part[LW];
        const double *sp = s_part(parity);
#pragma unroll
        for (int w = 0; w < LW; w++)
            part[w] = sp[w];
#if APEMOST_OB_PART_FIRST
        // The partial sums FIRST in the LDS queue: the add tree is the head of the step's chain, the proposal rows are
        // not needed before the select behind it (left to itself the compiler requests the threshold, the flag and the
        // four row reads ahead of them).  Worth nothing by itself beyond the skipped loops; kept with mov_dpp, with
        // which it is the fastest combination measured (2.141 / 2.394e8 at configs 2 / 4).
        asm volatile("" ::: "memory");
#endif
        double limit = *s_thr(parity);
        if constexpr (kSplit)
            limit = ObThreshold<kBase>::limit(*s_thx(parity), limit); // (one subtraction beside the partial sums' tree)
        const int pending = *s_flag(parity);
        m.fetch2(s_prop(parity, 0), s_prop(parity, 1));
#if !APEMOST_OB_FLAG_LATE
        if (__builtin_amdgcn_readfirstlane(pending) != 0) {
            // rare: the owner is replacing a proposal that could not be prepared (redraw path);
            // every wave of the workgroup takes this barrier, then the rows are read again
            __syncthreads();
            m.fetch2(s_prop(parity, 0), s_prop(parity, 1));
        }
#endif
#pragma unroll
        for (int span = 1; span < LW; span *= 2) {
#pragma unroll
            for (int w = 0; w + span < LW; w += 2 * span)
                part[w] += part[w + span];
        }
#if APEMOST_OB_FLAG_LATE
        // (measured and not taken, round 4: 2.06-2.14e8 at config 2 depending on what else is switched on, never above the
        // combinations without it) the redraw flag's test BEHIND the add tree: the decision needs the partial sums
        // and the threshold only, and a branch in front of the tree holds the wave until the flag has come back
        const bool first = part[0] < limit;
        asm volatile("" : "+v"(part[0]));
        if (__builtin_amdgcn_readfirstlane(pending) != 0) {
            __syncthreads();
            m.fetch2(s_prop(parity, 0), s_prop(parity, 1));
        }
        m.pick2(first, n_par);
#else
        m.pick2(part[0] < limit, n_par);
#endif
        const double mine = lik_partial();
        if (lane == 63)
            s_part(parity ^ 1)[hw] = mine;
    }

    // ---- owner ----
    // proposal attempts of the tick whose candidates are (cy, cs), from base point `from`:
    // first usable attempt per parameter -> LDS row `row`; returns the groups where none was usable
    // which >= 0 (markov_chain_step_for, src/markov_chain.c:317-333): only that parameter is proposed,
    // every other one keeps the base point's value
    __device__ __forceinline__ u64 attempts(double from, double cy, double cs, double *row, int which = -1) const {
        double prop = from + stepw * cy * cs;
        bool inside = !(prop > hi || prop < lo);
        if (circular != 0) { // uniform: a non-default proposal law, or some parameter is circular
            if (kVariants && proposal_law(circular) == kProposalFlat) {
                prop = from + flat_jump(stepw, cy);
                inside = !(prop > hi || prop < lo);
            }
            const bool wrap = !inside && ((circular >> grp) & 1);
            if (wrap)
                prop = wrap_circular(prop, lo, hi);
            inside = inside || wrap;
        }
        const bool active = cand() && (which < 0 || which == grp);
        const bool ok = active && (cs == cs) && inside;
        const u64 mask = __ballot(ok);
        if (ok && (mask & lowmask) == 0)
            row[grp] = prop;
        if (which >= 0 && cand() && !active && qidx == 0)
            row[grp] = from;
        return __ballot(active && qidx == 0 && (mask & grpmask) == 0);
    }
    // attempts() for both prepared proposals of the next step at once -- from `from_a` (the proposal in
    // flight) into row_a, from `from_r` (the current point) into row_r: the same arithmetic per variant,
    // but ONE uniform branch for the circular / proposal-law path instead of one per variant, and the two
    // variants' compare -> ballot -> predicated LDS write chains in one basic block, where the compiler
    // interleaves them (two calls are two regions, one behind the other)
    __device__ __forceinline__ void attempts2(double from_a, double from_r, double cy, double cs, double *row_a, double *row_r,
                                              int which, u64 &failed_a, u64 &failed_r) const {
        const double jump = stepw * cy * cs;
        double prop_a = from_a + jump, prop_r = from_r + jump;
        bool inside_a = !(prop_a > hi || prop_a < lo), inside_r = !(prop_r > hi || prop_r < lo);
        if (circular != 0) { // uniform: a non-default proposal law, or some parameter is circular
            if (kVariants && proposal_law(circular) == kProposalFlat) {
                const double fj = flat_jump(stepw, cy);
                prop_a = from_a + fj;
                prop_r = from_r + fj;
                inside_a = !(prop_a > hi || prop_a < lo);
                inside_r = !(prop_r > hi || prop_r < lo);
            }
            const bool circ = (circular >> grp) & 1;
            const bool wrap_a = !inside_a && circ, wrap_r = !inside_r && circ;
            if (wrap_a)
                prop_a = wrap_circular(prop_a, lo, hi);
            if (wrap_r)
                prop_r = wrap_circular(prop_r, lo, hi);
            inside_a = inside_a || wrap_a;
            inside_r = inside_r || wrap_r;
        }
        const bool active = cand() && (which < 0 || which == grp);
        const bool usable = active && (cs == cs);
        const bool ok_a = usable && inside_a, ok_r = usable && inside_r;
        const u64 mask_a = __ballot(ok_a), mask_r = __ballot(ok_r);
        if (ok_a && (mask_a & lowmask) == 0)
            row_a[grp] = prop_a;
        if (ok_r && (mask_r & lowmask) == 0)
            row_r[grp] = prop_r;
        if (which >= 0 && cand() && !active && qidx == 0) {
            row_a[grp] = from_a;
            row_r[grp] = from_r;
        }
        const bool leader = active && qidx == 0;
        failed_a = __ballot(leader && (mask_a & grpmask) == 0);
        failed_r = __ballot(leader && (mask_r & grpmask) == 0);
    }
    // Engine::propose's rare path: none of the Q prepared attempts of a parameter worked -> the
    // whole wave tries 64 more at a time (attempt indices continue at Q, as in the serial loop of
    // src/markov_chain.c:235-240)
    __device__ __forceinline__ void redraw(u64 failed, u64 t, double *row) {
        while (failed) {
            const int leader = __builtin_ctzll(failed);
            const int p = leader / Q;
            const double c0 = read_lane(cur, leader), w0 = read_lane(stepw, leader);
            const double lo0 = read_lane(lo, leader), hi0 = read_lane(hi, leader);
            for (unsigned qbase = (unsigned)Q;; qbase += kWave) {
                if (qbase >= (1u << kTickShift) - kWave) {
                    if (lane == 0) {
                        row[p] = c0;
                        *fail_flag() = 1;
                    }
                    break;
                }
                double jump;
                const bool v = jump_attempt(kVariants ? proposal_law(circular) : kProposalGaussian, seed, g, p, t,
                                            (u64)(qbase + (unsigned)lane),
                                            w0, jump);
                double pr = c0 + jump;
                bool inside = !(pr > hi0 || pr < lo0);
                if (!inside && ((circular >> p) & 1)) {
                    pr = wrap_circular(pr, lo0, hi0);
                    inside = true;
                }
                const u64 m2 = __ballot(v && inside);
                if (m2) {
                    if (lane == __builtin_ctzll(m2))
                        row[p] = pr;
                    break;
                }
            }
            failed &= failed - 1;
        }
    }

    // round start: the proposal of the first step from the current point (nothing to speculate on),
    // published as both variants; no partial sums yet, so the threshold slot says "take either"
    __device__ __forceinline__ void owner_first(int which = -1) {
        const double2 c0 = s_cand(tick)[lane];
        cand_y = c0.x;
        cand_s = c0.y;
        double *row = s_prop(0, 1);
        const u64 failed = attempts(cur, cand_y, cand_s, row, which);
        redraw(failed, tick, row);
        __builtin_amdgcn_wave_barrier();
        if (lane < n_par)
            s_prop(0, 0)[lane] = row[lane];
#ifdef APEMOST_EXP_NO_ATTEMPTS
        if (lane < n_par)
            s_prop(1, 0)[lane] = s_prop(1, 1)[lane] = row[lane];
#endif
        if (lane == 0) {
            *s_thr(0) = kSplit ? 0.0 : __builtin_inf();
            if constexpr (kSplit)
                *s_thx(0) = __builtin_inf(); // (X - Y = +inf: "take either")
            *s_flag(0) = 0;
        }
        if (lane < LW)
            s_part(0)[lane] = 0;
        fail_a = fail_r = 0;
        sel_a = sel_r = cand() ? row[grp] : 0.0;
    }

    // results of the step that just finished (tick - 1 from now on): check_accept through the
    // threshold, counters, mcmc_check_best, the sample row; then the rare redraw of the chosen variant
    bool accepted; // outcome of the step just finished (uniform)
    unsigned n_accepted; // accepted steps of this launch
    // accept / reject and the per-parameter counters after n_steps all-parameter steps
    // (inc_params_accepts / inc_params_rejects bump all of them together, src/mcmc_gettersetter.c:98-109)
    __device__ __forceinline__ void owner_settle_counters(u64 n_steps) {
        accept += n_accepted;
        reject += n_steps - n_accepted;
        pacc += n_accepted;
        prej += n_steps - n_accepted;
        n_accepted = 0;
    }
    // the calibration's block boundaries (same meaning as Engine's)
    __device__ __forceinline__ void check_best() {
        if (prob > prob_best) {
            prob_best = prob;
            best = cur;
        }
    }
    __device__ __forceinline__ void restart_from_best() {
        cur = best;
        prob = prob_best;
    }
    __device__ __forceinline__ void reset_accept_rejects() {
        pacc = prej = 0;
        accept = reject = 0;
    }
    // which >= 0: the step was a single-parameter update, which bumps that parameter's counters only
    // (quirk Q5); check_best = false: burn_in() looks at the best point once per block of 200 steps
    // (src/markov_chain.c:48-58), not after every step
    __device__ __forceinline__ void owner_results(int parity, double *sample, int which = -1, bool check_best = true) {
        const double sum = tree(parity);
        if constexpr (kSplit) {
            // the helper's half of the threshold and the prior it took for it, published by the barrier
            // that closed the step; `thr` holds the owner's own half since owner_publish
            thr = ObThreshold<kBase>::limit(*s_thx(parity), thr);
            prior_inflight = *s_pri(parity);
        }
        accepted = sum < thr;
        OB_SEG(0); // LDS batch, partial sums, decision
        // (the proposal's prior was computed for its threshold, a step ago: not again)
        const double prob_new = m.finish_known_prior(sum, beta_all, consts, prior_inflight);
        if (Model<kBase>::kHasPrior)
            prior = prior_inflight; // not restored on reject (quirk Q7)
        if (which < 0) {
            n_accepted += accepted ? 1u : 0u; // the four counters move together here (all-parameter steps): settled at the end
        } else {
            const u64 mine = which == grp ? 1u : 0u;
            pacc += accepted ? mine : 0u;
            prej += accepted ? 0u : mine;
        }
        if (accepted) {
            if (cand())
                cur = par_val;
            prob = prob_new;
        }
        if (check_best && prob > prob_best) { // mcmc_check_best
            prob_best = prob;
            best = cur;
        }
        if (sample) {
            if (lane == 63) {
                sample[0] = prob;
                sample[1] = prob - prior;
            } else {
                sample[0] = cur;
            }
        }
        cand_y = next_y; // the kernel's loop has advanced `tick` to the step now in flight
        cand_s = next_s;
        OB_SEG(1); // finish, counters, best point, sample row
    }

    // the step now in flight proposes s_prop(parity, variant): settle it (redraw path if the
    // prepared attempts of the chosen variant failed), keep its value, publish its threshold and the
    // two prepared proposals of the step after it into the other parity
    __device__ __forceinline__ void owner_choose(int parity, bool first) {
        double *row = s_prop(parity, (first || accepted) ? 0 : 1);
#ifdef APEMOST_EXP_OWNER_SLACK
        const u64 failed = 0;
#else
        const u64 failed = first ? 0 : (accepted ? fail_a : fail_r);
#endif
        // the value my parameter takes in the chosen proposal was read back when the proposals were
        // prepared (sel_a / sel_r): no trip to LDS between the accept decision and the next proposals
        par_val = (first || accepted) ? sel_a : sel_r;
        if (failed) { // (the workgroup takes an extra barrier after this: s_flag(parity) is set)
            redraw(failed, tick, row);
            __builtin_amdgcn_wave_barrier();
            par_val = cand() ? row[grp] : 0.0;
        }
        if constexpr (Model<kBase>::kHasPrior)
            m.load_offset(row, n_par); // (prior_only, offset and finish_known_prior read the heights and the additive parameter, nothing else)
        else
            m.load(row, n_par, x_abs_max);
        OB_SEG(2); // the proposal in flight
    }
    // candidates of the next tick: published by the barrier that opened this step; requested with the
    // step's first batch of LDS reads (APEMOST_HOIST_CAND: config 2 2.036 -> 2.066e8 steps/s, config 4
    // 2.29 -> 2.33e8, tools/experiments/gpu_exp_hoist.sh), not where the proposals need them
    __device__ __forceinline__ double2 owner_fetch_next_cand() const { return s_cand(tick + 1)[lane]; }
    // MERGED: both prepared proposals through attempts2() (measured, tools/experiments/gpu_exp_hoist.sh with
    // APEMOST_ATTEMPTS2: the calibration kernels 3-7 % faster and config 4's round kernel 2.345 -> 2.375e8
    // steps/s with it, config 2's round kernel 2.07 -> 2.04e8: the round kernel of the models without a
    // prior keeps the two calls)
    template <bool MERGED>
    __device__ __forceinline__ void owner_publish(int parity, double2 nx, int which_next = -1) {
        const int next = parity ^ 1;
        next_y = nx.x;
        next_s = nx.y;
        // the two proposals of the next step: from the proposal in flight, from the current point
#ifdef APEMOST_EXP_NO_ATTEMPTS
        // TIMING ONLY (results are garbage; tools/experiments/r04_session18.sh, 20): what the step costs when somebody
        // else prepares the proposals, before that somebody's own cost
        fail_a = fail_r = 0;
#else
        if constexpr (MERGED) {
            attempts2(par_val, cur, next_y, next_s, s_prop(next, 0), s_prop(next, 1), which_next, fail_a, fail_r);
        } else {
            fail_a = attempts(par_val, next_y, next_s, s_prop(next, 0), which_next);
            fail_r = attempts(cur, next_y, next_s, s_prop(next, 1), which_next);
        }
#endif
        OB_SEG(3); // next candidates, both prepared proposals
#if defined(APEMOST_OWNER_PRIO_PHASE) && APEMOST_OWNER_PRIO_PHASE == 1
        __builtin_amdgcn_s_setprio(0);
#endif
        // S_max of the step in flight (kSplit: the chain's half of it; the helper adds the proposal's)
        double prior_new = 0;
        if constexpr (Model<kBase>::kHasPrior && !kSplit) {
            prior_new = m.prior_only(consts);
            prior_inflight = prior_new;
        }
        OB_SEG(4); // the prior of the proposal in flight (kSplit: the helper's since round 4)
        const double lu = read_lane(cand_y, 63);
        if constexpr (kSplit)
            thr = thr_fn.y_part(prob + lu);
        else if constexpr (Model<kBase>::kHasPrior)
            thr = thr_fn.s_max(prob + lu, m, prior_new, m.offset());
        else
            thr = thr_fn.s_max(prob + lu, m, 0.0, 0.0);
        if (lane == 0) {
            *s_thr(next) = thr;
#ifdef APEMOST_EXP_OWNER_SLACK
            *s_flag(next) = 0;
#else
            *s_flag(next) = (fail_a | fail_r) != 0 ? 1 : 0;
#endif
        }
        OB_SEG(5); // threshold, flags
    }
    // first thing after the barrier, in one batch with the partial sums: what each prepared proposal
    // settled on for my parameter
    __device__ __forceinline__ void owner_fetch_selected(int parity) {
        sel_a = cand() ? s_prop(parity, 0)[grp] : 0.0;
        sel_r = cand() ? s_prop(parity, 1)[grp] : 0.0;
    }
};

} // namespace apemost
