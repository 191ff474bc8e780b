// pt_kernels.h -- the engine's kernels (templates over the likelihood model and the workgroup shape)
// and the helpers that turn a run-time (model, waves) into a compile-time instantiation.
//
// The library is built from one translation unit per model (apemost_model.hip, compiled once per
// value of APEMOST_TU_MODEL, in parallel) plus the host side of the C ABI (apemost_hip.hip): each
// model TU instantiates model_dispatch<MODEL> -- every kernel of that model for every workgroup
// shape -- and the ABI calls it through the declarations at the end of this file.
#pragma once
#include "pt_device.h"
#include "pt_onebarrier.h"

namespace apemost {
// ===========================================================================
// kernels
// ===========================================================================

// LDS carve (doubles): proposed params [2][64], wave partials [2][16], 8 control words, the logarithm's table
constexpr int kLdsLogTab = 2 * kWave + 32 + 8;
constexpr int kFixedLdsDoubles = kLdsLogTab + kLogTabLdsDoubles;

struct RoundArgs {
    DevArrays d;
    ChainShape sh;
    int cur;          // which half of the double-buffered fields is current
    int first;        // first local chain (calc_model on a range)
    int which;        // -1: all-parameter updates; p: update parameter p only (markov_chain_step_for)
    int apply_swap;   // fuse tempering_interaction() for swap-stream position `round`
    unsigned n_steps; // Metropolis steps per round
    unsigned n_rounds; // rounds in this launch; between them the swap attempts are exchanged in-kernel
    u64 round;
    double *samples; // [n_steps][n_chains][n_par+2] or nullptr
};

// candidate sets kept in LDS: 8-slot ring with producer waves, WAVES without
__host__ __device__ constexpr int cand_slots(int waves) { return waves > 8 ? waves : 8; }
// candidate production as a side duty of waves 1-3 (workgroups of at least 4 waves)
// (they pay when the chip has idle CUs: few chains; with many chains they only take wave slots)
__host__ __device__ constexpr bool has_producer(int waves) { return waves >= 4; }
// The one-barrier round kernel (pt_onebarrier.h) exists for 4 and 8 likelihood waves per chain
// (+ owner + three candidate producers: workgroups of 8 and 12 waves).  Measured on one MI355X
// (steps/s, one-barrier 4 / one-barrier 8 / two-phase 4; tools/experiments/gpu_exp_ob4.sh):
//   simplesin  128 x  1024 (n_swap 15): 1.85e8 / 1.66e8 / 1.27e8     64 x 1024: 0.99e8 / 0.88e8
//   simplesin  256 x  1024 (n_swap  7): 3.18e8 / 1.67e8 / 2.37e8     128 x 4096: 9.2e7 / 8.8e7
//   pulse      256 x  1024 (n_swap  1): 7.84e7 /   -    / 7.81e7     128 x 16384: 2.43e7 / 2.60e7
//   pulse      256 x  1024 (n_swap  7): 1.19e8 / 0.89e8 / 1.04e8     128 x 65536: 6.7e6 / 7.2e6
// With 2 it loses to 4 (128 x 1024: 1.48e8): the data vector no longer fits the registers.
#ifndef APEMOST_OB_WAVES_MASK
#define APEMOST_OB_WAVES_MASK 0x110
#endif
__host__ __device__ constexpr bool has_one_barrier(int waves) { return (APEMOST_OB_WAVES_MASK >> waves) & 1; }
__host__ __device__ constexpr int block_threads(int waves, bool) { return waves * kWave; }

template <int MODEL, int WAVES, bool LDS_DATA, bool PRODUCER>
__device__ __forceinline__ void engine_setup(Engine<MODEL, WAVES, LDS_DATA, PRODUCER> &e, const DevArrays &d,
                                             const ChainShape &sh, int c, double *lds) {
    constexpr int kThreads = WAVES * kWave;
    // Wave roles.  A workgroup's wavefronts are dealt to the CU's four SIMDs cyclically (wave w and
    // w+4 share one, tools/hwid_probe.hip): wave 0 owns the chain, waves 1-3 -- the other three
    // SIMDs -- produce the candidates as a side duty, so the owner's serial code does not share
    // issue slots with candidate generation.
    e.lane = threadIdx.x & (kWave - 1);
    e.wave = threadIdx.x / kWave;
    e.tid = e.wave * kWave + e.lane;
    e.n_par = sh.n_par;
    e.n_data = sh.n_data;
    e.consts = sh.consts;
    e.x_abs_max = sh.x_abs_max;
    e.circular = sh.circular;
    e.seed = sh.seed;
    e.g = (u64)(sh.chain_offset + c);
    e.parity = 0;
    e.s_par = lds;              // 2*64 doubles
    e.s_part = lds + 2 * kWave; // 2*16 doubles, then 8 control words
    e.s_cand = (double2 *)(lds + kFixedLdsDoubles);
    double *s_data = lds + kFixedLdsDoubles + cand_slots(WAVES) * 2 * kWave;
    e.setup_lanes();
    if (threadIdx.x == 0)
        *e.fail_flag() = 0; // ordered before its first use by the barrier every kernel has after setup
    if constexpr (Model<MODEL % kVariantModel>::kIndexed)
        e.m.set_data(d.data, sh.n_data, (int)((unsigned)sh.variant >> 16));
    if constexpr (Model<MODEL % kVariantModel>::kUsesLogTable) { // (published by the barrier every kernel has after setup)
        stage_logtab(lds + kLdsLogTab, threadIdx.x, kThreads);
        e.m.set_logtab(lds + kLdsLogTab);
    }
    if (d.f != nullptr) // a resident chain: its prior box may make the per-step argument check void
        e.m.set_box(d.pmin() + (size_t)c * sh.n_par, d.pmax() + (size_t)c * sh.n_par, sh.x_abs_max);
    if (LDS_DATA) {
        // stage the data vector once per launch: coalesced HBM/L2 reads, SoA in LDS
        for (int i = threadIdx.x; i < 2 * sh.n_data; i += kThreads)
            s_data[i] = d.data[i];
        e.xs = s_data;
        e.ys = s_data + sh.n_data;
    } else {
        e.xs = d.data;
        e.ys = d.data + sh.n_data;
    }
}

// Load the chain into wave 0's registers from the read half of the state.
template <class E>
__device__ __forceinline__ void chain_load(E &e, const DevArrays &d, const ChainShape &sh, int c,
                                           int cur) {
    const int row = c + 1, n = sh.n_par;
    e.beta_all = d.beta()[row];
    e.cur = e.best = e.stepw = e.lo = e.hi = 0;
    e.pacc = e.prej = 0;
    e.prob = d.prob(cur)[row];
    e.prior = d.prior(cur)[row];
    e.prob_best = d.prob_best(cur)[row];
    e.accept = d.accept()[c];
    e.reject = d.reject()[c];
    e.tick = d.ticks()[c];
    if (e.wave == 0 && e.cand()) {
        const size_t k = (size_t)c * n + e.grp;
        e.cur = d.params(cur)[(size_t)row * n + e.grp];
        e.best = d.params_best(cur)[(size_t)row * n + e.grp];
        e.stepw = d.step()[k];
        e.lo = d.pmin()[k];
        e.hi = d.pmax()[k];
        e.pacc = d.params_accepts()[k];
        e.prej = d.params_rejects()[k];
    }
}

template <class E>
__device__ __forceinline__ void chain_store(const E &e, const DevArrays &d, const ChainShape &sh,
                                            int c, int dst, bool store_step) {
    const int row = c + 1, n = sh.n_par;
    if (e.wave != 0)
        return;
    if (e.cand() && e.qidx == 0) {
        const size_t k = (size_t)c * n + e.grp;
        d.params(dst)[(size_t)row * n + e.grp] = e.cur;
        d.params_best(dst)[(size_t)row * n + e.grp] = e.best;
        d.params_accepts()[k] = e.pacc;
        d.params_rejects()[k] = e.prej;
        if (store_step)
            d.step()[k] = e.stepw;
    }
    if (e.lane == 63) {
        d.prob(dst)[row] = e.prob;
        d.prior(dst)[row] = e.prior;
        d.prob_best(dst)[row] = e.prob_best;
        d.accept()[c] = e.accept;
        d.reject()[c] = e.reject;
        d.ticks()[c] = e.tick;
    }
}

// ---- agent-scope accessors for words another workgroup of the same launch writes or reads
// (cdna_hip_programming.md Guideline 16: global address space, sc1, never plain) ----
typedef __attribute__((address_space(1))) u64 gu64;
__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store((gu64 *)p, (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(u64 *p, u64 v) {
    __hip_atomic_store((gu64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 ld_agent(const u64 *p) {
    return __hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// bounded relaxed poll by one wave on one word; on timeout the launch's timeout word is set and
// the host reports the failure (results of that launch are void)
// (kVariantTestShortWaits, a test hook: a few thousand polls, so that the timeout path can be walked
// in milliseconds -- apemost_hip.h APEMOST_HIP_FLAG_TEST_WITHHOLD_PUBLISH)
__device__ __forceinline__ unsigned handoff_spins(const ChainShape &sh) {
    return (sh.variant & kVariantTestWithhold) ? 20000u : 8000000u;
}
__device__ __forceinline__ bool wait_at_least(const u64 *word, u64 want, u64 *timeout_word, unsigned max_spins) {
    for (unsigned spins = 0; ld_agent(word) < want; spins++) {
        __builtin_amdgcn_s_sleep(2);
        if (spins > max_spins) {
            st_agent(timeout_word, 1);
            return false;
        }
    }
    return true;
}

// which neighbour's record a chain read at its latest use of each half of the state block, and for
// which swap index: before the chain overwrites its row in that half it waits for that reader's ack
struct SwapMemo { // two named slots, not arrays: a run-time subscript would put them in scratch
    int partner0, partner1;
    u64 index0, index1;
    __device__ __forceinline__ int partner(int half) const { return half ? partner1 : partner0; }
    __device__ __forceinline__ u64 index(int half) const { return half ? index1 : index0; }
    __device__ __forceinline__ void set(int half, int p, u64 i) {
        if (half) {
            partner1 = p;
            index1 = i;
        } else {
            partner0 = p;
            index0 = i;
        }
    }
};

// tempering_interaction() (src/parallel_tempering_interaction.c:25-42, 87-123, 125-141) as seen
// by one chain: every workgroup derives the same pair and uniforms from the replicated swap
// stream; the two workgroups of the pair evaluate the same expression on the same values and
// agree without negotiating.  Records are read from half `half` of the state block; `shared`
// selects agent-scope loads (records published inside this launch) over plain ones (records
// stored by the previous launch).  Returns the partner's local chain index, or -1.
// the draws of swap attempt `swap_index`: the lower chain of the pair (-1: no attempt) and the
// uniform of the acceptance test.  Default: parallel_tempering_decide_swap_now (:87-97), words 0
// and 1.  -DRANDOMSWAP: parallel_tempering_decide_swap_random(chains, n_beta, 1) (:47-64) draws
// swap_probability first and compares it with 1.0 / n_swap for the n_swap = 1 its caller passes.
template <bool VARIANTS>
__device__ __forceinline__ long long swap_draws(const ChainShape &sh, u64 swap_index, double &u_accept) {
    const uint4 b = philox_block(sh.seed, APEMOST_HIP_SWAP_SUBSEQUENCE, swap_index);
    const int nb = (int)sh.n_global;
    double u = u32_to_uniform(b.x);
    u_accept = u32_to_uniform(b.y);
    if (VARIANTS && (sh.variant & kVariantRandomSwap)) {
        if (!(u < 1.0 / 1))
            return -1;
        u = u32_to_uniform(b.y);
        u_accept = u32_to_uniform(b.z);
    }
    return (int)(nb * 1000 * u) % (nb - 1);
}

template <class E>
__device__ __forceinline__ int swap_apply(E &e, const DevArrays &d, const ChainShape &sh, int c, int half,
                                          u64 swap_index, bool shared) {
    double u_accept;
    const long long a = swap_draws<E::kVariants>(sh, swap_index, u_accept);
    const double lc = log(u_accept);
    const long long g = sh.chain_offset + c;
    if (a < 0 || (g != a && g != a + 1))
        return -1;
    const int n = sh.n_par;
    const int row = c + 1;
    const int row_a = (g == a) ? row : row - 1, row_b = row_a + 1;
    const int partner = (g == a) ? row_b : row_a;
    // own values come from registers, the partner's from memory
    const double p_prob = shared ? ld_agent(d.prob(half) + partner) : d.prob(half)[partner];
    const double p_best = shared ? ld_agent(d.prob_best(half) + partner) : d.prob_best(half)[partner];
    const double a_prob = (g == a) ? e.prob : p_prob, b_prob = (g == a) ? p_prob : e.prob;
    const double a_beta = d.beta()[row_a], b_beta = d.beta()[row_b];
    const double r = a_beta * b_prob / b_beta + b_beta * a_prob / a_beta - (a_prob + b_prob);
    if (r > lc) {
        // parallel_tempering_do_swap: params exchanged, prob is not (quirk Q1)
        const double a_best = (g == a) ? e.prob_best : p_best, b_best = (g == a) ? p_best : e.prob_best;
        const bool a_wins = a_best > b_best;
        const bool take_best = (g == a) != a_wins; // this chain receives the other one's best (quirk Q3)
        if (e.cand()) {
            const double *pp = d.params(half) + (size_t)partner * n + e.grp;
            const double *pb = d.params_best(half) + (size_t)partner * n + e.grp;
            e.cur = shared ? ld_agent(pp) : *pp;
            if (take_best)
                e.best = shared ? ld_agent(pb) : *pb;
        }
        if (take_best)
            e.prob_best = a_wins ? a_best : b_best;
        if (g == a && e.lane == 0)
            d.swapcount()[c] += 1; // inc_swapcount(chains[candidate])
    }
    return partner - 1;
}

// a chain's row in half `half` is about to be overwritten: its latest reader must be done
template <class E>
__device__ __forceinline__ void wait_for_reader(const DevArrays &d, const ChainShape &sh, const SwapMemo &memo,
                                                int half) {
    const int p = memo.partner(half);
    if (p >= 0 && p < sh.n_chains)
        wait_at_least(d.acked() + p, memo.index(half) + 1, d.timeout_word(), handoff_spins(sh));
}

// swap attempt at the start of a launch: both records were stored by the previous launch (or
// imported into a halo row by the host)
template <class E>
__device__ __forceinline__ void swap_at_launch_start(E &e, const DevArrays &d, const ChainShape &sh, int c, int half,
                                                     u64 swap_index, SwapMemo &memo) {
    if (sh.n_global <= 1 || e.wave != 0)
        return;
    const int partner = swap_apply(e, d, sh, c, half, swap_index, false);
    if (partner == -1)
        return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the reads above have landed
    if (e.lane == 0)
        st_agent(d.acked() + c, swap_index + 1);
    memo.set(half, partner, swap_index);
}

// swap attempt between two rounds of one launch: the two chains of the pair publish their
// records into half `half`, wait for each other, then decide (Guideline 16: payload and flag
// are agent-scope sc1 stores drained by the storing wave; the consumer polls the flag relaxed,
// takes one agent acquire, and reads the payload with agent-scope loads)
template <class E>
__device__ __forceinline__ void swap_in_launch(E &e, const DevArrays &d, const ChainShape &sh, int c, int half,
                                               u64 swap_index, SwapMemo &memo) {
    if (sh.n_global <= 1 || e.wave != 0)
        return;
    double u_accept;
    const long long a = swap_draws<E::kVariants>(sh, swap_index, u_accept);
    const long long g = sh.chain_offset + c;
    if (a < 0 || (g != a && g != a + 1))
        return;
    const int partner = (g == a) ? c + 1 : c - 1;
    if (partner < 0 || partner >= sh.n_chains) {
        st_agent(d.timeout_word(), 2); // the host must not schedule a shard-straddling pair in-launch
        return;
    }
    const int n = sh.n_par, row = c + 1;
    wait_for_reader<E>(d, sh, memo, half);
    if (e.cand() && e.qidx == 0) {
        st_agent(d.params(half) + (size_t)row * n + e.grp, e.cur);
        st_agent(d.params_best(half) + (size_t)row * n + e.grp, e.best);
    }
    if (e.lane == 63) {
        st_agent(d.prob(half) + row, e.prob);
        st_agent(d.prob_best(half) + row, e.prob_best);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every payload store of this wave has left
    // (test hook: the lower chain of the pair keeps the publish of swap attempt 3 to itself, so its
    // partner's bounded wait runs out)
    const bool withhold = (sh.variant & kVariantTestWithhold) && swap_index == 3 && g == a;
    if (e.lane == 0 && !withhold)
        st_agent(d.published() + c, swap_index + 1);
    if (!wait_at_least(d.published() + partner, swap_index + 1, d.timeout_word(), handoff_spins(sh)))
        return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    swap_apply(e, d, sh, c, half, swap_index, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (e.lane == 0)
        st_agent(d.acked() + c, swap_index + 1);
    memo.set(half, partner, swap_index);
}

// (the pulse model's one-wave kernel sits at the edge of three resident waves per SIMD, 168 registers:
// ladders of 1500+ chains run 2048: 1.29 vs 1.54e8 steps/s with two)
constexpr int round_min_waves(int model, int waves) { return model % kVariantModel == APEMOST_MODEL_PULSE && waves == 1 ? APEMOST_PULSE_MIN_WAVES : 1; }
template <int MODEL, int WAVES, bool LDS_DATA, bool PROD>
__global__ __launch_bounds__(block_threads(WAVES, PROD)) __attribute__((amdgpu_waves_per_eu(round_min_waves(MODEL, WAVES), 8)))
void pt_round_kernel(const RoundArgs a) {
    extern __shared__ __align__(16) double lds[];
    Engine<MODEL, WAVES, LDS_DATA, PROD> e;
    const int c = blockIdx.x;
    engine_setup(e, a.d, a.sh, c, lds);
    chain_load(e, a.d, a.sh, c, a.cur);
    e.pin_uniforms();
    SwapMemo memo;
    memo.partner0 = memo.partner1 = -1;
    memo.index0 = memo.index1 = 0;
    if (a.apply_swap)
        swap_at_launch_start(e, a.d, a.sh, c, a.cur, a.round, memo);
    e.producer_prologue();
    __syncthreads();
    e.cache_rows();
    e.producer_first_fetch();
#ifdef APEMOST_STAMPS
    e.stamps_begin();
#endif

    const int n = a.sh.n_par;
    // each lane's slot in the sample row of its chain, advanced by one row set per step
    double *my_sample = nullptr;
    if (a.samples && e.wave == 0 && (e.lane == 63 || (e.cand() && e.qidx == 0)))
        my_sample = a.samples + (size_t)c * (n + 2) + (e.lane == 63 ? n : e.grp);
    const size_t sample_stride = (size_t)a.sh.n_chains * (n + 2);
    for (unsigned r = 0; r < a.n_rounds; r++) {
        if (r > 0) // the swap attempt between round r-1 and round r
            swap_in_launch(e, a.d, a.sh, c, a.cur ^ (int)(r & 1), a.round + r - (a.apply_swap ? 0 : 1), memo);
        for (unsigned s = 0; s < a.n_steps; s++) {
            e.step(a.which);
            if (e.wave == 0) {
                e.check_best();
                if (my_sample) {
                    // the row the reference prints per step: params ("%.15e"), prob, prob-prior
                    if (e.lane == 63) {
                        my_sample[0] = e.prob;
                        my_sample[1] = e.prob - e.prior;
                    } else {
                        my_sample[0] = e.cur;
                    }
                    my_sample += sample_stride;
                }
            }
        }
    }
#ifdef APEMOST_STAMPS
    e.stamps_flush();
#endif
    if (e.wave == 0 && e.lane == 0)
        a.d.n_iter()[c] += (u64)a.n_steps * a.n_rounds; // mcmc_append_current_parameters, src/mcmc_calculate.c:30-33
    if (e.wave == 0)
        wait_for_reader<decltype(e)>(a.d, a.sh, memo, a.cur ^ 1);
    if (e.tid == 0 && *e.fail_flag())
        st_agent(a.d.timeout_word(), 3);
    chain_store(e, a.d, a.sh, c, a.cur ^ 1, false);
}

// The same rounds with one barrier per step (pt_onebarrier.h): LW likelihood wavefronts plus an
// owner and three candidate producers.  All-parameter steps only; launches with steps.
//
// Every role runs its own copy of the round/step loops (the registers a role carries from step to
// step are then live in its loop only); what the copies share is the barrier sequence: one at the
// start of a round, one per step, and one more in a step whose prepared proposal has to be redrawn
// (every wave reads the same LDS flag for that).
// Diagnostic build (-DAPEMOST_STAMPS): per wave of workgroup 0, the cycles between leaving a step's
// barrier and arriving at the next one (g_stamps[wave]); g_stamps[15] = whole steps of the owner,
// barrier to barrier.  Tells which role the others wait for.
#ifndef APEMOST_MERGED_ALL
#define APEMOST_MERGED_ALL 0 // (experiment: the sine models' round kernel through attempts2() too; see ObEngine::owner_publish)
#endif
#ifdef APEMOST_STAMPS
#define OB_STAMP_DECL u64 ob_busy = 0, ob_t0 = 0, ob_total = 0, ob_prev = 0
#define OB_STAMP_BEGIN ob_t0 = __builtin_amdgcn_s_memtime()
#define OB_STAMP_END ob_busy += __builtin_amdgcn_s_memtime() - ob_t0
#define OB_STAMP_FLUSH                                                                            \
    if (blockIdx.x == 0 && e.lane == 0)                                                           \
    atomicAdd(&g_stamps[e.is_helper() ? 14 : e.hw], ob_busy) /* (slots 8-13: the owner's segments when LW = 4) */
#else
#define OB_STAMP_DECL
#define OB_STAMP_BEGIN
#define OB_STAMP_END
#define OB_STAMP_FLUSH
#endif

// APEMOST_OWNER_PRIO_PHASE (experiment, round 4): the owner at APEMOST_OWNER_PRIO for the first part of its step only
// -- up to the decision and bookkeeping (3), up to the proposal in flight (2), up to the prepared proposals (1) --
// and at priority 0 for the rest: the likelihood wave on its SIMD is the last at the barrier, the owner is not.
#ifndef APEMOST_OWNER_PRIO_PHASE
#define APEMOST_OWNER_PRIO_PHASE 0
#endif
// The rounds base .. base+63 of this launch whose opening swap attempt involves chain c (bit r - base):
// lane l draws the pair of round base + l from the replicated swap stream.  A swap attempt touches
// its two chains only (src/parallel_tempering_interaction.c:99-141): for every other chain the
// boundary between two rounds is no event at all, and its pipeline of prepared proposals runs
// through it.  Every wave of the workgroup evaluates this for itself (one Philox block per 64
// rounds), so that all of them agree on where the extra barriers are.  Bit 0 of the launch's first
// block is always set: the pipeline starts there.
template <class E>
__device__ __forceinline__ u64 rounds_restarting(const E &e, const RoundArgs &a, int c, unsigned base) {
    bool involved = false;
    const unsigned r = base + (unsigned)e.lane;
    if (a.sh.n_global > 1 && r >= 1 && r < a.n_rounds) {
        double u_accept;
        const long long pair = swap_draws<E::kVariants>(a.sh, a.round + r - (a.apply_swap ? 0 : 1), u_accept);
        const long long g = a.sh.chain_offset + c;
        involved = pair >= 0 && (g == pair || g == pair + 1);
    }
    return __ballot(involved) | (base == 0 ? 1ull : 0ull);
}

template <int MODEL, int LW, bool LDS_DATA, bool HELPER>
__global__ __launch_bounds__(ob_block(LW, HELPER)) __attribute__((amdgpu_waves_per_eu(ob_waves_per_eu(LW, HELPER)))) void pt_round_ob_kernel(const RoundArgs a) {
    extern __shared__ __align__(16) double lds[];
    ObEngine<MODEL, LW, LDS_DATA, HELPER> e;
    const int c = blockIdx.x;
    e.setup_common(a.d, a.sh, c, lds);
#if APEMOST_OB_HELPER_SIMD
    if (HELPER && e.hw > LW + 3 && !e.is_helper())
        return; // (placeholders: see APEMOST_OB_HELPER_SIMD)
#endif
    OB_STAMP_DECL;
    // bit r % 64: the pipeline restarts at the start of round r (parity 0, the owner's first proposal
    // from the current point, one barrier more): at the launch's start and where a swap attempt moves
    // this chain.  Elsewhere a round's first step is a step like any other.
    u64 restart = 0;
    if (e.is_lik()) {
        e.setup_lik(a.d, a.sh, c);
        if (e.hw < 2)
            e.make_set(e.tick + (u64)e.hw); // the first two ticks' candidates, by waves with nothing else to do yet
        __syncthreads();
        e.cache_rows();
        // With a helper wavefront SIMD 0 holds three waves -- likelihood wave 0, the owner, the helper -- and that
        // likelihood wave is the one the step waits for (busy 1600 of 1580 ticks, the three others 1400, the owner
        // 1440, the helper 1490: profiles/r04_pulse_helper_wave.txt): it goes first on its SIMD.  Config 4 2.69 ->
        // 2.71e8 steps/s whatever the owner's priority.  Without the helper the owner is the last wave and this
        // costs 20 % (config 2: 2.14 -> 1.69e8).
#ifndef APEMOST_LIK0_PRIO
#define APEMOST_LIK0_PRIO 3
#endif
        if (HELPER && e.hw == 0)
            __builtin_amdgcn_s_setprio(APEMOST_LIK0_PRIO);
        int p = 0; // parity of the step about to start
        for (unsigned r = 0; r < a.n_rounds; r++) {
            if ((r & 63) == 0)
                restart = rounds_restarting(e, a, c, r);
            if ((restart >> (r & 63)) & 1) {
                p = 0;
                __syncthreads();
            }
            for (unsigned s = 0; s < a.n_steps; s++) {
                OB_STAMP_BEGIN;
                e.lik_step(p); // (takes one more barrier inside when a proposal has to be redrawn)
                OB_STAMP_END;
                __syncthreads();
                p ^= 1;
            }
        }
        OB_STAMP_FLUSH;
    } else if (e.is_producer()) {
        e.setup_lanes(a.sh, c);
        e.producer_prologue();
        __syncthreads();
        int p = 0;
        for (unsigned r = 0; r < a.n_rounds; r++) {
            if ((r & 63) == 0)
                restart = rounds_restarting(e, a, c, r);
            if ((restart >> (r & 63)) & 1) {
                p = 0;
                __syncthreads();
            }
            for (unsigned s = 0; s < a.n_steps; s++) {
                OB_STAMP_BEGIN;
#if defined(APEMOST_STAMPS) && defined(APEMOST_STAMP_PHASES)
                const int ph = e.pipe_phase; // (the first producer's step by phase: slots 8-10 instead of the owner's segments)
#endif
                // (the redraw flag requested first and looked at behind the phase's work -- the candidates do not depend
                // on it, only the barrier count does -- changes nothing: 2.131 vs 2.132e8 at config 2, session 20)
                if (e.redraw_pending(p))
                    __syncthreads();
                e.producer_step(p);
                OB_STAMP_END;
#if defined(APEMOST_STAMPS) && defined(APEMOST_STAMP_PHASES)
                if (blockIdx.x == 0 && e.lane == 0 && e.hw == LW + 1)
                    atomicAdd(&g_stamps[8 + ph], __builtin_amdgcn_s_memtime() - ob_t0);
#endif
                __syncthreads();
                p ^= 1;
            }
        }
        OB_STAMP_FLUSH;
    } else if (e.is_helper()) {
        // (HELPER) the helper alone: the barrier sequence of a producer, helper_step per step
        if constexpr (decltype(e)::kHelperWave) {
#ifdef APEMOST_HELPER_PRIO
            __builtin_amdgcn_s_setprio(APEMOST_HELPER_PRIO); // (experiment, round 4: tools/experiments/r04_session14.sh)
#endif
            e.setup_lanes(a.sh, c);
            e.setup_helper(a.d, a.sh, c);
            __syncthreads();
            int p = 0;
            for (unsigned r = 0; r < a.n_rounds; r++) {
                if ((r & 63) == 0)
                    restart = rounds_restarting(e, a, c, r);
                if ((restart >> (r & 63)) & 1) {
                    p = 0;
                    __syncthreads();
                }
                for (unsigned s = 0; s < a.n_steps; s++) {
                    OB_STAMP_BEGIN;
                    if (e.redraw_pending(p))
                        __syncthreads();
                    e.helper_step(p);
                    OB_STAMP_END;
                    __syncthreads();
                    p ^= 1;
                }
            }
            OB_STAMP_FLUSH;
        }
    } else {
        // the others wait for this wave at every barrier and it has little to issue: let it go first
#ifndef APEMOST_OWNER_PRIO
#define APEMOST_OWNER_PRIO 3
#endif
        __builtin_amdgcn_s_setprio(APEMOST_OWNER_PRIO);
        e.setup_lanes(a.sh, c);
        e.setup_owner(a.d, a.sh, c);
        chain_load(e, a.d, a.sh, c, a.cur);
        e.thr_fn.init(e.consts, e.beta_all);
        SwapMemo memo;
        memo.partner0 = memo.partner1 = -1;
        memo.index0 = memo.index1 = 0;
        if (a.apply_swap)
            swap_at_launch_start(e, a.d, a.sh, c, a.cur, a.round, memo);
        __syncthreads();
        const int n = a.sh.n_par;
        double *my_sample = nullptr;
        if (a.samples && (e.lane == 63 || (e.cand() && e.qidx == 0)))
            my_sample = a.samples + (size_t)c * (n + 2) + (e.lane == 63 ? n : e.grp);
        const size_t sample_stride = (size_t)a.sh.n_chains * (n + 2);
        int p = 0;         // parity of the step about to start
        bool open = false; // a step is in flight whose outcome is not settled yet
        for (unsigned r = 0; r < a.n_rounds; r++) {
            if ((r & 63) == 0)
                restart = rounds_restarting(e, a, c, r);
            if ((restart >> (r & 63)) & 1) {
                if (open) { // the last step of the previous round
                    e.owner_results(p, my_sample);
                    if (my_sample)
                        my_sample += sample_stride;
                    open = false;
                }
                if (r > 0) // the swap attempt between round r-1 and round r; the other waves wait at the barrier below
                    swap_in_launch(e, a.d, a.sh, c, a.cur ^ (int)(r & 1), a.round + r - (a.apply_swap ? 0 : 1), memo);
                p = 0;
                e.owner_first();
                __syncthreads();
            }
            for (unsigned s = 0; s < a.n_steps; s++) {
                OB_STAMP_BEGIN;
#ifdef APEMOST_STAMPS
                e.seg_last = ob_t0;
#endif
                // one batch of LDS reads: the redraw flag, what the prepared proposals settled on
                // for my parameter, and (owner_results) the partial sums
#if APEMOST_OWNER_PRIO_PHASE
                __builtin_amdgcn_s_setprio(APEMOST_OWNER_PRIO); // (dropped in the step's later part, see below)
#endif
                const int pending = *e.s_flag(p);
#if APEMOST_HOIST_CAND
                const double2 nx = e.owner_fetch_next_cand(); // (with the step's first batch of LDS reads)
#endif
                if (open) {
                    e.owner_fetch_selected(p);
                    e.owner_results(p, my_sample);
                    if (my_sample)
                        my_sample += sample_stride;
                }
                const bool redraw_pending = __builtin_amdgcn_readfirstlane(pending) != 0;
#if APEMOST_OWNER_PRIO_PHASE == 3
                __builtin_amdgcn_s_setprio(0);
#endif
                e.owner_choose(p, !open);
#if APEMOST_OWNER_PRIO_PHASE == 2
                __builtin_amdgcn_s_setprio(0);
#endif
                if (redraw_pending) // rare: a proposal in LDS has just been replaced
                    __syncthreads();
#if !APEMOST_HOIST_CAND
                const double2 nx = e.owner_fetch_next_cand();
#endif
#ifdef APEMOST_EXP_OWNER_SLACK
                // TIMING ONLY (results are garbage; tools/experiments/r04_slack.sh): the owner arrives at the
                // step's barrier with half of its work done and publishes behind it -- what a step costs when
                // nobody waits for the owner's whole program (the two-step look-ahead of DESIGN.md 9 would
                // give it that slack).  The redraw flag is never set in this build.
                __syncthreads();
                e.template owner_publish<Model<MODEL % kVariantModel>::kHasPrior || APEMOST_MERGED_ALL>(p, nx);
                e.tick++;
                OB_STAMP_END;
#else
                e.template owner_publish<Model<MODEL % kVariantModel>::kHasPrior || APEMOST_MERGED_ALL>(p, nx);
                e.tick++;
                OB_STAMP_END;
                __syncthreads();
#endif
#ifdef APEMOST_STAMPS
                ob_total += __builtin_amdgcn_s_memtime() - ob_t0;
#endif
                p ^= 1;
                open = true;
            }
        }
        if (open) { // the launch's last step
            e.owner_results(p, my_sample);
            if (my_sample)
                my_sample += sample_stride;
        }
        OB_STAMP_FLUSH;
#ifdef APEMOST_STAMPS
        if (blockIdx.x == 0 && e.lane == 0) {
            atomicAdd(&g_stamps[15], ob_total);
#ifndef APEMOST_STAMP_PHASES
            if (LW == 4) // (slots 8..13 are free with eight waves per workgroup)
#else
            if (false)
#endif
                for (int i = 0; i < 6; i++)
                    atomicAdd(&g_stamps[8 + i], e.seg_acc[i]);
        }
#endif
        e.owner_settle_counters((u64)a.n_steps * a.n_rounds);
        if (e.lane == 0)
            a.d.n_iter()[c] += (u64)a.n_steps * a.n_rounds;
        wait_for_reader<decltype(e)>(a.d, a.sh, memo, a.cur ^ 1);
        if (e.lane == 0 && *e.fail_flag())
            st_agent(a.d.timeout_word(), 3);
        chain_store(e, a.d, a.sh, c, a.cur ^ 1, false);
    }
}

// calc_model() for every resident chain, in place
template <int MODEL, int WAVES, bool LDS_DATA>
__global__ __launch_bounds__(WAVES *kWave) void pt_calc_model_kernel(const RoundArgs a) {
    extern __shared__ __align__(16) double lds[];
    Engine<MODEL, WAVES, LDS_DATA> e;
    const int c = a.first + blockIdx.x;
    engine_setup(e, a.d, a.sh, c, lds);
    e.m.clear_box(); // caller-supplied parameters may lie outside their prior box
    chain_load(e, a.d, a.sh, c, a.cur);
    __syncthreads();
    e.cache_rows();
    e.calc_model_current();
    if (e.wave == 0 && e.lane == 0) {
        a.d.prob(a.cur)[c + 1] = e.prob;
        a.d.prior(a.cur)[c + 1] = e.prior;
    }
}

// calc_model() at arbitrary points: params [n][n_par], beta [n] -> prob[n], prior[n]
struct EvalArgs {
    ChainShape sh;
    const double *data;
    const double *params;
    const double *beta;
    double *prob;
    double *prior;
};

template <int MODEL, int WAVES, bool LDS_DATA>
__global__ __launch_bounds__(WAVES *kWave) void pt_loglike_kernel(const EvalArgs a) {
    extern __shared__ __align__(16) double lds[];
    Engine<MODEL, WAVES, LDS_DATA> e;
    DevArrays d;
    d.f = nullptr;
    d.u = nullptr;
    d.n = a.sh.n_chains;
    d.np = a.sh.n_par;
    d.data = a.data;
    const int c = blockIdx.x;
    engine_setup(e, d, a.sh, c, lds);
    e.beta_all = a.beta[c];
    e.prior = 0;
    e.cur = (e.wave == 0 && e.cand()) ? a.params[(size_t)c * a.sh.n_par + e.grp] : 0.0;
    __syncthreads();
    e.cache_rows();
    e.calc_model_current();
    if (e.wave == 0 && e.lane == 0) {
        a.prob[c] = e.prob;
        a.prior[c] = e.prior;
    }
}

// ---- calibration: burn_in + markov_chain_calibrate_orig as a resumable per-chain state machine ----
//
// One workgroup calibrates one chain.  The two reference functions are cut at the places where they
// look at the chain anyway -- the end of a burn-in block of 200 steps (src/markov_chain.c:48-58), the
// end of ITER_READJUST single-parameter sweeps and the end of the ITER_READJUST all-parameter steps
// that follow them (src/markov_chain_calibrate.c:1064-1068, 1126-1131) -- into BLOCKS, and everything
// that lives across a block boundary (CalibRec + the chain's own state) is kept in HBM.  A launch runs
// whole blocks until the chain is done or has spent its budget of likelihood evaluations; the host
// collects which chains are done and launches the survivors again, with more wavefronts per chain as
// they get fewer (the trip count of a chain is data-dependent: 22 000 - 60 000 sweeps at the BASELINE
// configs, profiles/r03_calib_base.json).  Draws are addressed by tick, so a chain's result does not
// depend on where the launches were cut; the number of likelihood waves only moves the rounding of
// the data sum.
enum { CAL_INIT = 0, CAL_BURN1, CAL_BURN2, CAL_SWEEP, CAL_ALL, CAL_DONE };

struct CalibRec {
    int stage;    // CAL_*
    int status;   // 0, 1 = a step width became too large, 2 = iteration limit
    int nchecks;  // nchecks_without_rescaling
    int rescaled; // of the sweeps just judged, until the all-parameter steps behind them are judged too
    u64 iter;     // burn-in steps so far
    u64 sweeps;   // `iter` of markov_chain_calibrate_orig
    u64 evals;    // likelihood evaluations so far
    double rat_limit;
};

struct CalibArgs {
    DevArrays d;
    ChainShape sh;
    int cur;
    int first;        // first local chain of the calibrate_begin range
    int burn_in_only; // -DSKIP_CALIBRATE_ALLCHAINS
    int progress_slot; // the chain whose readjustments are logged (calibration_progress.data), or -1
    apemost_hip_calib_config cfg;
    const int *list;   // [grid] slots (chain - first) that are still calibrating
    CalibRec *rec;     // [count]
    double *orig_step; // [count][n_par] step widths burn_in() puts back when it ends
    double *progress;  // [progress_cap][1 + 2 n_par]: sweeps, then (normalised step, accept rate) per parameter
    int progress_cap;
    u64 budget;        // likelihood evaluations a chain may spend in this launch (whole blocks)
};

// Per-parameter rescaling after ITER_READJUST sweeps (src/markov_chain_calibrate.c:1084-1125),
// called by the chain's whole wavefront: the lanes of parameter p decide for parameter p, then the
// decisions are combined in parameter order like the reference's loop.  Returns `rescaled`.
template <class E>
__device__ __forceinline__ int calib_rescale(E &e, const apemost_hip_calib_config &cfg, double rat_limit, int n,
                                             int &fail) {
    int up = 0, clamped = 0, down = 0, too_large = 0;
    if (e.cand()) {
        const double ar = (double)e.pacc / ((double)e.prej + (double)e.pacc);
        if (ar > rat_limit + 0.05) {
            up = 1;
            e.stepw = e.stepw / cfg.mul;
            if (e.stepw / (e.hi - e.lo) > 1) {
                e.stepw = 1 * (e.hi - e.lo);
                clamped = 1;
            }
            if (e.stepw / (e.hi - e.lo) > 10000)
                too_large = 1;
        }
        if (ar < rat_limit - 0.05) {
            down = 1;
            e.stepw = e.stepw * cfg.mul;
        }
    }
    int rescaled = 0;
    fail = 0;
    for (int p = 0; p < n; p++) {
        const int src = p * e.Q; // first lane of parameter p's group
        const int up_p = __shfl(up, src, kWave), cl_p = __shfl(clamped, src, kWave);
        const int dn_p = __shfl(down, src, kWave), tl_p = __shfl(too_large, src, kWave);
        if (up_p) {
            if (rescaled == 0)
                rescaled = -1;
            if (cl_p && rescaled == -1)
                rescaled = 0;
            if (tl_p && !fail)
                fail = 1;
            if (rescaled == -1)
                rescaled = 1;
        }
        if (dn_p)
            rescaled = 1;
    }
    return rescaled;
}

// After the ITER_READJUST all-parameter steps (src/markov_chain_calibrate.c:1147-1173): 0 go on,
// 1 converged, 2 iteration limit; nudges rat_limit
template <class E>
__device__ __forceinline__ int calib_verdict(const E &e, const apemost_hip_calib_config &cfg, CalibRec &r) {
    const double delta = (double)e.accept / (double)(e.accept + e.reject) - cfg.target_global;
    int reached_perfection;
    if ((delta < 0 ? -delta : delta) < cfg.max_ar_deviation) {
        reached_perfection = 1;
    } else {
        reached_perfection = 0;
        if (delta < 0)
            r.rat_limit /= 0.99;
        else
            r.rat_limit *= 0.99;
    }
    if (r.nchecks >= cfg.no_rescaling_limit && reached_perfection == 1 && r.rescaled == 0)
        return 1;
    if (r.sweeps > cfg.iter_limit)
        return 2;
    return 0;
}

// the line block the reference appends to calibration_progress.data at every readjustment
// (src/markov_chain_calibrate.c:1141-1146), for the one chain whose file survives (progress_slot)
template <class E>
__device__ __forceinline__ void calib_log_progress(const E &e, const CalibArgs &a, int slot, const CalibRec &r, int n) {
    if (slot != a.progress_slot || a.cfg.iter_readjust == 0)
        return;
    const u64 k = r.sweeps / a.cfg.iter_readjust - 1;
    if (k >= (u64)a.progress_cap)
        return;
    double *row = a.progress + k * (size_t)(1 + 2 * n);
    if (e.lane == 63)
        row[0] = (double)r.sweeps;
    if (e.cand() && e.qidx == 0) {
        row[1 + 2 * e.grp] = e.stepw / (e.hi - e.lo);
        row[2 + 2 * e.grp] = (double)e.pacc / ((double)e.prej + (double)e.pacc);
    }
}

#ifndef APEMOST_CALIB_MIN_WAVES
#define APEMOST_CALIB_MIN_WAVES(waves) ((waves) == 1 ? 2 : 1)
#endif
// (one-wave workgroups: at least two of them per SIMD, i.e. at most 256 registers -- the state machine
// around the step costs the compiler a dozen registers more than the round kernel has, and a ladder of
// 2048 chains wants both of its waves per SIMD resident)
template <int MODEL, int WAVES, bool LDS_DATA, bool PROD>
__global__ __launch_bounds__(block_threads(WAVES, PROD)) __attribute__((amdgpu_waves_per_eu(APEMOST_CALIB_MIN_WAVES(WAVES), 8)))
void pt_calibrate_kernel(const CalibArgs a) {
    extern __shared__ __align__(16) double lds[];
    // The state machine's record lives in LDS, in the six control doubles behind the fail flag: it is
    // looked at between blocks only, and as registers it cost the one-wave kernels their second wave
    // per SIMD (pulse_vrot: 238 VGPRs with the record in LDS, 270 + 14 AGPRs without).  Thread 0
    // writes it, a barrier publishes it to every wave.
    static_assert(sizeof(CalibRec) == 6 * sizeof(double), "the record fills the control words of the LDS carve");
    volatile CalibRec &r = *(volatile CalibRec *)(lds + 2 * kWave + 32 + 2);
    Engine<MODEL, WAVES, LDS_DATA, PROD> e;
    const int slot = a.list[blockIdx.x];
    const int c = a.first + slot;
    const int n = a.sh.n_par;
    engine_setup(e, a.d, a.sh, c, lds);
    chain_load(e, a.d, a.sh, c, a.cur);
    e.pin_uniforms();
    e.producer_prologue();
    if (e.tid == 0) {
        const CalibRec in = a.rec[slot];
        r.stage = in.stage, r.status = in.status, r.nchecks = in.nchecks, r.rescaled = in.rescaled;
        r.iter = in.iter, r.sweeps = in.sweeps, r.evals = in.evals;
        r.rat_limit = in.rat_limit;
    }
    __syncthreads();
    e.cache_rows();
    e.producer_first_fetch();
    const bool w0 = (e.wave == 0);
    const apemost_hip_calib_config &cfg = a.cfg;
    double *my_orig = a.orig_step + (size_t)slot * n + e.grp; // what burn_in() puts back when it ends
    const u64 stop_at = r.evals + a.budget;

    for (;;) {
        // (every wave reads the same words: the branches below are uniform)
        const int stage = __builtin_amdgcn_readfirstlane(r.stage);
        const u64 evals_now = r.evals, iter_now = r.iter;
        // Every wave has read the record before thread 0 writes it again: three transitions below (INIT ->
        // BURN1, BURN1 -> BURN2, BURN2 -> SWEEP / DONE) take no step and so have no barrier of their own
        // between this read and the write at the loop's end -- a wave that left the previous barrier late
        // would read the NEXT stage and fall one barrier out of phase with wave 0.  One barrier
        // per block of >= 200 steps.
        if (WAVES > 1)
            __syncthreads();
        if (stage == CAL_DONE || evals_now >= stop_at)
            break;
        int next = stage; // (what wave 0 arrives at is what counts: thread 0 writes it)
        if (stage == CAL_INIT) {
            // ---- burn_in: src/markov_chain.c:34-79 ----
            if (w0 && e.cand() && e.qidx == 0)
                *my_orig = e.stepw;
            e.stepw = (e.hi - e.lo) * 0.1;
            next = CAL_BURN1;
            if (e.tid == 0)
                r.iter = 0;
        } else if (stage == CAL_BURN1 || stage == CAL_BURN2) {
            const u64 limit = stage == CAL_BURN1 ? cfg.burn_in_iterations / 2 : cfg.burn_in_iterations;
            if (iter_now < limit) {
                for (int sub = 0; sub < 200; sub++)
                    e.step(-1);
                if (w0)
                    e.check_best();
                if (e.tid == 0) {
                    r.iter += 200;
                    r.evals += 200;
                }
            } else if (stage == CAL_BURN1) {
                if (w0)
                    e.restart_from_best();
                e.stepw *= 0.5;
                next = CAL_BURN2;
            } else {
                if (w0 && e.cand()) {
                    __threadfence_block();
                    e.stepw = *(volatile double *)my_orig;
                }
                if (a.burn_in_only) {
                    next = CAL_DONE;
                } else {
                    // ---- markov_chain_calibrate_orig: src/markov_chain_calibrate.c:1039-1180 ----
                    e.stepw *= cfg.adjust_step;
                    e.reset_accept_rejects();
                    next = CAL_SWEEP;
                    if (e.tid == 0) {
                        r.rat_limit = pow(cfg.rat_limit, 1.0 / n);
                        r.nchecks = 0;
                        r.sweeps = 0;
                    }
                }
            }
        } else if (stage == CAL_SWEEP) {
            for (unsigned k = 0; k < cfg.iter_readjust; k++)
                for (int p = 0; p < n; p++) {
                    e.step(p);
                    if (w0)
                        e.check_best();
                }
            next = CAL_ALL;
            if (w0) {
                int fail = 0;
                const int rescaled = calib_rescale(e, cfg, r.rat_limit, n, fail);
                if (!fail) {
                    e.restart_from_best();
                    e.reset_accept_rejects();
                }
                if (e.tid == 0) {
                    r.sweeps += cfg.iter_readjust;
                    r.evals += (u64)cfg.iter_readjust * n;
                    r.rescaled = rescaled;
                    if (fail)
                        r.status = 1;
                    else if (rescaled == 0)
                        r.nchecks += 1;
                }
                if (fail)
                    next = CAL_DONE;
            }
        } else { // CAL_ALL
            for (unsigned sub = 0; sub < cfg.iter_readjust; sub++) {
                e.step(-1);
                if (w0)
                    e.check_best();
            }
            next = CAL_SWEEP;
            if (w0) {
                CalibRec now;
                now.nchecks = r.nchecks, now.rescaled = r.rescaled, now.sweeps = r.sweeps, now.rat_limit = r.rat_limit;
                calib_log_progress(e, a, slot, now, n);
                const int ctl = calib_verdict(e, cfg, now);
                if (ctl == 1)
                    e.reset_accept_rejects();
                if (ctl != 0)
                    next = CAL_DONE;
                if (e.tid == 0) {
                    r.evals += cfg.iter_readjust;
                    r.rat_limit = now.rat_limit;
                    if (ctl == 2)
                        r.status = 2;
                }
            }
        }
        // wave 0's verdicts (a failed rescaling, convergence, the iteration limit) reach the other
        // waves through the record
        if (e.tid == 0)
            r.stage = next;
        __syncthreads();
    }
    if (e.tid == 0) {
        CalibRec out;
        out.stage = r.stage, out.status = r.status, out.nchecks = r.nchecks, out.rescaled = r.rescaled;
        out.iter = r.iter, out.sweeps = r.sweeps, out.evals = r.evals;
        out.rat_limit = r.rat_limit;
        a.rec[slot] = out;
        if (*e.fail_flag())
            st_agent(a.d.timeout_word(), 3);
    }
    // calibration leaves the chain in place: same half of the double buffer
    chain_store(e, a.d, a.sh, c, a.cur, true);
}

// The same state machine on the one-barrier step (pt_onebarrier.h): LW likelihood wavefronts, the
// owner, three candidate producers.  A block is one run of the step pipeline -- it starts from the
// current point like a round does (owner_first, one barrier) -- and the owner alone follows the
// state machine: before every block it leaves the number of steps in an LDS word (0: this launch is
// over) that the other roles read behind the block's opening barrier.  In the single-parameter
// sweeps only the owner's work changes (attempts(which)): the likelihood waves see two prepared
// parameter vectors as ever.  The proposal of parameter p+1 does not depend on the outcome of
// parameter p's step, only the rest of the vector does, so both variants carry the same attempt.
template <int MODEL, int LW, bool LDS_DATA, bool HELPER>
__global__ __launch_bounds__(ob_block(LW, HELPER)) __attribute__((amdgpu_waves_per_eu(ob_waves_per_eu(LW, HELPER)))) void pt_calibrate_ob_kernel(const CalibArgs a) {
    extern __shared__ __align__(16) double lds[];
    ObEngine<MODEL, LW, LDS_DATA, HELPER> e;
    const int slot = a.list[blockIdx.x];
    const int c = a.first + slot;
    e.setup_common(a.d, a.sh, c, lds);
#if APEMOST_OB_HELPER_SIMD
    if (HELPER && e.hw > LW + 3 && !e.is_helper())
        return; // (placeholders: see APEMOST_OB_HELPER_SIMD)
#endif
    volatile int *s_steps = (volatile int *)(lds + kObCtl); // word 0 (word 2 is the fail flag)
    if (e.is_lik()) {
        e.setup_lik(a.d, a.sh, c);
        if (e.hw < 2)
            e.make_set(e.tick + (u64)e.hw);
        __syncthreads();
        e.cache_rows();
        // (likelihood wave 0 at a priority of its own, as in the round kernel: the calibration 0.32 -> 0.34 s at 3, no
        // change at 2 -- tools/experiments/r04_session17.sh -- not taken)
        for (;;) {
            __syncthreads(); // the block's opening barrier
            const int n_steps = __builtin_amdgcn_readfirstlane(*s_steps);
            if (n_steps == 0)
                break;
            int p = 0;
            for (int s = 0; s < n_steps; s++) {
                e.lik_step(p);
                __syncthreads();
                p ^= 1;
            }
        }
    } else if (e.is_producer()) {
        e.setup_lanes(a.sh, c);
        e.producer_prologue();
        __syncthreads();
        for (;;) {
            __syncthreads();
            const int n_steps = __builtin_amdgcn_readfirstlane(*s_steps);
            if (n_steps == 0)
                break;
            int p = 0;
            for (int s = 0; s < n_steps; s++) {
                if (e.redraw_pending(p))
                    __syncthreads();
                e.producer_step(p);
                __syncthreads();
                p ^= 1;
            }
        }
    } else if (e.is_helper()) {
        if constexpr (decltype(e)::kHelperWave) {
            e.setup_lanes(a.sh, c);
            e.setup_helper(a.d, a.sh, c);
            __syncthreads();
            for (;;) {
                __syncthreads();
                const int n_steps = __builtin_amdgcn_readfirstlane(*s_steps);
                if (n_steps == 0)
                    break;
                int p = 0;
                for (int s = 0; s < n_steps; s++) {
                    if (e.redraw_pending(p))
                        __syncthreads();
                    e.helper_step(p);
                    __syncthreads();
                    p ^= 1;
                }
            }
        }
    } else {
        __builtin_amdgcn_s_setprio(APEMOST_OWNER_PRIO);
        const int n = a.sh.n_par;
        const apemost_hip_calib_config &cfg = a.cfg;
        e.setup_lanes(a.sh, c);
        e.setup_owner(a.d, a.sh, c);
        chain_load(e, a.d, a.sh, c, a.cur);
        e.thr_fn.init(e.consts, e.beta_all);
        __syncthreads();
        CalibRec r = a.rec[slot];
        double original_step = e.cand() ? a.orig_step[(size_t)slot * n + e.grp] : 0.0;
        u64 done = 0;
        for (;;) {
            // the transitions that take no step, up to the next block
            int n_steps = 0;
            while (r.stage != CAL_DONE && done < a.budget) {
                if (r.stage == CAL_INIT) {
                    original_step = e.stepw;
                    e.stepw = (e.hi - e.lo) * 0.1;
                    r.iter = 0;
                    r.stage = CAL_BURN1;
                } else if (r.stage == CAL_BURN1 || r.stage == CAL_BURN2) {
                    const u64 limit = r.stage == CAL_BURN1 ? cfg.burn_in_iterations / 2 : cfg.burn_in_iterations;
                    if (r.iter < limit) {
                        n_steps = 200;
                        break;
                    }
                    if (r.stage == CAL_BURN1) {
                        e.restart_from_best();
                        e.stepw *= 0.5;
                        r.stage = CAL_BURN2;
                    } else {
                        e.stepw = original_step;
                        if (a.burn_in_only) {
                            r.stage = CAL_DONE;
                        } else {
                            r.rat_limit = pow(cfg.rat_limit, 1.0 / n);
                            r.nchecks = 0;
                            r.sweeps = 0;
                            e.stepw *= cfg.adjust_step;
                            e.reset_accept_rejects();
                            r.stage = CAL_SWEEP;
                        }
                    }
                } else {
                    n_steps = r.stage == CAL_SWEEP ? (int)cfg.iter_readjust * n : (int)cfg.iter_readjust;
                    break;
                }
            }
            const bool sweep = r.stage == CAL_SWEEP;
            const bool burn = r.stage == CAL_BURN1 || r.stage == CAL_BURN2;
            if (n_steps)
                e.owner_first(sweep ? 0 : -1);
            if (e.lane == 0)
                *s_steps = n_steps;
            __syncthreads(); // the block's opening barrier
            if (n_steps == 0)
                break;
            int p = 0;                   // parity of the step about to start
            bool open = false;           // a step is in flight whose outcome is not settled yet
            int which = sweep ? 0 : -1;  // what the step about to start proposes
            int which_prev = -1;         // ... and what the step in flight proposed
            for (int s = 0; s < n_steps; s++) {
                const int pending = *e.s_flag(p);
#if APEMOST_HOIST_CAND
                const double2 nx = e.owner_fetch_next_cand();
#endif
                if (open) {
                    e.owner_fetch_selected(p);
                    e.owner_results(p, nullptr, which_prev, !burn);
                }
                const bool redraw_pending = __builtin_amdgcn_readfirstlane(pending) != 0;
                e.owner_choose(p, !open);
                if (redraw_pending)
                    __syncthreads();
                const int which_next = !sweep ? -1 : (which + 1 == n ? 0 : which + 1);
#if !APEMOST_HOIST_CAND
                const double2 nx = e.owner_fetch_next_cand();
#endif
                e.template owner_publish<true>(p, nx, which_next);
                e.tick++;
                __syncthreads();
                p ^= 1;
                open = true;
                which_prev = which;
                which = which_next;
            }
            e.owner_results(p, nullptr, which_prev, !burn); // the block's last step
            if (!sweep)
                e.owner_settle_counters((u64)n_steps);
            done += (u64)n_steps;
            if (burn) {
                r.iter += 200;
                e.check_best();
            } else if (sweep) {
                r.sweeps += cfg.iter_readjust;
                int fail = 0;
                r.rescaled = calib_rescale(e, cfg, r.rat_limit, n, fail);
                if (fail) {
                    r.status = 1;
                    r.stage = CAL_DONE;
                } else {
                    if (r.rescaled == 0)
                        r.nchecks++;
                    e.restart_from_best();
                    e.reset_accept_rejects();
                    r.stage = CAL_ALL;
                }
            } else {
                calib_log_progress(e, a, slot, r, n);
                const int ctl = calib_verdict(e, cfg, r);
                r.stage = ctl == 0 ? CAL_SWEEP : CAL_DONE;
                if (ctl == 2)
                    r.status = 2;
                if (ctl == 1)
                    e.reset_accept_rejects();
            }
        }
        if (e.lane == 0) {
            r.evals += done;
            a.rec[slot] = r;
            if (*e.fail_flag())
                st_agent(a.d.timeout_word(), 3);
        }
        if (e.cand() && e.qidx == 0)
            a.orig_step[(size_t)slot * n + e.grp] = original_step;
        chain_store(e, a.d, a.sh, c, a.cur, true);
    }
}

// What the precompiled host code and a run-time compilation of these templates (a user-supplied model,
// apemost_hip.hip user_model_build) must agree on: the layouts of the kernel arguments and of the LDS carve.
// The hiprtc module exports the value it was compiled with (apemost_rtc_fingerprint below), the host compares
// it with its own at load: headers that drifted from the library (APEMOST_HIP_SOURCE_DIR, a stale .so, a
// development build) are an error, not silently wrong arguments.
constexpr unsigned long long kAbiFingerprint =
    (unsigned long long)sizeof(RoundArgs) ^ ((unsigned long long)sizeof(CalibArgs) << 10) ^ ((unsigned long long)sizeof(EvalArgs) << 20) ^
    ((unsigned long long)sizeof(CalibRec) << 28) ^ ((unsigned long long)kFixedLdsDoubles << 34) ^
    ((unsigned long long)kObFixedDoubles << 46) ^ ((unsigned long long)kTickShift << 58) ^ 0x4150454d6f535434ull;
#ifdef APEMOST_USER_MODEL
} // namespace apemost
extern "C" __device__ __attribute__((used)) const unsigned long long apemost_rtc_fingerprint = apemost::kAbiFingerprint;
namespace apemost {
#endif

#ifndef __HIPCC_RTC__ // (a run-time compilation of a user-supplied model holds the kernels only)
// ---- launch dispatch over (model, waves, lds) ----
enum KernelKind { K_ROUND, K_ROUND_OB, K_CALC, K_EVAL, K_CALIB, K_CALIB_OB };

// the one-barrier kernels of one (model, waves, staging, helper) shape
template <int MODEL, int WAVES, bool LDS, bool HELPER>
static hipError_t launch_ob(KernelKind kind, bool coop, int grid, size_t lds, hipStream_t st, const void *args) {
    const dim3 g(grid), bo(ob_block(WAVES, HELPER)); // likelihood waves + owner + three candidate producers (+ helper)
    if (kind == K_ROUND_OB) {
        if (coop) {
            void *params[] = {const_cast<void *>(args)};
            return hipLaunchCooperativeKernel((const void *)pt_round_ob_kernel<MODEL, WAVES, LDS, HELPER>, g, bo, params,
                                              (unsigned int)lds, st);
        }
        hipLaunchKernelGGL((pt_round_ob_kernel<MODEL, WAVES, LDS, HELPER>), g, bo, lds, st, *(const RoundArgs *)args);
    } else {
        // (the default proposal law and swap schedule only: the variants calibrate on the two-phase step)
        if constexpr (MODEL < kVariantModel)
            hipLaunchKernelGGL((pt_calibrate_ob_kernel<MODEL, WAVES, LDS, HELPER>), g, bo, lds, st, *(const CalibArgs *)args);
        else
            return hipErrorInvalidDeviceFunction;
    }
    return hipGetLastError();
}

template <int MODEL, int WAVES, bool LDS>
static hipError_t launch_one(KernelKind kind, bool producers, bool helper, bool coop, int grid, size_t lds, hipStream_t st,
                             const void *args) {
    const dim3 g(grid), b(WAVES * kWave);
    constexpr bool kCanProduce = has_producer(WAVES);
    const dim3 bp(block_threads(WAVES, kCanProduce)); // + producer waves
    switch (kind) {
    case K_ROUND:
        if (coop) {
            // the runtime places the whole grid at once or refuses the launch
            void *params[] = {const_cast<void *>(args)};
            return hipLaunchCooperativeKernel((const void *)pt_round_kernel<MODEL, WAVES, LDS, kCanProduce>, g, bp,
                                              params, (unsigned int)lds, st);
        }
        // (workgroups of four and more waves always carry their producer duty: the variant without
        // it is not instantiated)
        hipLaunchKernelGGL((pt_round_kernel<MODEL, WAVES, LDS, kCanProduce>), g, bp, lds, st, *(const RoundArgs *)args);
        break;
    case K_ROUND_OB:
    case K_CALIB_OB:
        if constexpr (has_one_barrier(WAVES)) {
            if constexpr (ob_can_help(MODEL)) {
                if (helper)
                    return launch_ob<MODEL, WAVES, LDS, true>(kind, coop, grid, lds, st, args);
            }
            if (helper)
                return hipErrorInvalidDeviceFunction;
            return launch_ob<MODEL, WAVES, LDS, false>(kind, coop, grid, lds, st, args);
        } else {
            return hipErrorInvalidDeviceFunction;
        }
    case K_CALC:
        hipLaunchKernelGGL((pt_calc_model_kernel<MODEL % kVariantModel, WAVES, LDS>), g, b, lds, st,
                           *(const RoundArgs *)args);
        break;
    case K_EVAL:
        hipLaunchKernelGGL((pt_loglike_kernel<MODEL % kVariantModel, WAVES, LDS>), g, b, lds, st, *(const EvalArgs *)args);
        break;
    case K_CALIB:
        hipLaunchKernelGGL((pt_calibrate_kernel<MODEL, WAVES, LDS, kCanProduce>), g, bp, lds, st,
                           *(const CalibArgs *)args);
        break;
    }
    return hipGetLastError();
}

// ---- run-time (model, waves) -> compile-time instantiation ----
// A development build can restrict what is instantiated (a full build compiles 4 models x 5
// workgroup shapes x every kernel and takes minutes): -DAPEMOST_DEV_MODELS=<bit per model>
// -DAPEMOST_DEV_WAVES=<bit per wave count>.  The product build has every bit set.
#ifndef APEMOST_DEV_MODELS
#define APEMOST_DEV_MODELS 0xF
#endif
#ifndef APEMOST_DEV_WAVES
#define APEMOST_DEV_WAVES 0x156 // 1, 2, 4, 6, 8
#endif
// the variant instantiations (MODEL + kVariantModel: non-default proposal law / swap schedule) exist
// for the workgroup shapes the engine chooses by itself: 1, 2, 4, 8 waves (build time)
#ifndef APEMOST_DEV_VARIANTS
#define APEMOST_DEV_VARIANTS 0x116
#endif
constexpr bool built(int model, int waves) {
    return ((APEMOST_DEV_MODELS >> (model % kVariantModel)) & 1) && ((APEMOST_DEV_WAVES >> waves) & 1) &&
           (model < kVariantModel || ((APEMOST_DEV_VARIANTS >> waves) & 1));
}

// f.template run<MODEL, WAVES>() for the sampler's model and workgroup shape
template <int MODEL, class F>
static hipError_t dispatch_w(int waves, const F &f) {
    switch (waves) {
    case 1:
        if constexpr (built(MODEL, 1))
            return f.template run<MODEL, 1>();
        break;
    case 2:
        if constexpr (built(MODEL, 2))
            return f.template run<MODEL, 2>();
        break;
    case 4:
        if constexpr (built(MODEL, 4))
            return f.template run<MODEL, 4>();
        break;
    case 6:
        if constexpr (built(MODEL, 6))
            return f.template run<MODEL, 6>();
        break;
    case 8:
        if constexpr (built(MODEL, 8))
            return f.template run<MODEL, 8>();
        break;
    }
    return hipErrorInvalidDeviceFunction; // not part of this (development) build
}

struct LaunchOp {
    KernelKind kind;
    bool lds_data, producers, helper, coop;
    int grid;
    size_t lds;
    hipStream_t st;
    const void *args;
    template <int MODEL, int WAVES>
    hipError_t run() const {
        return lds_data ? launch_one<MODEL, WAVES, true>(kind, producers, helper, coop, grid, lds, st, args)
                        : launch_one<MODEL, WAVES, false>(kind, producers, helper, coop, grid, lds, st, args);
    }
};

// kernels that stage > 64 KiB of data in LDS must opt in once per function
template <int MODEL, int WAVES, bool HELPER>
static hipError_t set_ob_lds_attr(size_t ob_bytes) {
    hipError_t e = hipFuncSetAttribute((const void *)pt_round_ob_kernel<MODEL, WAVES, true, HELPER>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ob_bytes);
    if (e != hipSuccess)
        return e;
    if constexpr (MODEL < kVariantModel)
        e = hipFuncSetAttribute((const void *)pt_calibrate_ob_kernel<MODEL, WAVES, true, HELPER>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ob_bytes);
    return e;
}

template <int MODEL, int WAVES>
static hipError_t set_lds_attr(size_t bytes, size_t ob_bytes) {
    hipError_t e;
    e = hipFuncSetAttribute((const void *)pt_round_kernel<MODEL, WAVES, true, has_producer(WAVES)>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess)
        return e;
    e = hipFuncSetAttribute((const void *)pt_calibrate_kernel<MODEL, WAVES, true, has_producer(WAVES)>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess)
        return e;
    e = hipFuncSetAttribute((const void *)pt_calc_model_kernel<MODEL % kVariantModel, WAVES, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess)
        return e;
    e = hipFuncSetAttribute((const void *)pt_loglike_kernel<MODEL % kVariantModel, WAVES, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess)
        return e;
    if constexpr (has_one_barrier(WAVES)) {
        e = set_ob_lds_attr<MODEL, WAVES, false>(ob_bytes);
        if (e != hipSuccess)
            return e;
        if constexpr (ob_can_help(MODEL)) {
            e = set_ob_lds_attr<MODEL, WAVES, true>(ob_bytes);
            if (e != hipSuccess)
                return e;
        }
    }
    return hipSuccess;
}

struct LdsAttrOp {
    size_t bytes, ob_bytes;
    template <int MODEL, int WAVES>
    hipError_t run() const {
        return set_lds_attr<MODEL, WAVES>(bytes, ob_bytes);
    }
};

// blocks of the round kernel one CU admits (occupancy API: registers, LDS, wave slots)
template <bool LDS>
struct OccupancyOp {
    bool producers;
    bool one_barrier, helper;
    size_t lds_bytes;
    int *blocks;
    template <int MODEL, int WAVES>
    hipError_t run() const {
        constexpr bool kCanProduce = has_producer(WAVES);
        if constexpr (has_one_barrier(WAVES)) {
            if constexpr (ob_can_help(MODEL)) {
                if (one_barrier && helper)
                    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, pt_round_ob_kernel<MODEL, WAVES, LDS, true>,
                                                                        ob_block(WAVES, true), lds_bytes);
            }
            if (one_barrier)
                return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, pt_round_ob_kernel<MODEL, WAVES, LDS, false>,
                                                                    ob_block(WAVES, false), lds_bytes);
        }
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, pt_round_kernel<MODEL, WAVES, LDS, kCanProduce>,
                                                            block_threads(WAVES, kCanProduce), lds_bytes);
    }
};

// ---- one entry point per model: what a model's translation unit exports ----
struct AnyOp {
    enum { LAUNCH, LDS_ATTR, OCCUPANCY_LDS, OCCUPANCY_PLAIN } what;
    LaunchOp launch;
    LdsAttrOp lds;
    OccupancyOp<true> occ_lds;
    OccupancyOp<false> occ_plain;
};

template <int MODEL>
hipError_t model_dispatch(int waves, const AnyOp &op) {
    switch (op.what) {
    case AnyOp::LAUNCH:
        return dispatch_w<MODEL>(waves, op.launch);
    case AnyOp::LDS_ATTR:
        return dispatch_w<MODEL>(waves, op.lds);
    case AnyOp::OCCUPANCY_LDS:
        return dispatch_w<MODEL>(waves, op.occ_lds);
    case AnyOp::OCCUPANCY_PLAIN:
        return dispatch_w<MODEL>(waves, op.occ_plain);
    }
    return hipErrorInvalidValue;
}

#endif // __HIPCC_RTC__

} // namespace apemost
