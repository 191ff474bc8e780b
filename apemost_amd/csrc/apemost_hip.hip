// apemost_hip.hip -- kernels and C ABI of the gfx950 parallel-tempering engine
// (declared in include/apemost_hip.h).  Written for MI355X only.
#include "pt_device.h"
#include "pt_onebarrier.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace apemost;

// ===========================================================================
// kernels
// ===========================================================================

// LDS carve (doubles): proposed params [2][64], wave partials [2][16], 8 control words
constexpr int kFixedLdsDoubles = 2 * kWave + 32 + 8;

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
// (steps/s, one-barrier 4 / one-barrier 8 / two-phase 4; tools/gpu_exp_ob4.sh):
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
__device__ __forceinline__ bool wait_at_least(const u64 *word, u64 want, u64 *timeout_word) {
    for (unsigned spins = 0; ld_agent(word) < want; spins++) {
        __builtin_amdgcn_s_sleep(2);
        if (spins > 8000000u) {
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
        wait_at_least(d.acked() + p, memo.index(half) + 1, d.timeout_word());
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
    if (e.lane == 0)
        st_agent(d.published() + c, swap_index + 1);
    if (!wait_at_least(d.published() + partner, swap_index + 1, d.timeout_word()))
        return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    swap_apply(e, d, sh, c, half, swap_index, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (e.lane == 0)
        st_agent(d.acked() + c, swap_index + 1);
    memo.set(half, partner, swap_index);
}

template <int MODEL, int WAVES, bool LDS_DATA, bool PROD>
__global__ __launch_bounds__(block_threads(WAVES, PROD)) void pt_round_kernel(const RoundArgs a) {
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
#ifdef APEMOST_STAMPS
#define OB_STAMP_DECL u64 ob_busy = 0, ob_t0 = 0, ob_total = 0, ob_prev = 0
#define OB_STAMP_BEGIN ob_t0 = __builtin_amdgcn_s_memtime()
#define OB_STAMP_END ob_busy += __builtin_amdgcn_s_memtime() - ob_t0
#define OB_STAMP_FLUSH                                                                            \
    if (blockIdx.x == 0 && e.lane == 0)                                                           \
    atomicAdd(&g_stamps[e.hw], ob_busy)
#else
#define OB_STAMP_DECL
#define OB_STAMP_BEGIN
#define OB_STAMP_END
#define OB_STAMP_FLUSH
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

template <int MODEL, int LW, bool LDS_DATA>
__global__ __launch_bounds__((LW + 4) * kWave) __attribute__((amdgpu_waves_per_eu(4))) void pt_round_ob_kernel(const RoundArgs a) {
    extern __shared__ __align__(16) double lds[];
    ObEngine<MODEL, LW, LDS_DATA> e;
    const int c = blockIdx.x;
    e.setup_common(a.d, a.sh, c, lds);
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
        e.setup_lanes(a.sh);
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
                if (e.redraw_pending(p))
                    __syncthreads();
                e.producer_step();
                OB_STAMP_END;
                __syncthreads();
                p ^= 1;
            }
        }
        OB_STAMP_FLUSH;
    } else {
        // the others wait for this wave at every barrier and it has little to issue: let it go first
#ifndef APEMOST_OWNER_PRIO
#define APEMOST_OWNER_PRIO 3
#endif
        __builtin_amdgcn_s_setprio(APEMOST_OWNER_PRIO);
        e.setup_lanes(a.sh);
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
                // one batch of LDS reads: the redraw flag, what the prepared proposals settled on
                // for my parameter, and (owner_results) the partial sums
                const int pending = *e.s_flag(p);
                if (open) {
                    e.owner_fetch_selected(p);
                    e.owner_results(p, my_sample);
                    if (my_sample)
                        my_sample += sample_stride;
                }
                const bool redraw_pending = __builtin_amdgcn_readfirstlane(pending) != 0;
                e.owner_choose(p, !open);
                if (redraw_pending) // rare: a proposal in LDS has just been replaced
                    __syncthreads();
                e.owner_publish(p);
                e.tick++;
                OB_STAMP_END;
                __syncthreads();
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
        if (blockIdx.x == 0 && e.lane == 0)
            atomicAdd(&g_stamps[15], ob_total);
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

// ---- calibration: burn_in + markov_chain_calibrate_orig as a per-chain state machine ----
struct CalibArgs {
    DevArrays d;
    ChainShape sh;
    int cur;
    int first;        // first local chain
    int burn_in_only; // -DSKIP_CALIBRATE_ALLCHAINS
    apemost_hip_calib_config cfg;
    int *status;      // [count]
    u64 *iters;       // [count]
};

template <int MODEL, int WAVES, bool LDS_DATA, bool PROD>
__global__ __launch_bounds__(block_threads(WAVES, PROD)) void pt_calibrate_kernel(const CalibArgs a) {
    extern __shared__ __align__(16) double lds[];
    // control word decided by wave 0, read by every wave (kept inside the dynamic
    // region so the carve base stays 16-byte aligned)
    volatile int &s_ctl = *(volatile int *)(lds + 2 * kWave + 32);
    Engine<MODEL, WAVES, LDS_DATA, PROD> e;
    const int c = a.first + blockIdx.x;
    const int n = a.sh.n_par;
    engine_setup(e, a.d, a.sh, c, lds);
    chain_load(e, a.d, a.sh, c, a.cur);
    e.pin_uniforms();
    e.producer_prologue();
    __syncthreads();
    e.cache_rows();
    e.producer_first_fetch();
    const bool w0 = (e.wave == 0);
    const apemost_hip_calib_config &cfg = a.cfg;

    // ---- burn_in: src/markov_chain.c:34-79 ----
    const double original_step = e.stepw;
    e.stepw = (e.hi - e.lo) * 0.1;
    unsigned long iter = 0;
    for (; iter < cfg.burn_in_iterations / 2;) {
        for (int sub = 0; sub < 200; sub++)
            e.step(-1);
        iter += 200;
        if (w0)
            e.check_best();
    }
    if (w0)
        e.restart_from_best();
    e.stepw *= 0.5;
    for (; iter < cfg.burn_in_iterations;) {
        for (int sub = 0; sub < 200; sub++)
            e.step(-1);
        iter += 200;
        if (w0)
            e.check_best();
    }
    e.stepw = original_step;

    int status = 0;
    unsigned long sweeps = 0;
    if (!a.burn_in_only) {
        // ---- markov_chain_calibrate_orig: src/markov_chain_calibrate.c:1039-1180 ----
        double rat_limit = pow(cfg.rat_limit, 1.0 / n);
        int nchecks_without_rescaling = 0;
        e.stepw *= cfg.adjust_step;
        e.reset_accept_rejects();
        while (true) {
            for (int p = 0; p < n; p++) {
                e.step(p);
                if (w0)
                    e.check_best();
            }
            sweeps++;
            if (sweeps % cfg.iter_readjust != 0)
                continue;
            // per-parameter rescaling; lane p decides for parameter p, the wave
            // combines the decisions in parameter order like the reference's loop
            int rescaled = 0, fail = 0;
            if (w0) {
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
                if (e.tid == 0)
                    s_ctl = fail;
            }
            __syncthreads();
            fail = s_ctl;
            __syncthreads();
            if (fail) {
                status = 1;
                break;
            }
            if (w0) {
                if (rescaled == 0)
                    nchecks_without_rescaling++;
                e.restart_from_best();
                e.reset_accept_rejects();
            }
            for (unsigned sub = 0; sub < cfg.iter_readjust; sub++) {
                e.step(-1);
                if (w0)
                    e.check_best();
            }
            int ctl = 0; // 0 continue, 1 converged, 2 iteration limit
            if (w0) {
                const double delta =
                    (double)e.accept / (double)(e.accept + e.reject) - cfg.target_global;
                int reached_perfection;
                if ((delta < 0 ? -delta : delta) < cfg.max_ar_deviation) {
                    reached_perfection = 1;
                } else {
                    reached_perfection = 0;
                    if (delta < 0)
                        rat_limit /= 0.99;
                    else
                        rat_limit *= 0.99;
                }
                if (nchecks_without_rescaling >= cfg.no_rescaling_limit && reached_perfection == 1 &&
                    rescaled == 0)
                    ctl = 1;
                else if (sweeps > cfg.iter_limit)
                    ctl = 2;
                if (e.tid == 0)
                    s_ctl = ctl;
            }
            __syncthreads();
            ctl = s_ctl;
            __syncthreads();
            if (ctl == 1)
                break;
            if (ctl == 2) {
                status = 2;
                break;
            }
        }
        if (status == 0 && w0)
            e.reset_accept_rejects();
    }
    if (e.tid == 0) {
        a.status[blockIdx.x] = status;
        a.iters[blockIdx.x] = sweeps;
        if (*e.fail_flag())
            st_agent(a.d.timeout_word(), 3);
    }
    // calibration leaves the chain in place: same half of the double buffer
    chain_store(e, a.d, a.sh, c, a.cur, true);
}

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
    int *d_status;
    u64 *d_iters;
    int calib_capacity;
    int calib_pending; // chains of a calibrate_begin whose results calibrate_end has not collected yet
    double *edge_out, *edge_in;  // edge records for in-process shard exchanges (created on first use)
    hipEvent_t ev_exported, ev_imported;
    hipStream_t copy_stream; // drains sample rows while the next launch runs (created on first use)
    hipEvent_t ev_copy;
    u64 *h_word;             // pinned copy of the launch error word, refreshed by every async read
    bool one_barrier;    // stepping launches use pt_round_ob_kernel
    int kmodel;          // template argument of this sampler's kernels: cfg.model, + kVariantModel when a
                         // non-default proposal law or swap schedule is asked for (pt_device.h)
    bool cooperative;    // multi-round launches through hipLaunchCooperativeKernel
    bool handoff_failed; // an in-launch hand-off timed out once: single-round launches from then on
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

static int enable_big_lds(apemost_hip_sampler *s);
template <bool LDS>
static hipError_t round_occupancy(int model, int waves, bool producers, bool one_barrier, size_t lds_bytes, int *blocks);
static size_t ob_lds_bytes(const apemost_hip_sampler *s, bool lds_data);

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
    // The pulse likelihood -- a loop over the modes that reads its parameters from LDS as it goes, a
    // longer chain per point than the others -- is the exception: eight waves up to 256 chains
    // (256 x 1024: 1.51 vs 1.41e8, 128 x 1024: 9.0 vs 8.2e7; pulse_vrot and sine3 stay with four:
    // 8.9 vs 8.4e7 and 1.05 vs 1.03e8 at 128 x 1024; tools/gpu_exp_w8.sh).
    // Two waves per chain only while such a ladder is still resident (three or four two-wave
    // workgroups per CU: up to 768 chains; 700 x 1024: 2.55 vs 2.46e8); beyond that a launch would
    // hold one round and one wave per chain does more (900 x 1024: 3.11 vs 1.68e8).
    int by_chip = (c.n_chains <= 128 && c.n_data >= 8192) ? 8 : c.n_chains <= 512 ? 4 : c.n_chains <= 768 ? 2 : 1;
    if (c.model == APEMOST_MODEL_PULSE && c.n_chains <= 256)
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
    if (s->d_status)
        hipFree(s->d_status);
    if (s->d_iters)
        hipFree(s->d_iters);
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
    s->lds_data = fixed_max + data_lds <= 160 * 1024 - 1024 && cfg->lds_policy != 2 &&
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
    s->kmodel = cfg->model + ((cfg->flags & (APEMOST_HIP_FLAG_PROPOSAL_LOGISTIC | APEMOST_HIP_FLAG_PROPOSAL_UNIFORM |
                                             APEMOST_HIP_FLAG_RANDOMSWAP))
                                  ? kVariantModel
                                  : 0);
    s->sh.x_abs_max = INFINITY; // until set_data
    HIP_TRY(hipStreamSynchronize(s->stream));
    if ((rc = enable_big_lds(s)))
        return rc;
    {
        // Multi-round launches need every workgroup resident at once.  Blocks per CU from the
        // occupancy query, one held back (the query over-reports by one block for scalar-register-
        // heavy kernels: MI355X_MICROARCH.md "Residency and cooperative launch").
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
        // the round kernel this sampler's stepping launches use: one barrier per step where that
        // variant exists (8 likelihood waves per chain), the classic two-phase step otherwise
        s->one_barrier = has_one_barrier(s->waves) && !(cfg->flags & APEMOST_HIP_FLAG_TWO_BARRIER_STEP);
        int b_lds = 0, b_plain = 0;
        if (s->lds_data)
            HIP_TRY(round_occupancy<true>(s->kmodel, s->waves, s->producers, s->one_barrier,
                                          s->one_barrier ? ob_lds_bytes(s, true) : s->lds_bytes, &b_lds));
        HIP_TRY(round_occupancy<false>(s->kmodel, s->waves, s->producers, s->one_barrier,
                                       s->one_barrier ? ob_lds_bytes(s, false) : s->lds_fixed_bytes, &b_plain));
        const long long cus = prop.multiProcessorCount;
        s->resident_lds = s->lds_data && (long long)cfg->n_chains <= (long long)(b_lds - 1) * cus;
        s->resident_plain = (long long)cfg->n_chains <= (long long)(b_plain - 1) * cus;
        if ((long long)cfg->n_chains * 2 <= cus) { // at most one workgroup per two CUs
            s->resident_lds = s->lds_data;
            s->resident_plain = true;
        }
        s->resident_ok = s->resident_lds || s->resident_plain;
        s->cooperative = (cfg->flags & APEMOST_HIP_FLAG_COOPERATIVE_LAUNCH) && prop.cooperativeLaunch;
        if (!s->resident_ok && prop.cooperativeLaunch && s->one_barrier) {
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
        if (cfg->flags & (APEMOST_HIP_FLAG_SINGLE_ROUND_LAUNCHES | APEMOST_HIP_FLAG_ADAPT))
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
    if (cfg->chain_offset < 0 || cfg->chain_offset + cfg->n_chains > cfg->n_chains_global)
        return fail(APEMOST_HIP_ERR_INVALID, "shard [%lld,%lld) outside ladder of %lld chains",
                    (long long)cfg->chain_offset, (long long)(cfg->chain_offset + cfg->n_chains),
                    (long long)cfg->n_chains_global);
    if (cfg->n_par < 64 && (cfg->circular_params >> cfg->n_par) != 0)
        return fail(APEMOST_HIP_ERR_INVALID, "circular_params names a parameter beyond n_par");
    if (cfg->flags & ~(APEMOST_HIP_FLAG_SINGLE_ROUND_LAUNCHES | APEMOST_HIP_FLAG_COOPERATIVE_LAUNCH |
                       APEMOST_HIP_FLAG_TWO_BARRIER_STEP | APEMOST_HIP_FLAG_PROPOSAL_LOGISTIC |
                       APEMOST_HIP_FLAG_PROPOSAL_UNIFORM | APEMOST_HIP_FLAG_RANDOMSWAP | APEMOST_HIP_FLAG_ADAPT |
                       APEMOST_HIP_FLAG_TEST_REFUSE_COOPERATIVE))
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
    default:
        return fail(APEMOST_HIP_ERR_UNSUPPORTED, "unknown device model %d", cfg->model);
    }
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
    s->d_status = nullptr;
    s->d_iters = nullptr;
    s->calib_capacity = 0;
    s->calib_pending = 0;
    s->stream = nullptr;
    s->ev0 = s->ev1 = nullptr;
    s->copy_stream = nullptr;
    s->ev_copy = nullptr;
    s->edge_out = s->edge_in = nullptr;
    s->ev_exported = s->ev_imported = nullptr;
    s->h_word = nullptr;
    s->handoff_failed = false;
    s->waves = choose_waves(*cfg);
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

// ---- launch dispatch over (model, waves, lds) ----
enum KernelKind { K_ROUND, K_ROUND_OB, K_CALC, K_EVAL, K_CALIB };

template <int MODEL, int WAVES, bool LDS>
static hipError_t launch_one(KernelKind kind, bool producers, bool coop, int grid, size_t lds, hipStream_t st,
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
        if constexpr (has_one_barrier(WAVES)) {
            const dim3 bo((WAVES + 4) * kWave); // + owner + three candidate producers
            if (coop) {
                void *params[] = {const_cast<void *>(args)};
                return hipLaunchCooperativeKernel((const void *)pt_round_ob_kernel<MODEL, WAVES, LDS>, g, bo, params,
                                                  (unsigned int)lds, st);
            }
            hipLaunchKernelGGL((pt_round_ob_kernel<MODEL, WAVES, LDS>), g, bo, lds, st, *(const RoundArgs *)args);
        } else {
            return hipErrorInvalidDeviceFunction;
        }
        break;
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

template <class F>
static hipError_t dispatch(int model, int waves, const F &f) {
    switch (model) {
    case APEMOST_MODEL_SIMPLESIN:
        return dispatch_w<APEMOST_MODEL_SIMPLESIN>(waves, f);
    case APEMOST_MODEL_PULSE:
        return dispatch_w<APEMOST_MODEL_PULSE>(waves, f);
    case APEMOST_MODEL_PULSE_VROT:
        return dispatch_w<APEMOST_MODEL_PULSE_VROT>(waves, f);
    case APEMOST_MODEL_SINE3:
        return dispatch_w<APEMOST_MODEL_SINE3>(waves, f);
    case kVariantModel + APEMOST_MODEL_SIMPLESIN:
        return dispatch_w<kVariantModel + APEMOST_MODEL_SIMPLESIN>(waves, f);
    case kVariantModel + APEMOST_MODEL_PULSE:
        return dispatch_w<kVariantModel + APEMOST_MODEL_PULSE>(waves, f);
    case kVariantModel + APEMOST_MODEL_PULSE_VROT:
        return dispatch_w<kVariantModel + APEMOST_MODEL_PULSE_VROT>(waves, f);
    case kVariantModel + APEMOST_MODEL_SINE3:
        return dispatch_w<kVariantModel + APEMOST_MODEL_SINE3>(waves, f);
    }
    return hipErrorInvalidDeviceFunction;
}

struct LaunchOp {
    KernelKind kind;
    bool lds_data, producers, coop;
    int grid;
    size_t lds;
    hipStream_t st;
    const void *args;
    template <int MODEL, int WAVES>
    hipError_t run() const {
        return lds_data ? launch_one<MODEL, WAVES, true>(kind, producers, coop, grid, lds, st, args)
                        : launch_one<MODEL, WAVES, false>(kind, producers, coop, grid, lds, st, args);
    }
};

static size_t ob_lds_bytes(const apemost_hip_sampler *s, bool lds_data) {
    return (size_t)kObFixedDoubles * sizeof(double) + (lds_data ? (size_t)2 * s->cfg.n_data * sizeof(double) : 0);
}

// stage_data: a launch that walks the data vector only a few times (n_swap < 4, single
// likelihood evaluations) reads it through L2 instead of copying it into LDS first
static int launch(apemost_hip_sampler *s, KernelKind kind, int grid, const void *args, bool stage_data = true,
                  bool coop = false) {
    LaunchOp op;
    op.kind = kind;
    op.lds_data = s->lds_data && stage_data;
    op.producers = s->producers;
    op.coop = coop;
    op.grid = grid;
    op.lds = kind == K_ROUND_OB ? ob_lds_bytes(s, op.lds_data) : op.lds_data ? s->lds_bytes : s->lds_fixed_bytes;
    op.st = s->stream;
    op.args = args;
    if (coop && (s->cfg.flags & APEMOST_HIP_FLAG_TEST_REFUSE_COOPERATIVE)) // test hook: see the header
        return fail(APEMOST_HIP_ERR_RUNTIME, "kernel launch failed: cooperative launch refused (test hook)");
    const hipError_t err = dispatch(s->kmodel, s->waves, op);
    if (err != hipSuccess)
        return fail(APEMOST_HIP_ERR_RUNTIME, "kernel launch failed: %s", hipGetErrorString(err));
    return APEMOST_HIP_OK;
}

static int enable_big_lds(apemost_hip_sampler *s);

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

// kernels that stage > 64 KiB of data in LDS must opt in once per function
template <int MODEL, int WAVES>
static hipError_t set_lds_attr(size_t bytes) {
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
        e = hipFuncSetAttribute((const void *)pt_round_ob_kernel<MODEL, WAVES, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes + 2048);
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

struct LdsAttrOp {
    size_t bytes;
    template <int MODEL, int WAVES>
    hipError_t run() const {
        return set_lds_attr<MODEL, WAVES>(bytes);
    }
};

// blocks of the round kernel one CU admits (occupancy API: registers, LDS, wave slots)
template <bool LDS>
struct OccupancyOp {
    bool producers;
    bool one_barrier;
    size_t lds_bytes;
    int *blocks;
    template <int MODEL, int WAVES>
    hipError_t run() const {
        constexpr bool kCanProduce = has_producer(WAVES);
        if constexpr (has_one_barrier(WAVES)) {
            if (one_barrier)
                return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, pt_round_ob_kernel<MODEL, WAVES, LDS>,
                                                                    (WAVES + 4) * kWave, lds_bytes);
        }
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, pt_round_kernel<MODEL, WAVES, LDS, kCanProduce>,
                                                            block_threads(WAVES, kCanProduce), lds_bytes);
    }
};

template <bool LDS>
static hipError_t round_occupancy(int model, int waves, bool producers, bool one_barrier, size_t lds_bytes, int *blocks) {
    OccupancyOp<LDS> op;
    op.producers = producers;
    op.one_barrier = one_barrier;
    op.lds_bytes = lds_bytes;
    op.blocks = blocks;
    return dispatch(model, waves, op);
}

static int enable_big_lds(apemost_hip_sampler *s) {
    if (!s->lds_data || s->lds_bytes <= 64 * 1024)
        return APEMOST_HIP_OK;
    LdsAttrOp op;
    op.bytes = s->lds_bytes;
    const hipError_t e = dispatch(s->kmodel, s->waves, op);
    if (e != hipSuccess)
        return fail(APEMOST_HIP_ERR_RUNTIME, "hipFuncSetAttribute(LDS %zu B): %s", s->lds_bytes,
                    hipGetErrorString(e));
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
    if ((s->cfg.flags & APEMOST_HIP_FLAG_ADAPT) && n_steps > 0 && which < 0) {
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
    int rc;
    for (uint64_t r = 0; r < n_rounds || (r == n_rounds && sh[0]->swap_pending);) {
        const bool finalise = r == n_rounds; // the swap attempt that closes the last round
        const int pending = sh[0]->swap_pending;
        const u64 first_inside = sh[0]->round + (pending ? 1 : 0);
        uint64_t limit = finalise || n_swap == 0 ? 1 : n_rounds - r;
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
}

// markov_chain_calibrate for chains [first, first+count): the launch ...
extern "C" int apemost_hip_calibrate_begin(apemost_hip_sampler *s, int32_t first, int32_t count,
                                           const apemost_hip_calib_config *c, int burn_in_only) {
    CHECK_S(s);
    if (!c || first < 0 || count < 1 || first + count > s->cfg.n_chains)
        return fail(APEMOST_HIP_ERR_INVALID, "calibrate_chains: chains [%d,%d) outside [0,%d)", first,
                    first + count, s->cfg.n_chains);
    if (c->iter_readjust == 0)
        return fail(APEMOST_HIP_ERR_INVALID, "iter_readjust must be > 0");
    if (s->calib_pending)
        return fail(APEMOST_HIP_ERR_INVALID, "calibrate_begin: the previous calibration was not collected (calibrate_end)");
    if (count > s->calib_capacity) {
        if (s->d_status)
            hipFree(s->d_status);
        if (s->d_iters)
            hipFree(s->d_iters);
        s->d_status = nullptr;
        s->d_iters = nullptr;
        s->calib_capacity = 0;
        HIP_TRY(hipMalloc((void **)&s->d_status, count * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&s->d_iters, count * sizeof(u64)));
        s->calib_capacity = count;
    }
    CalibArgs a;
    a.d = s->d;
    a.sh = s->sh;
    a.cur = s->cur;
    a.first = first;
    a.burn_in_only = burn_in_only ? 1 : 0;
    a.cfg = *c;
    a.status = s->d_status;
    a.iters = s->d_iters;
    const int rc = launch(s, K_CALIB, count, &a);
    if (rc)
        return rc;
    s->calib_pending = count;
    return APEMOST_HIP_OK;
}

// ... and its results: status[count] / iters[count] of the chains of the matching begin
extern "C" int apemost_hip_calibrate_end(apemost_hip_sampler *s, int32_t *status, uint64_t *iters) {
    CHECK_S(s);
    const int count = s->calib_pending;
    if (count <= 0)
        return fail(APEMOST_HIP_ERR_INVALID, "calibrate_end without calibrate_begin");
    s->calib_pending = 0;
    std::vector<int> st(count);
    std::vector<u64> it(count);
    HIP_TRY(hipMemcpyAsync(st.data(), s->d_status, count * sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(it.data(), s->d_iters, count * sizeof(u64), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    int rc = check_handoff(s);
    if (rc)
        return rc;
    int worst = 0;
    for (int i = 0; i < count; i++) {
        if (status)
            status[i] = st[i];
        if (iters)
            iters[i] = it[i];
        if (st[i] && !worst)
            worst = st[i];
    }
    if (worst)
        return fail(APEMOST_HIP_ERR_CALIBRATION, "calibration failed: %s",
                    worst == 1 ? "a step width became too large" : "iteration limit reached");
    return APEMOST_HIP_OK;
}

extern "C" int apemost_hip_calibrate_chains(apemost_hip_sampler *s, int32_t first, int32_t count,
                                            const apemost_hip_calib_config *c, int burn_in_only,
                                            int32_t *status, uint64_t *iters) {
    const int rc = apemost_hip_calibrate_begin(s, first, count, c, burn_in_only);
    return rc ? rc : apemost_hip_calibrate_end(s, status, iters);
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
