#!/usr/bin/env python3
"""bench.py -- Metropolis steps/s (all chains) of the parallel-tempering hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 under torch.distributed.run, one
rank per GPU over RCCL).  One bench "step" = one pass of the hot path over one batch of synthetic
work: LAUNCHES_PER_STEP launches of ROUNDS_PER_STEP rounds of {n_swap Metropolis steps per chain + one
swap attempt} (run_sampler's loop body, src/parallel_tempering.c:392-409), sample rows written to HBM
(20 steps are >= 5 s of GPU time).
With --gpus N > 1 and no torch.distributed environment, bench.py starts the N ranks itself (fresh child
processes, one per GPU; the parent never touches a GPU) and rank 0 prints the line.
Workload at every N: BASELINE config 2 per GPU -- simplesin, 128 chains x 1024 data points per
GPU (weak scaling: the ladder has 128*N chains, block-partitioned over the ranks).
`--config 3|4|5` selects the other GPU configs of BASELINE.json (per-GPU share of the ladder,
table CONFIGS below); the default line the driver records stays config 2.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6  # fp64 vector peak

# BASELINE.json configs as one GPU sees them: workload, chains per GPU, data points, n_swap
# (0 = the reference's rule 2000/n_beta of the per-GPU ladder; ladders beyond 2000 chains need an
# explicit value), burn-in of the device calibration that precedes the timed steps
# rounds_per_step: rounds of one bench step = of one launch where the ladder is resident (the C host
# launches up to apemost_hip_max_rounds_per_launch rounds at a time, bounded by its sample buffers):
# chosen so that a launch is ~1 ms of work and its fixed cost (launch, staging, pipeline start) is small
# launches_per_step: a bench step is that many such batches back to back, so that the driver's 20 steps are
# >= 5 s of GPU time at every config (its utilisation sampler and its own clock then see the kernel: the
# timed region was 25 ms in round 2 and ~1 s in round 3, where the sampler's three looks all read 0 %);
# `value` does not depend on it
CONFIGS = {
    2: dict(workload="simplesin", chains_per_gpu=128, n_data=1024, n_swap=0, burn_in=10000, rounds_per_step=128, launches_per_step=260),
    3: dict(workload="sine3", chains_per_gpu=1024, n_data=8192, n_swap=0, burn_in=2000, rounds_per_step=32, launches_per_step=480),
    4: dict(workload="pulse", chains_per_gpu=256, n_data=1024, n_swap=1, burn_in=2000, rounds_per_step=256, launches_per_step=1100),  # 2048 / 8 GPUs
    5: dict(workload="pulse_vrot", chains_per_gpu=2048, n_data=65536, n_swap=1, burn_in=600, rounds_per_step=32, launches_per_step=70),  # 16384 / 8 GPUs
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config number")
    ap.add_argument("--workload", default=None)
    ap.add_argument("--chains-per-gpu", type=int, default=None)
    ap.add_argument("--n-data", type=int, default=None)
    ap.add_argument("--n-swap", type=int, default=None, help="0 = reference rule 2000/n_beta of the per-GPU ladder")
    ap.add_argument("--burn-in", type=int, default=None, help="BURN_IN_ITERATIONS of the device calibration")
    ap.add_argument("--calib-iter-limit", type=int, default=None, help="ITER_LIMIT of the device calibration (experiments: a short one)")
    ap.add_argument("--flags", type=int, default=0, help="apemost_hip_config.flags")
    ap.add_argument("--rounds-per-step", type=int, default=None)
    ap.add_argument("--launches-per-step", type=int, default=None, help="batches of rounds-per-step rounds in one bench step")
    ap.add_argument("--n-swap-rule", default="per_gpu_ladder", choices=["per_gpu_ladder", "reference"],
                    help="--n-swap 0: 2000 / chains per GPU (the per-GPU work stays what it is at one GPU: weak "
                         "scaling), or the reference's 2000 / n_beta of the whole ladder (src/parallel_tempering.c:228-231)")
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--lds", type=int, default=0, help="0 choose, 1 stage the data vector in LDS, 2 read it through L2")
    ap.add_argument("--no-samples", action="store_true", help="do not write per-step sample rows")
    ap.add_argument("--no-calibrate", action="store_true", help="skip the device calibration before the timed steps")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg; 0 = skip")
    ap.add_argument("--calib-dump", default=None, help="write the calibration's per-chain status and sweep counts (JSON)")
    ap.add_argument("--maps-dump", default=None, help="debug: write /proc/self/maps there before exiting (to assign the "
                                                      "addresses of a native stack trace to libraries)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo + --same-device rehearses the N > 1 path on a one-GPU box "
                         "(RCCL refuses two ranks on one device)")
    ap.add_argument("--same-device", action="store_true", help="every rank uses cuda:0 (rehearsal of N > 1 on one GPU)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (nccl) even with one rank")
    ap.add_argument("--no-torch", action="store_true",
                    help="one rank without torch in the process: sample rows in a buffer of the engine's own, one HIP "
                         "runtime (what the profile scripts run: tools/profile_config.sh)")
    a = ap.parse_args()
    for k, v in CONFIGS[a.config].items():
        if getattr(a, k, None) is None:
            setattr(a, k, v)
    return a


def spawn_ranks(n):
    """`python bench.py --gpus N` without torch.distributed.run: N fresh child processes of this same command
    line, one rank per GPU, rendezvous on 127.0.0.1 at a free port.  This process never initialises a GPU
    (no exec after a GPU call, no fork of a process that holds one); rank 0's stdout is ours, so its ONE
    JSON line is the line.  Returns the exit code: the first failing rank's, after the others are ended."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:        # one rank is gone: the others would wait for it until a timeout
                    q.terminate()
        time.sleep(0.05)
    return rc


class EngineRows:
    """sample rows [rounds][doubles per round] in a device buffer of the engine's own (--no-torch): what
    ShardedLadder.run_sampler needs of a tensor -- slices along the rounds and their device address"""

    def __init__(self, sampler, rounds, steps_per_round, base=None, first=0):
        import ctypes as C
        self.s, self.rounds, self.steps, self.first = sampler, rounds, steps_per_round, first
        self.per_round = steps_per_round * sampler.n_chains * (sampler.n_par + 2)
        if base is None:
            from apemost_amd import capi
            p = C.c_void_p()
            capi.check(sampler.L.apemost_hip_samples_alloc(sampler._h, rounds * steps_per_round, C.byref(p)))
            base = p.value
        self.base = base

    def __getitem__(self, sl):
        lo, hi, _ = sl.indices(self.rounds)
        return EngineRows(self.s, hi - lo, self.steps, self.base, self.first + lo)

    def data_ptr(self):
        return self.base + 8 * self.first * self.per_round


def host_cores():
    """CPU threads this process may really use: the affinity mask capped by the cgroup CPU quota
    (the GPU box exposes 256 hardware threads but grants a share of them)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(w, st_template, n_swap, seconds):
    """Times the CPU oracle (a port of the reference's algorithm with per-chain RNG streams and the
    loop counter privatised) on all host cores, on a bounded sample of the workload."""
    from oracle import oracle as orc
    from tests.helpers import to_oracle
    cores = host_cores()
    lad = orc.Ladder(w.model, st_template.n_chain, w.n_par, w.data)
    to_oracle(st_template, lad)
    rng = orc.Rng(orc.RNG_STREAMS, 1, lad)
    t0 = time.time()
    orc.run_sampler(lad, rng, 8, n_swap, n_threads=cores)
    per_round = max((time.time() - t0) / 8, 1e-6)
    rounds = max(8, int(seconds / per_round))
    t0 = time.time()
    orc.run_sampler(lad, rng, rounds, n_swap, n_threads=cores)
    dt = time.time() - t0
    steps = rounds * n_swap * st_template.n_chain
    return {"value": steps / dt, "unit": "Metropolis steps/s", "cores": cores, "kind": "port",
            "sample": "%d rounds x %d steps x %d chains of the same workload in %.1f s (oracle, OpenMP over chains)"
                      % (rounds, n_swap, st_template.n_chain, dt)}


def kernel_sources_sha1():
    """sha1 over the kernel sources (csrc/*.h and the kernels' translation unit apemost_model.hip -- not
    apemost_hip.hip, the host side of the ABI, whose changes leave every kernel's code object what it was):
    tools/summarize_profile.py stores it with the counters it registers in profiles/pmc_traffic.json, so that a line can
    say whether they belong to this build"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "apemost_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")) and f != "apemost_hip.hip":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def expected_kernel(model, waves, lds, one_barrier, flags, helper=False):
    """the instantiation the sampler's run phase launches, as rocprofv3 names it (pt_kernels.h launch_one;
    the proposal-law / RANDOMSWAP variants are MODEL + 8; workgroups of >= 4 waves of the two-phase kernel
    carry the candidate producers)"""
    variant = 8 if flags & (8 | 16 | 32) else 0
    b = "true" if lds else "false"
    if one_barrier:
        return "apemost::pt_round_ob_kernel<%d, %d, %s, %s>" % (model + variant, waves, b, "true" if helper else "false")
    return "apemost::pt_round_kernel<%d, %d, %s, %s>" % (model + variant, waves, b, "true" if waves >= 4 else "false")


def calibration_block(wall_s, status, iters, ccfg, n_par):
    """What the device calibration before the timed region did (markov_chain_calibrate for every chain
    of the shard: burn_in + markov_chain_calibrate_orig, src/markov_chain_calibrate.c:1039-1180).
    A sweep = n_par single-parameter updates; every iter_readjust sweeps are followed by iter_readjust
    all-parameter updates; burn_in rounds its two halves up to blocks of 200 (src/markov_chain.c:34-79).
    Every update is one likelihood evaluation."""
    it = np.asarray(iters, dtype=np.int64)
    half, full = ccfg.burn_in_iterations // 2, ccfg.burn_in_iterations
    burn = -(-half // 200) * 200 if half > 0 else 0
    while burn < full:
        burn += 200
    cycles = it // ccfg.iter_readjust
    evals = burn + it * n_par + cycles * ccfg.iter_readjust
    q = [int(x) for x in np.percentile(it, [0, 25, 50, 75, 90, 99, 100])]
    return {"wall_s": wall_s, "chains": int(it.size), "ok": int((np.asarray(status) == 0).sum()),
            "sweeps_total": int(it.sum()), "evaluations_total": int(evals.sum()),
            "evaluations_per_s": float(evals.sum() / wall_s),
            "evaluations_slowest_chain": int(evals.max()),
            "sweeps_percentiles_0_25_50_75_90_99_100": q,
            # chains still running as a share of chain-time until the slowest one ends: 1 = no tail
            "active_share": float(evals.sum() / (evals.max() * it.size))}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1 and "RANK" not in os.environ:
            raise SystemExit(spawn_ranks(a.gpus))   # (before anything in this process has touched a GPU)
        a.gpus = world

    if os.environ.get("APEMOST_BENCH_TEST_FAIL_RANK") == str(rank):   # test hook (tests/test_gpu_bench_ranks.py)
        raise SystemExit(7)
    if a.no_torch:
        if world > 1 or a.force_dist:
            raise SystemExit("--no-torch is for one rank")
        os.environ["APEMOST_NO_TORCH"] = "1"
        torch = dist = None
    else:
        import torch
        import torch.distributed as dist
    from apemost_amd import capi, workloads as wl
    from apemost_amd.sampler import HipSampler, get_chain_beta
    from apemost_amd.state import LadderState
    from apemost_amd.distributed import HipShardEngine, ShardedLadder

    if a.same_device:
        local_rank = 0
    if torch is None:
        try:
            dev_name, dev_cus, _ = capi.device_info(local_rank)
        except capi.ApemostHipError as e:
            raise SystemExit("bench.py needs an MI355X; there is no CPU path (%s)" % e)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X; there is no CPU path")
        torch.cuda.set_device(local_rank)
    use_dist = world > 1 or a.force_dist
    comm_device = "cuda" if a.backend == "nccl" else "cpu"   # where the bench's own small collectives live
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    n_local = a.chains_per_gpu
    n_global = n_local * world
    lo = rank * n_local
    w = wl.by_name(a.workload, n_data=a.n_data, n_chain=n_global)
    n_swap = a.n_swap or max(1, 2000 // (n_local if a.n_swap_rule == "per_gpu_ladder" else n_global))

    R = a.rounds_per_step
    LPS = a.launches_per_step

    # a calibrated-looking ladder: chebyshev betas, steps = steps0 * beta^-1/2
    st = LadderState.from_params(n_local, w.start, w.pmin, w.pmax, w.step * 0.3)
    for i in range(n_local):
        b = get_chain_beta(0, lo + i, n_global, 0.02)
        st.beta[i] = b
        st.step[i] = np.minimum(st.step[i] * b ** -0.5, w.pmax - w.pmin)

    s = HipSampler(w.model, w.n_par, n_local, w.data, seed=2024, device=local_rank, chain_offset=lo,
                   n_chains_global=n_global, waves_per_chain=a.waves, lds_policy=a.lds, flags=a.flags)
    s.set_state(st)
    calibrated = None
    calibration = None
    if not a.no_calibrate:
        # markov_chain_calibrate (burn-in + step-width calibration towards TARGET_ACCEPTANCE_RATE) for
        # every chain at its own beta, as calibrate_rest leaves a ladder: the timed steps then accept at
        # the rate of a production run instead of the ~0 of guessed step widths
        s.calc_model(0, n_local)
        s.synchronize()
        ccfg = capi.calib_defaults(burn_in_iterations=a.burn_in)
        if a.calib_iter_limit:
            ccfg.iter_limit = a.calib_iter_limit
        tc = time.perf_counter()
        status, iters = s.markov_chain_calibrate(0, n_local, ccfg)
        calib_wall = time.perf_counter() - tc
        calibrated = int((status == 0).sum())
        calibration = calibration_block(calib_wall, status, iters, ccfg, w.n_par)
        seg, ev, by_waves = s.calibrate_stats()
        calibration.update({"segments": seg, "evaluations_counted_on_device": ev,
                            "launches_by_waves_per_chain": {str(k): v for k, v in by_waves.items()},
                            "seconds_by_waves_per_chain": {str(k): round(v, 4) for k, v in s.calibrate_seconds_by_waves.items()}})
        if a.calib_dump and rank == 0:
            json.dump({"config": a.config, "burn_in": a.burn_in, "wall_s": calib_wall, "status": status.tolist(),
                       "sweeps": [int(x) for x in iters]}, open(a.calib_dump, "w"))
    acc0 = s.get_state()
    waves, lds = s.geometry
    samples = None
    if not a.no_samples and torch is not None:
        samples = torch.zeros((R, n_swap, n_local, w.n_par + 2), dtype=torch.float64, device="cuda")
    elif not a.no_samples:
        samples = EngineRows(s, R, n_swap)
    eng = HipShardEngine(s, torch)
    ladder = ShardedLadder(eng, n_global, lo, n_local, rank, world, dist if use_dist else None)

    ladder.prime()

    def one_step():
        # the swap attempt that closes a batch is applied at the start of the next one
        for _ in range(LPS):
            ladder.run_sampler(R, n_swap, samples, finalize=False)

    def sync_all():
        s.synchronize()
        if torch is not None:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_step()
    sync_all()
    capi.check(s.L.apemost_hip_timer_begin(s._h))
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    s.synchronize()
    import ctypes as C
    ev_ms, launches = C.c_float(0), C.c_uint64(0)
    capi.check(s.L.apemost_hip_timer_end(s._h, C.byref(ev_ms), C.byref(launches)))
    dt_rank = time.perf_counter() - t0   # this rank's own K steps, before it waits for the others
    sync_all()
    dt = time.perf_counter() - t0
    per_rank = None
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor([dt_rank, float(ladder.exchanges)], dtype=torch.float64, device=comm_device)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        per_rank = np.array([e.cpu().numpy() for e in every])

    acc1 = s.get_state()
    d_acc = (acc1.accept - acc0.accept).astype(np.float64)
    d_rej = (acc1.reject - acc0.reject).astype(np.float64)
    acceptance = float(np.mean(d_acc / np.maximum(d_acc + d_rej, 1)))
    total_steps = a.steps * LPS * R * n_swap * n_global
    value = total_steps / dt
    # roofline of the dominant kernel (pt_round_kernel): ALGORITHMIC bytes per launch / avg launch duration
    bytes_per_step = w.bytes_per_step()
    launch_ms = ev_ms.value / max(launches.value, 1)
    steps_per_launch = (a.steps * LPS * R * n_swap * n_local) / max(launches.value, 1)
    achieved_gbs = bytes_per_step * steps_per_launch / (launch_ms * 1e-3) / 1e9
    # fp64 operations per data point as the kernels issue them, an FMA counted as two:
    # a sine with its argument is 28 (3 + reduction 7 + polynomial 18); simplesin adds 5 around it,
    # sine3 2 per sine and 4 per point; the pulse models since round 3 (spectrum over a common
    # denominator, one reciprocal per two points, one logarithm per lane): two modes 10 additions /
    # products and 5.5 FMAs per point, pulse_vrot 13.5 and 9 (more than three modes: a division by
    # reciprocal per mode, 13 each, and 9 for the quotient)
    modes = max(1, (w.n_par - 2) // 2)
    flops_per_point = {"simplesin": 33.0, "sine3": 3 * 30.0 + 4, "pulse": 21.0 if modes <= 3 else 13.0 * modes + 9, "pulse_vrot": 31.5}
    flops_per_step = flops_per_point.get(w.name, 40.0) * w.n_data
    # What the PMC counters say cannot be read inside this process: `traffic` stays null in this line;
    # the HBM bytes and the VALU instruction count profiled with `rocprofv3 --pmc` on this same command
    # (profiles/README.md, profiles/pmc_traffic.json) are quoted when the workload is the one they
    # were measured on.
    # Counters are only quoted for the kernel they were measured on: same workload key, same waves per chain
    # AND the same kernel instantiation as this sampler launches; `profile_kernel_sources_changed` says whether
    # the kernel sources have changed since that profile was taken (then the instruction count is that of an
    # older build of the same kernel: re-profile with tools/profile_config.sh).
    one_barrier = waves in (4, 8) and not (a.flags & 4)
    helper = one_barrier and s.ob_helper
    kernel_name = expected_kernel(w.model, waves, lds, one_barrier, a.flags, helper)
    traffic_profiled = valu_insts = profile_src = profile_stale = None
    key = "%s/%d/%d/%d/%d/%s" % (w.name, n_local, w.n_data, n_swap, R, "nosamples" if a.no_samples else "samples")
    try:
        for pmc in json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))):
            if (pmc.get("workload_key") == key and pmc.get("waves_per_chain") == waves
                    and pmc.get("kernel", "").replace("void ", "").split("(")[0] == kernel_name):
                traffic_profiled = pmc.get("hbm_bytes_per_launch")
                valu_insts = pmc.get("valu_wave_insts_per_launch", valu_insts)
                profile_src = pmc.get("tag", profile_src)
                profile_stale = pmc.get("kernel_sources_sha1") != kernel_sources_sha1()
    except (OSError, ValueError, KeyError, TypeError):
        pass
    # The roofline that binds.  The data vector is resident in LDS or L2 by design (HBM traffic is
    # well under 1 % of the algorithmic bytes), so no kernel of this engine is bound by HBM: what a
    # step costs is instructions -- one wave64 VALU instruction occupies its SIMD for four cycles,
    # integer or fp64 -- and, on ladders smaller than the chip, the CUs that have no chain.  `frac` is
    # against the whole chip; `frac_on_occupied_simds` says how busy the SIMDs that hold a chain are.
    if torch is not None:
        props = torch.cuda.get_device_properties(local_rank)
        clock_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
        cus = props.multi_processor_count
    else:
        clock_hz, cus = 2400e6, dev_cus     # (MI355X_MICROARCH.md: 2.4 GHz peak engine clock)
    waves_per_wg = waves + 4 + (1 if helper else 0) if one_barrier else waves
    cus_occupied = min(cus, n_local)
    simds_occupied = min(4 * cus, n_local * min(4, waves_per_wg))
    launch_s = launch_ms * 1e-3
    fp64_frac = flops_per_step * steps_per_launch / launch_s / 1e12 / FP64_VALU_PEAK_TF
    roof = {"traffic": None, "traffic_profiled": traffic_profiled,
            "kernel": kernel_name, "profile_kernel_sources_changed": profile_stale, "launch_us": launch_ms * 1e3,
            "cus_occupied": cus_occupied, "cus": cus, "simds_occupied": simds_occupied, "clock_mhz_peak": clock_hz / 1e6,
            # SURVEY 8(d)'s nominal figure: every step "streams" the data vector once (it does, from LDS / L2)
            "hbm_nominal_achieved": achieved_gbs, "hbm_nominal_peak": HBM_PEAK_GBS, "hbm_nominal_unit": "GB/s",
            "hbm_nominal_frac": achieved_gbs / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": bytes_per_step * steps_per_launch,
            "fp64_valu_frac": fp64_frac}
    if valu_insts:
        issue_peak = 4 * cus * clock_hz / 4.0
        issue = valu_insts / launch_s
        roof.update({"bound": "valu_issue", "achieved": issue / 1e9, "peak": issue_peak / 1e9,
                     "unit": "G wave-instructions/s", "frac": issue / issue_peak,
                     "frac_on_occupied_simds": issue / (simds_occupied * clock_hz / 4.0),
                     "valu_wave_insts_per_step_per_occupied_simd": valu_insts / steps_per_launch * n_local / simds_occupied,
                     "issue_source": "SQ_INSTS_VALU of this kernel under rocprofv3 --pmc, same command (profiles/%s)" % profile_src})
    else:
        roof.update({"bound": "fp64_valu", "achieved": fp64_frac * FP64_VALU_PEAK_TF, "peak": FP64_VALU_PEAK_TF,
                     "unit": "TFLOP/s", "frac": fp64_frac,
                     "issue_source": "no profiled instruction count for this workload: fp64 flops of the likelihood as issued (an FMA = 2)"})
    parallel = {"ranks": world, "backend": dist.get_backend() if use_dist else None,
                "n_swap_rule": "explicit" if a.n_swap else a.n_swap_rule}
    if per_rank is not None:
        rates = a.steps * LPS * R * n_swap * n_local / per_rank[:, 0]
        parallel.update({"rank_steps_per_s_min": float(rates.min()), "rank_steps_per_s_max": float(rates.max()),
                         "edge_exchanges_all_ranks": int(per_rank[:, 1].sum())})
    out = {
        "metric": "MCMC steps/sec (all chains) on %s, 1/2/4/8 MI355X + HBM-roofline %%" % w.name,
        "baseline_config": a.config,
        "value": value, "unit": "Metropolis steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d beta-chains/GPU x %d GPU, %d data points, n_par=%d, n_swap=%d, "
                               "%d launches x %d rounds per bench step, sample rows %s" %
                               (w.name, n_local, world, w.n_data, w.n_par, n_swap, LPS, R,
                                "off" if a.no_samples else "on"),
                   "chains_per_gpu": n_local, "n_data": w.n_data, "n_swap": n_swap, "rounds_per_step": R,
                   "launches_per_step": LPS,
                   "waves_per_chain": waves, "data_in_lds": lds, "parallelism": "ladder-sharded x%d" % world,
                   "edge_exchanges_rank0": ladder.exchanges, "distributed": parallel,
                   "device_calibrated_chains_rank0": calibrated, "acceptance_rate_rank0": acceptance},
        "roofline": roof,
    }
    if calibration is not None:
        # the calibration's likelihood evaluations per second against the round kernel's on the same ladder
        calibration["rate_vs_round_kernel"] = calibration["evaluations_per_s"] / (value / world)
        # its own issue roofline: SQ_INSTS_VALU of every calibration launch of the profiled run of this command
        # over their summed durations (profiles/*_calib_counters.json, registered by tools/summarize_profile.py)
        ckey = "%s/%d/%d/calibration" % (w.name, n_local, w.n_data)
        try:
            for pmc in json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))):
                if pmc.get("workload_key") == ckey:
                    calibration.update({"valu_issue_frac_profiled": pmc.get("valu_issue_frac"),
                                        "valu_wave_insts_profiled": pmc.get("valu_wave_insts"),
                                        "profile": pmc.get("tag"),
                                        "profile_kernel_sources_changed": pmc.get("kernel_sources_sha1") != kernel_sources_sha1()})
        except (OSError, ValueError, KeyError, TypeError):
            pass
        out["calibration"] = calibration
    if rank == 0:
        if world == 1 and a.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(w, acc0, n_swap, a.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if a.maps_dump:
        open(a.maps_dump, "w").write(open("/proc/self/maps").read())
    s.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
