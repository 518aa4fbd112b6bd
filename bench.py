#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

A "step" is one pass of the hot path over one device-resident synthetic batch: the fp64 MVN
log-density of N = 1e6 particles of dimension d = 64 per GPU (BASELINE.json `metric`:
"MVN log-pdf evals/sec (1e6 particles, d=64) + MH steps/sec; 1/2/4/8 GPUs").  Particles shard
embarrassingly: one process per GPU, every rank owns its own 1e6 particles, no collective in the
data path (weak scaling).  The Metropolis-resampler rate (BASELINE configs[1]:
1e5 chains, weights = d=32 MVN densities, 1e3 iterations) is measured after the timed region and
reported as `mh_steps_per_s`.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms run N ranks.  Started WITHOUT a launcher (`WORLD_SIZE` unset) and with --gpus N > 1, this
process never touches the GPU: it starts `torch.distributed.run` with N fresh child ranks and exits
with their code.  Under a launcher, `--gpus` must equal WORLD_SIZE (exit code 2 otherwise), so a
mismatch can no longer measure one GPU and call it N.

Prints ONE JSON line on rank 0 (contract: the task statement; roofline terms: DESIGN.md).  Beyond the
contract's keys the line carries `strong` (fixed-total-work legs: the headline's 1e6 x 64 and
BASELINE configs[4]'s 4e6 x 256 split over the ranks), `mh` (configs[1], weight all-gather timed) and
`filter_step` (one sharded bootstrap-filter step with its two exchanges timed) and `proposal` (propagate_K's draws
for the headline shape, dense G, Q full or a Cholesky factor, Normal and Student-t).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_PER_GPU = 1_000_000
D = 64
ALGO_BYTES_PER_EVAL = 8 * D + 8  # read the particle once, write one log-density
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MH_N, MH_B, MH_D = 100_000, 1000, 32
EXPECTED_KERNEL = "cusmc_logpdf_nb4_asm"  # kernels/logpdf_nb4_gfx950.s (CUSMC_NB4_ASM=0: cusmc::logpdf_mfma_kernel<4, true, false, 0, 1, false>)
C5_N, C5_D = 4_000_000, 256      # BASELINE configs[4]: d=256 MVN, 4e6 particles over the node
F64_MFMA_PEAK_TFLOPS = 78.6      # public spec; measured 77.7 (profiles/r01_calibration.txt)
PF_N = 1_000_000                 # BASELINE configs[2]: particle filter, 1e6 particles
AUX_TIMEOUT_S = 300              # rank 0 prints the headline line by itself if the legs after it hang this long


def make_sigma(d, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d))
    return A @ A.T / d + np.eye(d)


def pmc_child(n, d, launches):
    """Runs under rocprofv3 --pmc: a few launches of the headline kernel, nothing else."""
    import numpy as np
    import torch
    import cusmc_amd
    torch.cuda.set_device(0)
    X = torch.randn(n, d, dtype=torch.float64, device="cuda")
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    D_ = cusmc_amd.MultiVariateNormalDistribution(np.zeros(d), make_sigma(d, 1))
    D_.ctx.use_torch_stream()
    for _ in range(launches):
        D_.pdf_dev(X, out)
    torch.cuda.synchronize()
    D_.close()


def collect_hbm_traffic(n, d, kernel_substr=("logpdf_nb4", "logpdf_mfma")):
    """HBM bytes per launch of the dominant kernel from the PMC counters, collected as
    MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
    (4 TCC slots; 3 + 2 do not fit), counter unit KiB, and the gfx950 correction: FETCH_SIZE
    reports exactly half the bytes of a wide coalesced streaming read (16 B/lane), so it is
    doubled; WRITE_SIZE is exact.  Runs rocprofv3 on a child copy of this script, BEFORE this
    process touches the GPU.  Returns None if rocprofv3 is unavailable."""
    import csv
    import glob
    rocprof = "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    total, names = {}, set()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
                env = dict(os.environ, TMPDIR="/tmp", CUSMC_PMC_CHILD="%d,%d,%d" % (n, d, 3))
                cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp,
                       "--", sys.executable, os.path.abspath(__file__)]
                subprocess.run(cmd, cwd="/tmp", env=env, check=True, capture_output=True, timeout=300)
                vals = []
                for path in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                    with open(path) as f:
                        for row in csv.DictReader(f):
                            if any(k in row.get("Kernel_Name", "") for k in kernel_substr) and row.get("Counter_Name") == counter:
                                vals.append(float(row["Counter_Value"]))
                                names.add(row["Kernel_Name"])
                if not vals:
                    return None
                total[counter] = sum(vals) / len(vals) * 1024.0
        return {"fetch_bytes": 2.0 * total["FETCH_SIZE"], "write_bytes": total["WRITE_SIZE"],
                "hbm_bytes": 2.0 * total["FETCH_SIZE"] + total["WRITE_SIZE"],
                "kernel": " | ".join(sorted(n.split("(")[0] for n in names))}
    except Exception as e:  # profiler missing / refused: report null, never a guess
        sys.stderr.write("bench: PMC traffic pass unavailable (%s)\n" % e)
        return None


def cpu_baseline(d, budget_s=12.0):
    """The reference's CPU cost structure on this host's cores: per particle a fresh
    distribution, an LU determinant and an LU inverse of Sigma, then the dense quadratic form
    (src/statistics.cc.cpp:171-196 as called from src/mcmc.cpp:193-215) -- the oracle's
    reference-faithful mode, OpenMP over particles.  Bounded sample, sized from a calibration."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(2)
    sigma, mu = make_sigma(d, 1), np.zeros(d)
    F = np.eye(d)
    cal = rng.standard_normal((512, d))
    O.pdf_batch(cal[:64], mu, sigma, F)  # warm (thread pool, page faults)
    t0 = time.perf_counter()
    O.pdf_batch(cal, mu, sigma, F)
    rate = 512 / (time.perf_counter() - t0)
    n = int(min(N_PER_GPU, max(2048, rate * budget_s)))
    X = rng.standard_normal((n, d))
    t0 = time.perf_counter()
    O.pdf_batch(X, mu, sigma, F)
    dt = time.perf_counter() - t0
    # second line: the same cores with Sigma factored ONCE (algorithmic gain vs hardware gain)
    Xh = rng.standard_normal((min(N_PER_GPU, 400_000), d))
    O.logpdf_hoisted(Xh[:1000], mu, sigma)
    t1 = time.perf_counter()
    O.logpdf_hoisted(Xh, mu, sigma)
    dth = time.perf_counter() - t1
    return {"value": n / dt, "unit": "evals/s", "cores": O.num_threads(), "kind": "port",
            "sample": "%d of the %d particles (d=%d), reference-faithful per-particle LU det+inverse, "
                      "%.1f s" % (n, N_PER_GPU, d, dt),
            "hoisted_factor_value": Xh.shape[0] / dth}


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start N ranks through
    torch.distributed.run as a CHILD process (never an exec: this process may not have touched the
    GPU, but the rule on this pool is child processes only) and pass its exit code on."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.no_pmc:
        cmd.append("--no-pmc")
    if args.no_cpu:
        cmd.append("--no-cpu")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def main():
    if os.environ.get("CUSMC_PMC_CHILD"):
        n, d, launches = (int(v) for v in os.environ["CUSMC_PMC_CHILD"].split(","))
        pmc_child(n, d, launches)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500,
                    help="the clock governor needs ~50 ms of load to settle (profiles/r01_calibration.txt)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC traffic passes")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.stderr.write("bench: --gpus must be >= 1\n")
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)  # the parent never touches the GPU; the ranks are fresh child processes
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench: --gpus %d but WORLD_SIZE is %d: refusing to report one as the other\n"
                         % (args.gpus, world))
        return 2

    traffic = None
    if world == 1 and not args.no_pmc:
        traffic = collect_hbm_traffic(N_PER_GPU, D)  # child processes; this one has not touched the GPU yet

    import numpy as np
    import torch
    import torch.distributed as dist

    import cusmc_amd

    # Rehearsal switch for a one-GPU box: CUSMC_BENCH_REHEARSAL=1 puts every rank on device 0 and uses
    # gloo, to exercise the world > 1 code path (barriers, MAX reduction, the sharded resampler's
    # all-gather).  Its numbers mean nothing; the driver's multi-GPU runs never set it.
    rehearsal = os.environ.get("CUSMC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # CUSMC_BENCH_FORCE_DIST=1 under a one-rank launcher: the RCCL group is created and every barrier / reduction
    # below goes through it -- the closest a one-GPU box gets to the multi-GPU control path (tests/test_bench_cli.py)
    dist_on = world > 1 or (os.environ.get("CUSMC_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # synthetic batch, resident in HBM before the timed region; every rank owns its own particles
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    X = torch.randn(N_PER_GPU, D, dtype=torch.float64, device="cuda", generator=g)
    out = torch.empty(N_PER_GPU, dtype=torch.float64, device="cuda")
    sigma, mu = make_sigma(D, 1), np.zeros(D)
    mvn = cusmc_amd.MultiVariateNormalDistribution(mu, sigma)
    mvn.ctx.use_torch_stream()  # launches go on torch's current stream, so torch events see them

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # The clock governor needs ~50 ms of this load to settle (the first ~20 ms of a burst run 15 % slow:
    # profiles/r01_calibration.txt).  With the default W that is the warm-up itself; when a caller asks
    # for a shorter one, the difference is made up here, untimed and reported as "clock_settle_launches".
    settle = max(0, 600 - args.warmup)
    for _ in range(settle):
        mvn.pdf_dev(X, out)
    for _ in range(args.warmup):
        mvn.pdf_dev(X, out)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        mvn.pdf_dev(X, out)
    ev1.record()
    while not ev1.query():  # busy-poll: a blocking synchronize wakes tens of microseconds late, which K = 20 steps of
        pass                # ~90 us would carry as 3 %; the barrier + synchronize below then return at once
    barrier()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream

    t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if dist_on:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())

    # a 2048-row sample of the timed launch's inputs and outputs, for the CPU leg's check below
    idx = torch.arange(0, N_PER_GPU, N_PER_GPU // 2048, device="cuda")[:2048]
    sample_in, sample_out = X[idx].cpu().numpy(), out[idx].cpu().numpy()

    # Metropolis resampler rate, BASELINE configs[1] shape (outside the timed region).  Weak scaling
    # like the headline: every rank owns MH_N chains of a world x MH_N vector; the one real exchange
    # step of this path -- the all-gather of the weight shards (cusmc_amd/sharding.py) -- is inside
    # the timed loop.  A failure here must not cost the headline line: it is reported in "mh".
    # ---- the line: everything the contract asks for is known here; the legs below only add to it --------------
    line = None
    if rank == 0:
        evals = world * N_PER_GPU * args.steps
        achieved = N_PER_GPU * ALGO_BYTES_PER_EVAL / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "MVN log-pdf evals/sec (1e6 particles, d=64)",
            "value": evals / wall_max,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "clock_settle_launches": settle,
            "ms_per_step": wall_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one device, gloo)",
            "config": {"workload": "mvn_logpdf fp64: N=%d particles per GPU x d=%d, Sigma = AA^T/d + I "
                                   "(seed 1), device-resident, one launch per step" % (N_PER_GPU, D),
                       "particles_per_gpu": N_PER_GPU, "d": D, "parallelism": "particle-sharded x%d, "
                       "no data-path collective" % world,
                       # the clock governor's transient: --warmup below 600 is topped up to 600 untimed launches
                       "untimed_launches_before_the_timed_region": settle + args.warmup,
                       "kernel_source": "hand-written assembly (kernels/logpdf_nb4_gfx950.s)"
                       if os.environ.get("CUSMC_NB4_ASM", "1") != "0" else "compiled HIP (CUSMC_NB4_ASM=0)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if traffic is None else traffic["hbm_bytes"],
                         "algorithmic_bytes_per_launch": N_PER_GPU * ALGO_BYTES_PER_EVAL,
                         # the name the PMC pass matched; without that pass, the instantiation this
                         # workload dispatches to (logpdf_mfma.hip: NB = 4, centred, no shift, MVN epilogue)
                         "kernel": traffic["kernel"] if traffic else EXPECTED_KERNEL,
                         "kernel_name_source": "rocprofv3 counter_collection.csv" if traffic else "expected (no PMC pass)",
                         "kernel_ms": kernel_ms,
                         "frac_of_measured_copy_peak": achieved / 6290.0},
            "cpu_baseline": None,
            "mh_steps_per_s": None,
            "mh": None,
            "strong": None,
            "filter_step": None,
            "proposal": None,
            "ranks": int(dist.get_world_size()) if dist_on else world,
            "backend": dist.get_backend() if dist_on else "single process",
            "parity_max_rel_err_vs_oracle": None,
        }
        if traffic is not None:
            line["roofline"]["traffic_detail"] = {k: v for k, v in traffic.items() if k != "kernel"}
    emitted = []

    def emit():
        if rank == 0 and not emitted:
            emitted.append(True)
            print(json.dumps(line), flush=True)

    # The legs below run collectives on N > 1 ranks.  Whatever happens in them must not cost the headline
    # line: on one rank an exception is recorded in the line; on several, a rank that threw can no longer keep
    # step with its peers' collectives, so it reports (rank 0: prints the line with the error) and leaves the
    # job at once, and a watchdog on rank 0 prints the line if its peers never come back.
    watchdog = None
    if world > 1 and rank == 0:
        import threading

        def _late():
            line["aux_error"] = "auxiliary legs did not finish within %d s: headline reported without them" % AUX_TIMEOUT_S
            emit()
            os._exit(3)  # (non-zero: the launcher tears the peers down and the driver sees the failure)
        watchdog = threading.Timer(AUX_TIMEOUT_S, _late)
        watchdog.daemon = True
        watchdog.start()

    def aux_failed(name, exc):
        import traceback
        sys.stderr.write("bench: leg '%s' failed on rank %d:\n%s\n" % (name, rank, traceback.format_exc()))
        if world > 1:
            if rank == 0:
                line[name] = {"error": repr(exc)}
                line["aux_error"] = "leg '%s' failed; later legs not run" % name
                emit()
            sys.stderr.flush()
            os._exit(3)  # (the headline line is out; the exit code says a leg failed and takes the peers down with it)
        return {"error": repr(exc)}

    def timed_max(fn, reps, warm=3):
        """seconds per repetition of fn(), barrier + synchronize on both sides, MAX over ranks"""
        for _ in range(warm):
            fn()
        barrier()
        t_ = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        tt = torch.tensor([(time.perf_counter() - t_) / reps], dtype=torch.float64, device="cuda")
        if dist_on:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    mh = None
    try:
        from cusmc_amd import sharding
        sig32 = make_sigma(MH_D, 2)
        d32 = cusmc_amd.MultiVariateNormalDistribution(np.zeros(MH_D), sig32)
        Xw = torch.randn(MH_N, MH_D, dtype=torch.float64, device="cuda", generator=g) @ \
            torch.from_numpy(np.linalg.cholesky(sig32).T).cuda()
        w = torch.empty(MH_N, dtype=torch.float64, device="cuda")
        d32.pdf_dev(Xw.contiguous(), w, log=False)
        a = torch.empty(MH_N, dtype=torch.int32, device="cuda")
        ctx = mvn.ctx
        n_total = world * MH_N
        step = [0]

        def resample_once():
            step[0] += 1

            def fn(w_full, first, count):
                return cusmc_amd.Sampler.metropolis_hastings_dev(w_full, a, B=MH_B, t=step[0], seed=1, first=first, ctx=ctx)
            if world > 1:
                sharding.sharded_resample(w, n_total, fn)
            else:
                fn(w, 0, MH_N)
        resample_once()
        barrier()
        reps = 5
        t1 = time.perf_counter()
        for _ in range(reps):
            resample_once()
        barrier()
        tm = torch.tensor([(time.perf_counter() - t1) / reps], dtype=torch.float64, device="cuda")
        if dist_on:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        mh_s = float(tm.item())
        sps = n_total * MH_B / mh_s
        # Yardstick (counters in profiles/r03_pmc_mh.md): the chain is bound by the rate of its random
        # gathers -- one L2 request per lane and step (TCC_REQ / step = 0.96), 2.65e11 / s with every SIMD loaded --
        # not by HBM and not by VALU issue: 50 VALU instructions per wave-step (SQ_INSTS_VALU), ~270 cycles at the
        # measured issue costs, against the cycles a SIMD spends per wave-step at this rate.
        cyc_per_wave_step = 2.3e9 * 65536.0 / (sps / world)
        # (up to 4e5 weights the head of the truncated table -- 40 000 words, 160 KB -- sits in LDS: those gathers
        # are no L2 requests; resample.hip: metropolis_hi_lds_kernel)
        lds_frac = min(n_total, 40000) / n_total if (n_total <= 400000 and 90000 <= MH_N <= 262144) else 0.0
        l2_req = (1.0 - lds_frac) * sps / world
        mh = {"steps_per_s": sps, "ms_per_resample": mh_s * 1e3,
              "workload": "metropolis_hastings %d x N=%d chains, B=%d iters, weights = d=%d MVN densities%s"
                          % (world, MH_N, MH_B, MH_D, "" if world == 1 else "; all-gather of the weight shards timed"),
              "bound": "L2 gather requests (one per lane-step not served from the LDS head of the table)",
              "gathers_from_lds_frac": lds_frac,
              "l2_gather_requests_per_s_per_gpu": l2_req, "l2_gather_saturated_requests_per_s": 2.65e11,
              "l2_gather_frac_of_saturated": l2_req / 2.65e11,
              "gather_useful_GBps_per_gpu": 4.0 * sps / world / 1e9, "l2_gather_sector_GBps_per_gpu": 64.0 * l2_req / 1e9,
              "valu_instr_per_wave_step": 50, "valu_issue_cycles_per_wave_step": 270,
              "valu_issue_frac": 270.0 / cyc_per_wave_step}
        d32.close()
    except Exception as exc:  # noqa: BLE001
        mh = aux_failed("mh", exc)

    # Strong scaling (fixed TOTAL work split over the ranks, no data-path collective): the headline's
    # 1e6 x 64 batch, and BASELINE configs[4]'s 4e6 x 256 (MFMA-bound: reported against the f64 matrix peak).
    strong = None
    try:
        from cusmc_amd import sharding
        _, cnt = sharding.shard_range(N_PER_GPU, rank, world)
        Xs, outs = X[:cnt], out[:cnt]
        # (warm-up: the clock governor's transient, as for the headline -- 200 launches straight after an idle
        # period run ~10 % slow, profiles/r01_calibration.txt)
        s_head = timed_max(lambda: mvn.pdf_dev(Xs, outs), max(args.steps, 200), warm=600)
        strong = {"headline_1e6x64": {"total_particles": N_PER_GPU, "particles_per_gpu": cnt, "evals_per_s": N_PER_GPU / s_head,
                                      "us_per_launch": s_head * 1e6,
                                      "hbm_frac_per_gpu": cnt * ALGO_BYTES_PER_EVAL / s_head / 1e9 / HBM_PEAK_GBS}}
        _, cnt5 = sharding.shard_range(C5_N, rank, world)
        X5 = torch.randn(cnt5, C5_D, dtype=torch.float64, device="cuda", generator=g)
        out5 = torch.empty(cnt5, dtype=torch.float64, device="cuda")
        d256 = cusmc_amd.MultiVariateNormalDistribution(np.zeros(C5_D), make_sigma(C5_D, 5))
        s5 = timed_max(lambda: d256.pdf_dev(X5, out5), 10)
        nb = C5_D // 16
        flop5 = 2.0 * nb * (nb + 1) * 2048 / 16  # per particle: 2 NB (NB+1) MFMAs of 2048 flop per 16 particles
        strong["c5_4e6x256"] = {"total_particles": C5_N, "particles_per_gpu": cnt5, "evals_per_s": C5_N / s5,
                                "ms_per_pass": s5 * 1e3, "bound": "mfma",
                                "tflops_per_gpu": cnt5 * flop5 / s5 / 1e12,
                                "mfma_frac_per_gpu": cnt5 * flop5 / s5 / 1e12 / F64_MFMA_PEAK_TFLOPS}
        d256.close()
        del X5, out5
    except Exception as exc:  # noqa: BLE001
        strong = aux_failed("strong", exc)

    # One bootstrap-filter time step with the particles sharded (BASELINE configs[2]: 1e6 particles in all):
    # all-gather of w_{t-1}, resample, all-to-all of the ancestor rows, propagate, reweight -- both
    # exchanges inside the timed region; the all-gather-of-x form of round 1 beside it.
    filt = None
    try:
        from cusmc_amd import sharding
        filt = {}
        for dd in (2, 64):
            I = np.eye(dd)
            Tf = 6
            Yf = np.cumsum(0.05 * np.random.default_rng(3).standard_normal((dd, Tf)), axis=1)
            model = (Yf, np.zeros(dd), I, I, I, 0.001 * I, 0.001 * I)
            ini, res, mov, obs = sharding.gpu_filter_callables_exchange(*model, seed=11, ctx=mvn.ctx)
            ini2, stp, obs2 = sharding.gpu_filter_callables(*model, seed=11, ctx=mvn.ctx)
            st = {}
            sharding.run_filter_sharded(PF_N, 2, ini, resample_fn=res, move_fn=mov)  # warm
            s_x = timed_max(lambda: sharding.run_filter_sharded(PF_N, Tf, ini, resample_fn=res, move_fn=mov, stats=st), 1, warm=0)
            s_g = timed_max(lambda: sharding.run_filter_sharded(PF_N, Tf, ini2, stp), 1, warm=1)
            filt["d%d" % dd] = {"N_total": PF_N, "steps": Tf - 1,
                                "ms_per_step_row_exchange": s_x / (Tf - 1) * 1e3,
                                "ms_per_step_allgather_x": s_g / (Tf - 1) * 1e3,
                                "bytes_in_per_step_row_exchange": {k: v / (Tf - 1) for k, v in st.items()},
                                "bytes_in_per_step_allgather_x": (PF_N - sharding.shard_range(PF_N, rank, world)[1]) * 8 * (dd + 1)
                                if world > 1 else 0}
            obs.close()
            obs2.close()
    except Exception as exc:  # noqa: BLE001
        filt = aux_failed("filter_step", exc)

    # The proposal draws north_star names ("the MVN proposal RNG"): propagate_K's x_t = Q xi + G x_{t-1}[a] for the
    # headline shape, every rank its own 1e6 x 64 particles (weak, no collective), dense G; Q once as eigenSolver's full
    # square root and once as a Cholesky factor (multiplied as a triangle), Normal and Student-t (nu = 4).
    prop = None
    try:
        rngp = np.random.default_rng(64)
        Gp = 0.9 * np.eye(D) + 0.1 * rngp.standard_normal((D, D)) / np.sqrt(D)
        Sp = make_sigma(D, 6)
        Qe, Qc = cusmc_amd.eigenSolver(Sp), np.linalg.cholesky(Sp)
        anc = torch.randint(0, N_PER_GPU, (N_PER_GPU,), dtype=torch.int32, device="cuda", generator=g)
        Xo = torch.empty_like(X)
        stp = [0]
        prop = {"particles_per_gpu": N_PER_GPU, "d": D}
        for key, Qm, kind, nu in (("mvn_full_q", Qe, "mvn", 0.0), ("mvn_cholesky_q", Qc, "mvn", 0.0),
                                  ("mvt4_full_q", Qe, "mvt", 4.0), ("mvt4_cholesky_q", Qc, "mvt", 4.0)):
            def draw(Qm=Qm, kind=kind, nu=nu):
                stp[0] += 1
                cusmc_amd.api.propagate_dev(X, anc, Gp, Qm, Xo, kind, nu, 1.0, seed=5, step=stp[0], ctx=mvn.ctx)
            s_p = timed_max(draw, 40, warm=20)
            prop[key] = {"us_per_launch": s_p * 1e6, "particles_per_s": world * N_PER_GPU / s_p}
        del anc, Xo
    except Exception as exc:  # noqa: BLE001
        prop = aux_failed("proposal", exc)

    # The CPU leg is the only place bench.py touches oracle/: it times the reference-faithful port
    # and, while it has it loaded, checks the sample of the GPU's outputs against it.
    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(D)
        from oracle import oracle as O
        want = O.logpdf_hoisted(sample_in, mu, sigma)
        parity = float(np.max(np.abs(sample_out - want) / np.abs(want)))

    if rank == 0:
        line.update({"cpu_baseline": cpu, "mh_steps_per_s": None if not mh else mh.get("steps_per_s"), "mh": mh,
                     "strong": strong, "filter_step": filt, "proposal": prop, "parity_max_rel_err_vs_oracle": parity})
        emit()
    if dist_on:
        dist.barrier()
    if watchdog is not None:
        watchdog.cancel()
    mvn.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
