#!/usr/bin/env python3
"""Headline benchmark: LU GFLOP/s of the MPF hot path at N=32768, nb=256 on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 32768] [--nb 256]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one complete MPF factorization (fp16 pre-pivoting panel, row interchanges, fp64 no-pivot
panel, TRSM, f64-MFMA GEMM) of a matrix already resident in HBM.  value = (2/3 N^3) / t / 1e9.
Rank 0 prints ONE JSON line (see the repo's DESIGN.md, section Measurement, for every field).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense fp64 matrix peak (AMD spec)


def kernel_source_sha(name):
    """sha256[:16] of a kernel source file: PMC summaries under profiles/ are only quoted for the source they measured."""
    import hashlib
    with open(os.path.join(ROOT, "mixed-precision_lu_factorization_amd", "csrc", name), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


_THREAD_ENV = ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS")


def cpu_baseline(n_cpu):
    """The reference's CPU path (benchmark.cpp:239-242): LAPACK dgetrf on the host cores.  When the environment pins the BLAS
    thread count (torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks) the leg runs in a child process with those
    variables removed: LAPACK must see the box's cores, and raising OpenBLAS's thread count after start-up is not safe."""
    if any(k in os.environ for k in _THREAD_ENV):
        import subprocess
        env = {k: v for k, v in os.environ.items() if k not in _THREAD_ENV}
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-only", "--cpu-n", str(n_cpu)], env=env,
                                 capture_output=True, text=True, timeout=1200)
            return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        except Exception as e:  # pragma: no cover
            return {"value": None, "unit": "GFLOP/s", "cores": None, "kind": "reference", "sample": f"CPU leg failed: {e}"}
    return _cpu_baseline_here(n_cpu)


def _cpu_baseline_here(n_cpu):
    """LAPACK dgetrf on the host cores of this process, on the reference generator's matrix, bounded sample size.
    Returns the cpu_baseline object of the JSON line."""
    import numpy as np
    try:
        import scipy.linalg as sl
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "GFLOP/s", "cores": os.cpu_count(), "kind": "reference", "sample": f"scipy missing: {e}"}
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    try:  # the box's CPU share (cgroup quota) is what the host cores really are for this process
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) // int(per))))
    except Exception:
        pass
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=cores)
    except Exception:
        pass
    rng = np.random.default_rng(0)
    warm = np.asfortranarray(rng.integers(0, 100, (512, 512)) / 10.0)
    sl.lu_factor(warm, overwrite_a=True, check_finite=False)
    try:   # the reference generator's own stream: at n_cpu == N the very matrix the GPU factored (`matgen f N (N-2) lin`)
        from oracle import oracle as O
        A = O.matgen_skip(n_cpu)
        what = f"the N={n_cpu} matrix of the reference generator's stream (`matgen f N (N-2) lin`: the GPU run's input)"
    except Exception:
        A = np.asfortranarray(rng.integers(0, 100, (n_cpu, n_cpu)) / 10.0)
        what = f"one N={n_cpu} generator-distributed matrix (numpy stream: oracle unavailable)"
    t = time.perf_counter()
    sl.lu_factor(A, overwrite_a=True, check_finite=False)
    dt = time.perf_counter() - t
    del A
    out = {"value": round(2.0 / 3.0 * n_cpu ** 3 / dt / 1e9, 1), "unit": "GFLOP/s", "cores": cores, "kind": "reference",
           "sample": f"LAPACK dgetrf (scipy / OpenBLAS), the routine benchmark.cpp:240 calls, on {what}, {dt:.2f} s"}
    # the repo's own CPU restatement of MPF (oracle, 'port'), smaller sample: for context only
    try:
        from oracle import oracle as O
        n_o = 2048
        Ao = O.matgen_skip(n_o)
        t = time.perf_counter()
        O.mpf(Ao, 256)
        dto = time.perf_counter() - t
        out["port_value"] = round(2.0 / 3.0 * n_o ** 3 / dto / 1e9, 1)
        out["port_sample"] = f"oracle/mpf_oracle.c MPF restatement, N={n_o} nb=256, {dto:.2f} s"
    except Exception as e:  # pragma: no cover
        out["port_value"] = None
        out["port_sample"] = f"oracle unavailable: {e}"
    return out


def self_launch(args):
    """`python bench.py --gpus N ...` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W ...` as a child process, relay its output
    (rank 0's one JSON line) and return its exit code.  Sizes travel through the environment (torch.distributed.run's own
    parser trips over `--n`)."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:   # a free port on the loopback interface
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    for flag in ("no_cpu", "no_ir", "no_mxp", "no_phases", "no_config5", "no_ref_style", "no_configs", "no_check"):
        if getattr(args, flag):
            cmd.append("--" + flag.replace("_", "-"))
    env = dict(os.environ)
    env["MPF_BENCH_N"], env["MPF_BENCH_NB"], env["MPF_BENCH_CPU_N"] = str(args.n), str(args.nb), str(args.cpu_n)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in proc.stdout:          # relay as it comes (progress lines, then rank 0's JSON line)
        sys.stdout.write(ln)
        sys.stdout.flush()
    return proc.wait()


def _finite(o):
    """JSON has no inf/nan: replace non-finite floats (a diverged refinement history) by strings."""
    import math
    if isinstance(o, float) and not math.isfinite(o):
        return "inf" if o > 0 else ("-inf" if o < 0 else "nan")
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    # (torch.distributed.run's own parser trips over "--n ..." even after the script name: tests set the size through the environment)
    ap.add_argument("--n", type=int, default=int(os.environ.get("MPF_BENCH_N", "32768")))
    ap.add_argument("--nb", type=int, default=int(os.environ.get("MPF_BENCH_NB", "256")))
    ap.add_argument("--cpu-n", type=int, default=int(os.environ.get("MPF_BENCH_CPU_N", "32768")), help="size of the CPU-baseline sample (32768 = the GPU workload; ~15 s on 16 cores)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ir", action="store_true")
    ap.add_argument("--no-mxp", action="store_true")
    ap.add_argument("--no-phases", action="store_true", help="skip the per-phase (host-synchronised) repetition: profile runs then contain only look-ahead launches")
    ap.add_argument("--no-config5", action="store_true")
    ap.add_argument("--no-ref-style", action="store_true", help="skip the host-buffer run timed the way benchmark.cpp times MPF()")
    # SURVEY 5 bench CLI (the reference's own flags are -v / --no-check, benchmark.cpp:153-158): single legs on chosen inputs
    ap.add_argument("--trailing", choices=("fp64", "fp16", "fp16x3"), default="fp64",
                    help="arithmetic of the trailing update in the TIMED steps (fp64 = the reference's, the headline configuration)")
    ap.add_argument("--gen", choices=("ref", "diagdom", "kappa"), default="ref",
                    help="input of the timed steps: ref = the reference generator's stream, diagdom = + diag(rowsum), kappa = diagdom with rows scaled by logspace(0, 8)")
    ap.add_argument("--seed", type=int, default=0, help="extra rand() draws skipped before the generator's stream (0 = the reference's own matrix)")
    ap.add_argument("--check", action="store_true", help="(default since round 5) run the reference's acceptance test max|A - P L U| <= 1e-10 (benchmark.cpp:97-144) on the last timed factorization")
    ap.add_argument("--no-check", action="store_true", help="skip it (the reference's own flag, benchmark.cpp:153-158)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object: BASELINE configs C1 / C2 / C4-size on this one GPU (C3 = the timed steps, C5 = config5)")
    ap.add_argument("--ir-steps", type=int, default=10, help="at most that many refinement sweeps in the `ir` leg")
    ap.add_argument("--legs", default="", help="comma-separated POSITIVE list of the extra legs to run (ir, phases, mxp, config5, ref_style, cpu, configs, check); "
                                               "default: all that no --no-* flag removes.  `--legs none` runs the timed steps only")
    ap.add_argument("--cpu-only", action="store_true", help="print the cpu_baseline object (CPU LAPACK leg alone) and exit: no GPU, no torch")
    args = ap.parse_args()
    if args.legs:
        want = {w.strip() for w in args.legs.split(",") if w.strip() and w.strip() != "none"}
        for leg in ("ir", "phases", "mxp", "config5", "ref_style", "cpu", "configs", "check"):
            if leg not in want:
                setattr(args, "no_" + leg, True)
    if args.cpu_only:
        print(json.dumps(cpu_baseline(args.cpu_n)))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  The ranks are started as a CHILD process (never exec'ed)
        # and before this process has imported torch or touched the GPU; its stdout (rank 0's one JSON line) and its
        # return code are relayed unchanged.
        raise SystemExit(self_launch(args))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = os.environ.get("MPF_BENCH_REHEARSAL") == "1"   # ranks share the visible GPU(s), gloo through host memory
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MPF hot path is HIP-only (no CPU fallback)")
    mpf = importlib.import_module("mixed-precision_lu_factorization_amd")

    if world > 1:
        import bench_dist   # (benchmark code beside this file, not in the package)
        return bench_dist.bench_main(args, rank, world, local_rank, rehearsal=rehearsal)

    dev = torch.device("cuda", local_rank)
    ctx = mpf.MPFContext(local_rank)
    n, nb = args.n, args.nb
    A0 = ctx.matgen(n, skip=4 + args.seed)   # the reference generator's stream on the device (mpf_matgen_dev): the oracle's N=32768 input
    headline = args.trailing == "fp64" and args.gen == "ref" and args.seed == 0
    if args.gen != "ref":
        idx0 = torch.arange(n, device=dev)
        A0[idx0, idx0] += A0.sum(dim=1)
        if args.gen == "kappa":
            A0 *= torch.logspace(0, 8, n, dtype=torch.float64, device=dev)[:, None]
    tmode = {"fp64": mpf.TRAIL_FP64, "fp16": mpf.TRAIL_FP16, "fp16x3": mpf.TRAIL_FP16X3}[args.trailing]
    free_b, _ = torch.cuda.mem_get_info(dev)
    per = n * n * 8
    ncopies = max(1, min(args.steps + args.warmup, int((free_b * 0.8) // per)))
    work = [torch.empty((n, n), dtype=torch.float64, device=dev).t() for _ in range(ncopies)]

    def fresh(i):
        w = work[i % ncopies]
        w.copy_(A0)
        return w

    ipiv = None
    for i in range(args.warmup):
        ipiv, info = ctx.factor(fresh(i), nb, trailing=tmode)
    # inputs for the timed steps are staged in HBM before the clock starts
    staged = args.steps <= ncopies
    if staged:
        mats = [fresh(i) for i in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev_ms = 0.0
    for i in range(args.steps):
        w = mats[i] if staged else fresh(i)
        ipiv, info = ctx.factor(w, nb, trailing=tmode)
        st_ = ctx.stats()
        dev_ms += st_.ms_total
        last_stats = {"ms_gemm": st_.ms_gemm, "gemm_launches": st_.gemm_launches, "lookahead": st_.lookahead,
                      "ms_hpanel": st_.ms_hpanel, "ms_trsm": st_.ms_trsm, "ms_laswp": st_.ms_laswp, "ms_dpanel": st_.ms_dpanel,
                      "gemm_flops": st_.gemm_flops, "gemm_bytes": st_.gemm_bytes, "superpanel": st_.superpanel}
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms_per_step = dt * 1e3 / args.steps
    flops = 2.0 / 3.0 * n ** 3
    value = flops / (ms_per_step * 1e-3) / 1e9
    LU = w
    # The timed steps run with the library's default timers: HIP-event pairs around the trailing-update launches only (what the
    # roofline needs).  The other per-phase timers cost ~8 ms per step (an event pair around every small launch of the chain): one
    # extra, untimed step with all of them on (option event_timers = 2) gives the breakdown below.
    ctx.set_option("event_timers", 2)
    ctx.factor(fresh(0), nb, trailing=tmode)
    sd = ctx.stats()
    ctx.set_option("event_timers", 1)
    diag_stats = {"ms_hpanel": sd.ms_hpanel, "ms_trsm": sd.ms_trsm, "ms_laswp": sd.ms_laswp, "ms_dpanel": sd.ms_dpanel, "ms_gemm": sd.ms_gemm,
                  "ms_total": sd.ms_total}

    # ---- refinement solve on the last factorization (metric: IR iterations to ||r||/||b|| < 1e-12) ----
    ir = None
    if not args.no_ir:
        xs = torch.ones(n, dtype=torch.float64, device=dev)
        b = A0 @ xs
        x, st = ctx.solve_ir(A0, LU, ipiv, b, max_iter=args.ir_steps, tol=1e-12)
        ir = {"iterations": int(st.iterations), "rel_residual": float(st.rel_residual), "converged": bool(st.converged),
              "ms": round(float(st.ms_total), 2)}

    timed_mode_stats = st_   # library counters of the last timed step (the fp16 modes' roofline object is built from them below)
    check = None
    if not args.no_check:
        mx, fro = ctx.check_plu(A0, LU, ipiv)
        check = {"max_abs_A_minus_PLU": mx, "fro_rel": fro, "criterion": 1e-10, "passed": bool(mx <= 1e-10),
                 "what": "the reference's acceptance test (benchmark.cpp:97-144), L U on the device; the fp16 trailing modes are not "
                         "expected to meet it (their answer is the refined solve)"}

    # ---- roofline of the dominant kernel (dgemm_minus_kernel8d, f64 MFMA): HIP-event pairs around every GEMM
    #      launch of the LAST TIMED step, on the stream the kernel was launched on (no host sync in between;
    #      the concurrent look-ahead panel work is included in the durations) ------------------------------
    gflops_total = last_stats["gemm_flops"]              # sum of 2 m n k over the launches timed under ms_gemm (library count)
    launches = max(int(last_stats["gemm_launches"]), 1)
    ms_gemm = last_stats["ms_gemm"]
    achieved = gflops_total / (ms_gemm * 1e-3) / 1e12 if ms_gemm > 0 else 0.0
    roofline = {"kernel": "dgemm_minus_kernel8d", "bound": "mfma", "achieved": round(achieved, 2), "peak": F64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / F64_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                "launches": launches, "avg_launch_ms": round(ms_gemm / launches, 4),
                "flop_per_launch_avg": gflops_total / launches,
                "algorithmic_bytes_per_launch_avg": last_stats["gemm_bytes"] / launches,
                "superpanel": int(last_stats["superpanel"]),
                "frac_source": "HIP-event pairs around the update launches of the last timed step, on the stream they are launched on (this run); "
                               "`rocprof`: the same kernel's average duration in the committed rocprofv3 --kernel-trace --stats summary of this command"}
    # What the f64 matrix pipe of THIS box sustains with no memory traffic at all (register-only loops, measured in this
    # process; profiles/r02_mfma_f64_issue.txt): one dependent accumulator chain per wave and the 4x4x4_4b form, both with
    # four waves per SIMD (the occupancy the update kernel runs at), and round 1's loop (8 accumulators, 2 waves per SIMD).
    # (the microbenchmarks live in libmpf_probe.so -- include/mpf_probe.h -- not in the product library)
    pctx = mpf.MPFContext(local_rank, probe=True)
    measured = {"v_mfma_f64_16x16x4 one accumulator, 4 waves/SIMD": round(pctx.microbench(234), 1),
                "v_mfma_f64_4x4x4_4b 16 accumulators, 4 waves/SIMD": round(pctx.microbench(214), 1),
                "v_mfma_f64_16x16x4 8 accumulators, 2 waves/SIMD": round(pctx.microbench(0), 1)}
    roofline["peak_measured_register_only_tflops"] = measured
    roofline["frac_of_measured_peak"] = round(achieved / max(measured.values()), 4)
    # The same kernel ALONE on the chip (no pivot chain, no small launches beside it) on two of the factorization's own shapes:
    # what `achieved` loses to the schedule.  Since round 3 the schedule runs every small launch UNDER an update on purpose
    # (two lanes): the step gets shorter while the update launches themselves get longer.
    try:
        alone = {}
        for nn in (28672, 16384):
            Cm = ctx.colmajor(nn, nn); Am = ctx.colmajor(nn, nb); Bm = ctx.colmajor(nb, nn)
            Cm.normal_(); Am.normal_(); Bm.normal_()
            for _ in range(2):
                ctx.dgemm_minus(Cm, Am, Bm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ctx.dgemm_minus(Cm, Am, Bm)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            tf = 2.0 * nn * nn * nb / (ms * 1e-3) / 1e12
            alone[f"m=n={nn} k={nb}"] = {"ms": round(ms, 3), "tflops": round(tf, 1), "frac": round(tf / F64_MFMA_PEAK_TFLOPS, 4)}
            del Cm, Am, Bm
        roofline["kernel_alone"] = alone
        torch.cuda.empty_cache()
    except Exception as ex:  # (diagnostic only)
        roofline["kernel_alone"] = {"error": str(ex)}
    roofline["mfma_f64_cycles_one_wave_16_accumulators"] = round(pctx.microbench(60), 1)
    roofline["mfma_f64_cycles_one_wave_one_accumulator"] = round(pctx.microbench(130), 1)
    # the other two roofs, measured on this box next to their specification values (SURVEY 8d "print both")
    hbm_copy_tbps = pctx.microbench(2)       # 2 GiB read + 2 GiB write stream copy
    f16_mfma_tflops = pctx.microbench(1)     # register-only v_mfma_f32_32x32x16_f16 loop
    # (the stream copy is information, not a roof: sixteen forms of it read 5.0-5.7 TB/s on one box -- gpurun_out/r04_s_hbm.log --
    #  where the big fp16 update's own C stream reaches 5.8-6.9 and the guide measures 6.29: nothing is divided by it any more)
    peaks = {"hbm_spec_TBps": 8.0, "hbm_stream_copy_measured_TBps": round(hbm_copy_tbps, 2),
             "fp16_mfma_spec_tflops": 2500.0, "fp16_mfma_register_only_measured_tflops": round(f16_mfma_tflops, 1),
             "fp64_mfma_spec_tflops": F64_MFMA_PEAK_TFLOPS, "fp64_mfma_register_only_measured_tflops": max(measured.values())}
    pctx.close()
    # HBM-side traffic of this kernel: separate rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE, the guide's gfx950
    # correction), summarised in profiles/r04_pmc_nongemm_summary.json together with the sha of the kernel source they measured.
    # Quoted only when that sha is the source this library was built from; otherwise null (never a stale replay).
    roofline["traffic_source"] = ("profiles/r05_pmc_nongemm_summary.json, else r04_pmc_nongemm_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over whole factorizations, "
                                  "tools/pmc_factor.sh; FETCH_SIZE x 2: the guide's gfx950 correction)")
    # rocprofv3's own figure for the same kernel (profiles/r05_rocprof_summary.json, written by tools/rocprof_summarize.py from the
    # kernel-stats CSV of `rocprofv3 --kernel-trace --stats -- python3 bench.py`), quoted only for the source it was taken on
    try:
        with open(os.path.join(ROOT, "profiles", "r05_rocprof_summary.json")) as f:
            rsum = json.load(f)
        rk = rsum["kernels"].get("dgemm_minus_kernel8d<16, 2, 1>")
        if rk and rsum["sources_sha16"].get("trailing_f64.hip") == kernel_source_sha("trailing_f64.hip") and headline and rsum["n"] == n and rsum["nb"] == nb:
            tf_r = roofline["flop_per_launch_avg"] / (rk["avg_ns"] * 1e-9) / 1e12
            roofline["rocprof"] = {"avg_launch_ms": round(rk["avg_ns"] * 1e-6, 4), "calls": rk["calls"], "achieved": round(tf_r, 2),
                                   "frac": round(tf_r / F64_MFMA_PEAK_TFLOPS, 4), "file": "profiles/r05_kernel_stats.csv"}
        else:
            roofline["rocprof"] = None
    except Exception:
        roofline["rocprof"] = None
    pmc_sum = None
    try:
        with open(os.path.join(ROOT, "profiles", "r05_pmc_nongemm_summary.json" if os.path.exists(os.path.join(ROOT, "profiles", "r05_pmc_nongemm_summary.json")) else "r04_pmc_nongemm_summary.json")) as f:
            pmc_sum = json.load(f)
        pm = pmc_sum["kernels"]["dgemm_minus_kernel8d<16, 2, 1>"]
        # (the PMC run factors the same matrix size once in fp64 with the chain not pipelined: the same updates in slightly fewer
        #  launches -- the kernel's bytes over the whole factorization against this run's algorithmic bytes, then per launch)
        if (pmc_sum["sources_sha16"].get("trailing_f64.hip") == kernel_source_sha("trailing_f64.hip") and headline
                and pmc_sum["probe"]["n"] == n and pmc_sum["probe"]["nb"] == nb):
            ratio = (pm["fetch_bytes"] + pm["write_bytes"]) / last_stats["gemm_bytes"]
            roofline["traffic"] = round(ratio * last_stats["gemm_bytes"] / launches)
            roofline["traffic_over_algorithmic"] = round(ratio, 3)
            roofline["traffic_launches_in_pmc_run"] = pm["launches"]
        else:
            roofline["traffic_note"] = "PMC summary was taken on another version of trailing_f64.hip (or another configuration): not quoted"
    except Exception as e:
        roofline["traffic_note"] = f"PMC summary not usable ({type(e).__name__}: {e}): not quoted"
    overlap = {"what": "one extra step with every timer on (option event_timers = 2; the timed steps keep the update timers only)",
               "lookahead": bool(last_stats["lookahead"]), "step_ms_with_all_timers": round(diag_stats["ms_total"], 2),
               "panel_chain_ms": round(diag_stats["ms_hpanel"] + diag_stats["ms_dpanel"], 2),
               "chain_hgetf2_ms": round(diag_stats["ms_hpanel"], 2),
               # event pair around the fp64 panel's launches INCLUDING the gates' waiting for the pivot kernel (pipelined
               # chain): a span, not a measure of work
               "chain_dpanel_span_incl_gate_wait_ms": round(diag_stats["ms_dpanel"], 2),
               "trsm_ms": round(diag_stats["ms_trsm"], 2), "laswp_others_ms": round(diag_stats["ms_laswp"], 2),
               "gemm_ms": round(diag_stats["ms_gemm"], 2), "gemm_ms_timed_step": round(ms_gemm, 2)}
    # per-phase times with every phase alone on the chip (single stream, host sync between phases)
    phases = None
    if not args.no_phases:
        wprof = fresh(0)
        ctx.factor(wprof, nb, sync_timing=True)
        s = ctx.stats()
        phases = {"hpanel_ms": round(s.ms_hpanel, 2), "laswp_ms": round(s.ms_laswp, 2), "dpanel_ms": round(s.ms_dpanel, 2),
                  "trsm_ms": round(s.ms_trsm, 2), "gemm_ms": round(s.ms_gemm, 2), "total_ms": round(s.ms_total, 2)}

    # ---- speed mode (north_star): fp16-in/fp32-acc MFMA trailing update + fp64 refinement, on the
    #      IR-friendly input of SURVEY 8d (generator matrix + diag(rowsum)) -----------------------------------
    def run_mxp(mode, label, Aorig, matrix_desc):
        """factor a copy of Aorig with the given trailing mode, then refine to 1e-12; all buffers resident in HBM"""
        Ad = fresh(1 % ncopies)
        Ad.copy_(Aorig)
        ctx.factor(Ad, nb, trailing=mode)            # warm-up of this mode
        Ad.copy_(Aorig)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ipiv16, info16 = ctx.factor(Ad, nb, trailing=mode)
        torch.cuda.synchronize()
        t_fact = time.perf_counter() - t1
        s16 = ctx.stats()
        # one more factorization with every timer on: the breakdown fields (its matrix: a spare copy, or a new one)
        sdg = None
        try:
            Ad2 = work[3] if ncopies > 3 else torch.empty((n, n), dtype=torch.float64, device=dev).t()
            if Ad2.data_ptr() not in (Ad.data_ptr(), Aorig.data_ptr()):
                Ad2.copy_(Aorig)
                ctx.set_option("event_timers", 2)
                ctx.factor(Ad2, nb, trailing=mode)
                sdg = ctx.stats()
            del Ad2
        except Exception:
            sdg = None
        finally:
            ctx.set_option("event_timers", 1)
        xs = torch.ones(n, dtype=torch.float64, device=dev)
        b16 = Aorig @ xs
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        x16, st16 = ctx.solve_ir(Aorig, Ad, ipiv16, b16, max_iter=20, tol=1e-12)
        torch.cuda.synchronize()
        t_ir = time.perf_counter() - t2
        return {"trailing": label, "matrix": matrix_desc,
                "factor_ms": round(t_fact * 1e3, 2), "factor_gflops": round(flops / t_fact / 1e9, 1),
                "ir_iterations": int(st16.iterations), "ir_rel_residual": float(st16.rel_residual), "ir_converged": bool(st16.converged),
                "ir_ms": round(t_ir * 1e3, 2), "solve_gflops_incl_ir": round(flops / (t_fact + t_ir) / 1e9, 1),
                # every trailing-update kernel launch (inner-region K = nb updates and the K = sb * nb updates; conversions are
                # booked under cvt_ms) is timed under gemm_ms; flops and algorithmic bytes are the library's own counts for
                # exactly those launches.  The big-K launches alone: "roofline" below.
                "superpanel": int(s16.superpanel), "gemm_launches": int(s16.gemm_launches),
                "gemm_ms": round(s16.ms_gemm, 2), "gemm_tflops": round(s16.gemm_flops / (s16.ms_gemm * 1e-3) / 1e12, 1) if s16.ms_gemm > 0 else None,
                "gemm_frac_of_fp16_mfma_peak": round(s16.gemm_flops / (s16.ms_gemm * 1e-3) / 2.5e15, 4) if s16.ms_gemm > 0 else None,
                "gemm_frac_of_fp16_mfma_peak_measured": round(s16.gemm_flops / (s16.ms_gemm * 1e-3) / (peaks["fp16_mfma_register_only_measured_tflops"] * 1e12), 4) if s16.ms_gemm > 0 else None,
                "gemm_hbm_algorithmic_TBps": round(s16.gemm_bytes / (s16.ms_gemm * 1e-3) / 1e12, 2) if s16.ms_gemm > 0 else None,
                "gemm_frac_of_hbm_peak": round(s16.gemm_bytes / (s16.ms_gemm * 1e-3) / 8e12, 4) if s16.ms_gemm > 0 else None,
                # (from the extra run with every timer on; null when the device had no room for another copy of the matrix)
                "cvt_ms": round(sdg.ms_cvt, 2) if sdg else None, "blockrow_ms": round(sdg.ms_blockrow, 2) if sdg else None,
                "trsm_ms": round(sdg.ms_trsm, 2) if sdg else None, "laswp_ms": round(sdg.ms_laswp, 2) if sdg else None,
                "chain_hgetf2_ms": round(sdg.ms_hpanel, 2) if sdg else None,
                "factor_ms_with_all_timers": round(sdg.ms_total, 2) if sdg else None,
                "roofline": mxp_roofline(s16, split=(mode == mpf.TRAIL_FP16X3)),
                "info": int(info16)}

    def mxp_roofline(s, split):
        """Roofline object of the fp16 modes' dominant kernel: the K = sb * nb update launches ALONE (library counters: HIP-event
        time on the launch stream, 2 m n K flops, algorithmic bytes = the fp32 working copy read + written once + the fp16
        operand images once).  The kernel is bound by whichever roof is lower at its intensity: HBM x flop/byte or the MFMA peak."""
        if s.ms_gemm_big <= 0 or s.gemm_big_launches <= 0:
            return None
        t = s.ms_gemm_big * 1e-3
        tf = s.gemm_big_flops / t / 1e12
        tbps = s.gemm_big_bytes / t / 1e12
        intensity = s.gemm_big_flops / s.gemm_big_bytes
        mfma_work = 3.0 if split else 1.0          # MFMA products issued per counted flop (hi*hi + hi*lo + lo*hi)
        out = {"kernel": ("hgemm_big_kernel<SPLIT=true, C32=true> (v_mfma_f32_32x32x16_f16, 256 x 128 tiles, C stream pipelined)" if split else
                          "hgemm16_big_kernel<C32=true> (v_mfma_f32_16x16x32_f16, 256 x 256 tiles, permuted L rows + dwordx4 C stream, first batch "
                          "requested inside the K loop)"),
               "launches": int(s.gemm_big_launches), "avg_launch_ms": round(s.ms_gemm_big / s.gemm_big_launches, 4),
               "flop_per_launch_avg": s.gemm_big_flops / s.gemm_big_launches,
               "algorithmic_bytes_per_launch_avg": s.gemm_big_bytes / s.gemm_big_launches,
               "flop_per_byte": round(intensity, 1), "achieved": round(tf, 1), "unit": "TFLOP/s",
               "mfma_products_per_flop": mfma_work,
               "hbm_algorithmic_TBps": round(tbps, 2),
               "frac_of_fp16_mfma_peak_spec": round(tf * mfma_work / peaks["fp16_mfma_spec_tflops"], 4),
               "frac_of_fp16_mfma_peak_measured": round(tf * mfma_work / peaks["fp16_mfma_register_only_measured_tflops"], 4),
               "frac_of_hbm_peak_spec": round(tbps / peaks["hbm_spec_TBps"], 4)}
        roof_spec = min(peaks["fp16_mfma_spec_tflops"] / mfma_work, intensity * peaks["hbm_spec_TBps"])
        out["bound"] = "mfma" if peaks["fp16_mfma_spec_tflops"] / mfma_work <= intensity * peaks["hbm_spec_TBps"] else "hbm"
        out["roof_spec_tflops"] = round(roof_spec, 1)
        out["frac"] = round(tf / roof_spec, 4)
        # HBM-side bytes per launch from the PMC passes (same N, same schedule), quoted only for the source they were taken on
        out["traffic"] = None
        try:
            src = "trailing_f16.hip" if split else "hgemm16.hip"
            if pmc_sum and pmc_sum["sources_sha16"].get(src) == kernel_source_sha(src) and n == pmc_sum["probe"]["n"]:
                # the K = sb * nb launches of the PMC run's fp16 factorization: the instantiation with the most bytes
                want = "hgemm_big_kernel<true, true" if split else "hgemm16_big_kernel<true"
                cands = [v for k_, v in pmc_sum["kernels"].items() if k_.startswith(want)]
                if cands:
                    pk = max(cands, key=lambda v: v["fetch_bytes"] + v["write_bytes"])
                    out["traffic"] = round((pk["fetch_bytes"] + pk["write_bytes"]) / pk["launches"])
                    out["traffic_launches_in_pmc_run"] = pk["launches"]
                    out["traffic_over_algorithmic"] = round(out["traffic"] / out["algorithmic_bytes_per_launch_avg"], 3)
        except Exception:
            pass
        return out

    if args.trailing != "fp64":   # the timed steps ran an fp16 mode: their dominant kernel is the big-K fp16 update
        r16 = mxp_roofline(timed_mode_stats, split=(args.trailing == "fp16x3"))
        if r16:
            roofline = dict(r16, peak=peaks["fp16_mfma_spec_tflops"], traffic=None,
                            note="timed steps in an fp16 trailing mode (--trailing): roofline of the K = sb * nb update launches")
    mxp = mxp_x3 = mxp_gmres = None
    if not args.no_mxp:
        Aorig = work[2 % ncopies] if ncopies > 2 else torch.empty((n, n), dtype=torch.float64, device=dev).t()
        if Aorig.data_ptr() == work[1 % ncopies].data_ptr():
            Aorig = torch.empty((n, n), dtype=torch.float64, device=dev).t()
        # (a) fp16 operands, on the IR-friendly input of SURVEY 8d (generator matrix + diag(rowsum))
        Aorig.copy_(A0)
        idx = torch.arange(n, device=dev)
        Aorig[idx, idx] += A0.sum(dim=1)
        mxp = run_mxp(mpf.TRAIL_FP16, "fp16-in/fp32-acc MFMA", Aorig, "generator + diag(rowsum) (diagonally dominant)")
        # (b) split fp16 operands (hi + 2^-11 lo, three MFMA products), on the reference generator's matrix itself,
        #     where plain fp16 operands make the refinement diverge for N >= 8192
        mxp_x3 = run_mxp(mpf.TRAIL_FP16X3, "fp16x3 (hi/lo split operands, fp32-acc MFMA)", A0, "generator matrix (matrix_generator.cpp:66 distribution)")

        # (c) plain fp16 operands on the generator's matrix itself: classical refinement diverges there (kappa ~ 3e6), GMRES
        #     preconditioned with the same factors converges (mpf_solve_gmres_ir)
        Ad = fresh(1 % ncopies)
        Ad.copy_(A0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ipg, infog = ctx.factor(Ad, nb, trailing=mpf.TRAIL_FP16)
        torch.cuda.synchronize()
        t_f = time.perf_counter() - t1
        bg = A0 @ torch.ones(n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        xg, gm = ctx.solve_gmres_ir(A0, Ad, ipg, bg, max_outer=10, restart=50, tol=1e-12)
        torch.cuda.synchronize()
        t_g = time.perf_counter() - t2
        mxp_gmres = {"trailing": "fp16-in/fp32-acc MFMA", "matrix": "generator matrix (matrix_generator.cpp:66 distribution)",
                     "solver": "GMRES-IR: GMRES(50) on the correction equation, preconditioned with the fp16-mode factors, fp64",
                     "factor_ms": round(t_f * 1e3, 2), "outer_iterations": int(gm.outer_iterations), "inner_iterations": int(gm.inner_iterations),
                     "rel_residual": float(gm.rel_residual), "converged": bool(gm.converged), "solve_ms": round(t_g * 1e3, 2),
                     "solve_gflops_incl_refinement": round(flops / (t_f + t_g) / 1e9, 1)}

    # ---- BASELINE config 5: kappa ~ 1e8 row-scaled diagonally dominant matrix through mpf_gesv (fp16 path first,
    #      automatic fp64 fallback when the refinement stalls) -------------------------------------------------------
    config5 = None
    if not args.no_config5:
        Ak = fresh(0)
        idx = torch.arange(n, device=dev)
        Ak[idx, idx] += A0.sum(dim=1)
        Ak *= torch.logspace(0, 8, n, dtype=torch.float64, device=dev)[:, None]
        xs = torch.ones(n, dtype=torch.float64, device=dev)
        bk = Ak @ xs
        wk = work[1 % ncopies] if ncopies > 1 else torch.empty((n, n), dtype=torch.float64, device=dev).t()
        torch.cuda.synchronize()
        xk, gs, _, _ = ctx.gesv(Ak, bk, nb, max_iter=10, tol=1e-12, work=wk)
        config5 = {"matrix": "(generator + diag(rowsum)) rows scaled by logspace(0, 8): kappa ~ 1e8",
                   "path": "fp16 trailing + IR" if gs.path == 1 else "fp64 fallback after the fp16 refinement stalled",
                   "fp16_attempt_ms": round(gs.ms_factor_fp16 + gs.ms_ir_fp16, 1), "fp16_ir_history": [float(v) for v in list(gs.ir_fp16.history)[:3]],
                   "fp64_factor_ms": round(gs.ms_factor_fp64, 1), "final_rel_residual": float(gs.ir_final.rel_residual),
                   "final_ir_iterations": int(gs.ir_final.iterations), "converged": bool(gs.ir_final.converged), "total_ms": round(gs.ms_total, 1)}

    # ---- the reference's own way of timing (benchmark.cpp:219-222: clock around the whole MPF() call -- handle creation,
    #      device allocation, H2D of the matrix, factorization, D2H, release): one run on host buffers, never `value` --------
    ref_style = None
    if not args.no_ref_style:
        import numpy as np
        Ah = ctx.to_numpy_f(A0)                        # pageable host memory, column-major, as benchmark.cpp holds it
        ip_h = np.arange(1, n + 1, dtype=np.int32)     # benchmark.cpp:215-217
        torch.cuda.empty_cache()                       # (the fresh context allocates from the device, not from torch's pool)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        hctx = mpf.MPFContext(local_rank)
        Ah1 = Ah.copy(order="F")
        hctx.factor_host(Ah1, nb, ip_h.copy())
        sh1 = hctx.stats()
        t_first = time.perf_counter() - t1
        del Ah1
        t2 = time.perf_counter()                       # the second matrix of benchmark.cpp:181-266's loop: the context and its
        hctx.factor_host(Ah, nb, ip_h)                 # device buffers are there (MPF() keeps them for the life of the process)
        sh = hctx.stats()
        t_ref = time.perf_counter() - t2
        hctx.close()
        ref_style = {"ms": round(t_ref * 1e3, 1), "gflops": round(flops / t_ref / 1e9, 1), "h2d_ms": round(sh.ms_h2d, 1),
                     "d2h_ms": round(sh.ms_d2h, 1), "factor_ms": round(sh.ms_total, 1), "rows_streamed": int(sh.host_rows_streamed), "late_segments": int(sh.host_late_segments),
                     "first_call_ms": round(t_first * 1e3, 1), "first_call_factor_ms": round(sh1.ms_total, 1),
                     "what": "wall clock around the host-buffer entry point, what benchmark.cpp:219-222 times around MPF(): `ms` = the SECOND "
                             "call of a context (`h2d_ms`: the FIRST part of the 8 GiB matrix from pageable memory; `late_segments` more column segments go up "
                             "while the factorization has started on it and receive the panels they missed afterwards, MPF_HOST_LATE_PARTS=0: the whole matrix first, as MPF.cu:82; "
                             "then factor, and the way home: `rows_streamed` block rows "
                             "of the factors leave for the caller's pageable matrix WHILE the factorization runs (rowsink.hip), `d2h_ms` is what is left "
                             "of that after the last kernel; MPF_HOST_SINK=0: one copy afterwards, as MPF.cu:245-247; the device copy and the "
                             "row-major working copy are kept between calls, as MPF() keeps them for the life of the process); "
                             "`first_call_ms` = mpf_create + the first call (adds two 8 GiB hipMallocs and handle set-up)"}
        del Ah

    # ---- every BASELINE config that fits one GPU gets a driver-run number.  C3 = the timed steps above, C5 =
    #      config5; here C1 (N = 1024, r = 32: the reference's own CPU-runnable case, on the GPU), C2 (N = 8192, nb = 128, fp16 panel +
    #      fp16-MFMA trailing update + 3-step refinement) and C4's size (N = 65536, nb = 256) on this one GPU when HBM allows ------
    configs = None
    if not args.no_configs:
        configs = {"C3": {"what": "the timed steps of this line", "n": n, "nb": nb, "ms": round(ms_per_step, 3), "gflops": round(value, 1),
                          "check_passed": (check or {}).get("passed")},
                   "C5": {"what": "the `config5` object of this line"}}

        def one_config(nn, r, mode, diagdom, ir_steps, warm=True):
            """one factorization of the generator's N = nn matrix (+ diag(rowsum) when diagdom) with panel width r, device-event time,
            then the refinement (ir_steps > 0) and the reference's PLU test where the arithmetic is the reference's"""
            A = ctx.matgen(nn)
            if diagdom:
                ii = torch.arange(nn, device=dev)
                A[ii, ii] += A.sum(dim=1)
            W = A.clone()
            if warm:
                ctx.factor(W, r, trailing=mode)
                W.copy_(A)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ipv, inf = ctx.factor(W, r, trailing=mode)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t1
            sct = ctx.stats()
            out = {"n": nn, "nb": r, "trailing": {mpf.TRAIL_FP64: "fp64", mpf.TRAIL_FP16: "fp16", mpf.TRAIL_FP16X3: "fp16x3"}[mode],
                   "matrix": "generator + diag(rowsum)" if diagdom else "generator", "ms": round(sct.ms_total, 3), "wall_ms": round(wall * 1e3, 3),
                   "tflops": round(2.0 / 3.0 * nn ** 3 / (sct.ms_total * 1e-3) / 1e12, 2), "info": int(inf), "hpanel_timeouts": int(sct.hpanel_timeouts)}
            if mode == mpf.TRAIL_FP64 and nn <= 32768:   # (the test's L U product is 2 N^3 flops: ~9 s at N = 65536)
                mx_, _ = ctx.check_plu(A, W, ipv)
                out["max_abs_A_minus_PLU"] = mx_
                out["check_passed"] = bool(mx_ <= 1e-10)
            if ir_steps > 0:
                bb = A @ torch.ones(nn, dtype=torch.float64, device=dev)
                _, sti = ctx.solve_ir(A, W, ipv, bb, max_iter=ir_steps, tol=1e-12)
                out["ir"] = {"max_steps": ir_steps, "iterations": int(sti.iterations), "rel_residual": float(sti.rel_residual),
                             "converged": bool(sti.converged), "ms": round(float(sti.ms_total), 2)}
            del A, W
            return out, sct

        try:
            configs["C1"], _ = one_config(1024, 32, mpf.TRAIL_FP64, False, 0)
            configs["C1"]["what"] = ("N = 1024, r = 32 as benchmark.cpp:220 calls MPF(): on the GPU (generic pivot path: 32-column panels); the "
                                     "reference's CPU run of this case: cpu_baseline.port_value (oracle) and tests/test_harness.py")
            c2, _ = one_config(8192, 128, mpf.TRAIL_FP16, True, 3)
            # the chain per column: one more run with every timer on (pivot-kernel span / columns)
            ctx.set_option("event_timers", 2)
            c2t, sc2 = one_config(8192, 128, mpf.TRAIL_FP16, True, 0, warm=False)
            ctx.set_option("event_timers", 1)
            c2["chain_hgetf2_us_per_column"] = round(sc2.ms_hpanel * 1e3 / 8192, 3)
            c2["ms_with_all_timers"] = c2t["ms"]
            c2["what"] = "N = 8192, nb = 128, fp16 pivot panel + fp16-MFMA trailing update + at most 3 fp64 refinement sweeps, diagonally dominant input"
            configs["C2"] = c2
        except Exception as ex:
            configs["error_C1_C2"] = repr(ex)[:300]
        finally:
            ctx.set_option("event_timers", 1)
        try:
            # C4's size on ONE GPU: free the N = 32768 copies first (the context keeps its own working copies)
            LU = None; w = None; mats = None
            del work[:]
            torch.cuda.empty_cache()
            free_b, _ = torch.cuda.mem_get_info(dev)
            n4 = 65536
            if free_b > 3.6 * 8 * n4 * n4:
                c464, _ = one_config(n4, 256, mpf.TRAIL_FP64, False, 0, warm=False)
                c416, _ = one_config(n4, 256, mpf.TRAIL_FP16, True, 3, warm=False)
                configs["C4_size_on_one_gpu"] = {"what": "N = 65536, nb = 256 on ONE MI355X (BASELINE C4 is this size over 8 GPUs: bench.py --gpus 8); one "
                                                         "run each, device events, first call at this size (buffers allocated outside the events)",
                                                 "fp64": c464, "fp16": c416}
            else:
                configs["C4_size_on_one_gpu"] = {"skipped": f"{free_b / 1e9:.0f} GB of HBM free, {3.6 * 8 * n4 * n4 / 1e9:.0f} GB needed"}
        except Exception as ex:
            configs["error_C4"] = repr(ex)[:300]
        torch.cuda.empty_cache()

    line = {
        "metric": "LU GFLOP/s at N=32768 (1/2/4/8 GPUs); IR iterations to ||r||/||b||<1e-12",
        "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64" if args.trailing == "fp64" else "f16 operands, f32 accumulation in the trailing update (fp64 panels and TRSM)",
        "data": "synthetic: the reference generator's own stream (`matgen f N (N-2) lin`, matrix_generator.cpp:55-80: "
                                "glibc rand() seed 1, 4 draws skipped, (rand() % 100) / 10.0), produced on the device by mpf_matgen_dev",
        "config": {"workload": (f"N={n} nb={nb} MPF LU: fp16 pre-pivot panel + fp64 no-pivot panel + fp64 TRSM/MFMA-GEMM "
                                f"trailing update (reference arithmetic), 1 MI355X, matrix resident in HBM") if headline else
                               (f"N={n} nb={nb} MPF LU, trailing update {args.trailing}, input {args.gen} (seed {args.seed}), 1 MI355X, matrix resident "
                                f"in HBM -- NOT the headline configuration (--trailing / --gen / --seed given)"),
                   "n": n, "nb": nb, "trailing": args.trailing, "gen": args.gen, "seed": args.seed, "parallelism": "1 GPU",
                   "headline_configuration": headline},
        "device_ms_per_step": round(dev_ms / args.steps, 3), "info": int(info), "ir": ir, "timed_step_events": overlap,
        "phases_sync_timed": phases, "mxp": mxp, "mxp_x3": mxp_x3, "mxp_gmres": mxp_gmres, "config5": config5,
        "roofline": roofline, "peaks": peaks, "reference_style": ref_style, "check": check, "configs": configs,
    }
    if not args.no_cpu:
        line["cpu_baseline"] = cpu_baseline(args.cpu_n)
    print(json.dumps(_finite(line), allow_nan=False))


if __name__ == "__main__":
    main()
