#!/usr/bin/env python
"""Benchmark of the VI iteration (M-step -> gamma-step -> prior table -> E-step, the body of
/root/reference/src/var_inf.py:135-137) on synthetic Gaussian-mixture latents.

  python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1: under a launcher (torch.distributed.run sets WORLD_SIZE) this
process is one rank; from a bare shell the parent starts N fresh children itself -- before it
imports torch or touches a GPU -- one rank per GPU over RCCL, and prints rank 0's JSON line.

A "step" is one full VI iteration over the rank's resident latents (Z, phi, video ids already in
HBM) plus the all-reduce of the split/merge statistics the drop-in issues after every E-step.
Headline workload = BASELINE.json configs[1]: N=1M frames per rank, D=64, 8-ary tree with K=73
nodes / L=64 leaves, videos of 256 frames (weak scaling: every rank holds its own 1M frames).  The
same JSON line carries `config3_strong`: BASELINE configs[2] (16M frames in total, D=64, K=585)
divided over the ranks -- the strong-scaling case of north_star's ">= 7x at 8 GPUs on N=16M".
Collectives per step: one fp64 all-reduce of the L x (1+D) leaf statistics (the M-step) and one of
the L x 2 split/merge statistics.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (N, D, branching, levels below root)   -- BASELINE.json configs / SURVEY.md section 0
    "config1": (60_000, 32, 5, 2),        # K=31,  L=25
    "config2": (1_000_000, 64, 8, 2),     # K=73,  L=64   (headline, memory-bound E/M-step; N per rank)
    "config3": (16_000_000, 64, 8, 3),    # K=585, L=512  (total N; divided over ranks)
    "config4": (4_000_000, 256, 6, 5),    # K=9331, L=7776 (total N; divided over ranks)
}
FRAMES_PER_VIDEO = 256
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16 / f16 MFMA


def regular_tree(bf, levels):
    """Pre-order parent / is_leaf arrays of the full bf-ary tree."""
    parent, is_leaf = [], []

    def rec(p, depth):
        k = len(parent)
        parent.append(p)
        is_leaf.append(1 if depth == levels else 0)
        if depth < levels:
            for _ in range(bf):
                rec(k, depth + 1)

    rec(-1, 0)
    return np.asarray(parent, np.int32), np.asarray(is_leaf, np.uint8)


def make_latents(torch, device, N, D, L, rank):
    """L centres ~ N(0, 4^2 I), uniform component, unit noise, fp32 (device RNG)."""
    mu = torch.from_numpy(np.random.RandomState(13).normal(scale=4.0, size=(L, D)).astype(np.float32)).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(13 + 1000 * rank)
    Z = torch.empty((N, D), dtype=torch.float32, device=device)
    step = 1 << 21                                   # bounded temporaries at 16M frames
    for lo in range(0, N, step):
        hi = min(N, lo + step)
        comp = torch.randint(0, L, (hi - lo,), generator=g, device=device)
        Z[lo:hi] = mu[comp] + torch.randn((hi - lo, D), generator=g, device=device, dtype=torch.float32)
    vid = (torch.arange(N, device=device) // FRAMES_PER_VIDEO).to(torch.int32)
    V = int((N + FRAMES_PER_VIDEO - 1) // FRAMES_PER_VIDEO)
    return Z, vid, V


# ---------------------------------------------------------------------------------------------
# CPU baseline (runs BEFORE this process touches the GPU, so forking worker processes is safe)
def _oracle_iteration(args):
    D, bf, levels, sample_n, faithful, seed = args
    from oracle.vi_oracle import OracleVarInf, make_latent_dict, gaussian_mixture
    Z = gaussian_mixture(sample_n, D, bf ** levels, seed=seed, spread=4.0)
    lc = make_latent_dict(Z, [FRAMES_PER_VIDEO])
    np.random.seed(13)
    vi = OracleVarInf(latent_code_size=D, branching=bf, levels=levels, faithful_loops=faithful)
    Zm = vi.get_matrix_from_dict(lc)
    vi.build_initial_tree()
    vi.build_video_index()
    vi.one_hot_init(Zm)
    t0 = time.perf_counter()
    vi.compute_sigma_alpha(Zm)
    vi.compute_gamma()
    vi.compute_phi(Zm)
    return time.perf_counter() - t0, len(lc)


def cpu_baseline(D, bf, levels, sample_n):
    """The reference-faithful CPU port (oracle with the reference's per-frame loop and list.index
    lookups, single thread like the reference) on a bounded sample; beside it the vectorised
    NumPy oracle on one core and on every core of this box's share (one process per core, each
    with its own shard of `sample_n` frames, as a sample-sharded CPU run would)."""
    import multiprocessing as mp
    t_faithful, nvid = _oracle_iteration((D, bf, levels, sample_n, True, 13))
    t_vec, _ = _oracle_iteration((D, bf, levels, sample_n, False, 13))
    cores = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or min(16, os.cpu_count() or 1)
    cores = max(1, min(cores, os.cpu_count() or 1))
    t_all = None
    try:
        ctx = mp.get_context("fork")
        with ctx.Pool(cores) as pool:
            t0 = time.perf_counter()
            pool.map(_oracle_iteration, [(D, bf, levels, sample_n, False, 13 + i) for i in range(cores)])
            t_all = time.perf_counter() - t0            # includes each worker's set-up: a lower bound on the rate
    except Exception:
        t_all = None
    return {
        "value": sample_n / t_faithful, "unit": "samples/s", "cores": 1, "kind": "port",
        "sample": "first %d frames (%d videos) of the workload, one M+gamma+E iteration of "
                  "oracle/vi_oracle.py with the reference's per-frame loop" % (sample_n, nvid),
        "vectorized_value": sample_n / t_vec,
        "vectorized_note": "same oracle with the per-frame loop vectorised (NumPy), ONE core",
        "vectorized_all_cores_value": (cores * sample_n / t_all) if t_all else None,
        "vectorized_all_cores": cores,
        "vectorized_all_cores_note": "%d processes x %d frames each (sample-sharded), wall time incl. per-process set-up"
                                     % (cores, sample_n),
    }


# ---------------------------------------------------------------------------------------------
def launch_children(args):
    """`python bench.py --gpus N` from a bare shell: start N fresh rank processes (this parent has not imported
    torch and never touches a GPU), watch ALL of them, and relay rank 0's JSON line.

    Fail-safe: the parent polls every child; as soon as one exits non-zero (a rank that dies before or inside
    init_process_group would otherwise leave the others waiting for the rendezvous until the caller's timeout)
    the rest are terminated, every rank's stderr is kept in its own file and the tails of the failed ranks' are
    shown, and the parent exits non-zero.  BNPVI_BENCH_LOGDIR names the directory of the per-rank logs
    (default: a fresh temporary directory, removed when every rank succeeded)."""
    import shutil
    import socket
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    logdir = os.environ.get("BNPVI_BENCH_LOGDIR")
    own_logdir = not logdir
    if own_logdir:
        logdir = tempfile.mkdtemp(prefix="bnpvi_bench_")
    os.makedirs(logdir, exist_ok=True)
    procs, files = [], []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = open(os.path.join(logdir, "rank%d.out" % r), "w+")
        err = open(os.path.join(logdir, "rank%d.err" % r), "w+")
        files.append((out, err))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out, stderr=err, text=True))
    deadline = time.time() + float(os.environ.get("BNPVI_BENCH_TIMEOUT", "3000"))
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = "rank %d exited with code %d" % (bad[0], codes[bad[0]])
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            failed = "timeout"
            break
        time.sleep(0.05)
    if failed:
        for p in procs:                       # fresh children of this parent only, by their exact PIDs
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    codes = [p.returncode for p in procs]
    files[0][0].seek(0)
    sys.stdout.write(files[0][0].read())
    sys.stdout.flush()
    if failed:
        sys.stderr.write("bench.py: %s; the other ranks were terminated (exit codes %s); per-rank logs in %s\n"
                         % (failed, codes, logdir))
        for r, (_o, e) in enumerate(files):
            if codes[r] not in (0, -15):
                e.seek(0)
                tail = e.read().strip().splitlines()[-12:]
                sys.stderr.write("---- rank %d stderr (tail) ----\n%s\n" % (r, "\n".join(tail)))
    for o, e in files:
        o.close()
        e.close()
    if own_logdir and not failed:
        shutil.rmtree(logdir, ignore_errors=True)
    return 1 if failed else 0


class Runner(object):
    """One rank's timing of one workload."""

    def __init__(self, torch, dist, device, local_rank, rank, world, group, rehearsal):
        self.torch, self.dist, self.device = torch, dist, device
        self.local_rank, self.rank, self.world, self.group, self.rehearsal = local_rank, rank, world, group, rehearsal

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def run(self, cfg, N, steps, warmup, want_estats):
        import ctypes as C
        from bnp_vae_amd import _lib
        from bnp_vae_amd.engine import DeviceVI
        torch = self.torch
        _n, D, bf, levels = CONFIGS[cfg]
        N = (N // FRAMES_PER_VIDEO) * FRAMES_PER_VIDEO or N     # shards cut on video boundaries
        parent, is_leaf = regular_tree(bf, levels)
        K, L = len(parent), int(is_leaf.sum())
        dev = DeviceVI(device=self.local_rank, group=self.group)
        Z, vid, V = make_latents(torch, self.device, N, D, L, self.rank)
        dev.set_latents(Z, vid, V)
        del Z
        alpha0 = np.random.RandomState(13).normal(size=(K, D))      # Node.__init__ draw, src/node.py:12
        dev.set_tree(parent, is_leaf, alpha0, np.ones(K))
        dev.default_gamma()
        dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)            # phi_in = one-hot nearest leaf

        def step():
            # incl. the all-reduce of the split / merge statistics and, for trees of more than 100 leaves, the
            # gated S2 pass (src/var_inf.py:217-220)
            dev.iteration(want_estats=want_estats, n_total=N * self.world)

        for _ in range(warmup):
            step()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.barrier()
        elapsed = time.perf_counter() - t0

        # Kernel durations for the roofline: a second, UNTIMED run of the same steps with the library's
        # HIP-event pairs around the two data-pass kernels (recorded on the launch stream; the event
        # records cost a few microseconds per launch, so they stay out of the timed region above).
        def read(which):
            ms, cnt = C.c_double(), C.c_int()
            _lib.check(_lib.lib.bnpvi_profile_read(dev.ctx, which, C.byref(ms), C.byref(cnt)))
            return (ms.value / cnt.value) if cnt.value else float("nan")

        _lib.check(_lib.lib.bnpvi_profile_enable(dev.ctx, 1))
        for _ in range(steps):
            step()
        self.barrier()
        e_ms, m_ms = read(0), read(1)
        _lib.check(_lib.lib.bnpvi_profile_enable(dev.ctx, 0))

        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if self.rehearsal else self.device)
        if self.world > 1:
            self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

        # sanity of the state the timed loop left behind (cheap, outside the timed region)
        rows = dev.phi[: min(N, 4096), :L].sum(dim=1)
        assert torch.allclose(rows, torch.ones_like(rows), atol=1e-4), "phi rows do not sum to 1"
        dev.close()
        del dev
        torch.cuda.empty_cache()

        ms_per_step = elapsed / steps * 1e3
        kern_bytes = N * 4.0 * (D + L)            # per launch: E reads Z + writes phi; M reads Z + phi
        dom, dom_ms = ("estep", e_ms) if e_ms >= m_ms else ("mstep_stats", m_ms)
        achieved = kern_bytes / (dom_ms * 1e-3) / 1e9
        traffic, traffic_at = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.isfile(pmc):
            try:
                rec = json.load(open(pmc)).get("%s:%s" % (cfg, dom))
                if rec:      # counters of a streaming kernel: linear in the frame count
                    traffic = rec["hbm_bytes_per_launch"] * (N / float(rec["N"]))
                    traffic_at = rec["N"]
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": "profiles/pmc_traffic.json (rocprofv3 PMC passes of an earlier run, scaled "
                                  "linearly in the frame count; not measured in this run)" if traffic else None,
                "traffic_measured_at_N": traffic_at,
                "algorithmic_bytes_per_launch": kern_bytes, "avg_launch_ms": dom_ms,
                "estep_ms": e_ms, "mstep_stats_ms": m_ms,
                "iteration_frac": N * 8.0 * (D + L) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if L * D >= 256 * 1024:
            # large trees (config 4): both data passes are matrix-pipe problems.  Algorithmic flops per launch =
            # the one dense contraction of the kernel, 2 L D per frame (E: z . alpha^T, M: R^T Z); the kernels
            # issue 3 products per algorithmic one (f16 hi/lo split; the M-step's wide kernel takes that split where
            # the padded tree has >= 512 x 256 leaf x dim entries, else the exact bf16 x 3 split with 6) to keep
            # fp32 accuracy.  Peak: dense bf16 / f16 MFMA, MI355X_MICROARCH.md.  (The E-step timer covers the fragment
            # pre-pass and the logits kernel -- pass A -- of the streamed regime only.)
            flops = 2.0 * N * L * D
            Dp = 32 if D <= 32 else 64 if D <= 64 else 128 if D <= 128 else 256
            m_issued = 3.0 if Dp >= 128 and (-(-L // 32) * 32) * Dp >= 512 * 256 else 6.0
            issued = {"estep": 3.0, "mstep_stats": m_issued}[dom]
            roof.update({"bound": "mfma", "achieved": flops / (dom_ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": flops / (dom_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                         "algorithmic_flops_per_launch": flops, "issued_products_per_algorithmic": issued,
                         "issued_frac": issued * flops / (dom_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                         "hbm_frac": achieved / HBM_PEAK_GBS,
                         "estep_issued_tflops": 3.0 * flops / (e_ms * 1e-3) / 1e12,
                         "mstep_issued_products_per_algorithmic": m_issued,
                         "mstep_issued_tflops": m_issued * flops / (m_ms * 1e-3) / 1e12})
        return {
            "N": N, "D": D, "bf": bf, "K": K, "L": L, "V": V, "ms_per_step": ms_per_step,
            "value": N * self.world / (elapsed / steps),
            "allreduce_bytes_per_step": (L * (1 + D) * 8 + (L * 2 * 8 if want_estats else 0)) if self.world > 1 else 0,
            "ranks_seen": int(self.dist.get_world_size()) if self.world > 1 else 1,
            "roofline": roof,
        }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="config2", choices=sorted(CONFIGS))
    ap.add_argument("--frames", type=int, default=0,
                    help="frames per rank (override; not `--n`: torch.distributed.run reads that as its own)")
    ap.add_argument("--no-estats", action="store_true",
                    help="skip the split/merge statistics epilogue of the E-step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config3", action="store_true", help="skip the config3_strong sub-record")
    ap.add_argument("--cpu-sample", type=int, default=60_000)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_children(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    if os.environ.get("BNPVI_BENCH_FAIL_RANK") == str(rank) and world > 1:
        raise RuntimeError("rank %d: failure injected by BNPVI_BENCH_FAIL_RANK (watchdog rehearsal)" % rank)

    _n, D, bf, levels = CONFIGS[args.config]
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(D, bf, levels, args.cpu_sample)        # before torch / the GPU: it forks workers

    import torch
    import torch.distributed as dist

    # Rehearsal of the N > 1 control flow on one GPU (never a measured configuration; only with
    # BNPVI_BENCH_REHEARSAL=1): every rank on cuda:0, gloo with host-staged collectives.  Its record says
    # n_gpus = 1 and value = null; the rate the ranks reached together is kept as `rehearsal_value`.
    rehearsal = world > 1 and os.environ.get("BNPVI_BENCH_REHEARSAL") == "1"
    if world > 1 and not rehearsal and torch.cuda.device_count() < world:
        # every rank says so (each has its own stderr file under the self-launching parent, which shows the tail
        # of the first rank to fail)
        print("bench.py: --gpus %d but only %d GPU(s) visible (BNPVI_BENCH_REHEARSAL=1 rehearses the N > 1 "
              "control flow on one GPU over gloo; it is never a measurement)" % (world, torch.cuda.device_count()),
              file=sys.stderr)
        sys.exit(3)
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    backend = "none"
    if world > 1:
        backend = "gloo" if rehearsal else "nccl"
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        group = dist.group.WORLD
    elif os.environ.get("BNPVI_DIST_FORCE_COLLECTIVES") == "1":
        # measurement aid for the N > 1 budget: ONE rank in an RCCL group, every collective of the step
        # issued anyway (bnp_vae_amd/dist.py) -- what the two all-reduces cost a step before any link time
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        group = dist.group.WORLD
        backend = "nccl (one rank, collectives forced)"

    run = Runner(torch, dist, device, local_rank, rank, world, group, rehearsal)
    want_estats = not args.no_estats
    N = CONFIGS[args.config][0]
    scaling = "weak"
    if args.config in ("config3", "config4"):
        N = N // world                       # fixed total problem: strong scaling
        scaling = "strong"
    if args.frames:
        N = args.frames
    head = run.run(args.config, N, args.steps, args.warmup, want_estats)

    strong = None
    if args.config == "config2" and not args.no_config3 and not args.frames:
        n3 = CONFIGS["config3"][0] // world
        if rehearsal:
            n3 = min(n3, 262_144)            # all ranks share one GPU's memory in a rehearsal
        s3 = max(3, min(args.steps, 10))
        rec = run.run("config3", n3, s3, min(args.warmup, 2), want_estats)
        strong = {"workload": "config3: %d frames in total over %d rank(s) (%d per rank), D=%d, K=%d, L=%d"
                              % (rec["N"] * world, world, rec["N"], rec["D"], rec["K"], rec["L"]),
                  "frames_total": rec["N"] * world, "frames_per_rank": rec["N"], "steps": s3, "scaling": "strong",
                  "ms_per_step": rec["ms_per_step"], "value": rec["value"], "unit": "samples/s",
                  "allreduce_bytes_per_step": rec["allreduce_bytes_per_step"], "ranks_seen": rec["ranks_seen"],
                  "collective_backend": backend, "rehearsal_on_one_gpu": bool(rehearsal),
                  "roofline": rec["roofline"]}

    if rank == 0:
        out = {
            "metric": "latent samples/sec per VI E+M iteration (whole node)",
            "value": None if rehearsal else head["value"], "unit": "samples/s",
            "n_gpus": 1 if rehearsal else world, "ranks": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: N=%d frames/rank, D=%d, %d-ary tree K=%d nodes L=%d leaves, "
                                   "%d-frame videos (V=%d), Gaussian-mixture latents, estats=%s"
                                   % (args.config, head["N"], head["D"], head["bf"], head["K"], head["L"],
                                      FRAMES_PER_VIDEO, head["V"], want_estats),
                       "parallelism": "sample-sharded x%d; per step 1 fp64 all-reduce of L x (1+D) leaf statistics"
                                      " + 1 of L x 2 split/merge statistics" % world,
                       "collective_backend": backend, "rehearsal_on_one_gpu": bool(rehearsal),
                       "allreduce_bytes_per_step": head["allreduce_bytes_per_step"],
                       "ranks_seen": head["ranks_seen"]},
            "roofline": head["roofline"],
        }
        if rehearsal:
            out["rehearsal_value"] = head["value"]
        if strong is not None:
            if rehearsal:
                strong["rehearsal_value"], strong["value"] = strong["value"], None
            out["config3_strong"] = strong
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
        sys.stdout.flush()
    if group is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
