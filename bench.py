#!/usr/bin/env python
"""bench.py - train samples/s of the VAE hot path on MI355X (driver contract in the task).

N=1 : BASELINE.json configs[1]: notebook MNIST Conv2d VAE, batch 4096, fp32, BCE ELBO, Adam ("weak": the per-GPU
      batch is the fixed quantity of this point).  The same line carries `strong_scaling_ref`: one GPU at global
      batch 65536, the N=1 point of the N>1 curve.
N>1 : configs[3]: same model, data-parallel, global batch 65536 split evenly ("strong"), gradient all-reduce SUM +
      synchronised BatchNorm statistics over RCCL.  `--batch B` fixes the per-GPU batch instead ("weak"),
      `--global-batch G` the global one ("strong"), at any N.
One step = eps draw + forward + ELBO + backward + Adam on a synthetic batch resident in HBM.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (child processes
through torch.distributed.run on 127.0.0.1, before this process touches a GPU) and relays rank 0's JSON line.
"""
import argparse
import csv
import ctypes
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic figures per sample-step for the MNIST model
M_FLOP_PER_SAMPLE = 1.89e6
M_BYTES_PER_SAMPLE = 178e3
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (weak scaling: fixed as N grows)")
    ap.add_argument("--global-batch", type=int, default=0, help="global batch, split evenly over the GPUs (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong-ref", action="store_true", help="skip the N=1 measurement at global batch 65536")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--dp-graph", action="store_true",
                    help="N > 1 on RCCL: capture the data-parallel step (collectives included) into one hipGraph")
    ap.add_argument("--roofline-iters", type=int, default=20)
    ap.add_argument("--rank-timeout", type=int, default=900,
                    help="--gpus N > 1 started by this script: seconds after which the ranks are killed (0: never)")
    ap.add_argument("--dp-overlap", action="store_true",
                    help="N > 1 on RCCL: two communicators, the decoder's gradient bucket all-reduced on a side stream while "
                         "the encoder's backward runs.  Off by default: no box with more than one GPU was in reach of the "
                         "build, so the default is the arrangement with no two collectives in flight together")
    ap.add_argument("--dp-single-comm", action="store_true",
                    help="(the default since round 3; kept so that older command lines still parse)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="compute dtype of the matrix kernels: f32 (the headline) or bf16 operands with fp32 accumulate "
                         "and fp32 storage (vae_set_compute_dtype; BASELINE configs[1] 'fp32 then bf16')")
    ap.add_argument("--model", default="mnist", choices=["mnist", "wide", "multiscale", "tcvae"],
                    help="mnist = the contract workload (BASELINE configs[1]/[3]); multiscale / tcvae = configs[2] "
                         "(dataGen-shaped windows, MSE ELBO); wide = configs[4]")
    args = ap.parse_args()
    if args.batch and args.global_batch:
        ap.error("--batch (per GPU) and --global-batch exclude each other")
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    return args


def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks as children (never an exec: this process may not be
    replaced once anything touched the GPU, and it has not touched it yet) and exit with their status.  With fewer
    visible GPUs than ranks the ranks share the cards and rendezvous over gloo (a rehearsal of the N>1 code path)."""
    import torch  # (device_count() does not initialise the GPU)
    ngpu = torch.cuda.device_count()
    env = dict(os.environ)
    if ngpu < args.gpus and "VAE_BENCH_BACKEND" not in env:
        env["VAE_BENCH_BACKEND"] = "gloo"
        print(f"bench.py: {ngpu} GPU(s) visible for {args.gpus} ranks: ranks share the card(s), backend gloo (rehearsal)",
              file=sys.stderr, flush=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import shutil
    import signal
    import tempfile
    logdir = tempfile.mkdtemp(prefix="vae_bench_ranks_")
    # stderr of every rank goes to a file of its own (stdout stays ours: rank 0's JSON line goes straight out)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), "--redirects", "2", "--log-dir", logdir,
           os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)

    def tails(n):
        for root, _, files in sorted(os.walk(logdir)):
            for f in files:
                if f == "stderr.log":
                    with open(os.path.join(root, f), errors="replace") as fh:
                        lines = fh.read().splitlines()[-n:]
                    if lines:
                        print(f"---- rank {os.path.basename(root)} stderr (last {len(lines)} lines)", file=sys.stderr)
                        print("\n".join(lines), file=sys.stderr, flush=True)
    try:
        rc = proc.wait(timeout=args.rank_timeout if args.rank_timeout > 0 else None)
    except subprocess.TimeoutExpired:
        # a rank that deadlocks in a collective never returns: this process has not touched a GPU, so it can end the
        # ranks (their whole process group) and say what each was doing
        print(f"bench.py: the ranks did not finish within --rank-timeout {args.rank_timeout}s: killing them",
              file=sys.stderr, flush=True)
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.wait()
        tails(30)
        shutil.rmtree(logdir, ignore_errors=True)
        sys.exit(124)
    if rc != 0:
        print(f"bench.py: a rank failed (launcher exit status {rc})", file=sys.stderr, flush=True)
        tails(30)
    shutil.rmtree(logdir, ignore_errors=True)
    sys.exit(rc if 0 <= rc < 256 else 1)


def make_model(name, B, dev, gen):
    import torch
    import vae_amd
    if name in ("mnist", "wide"):
        if name == "mnist":
            model = vae_amd.mnist_vae.VariationalAutoencoder(15)
            workload = "MNIST 28x28 Conv2d VAE (notebook model, 103,343 params), BCE ELBO, Adam"
        else:  # BASELINE configs[4]; hidden width kept at 100 (SURVEY 8d asks to state the choice)
            model = vae_amd.mnist_vae.VariationalAutoencoder(256, channels=(32, 64, 128), hidden=100)
            workload = "widened MNIST Conv2d VAE (channels 32/64/128, latent 256, hidden 100), BCE ELBO, Adam"
        x = torch.rand(B, 1, 28, 28, device=dev, generator=gen)
    else:
        if name == "multiscale":
            model = vae_amd.multiscale_vae.vae(3, 60, 15)
            workload = "multiscale_vae.vae(3, 60, 15) (12,255 params), MSE ELBO, Adam"
        else:
            model = vae_amd.VariationalAutoencoder(input_size=3, hidden_size=30, num_layers=3, latent_dims=15,
                                                   v_encoder=vae_amd.Encoders.LongShort_TCVAE_Encoder,
                                                   v_decoder=vae_amd.Decoders.LongShort_TCVAE_Decoder, L=60, slope=0.2,
                                                   first_kernel=21)
            workload = "TCVAE C=3 N=3 latent 15 L=60 first_kernel 21 (9,549 params), MSE ELBO, Adam"
        x = 50 + 10 * torch.randn(B, 3, 60, device=dev, generator=gen)  # dataGen-shaped windows (SURVEY 8d cfg 2)
    return model, x, workload


# ------------------------------------------------------------------------------------------- CPU baseline
def _cpu_leg(name, B, seconds):
    """The CPU oracle (oracle/vae_oracle.py, a port of the reference train loop: zero_grad / forward / criterion /
    backward / Adam / loss.item() per step) on the host cores for about `seconds`."""
    import torch
    from oracle import vae_oracle as O
    torch.manual_seed(0)
    model, x, _ = make_model(name, B, "cpu", torch.Generator().manual_seed(0))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    family, kw = {"mnist": ("mnist", {}), "wide": ("mnist", {}), "multiscale": ("multiscale", {}),
                  "tcvae": ("tcvae", {"num_layers": 3})}[name]
    lat = model.latent_total
    opt = O.Adam(sd, lr=1e-3)
    g = torch.Generator().manual_seed(1)
    O.train_step(family, sd, opt, x, torch.randn(B, lat, generator=g), **kw)  # warm-up
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        O.train_step(family, sd, opt, x, torch.randn(B, lat, generator=g), **kw)
        n += 1
    dt = time.perf_counter() - t0
    return n, dt


def cpu_baseline(name, B):
    """Bounded sample of the same workload on the host cores (kind "port": the oracle restates the reference's
    loop), plus, for the MNIST model, the reference's own batch size 128 (BASELINE configs[0], notebook loaders)."""
    import torch
    cores = torch.get_num_threads()
    n, dt = _cpu_leg(name, B, 12.0)
    out = {"value": n * B / dt, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": f"{n} steps of batch {B} ({name} VAE fwd + ELBO + bwd + Adam) in {dt:.1f}s, torch CPU threads={cores}"}
    if name == "mnist":
        n2, dt2 = _cpu_leg(name, 128, 6.0)
        out["reference_batch_128"] = {"value": n2 * 128 / dt2, "unit": "samples/s", "cores": cores,
                                      "sample": f"{n2} steps of batch 128 (configs[0], the notebook's loader batch) in {dt2:.1f}s"}
    return out


# ------------------------------------------------------------------------------------------- roofline
def pmc_traffic(model_name, kernel, B):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/rNN_pmc_summary.csv
    for the MNIST model, profiles/rNN_<model>_pmc_summary.csv for the others; made by tools/pmc_summary.py from separate
    --pmc passes at batch 4096, corrected as DESIGN.md states).  (None, reason) when there is none for this run."""
    if B != 4096:
        return None, f"PMC summaries were collected at batch 4096, this run uses {B}"
    pat = "r[0-9][0-9]_pmc_summary.csv" if model_name == "mnist" else f"r[0-9][0-9]_{model_name}_pmc_summary.csv"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
    if not files:
        return None, f"no profiles/{pat}"
    with open(files[-1], newline="") as f:
        for row in csv.reader(f):
            if row and row[0].replace("void ", "").startswith(kernel):
                return int(float(row[-1])), "profiles/" + os.path.basename(files[-1])
    return None, f"kernel {kernel} not in profiles/{os.path.basename(files[-1])}"


def roofline_step(stepper, iters, model_name):
    """Roofline of the dominant kernel FAMILY of the whole train step, whichever it is: every launch of forward +
    backward + Adam is timed live with a HIP event behind it on the launch stream (vae_time_step; eager launches of
    real steps, after the timed region), launches are grouped by kernel name as rocprofv3's --stats groups them, and
    the family with the largest total time per step is reported.  achieved = the family's algorithmic bytes per step
    (declared by each launch site from the op geometry: inputs read once, outputs written once, weights included) /
    its time per step; matrix-bound families (wide.h) report flops against the fp32 MFMA peak instead."""
    prog = stepper.prog
    ext = [stepper.x.view(stepper.B, -1), stepper.eps]
    acc = None
    for it in range(iters + 2):
        rows = prog.time_step(stepper.work, stepper.flat, stepper.grads, stepper.sflat, ext, stepper.m, stepper.v,
                              stepper.hyper, stepper.tick)
        if it >= 2:
            if acc is None:
                acc = [[n, 0.0, b, f] for n, _, b, f in rows]
            assert len(rows) == len(acc)
            for a, (_, ms, _, _) in zip(acc, rows):
                a[1] += ms
    fams = {}
    for name, ms, b, f in acc:
        fam = name.split(" (phase")[0]
        d = fams.setdefault(fam, {"kernel": fam, "launches_per_step": 0, "us_per_step": 0.0, "bytes_per_step": 0.0,
                                  "flops_per_step": 0.0, "per_launch_us": []})
        d["launches_per_step"] += 1
        d["us_per_step"] += ms / iters * 1e3
        d["bytes_per_step"] += b
        d["flops_per_step"] += f
        d["per_launch_us"].append(round(ms / iters * 1e3, 2))
    step_us = sum(d["us_per_step"] for d in fams.values())
    order = sorted(fams.values(), key=lambda d: -d["us_per_step"])
    top = order[0]
    n = top["launches_per_step"]
    if top["flops_per_step"] > 0 and top["kernel"].startswith("wide_"):
        achieved = top["flops_per_step"] / (top["us_per_step"] * 1e-6) / 1e12
        out = {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
               "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_launch": top["flops_per_step"] / n}
    else:
        achieved = top["bytes_per_step"] / (top["us_per_step"] * 1e-6) / 1e9
        out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}
    traffic, src = pmc_traffic(model_name, top["kernel"], stepper.B)
    out.update({"traffic": traffic, "traffic_source": src, "kernel": top["kernel"], "launches_per_step": n,
                "avg_launch_us": top["us_per_step"] / n, "per_launch_us": top["per_launch_us"],
                "algorithmic_bytes_per_launch": top["bytes_per_step"] / n,
                "share_of_step": top["us_per_step"] / step_us,
                "step_launches": len(acc), "step_sum_of_launches_us": step_us,
                "families": [{"kernel": d["kernel"], "launches": d["launches_per_step"], "us_per_step": round(d["us_per_step"], 2),
                              "GBps": round(d["bytes_per_step"] / (d["us_per_step"] * 1e-6) / 1e9, 1)} for d in order[:int(os.environ.get("VAE_BENCH_FAMILIES", "16"))]],
                "timing": f"HIP event behind every launch on the launch stream, {iters} eager train steps (vae_time_step)"})
    return out


# ------------------------------------------------------------------------------------------- main
def timed_run(stepper, steps, warmup, barrier):
    import torch
    for _ in range(warmup):
        stepper.step()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        stepper.step()
    e1.record()
    barrier()
    dt = time.perf_counter() - t0
    return dt, e0.elapsed_time(e1) / steps


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # (does not return)

    import torch
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    out = sys.stdout
    if world > 1:
        # stdout carries ONE line, rank 0's JSON: whatever else writes to file descriptor 1 in a rank (gloo prints its
        # connection messages there, from C++) goes to stderr instead
        out = os.fdopen(os.dup(1), "w")
        sys.stdout.flush()
        os.dup2(2, 1)
    watchdog = None
    if world > 1 and args.rank_timeout > 0:
        # a rank started by somebody else's launcher (torchrun) has no parent of ours to end it: a collective that never
        # completes must not hang the job for good
        import threading

        def _expired():
            print(f"bench.py rank {rank}: not finished within --rank-timeout {args.rank_timeout}s (a collective that "
                  "never completed?): exiting; try VAE_BENCH_BACKEND=gloo (collectives staged through the host)",
                  file=sys.stderr, flush=True)
            os._exit(3)
        watchdog = threading.Timer(args.rank_timeout, _expired)
        watchdog.daemon = True
        watchdog.start()
    ngpu = max(torch.cuda.device_count(), 1)
    backend = os.environ.get("VAE_BENCH_BACKEND") or ("nccl" if ngpu >= world else "gloo")
    local_rank %= ngpu
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import vae_amd
    from vae_amd.train import FusedTrainStep
    vae_amd.set_compute_dtype(args.dtype)

    if args.batch:
        B, scaling = args.batch, "weak"
    elif args.global_batch:
        B, scaling = args.global_batch // world, "strong"
    elif world == 1:
        B, scaling = 4096, "weak"           # configs[1]
    else:
        B, scaling = 65536 // world, "strong"  # configs[3]
    if B < 2:
        raise SystemExit(f"per-GPU batch {B}: too small")
    if args.global_batch and B * world != args.global_batch:
        raise SystemExit(f"--global-batch {args.global_batch} does not split evenly over {world} GPUs")

    torch.manual_seed(0)
    g = torch.Generator(device=dev).manual_seed(rank)
    model, x, workload = make_model(args.model, B, dev, g)
    model = model.to(dev).train()
    opt = vae_amd.optim.Adam(model.parameters(), lr=1e-3)
    if world == 1:
        stepper = FusedTrainStep(model, opt, B, dev, use_graph=not args.no_graph)
    else:
        from vae_amd.parallel import DataParallelTrainStep
        stepper = DataParallelTrainStep(model, opt, B, dev, use_graph=args.dp_graph and backend == "nccl",
                                        single_comm=not args.dp_overlap)
    stepper.step(x)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dt, dev_ms = timed_run(stepper, args.steps, args.warmup, barrier)
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(stepper.loss.item())
    if (not (loss == loss) or loss <= 0) and not os.environ.get("VAE_BENCH_NOCHECK"):  # (diagnostic kernel builds only)
        raise SystemExit(f"loss is {loss}: the step is not training")

    roof = None
    if rank == 0:
        # (N > 1: this rank's launches only, BatchNorm on local statistics, no collectives; the step it takes is local too,
        # which no longer matters behind the timed region)
        roof = roofline_step(stepper, args.roofline_iters, args.model)
    barrier()

    strong_ref = None
    if rank == 0 and world == 1 and scaling == "weak" and not args.batch and not args.no_strong_ref and args.model == "mnist":
        # the N=1 point of the N>1 curve (configs[3]: global batch 65536), so scaling efficiency has a denominator
        G = 65536
        del stepper
        torch.cuda.empty_cache()
        m2, x2, _ = make_model(args.model, G, dev, g)
        m2 = m2.to(dev).train()
        st2 = FusedTrainStep(m2, vae_amd.optim.Adam(m2.parameters(), lr=1e-3), G, dev, use_graph=not args.no_graph)
        st2.step(x2)
        n2 = max(5, min(args.steps, 20))
        dt2, dev2 = timed_run(st2, n2, 3, barrier)
        strong_ref = {"n_gpus": 1, "global_batch": G, "value": G * n2 / dt2, "unit": "samples/s", "ms_per_step": dt2 / n2 * 1e3,
                      "steps": n2, "note": "one GPU at the global batch of the N>1 (strong-scaling) runs"}
        del st2, m2, x2

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        if world == 1:
            par = "single GPU, whole step as one hipGraph" if not args.no_graph else "single GPU, eager launches"
        else:
            native = getattr(stepper, "native", False)
            how = ("RCCL through torch.distributed from the SyncBN callback" if not native else
                   "RCCL enqueued by the library, no host callback, " +
                   ("one communicator on the compute stream, the gradient in one all-reduce behind the backward pass"
                    if getattr(stepper, "single_comm", False) else
                   "two communicators: the decoder's gradient bucket reduced on a side stream beside the encoder's backward"))
            coll = ((how + (", whole step captured as one hipGraph" if stepper.use_graph else ", eager launches"))
                    if backend == "nccl" else f"{backend} (rehearsal: ranks share {ngpu} GPU(s), collectives staged through the host)")
            par = f"dp{world}: batch shards, SyncBN all-gather per BatchNorm, grad all-reduce SUM ({coll})"
        line = {
            "metric": "train samples/sec (whole node)", "value": value, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{workload}, batch {B} per GPU, global batch {B * world}",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": par, "backend": backend if world > 1 else None},
            "loss_per_sample": loss / B,
            "device_ms_per_step": dev_ms,
        }
        if args.dtype == "bf16":
            line["dtype_note"] = "bf16 MFMA operands (rounded as they leave LDS), fp32 accumulate, fp32 storage, statistics, ELBO and Adam"
        if args.model == "mnist":
            # whole-step figure against the materialised-activation traffic model of SURVEY 8d (178 KB/sample)
            line["step_hbm_model_GBps"] = M_BYTES_PER_SAMPLE * B / (dev_ms * 1e-3) / 1e9
        if roof is not None:
            line["roofline"] = roof
        if strong_ref is not None:
            line["strong_scaling_ref"] = strong_ref
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.model, B)
        print(json.dumps(line), file=out, flush=True)
    if watchdog is not None:
        watchdog.cancel()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
