#!/usr/bin/env python
"""bench.py - train samples/s of the VAE hot path on MI355X (driver contract in the task).

N=1 : BASELINE.json configs[1]: notebook MNIST Conv2d VAE, batch 4096, fp32, BCE ELBO, Adam.
N>1 : configs[3]: same model, data-parallel, global batch 65536 split evenly, gradient
      all-reduce SUM + synchronised BatchNorm statistics over RCCL.
One step = eps draw + forward + ELBO + backward + Adam on a synthetic batch resident in HBM.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic figures per sample-step for the MNIST model
M_FLOP_PER_SAMPLE = 1.89e6
M_BYTES_PER_SAMPLE = 178e3
HBM_PEAK_GBS = 8000.0
# HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (profiles/r01_pmc_*.csv, corrected
# as DESIGN.md states), measured at batch 4096 per launch (None for any other batch)
PMC_TRAFFIC_BYTES_PER_LAUNCH = 32271264  # (2*7288.1 + 16938.7) KiB, profiles/r01_pmc_summary.csv


def cpu_baseline(seconds=15.0, B=4096):
    """CPU oracle (oracle/vae_oracle.py, a port of the reference train loop) on the host cores:
    zero_grad/forward/criterion/backward/Adam/loss.item() per step, same workload, bounded sample."""
    from oracle import vae_oracle as O
    import vae_amd
    torch.manual_seed(0)
    m = vae_amd.mnist_vae.VariationalAutoencoder(15)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    opt = O.Adam(sd, lr=1e-3)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 1, 28, 28, generator=g)
    cores = torch.get_num_threads()
    O.train_step("mnist", sd, opt, x, torch.randn(B, 15, generator=g))  # warm-up
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        O.train_step("mnist", sd, opt, x, torch.randn(B, 15, generator=g))
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n * B / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of batch {B} (MNIST VAE fwd+BCE ELBO+bwd+Adam) in {dt:.1f}s, torch CPU threads={cores}"}


def dominant_kernel_roofline(stepper, iters):
    """Roofline of the dominant kernel, igemm_gather_k<1, 2, false, 0> (the forward implicit-GEMM gather on MFMA
    tiles of the conv / transposed-conv layers with <= 16 output channels that are not thin enough for the
    vector-ALU variant: conv2, convT1, convT2 of this model; the largest line of profiles/*kernel_stats.csv).  Its launches are timed live with HIP events on the launch
    stream (vae_time_forward puts an event between consecutive ops; a fused conv op is exactly one launch).
    achieved = algorithmic bytes per launch / average launch duration; algorithmic bytes of a launch =
    4 * B * (Cin*H*W + Cout*OH*OW) + 4 * weights + bias (input read once, output written once: DESIGN.md)."""
    from vae_amd import _lib as L
    prog = stepper.prog
    ext = [stepper.x.view(stepper.B, -1), stepper.eps]
    sel = []
    for i in range(prog.n_ops):
        op = prog.ops[i]
        if op.kind in (L.OP_CONV, L.OP_CONVT) and (op.flags & L.F_FUSED) and \
                L.lib().vae_conv_forward_variant(ctypes.byref(op), stepper.B) == 100:
            cin, H, W, cout, OH, OW, KH, KW = [op.d[j] for j in range(8)]
            nbytes = 4 * (stepper.B * (cin * H * W + cout * OH * OW) + cin * cout * KH * KW + cout)
            sel.append((i, nbytes))
    if not sel:
        return None
    tot_ms = [0.0] * len(sel)
    for it in range(iters + 2):
        ms = prog.time_forward(stepper.work, stepper.flat, stepper.sflat, ext, True)
        if it >= 2:
            for k, (i, _) in enumerate(sel):
                tot_ms[k] += ms[i]
    avg_us = sum(tot_ms) / iters / len(sel) * 1e3
    avg_bytes = sum(b for _, b in sel) / len(sel)
    achieved = avg_bytes / (avg_us * 1e-6) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": PMC_TRAFFIC_BYTES_PER_LAUNCH if stepper.B == 4096 else None, "kernel": "igemm_gather_k<1, 2, false, 0>",
            "launches_per_step": len(sel), "avg_launch_us": avg_us, "algorithmic_bytes_per_launch": avg_bytes,
            "per_launch_us": [t / iters * 1e3 for t in tot_ms],
            "timing": f"HIP events on the launch stream, {iters} eager forward passes"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--roofline-iters", type=int, default=20)
    ap.add_argument("--model", default="mnist", choices=["mnist", "wide", "multiscale", "tcvae"],
                    help="mnist = the contract workload (BASELINE configs[1]/[3]); multiscale / tcvae = configs[2] "
                         "(dataGen-shaped windows, MSE ELBO) for the record in DESIGN.md")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = os.environ.get("VAE_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N>1 path on a box with one GPU
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import vae_amd
    from vae_amd.train import FusedTrainStep

    B = args.batch or (4096 if world == 1 else 65536 // world)
    torch.manual_seed(0)
    g = torch.Generator(device=dev).manual_seed(rank)
    if args.model in ("mnist", "wide"):
        if args.model == "mnist":
            model = vae_amd.mnist_vae.VariationalAutoencoder(15)
            workload = "MNIST 28x28 Conv2d VAE (notebook model, 103,343 params), BCE ELBO, Adam"
        else:  # BASELINE configs[4]; hidden width kept at 100 (SURVEY 8d asks to state the choice)
            model = vae_amd.mnist_vae.VariationalAutoencoder(256, channels=(32, 64, 128), hidden=100)
            workload = "widened MNIST Conv2d VAE (channels 32/64/128, latent 256, hidden 100), BCE ELBO, Adam"
        x = torch.rand(B, 1, 28, 28, device=dev, generator=g)
    else:
        if args.model == "multiscale":
            model = vae_amd.multiscale_vae.vae(3, 60, 15)
            workload = "multiscale_vae.vae(3, 60, 15) (12,255 params), MSE ELBO, Adam"
        elif args.model == "tcvae":
            model = vae_amd.VariationalAutoencoder(input_size=3, hidden_size=30, num_layers=3, latent_dims=15,
                                                   v_encoder=vae_amd.Encoders.LongShort_TCVAE_Encoder,
                                                   v_decoder=vae_amd.Decoders.LongShort_TCVAE_Decoder, L=60, slope=0.2,
                                                   first_kernel=21)
            workload = "TCVAE C=3 N=3 latent 15 L=60 first_kernel 21 (9,549 params), MSE ELBO, Adam"
        x = 50 + 10 * torch.randn(B, 3, 60, device=dev, generator=g)  # dataGen-shaped windows (SURVEY 8d cfg 2)
    model = model.to(dev).train()
    opt = vae_amd.optim.Adam(model.parameters(), lr=1e-3)
    if world == 1:
        stepper = FusedTrainStep(model, opt, B, dev, use_graph=not args.no_graph)
    else:
        from vae_amd.parallel import DataParallelTrainStep
        stepper = DataParallelTrainStep(model, opt, B, dev)
    stepper.step(x)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        stepper.step()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        stepper.step()
    e1.record()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(stepper.loss.item())
    if not (loss == loss) or loss <= 0:
        raise SystemExit(f"loss is {loss}: the step is not training")

    roof = None
    if rank == 0 and args.model == "mnist":
        roof = dominant_kernel_roofline(stepper, args.roofline_iters)  # (local launches only)
    barrier()

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        dev_ms = e0.elapsed_time(e1) / args.steps
        line = {
            "metric": "train samples/sec (whole node)", "value": value, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak" if world == 1 else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{workload}, batch {B} per GPU, global batch {B * world}",
                       "per_gpu_batch": B, "global_batch": B * world,
                       "parallelism": ("single GPU, whole step as one hipGraph" if world == 1 else
                                       f"dp{world}: batch shards, SyncBN all-gather per BatchNorm, grad all-reduce SUM (RCCL)")},
            "loss_per_sample": loss / B,
            "device_ms_per_step": dev_ms,
            # whole-step figure against the materialised-activation traffic model of SURVEY 8d (178 KB/sample)
        }
        if args.model == "mnist":
            line["step_hbm_model_GBps"] = M_BYTES_PER_SAMPLE * B / (dev_ms * 1e-3) / 1e9
        if roof is not None:
            line["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
