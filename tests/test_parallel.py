"""Data-parallel path: world_size-2 gloo runs of tests/dp_worker.py."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_world(mode, stem, world=2, timeout=240):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", PYTHONDONTWRITEBYTECODE="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), mode, stem], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"DP-OK {rank}" in out, f"rank {rank} failed:\n{out[-3000:]}"


def test_shard_bounds():
    from vae_amd.parallel import shard_bounds
    assert [shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_bounds(65536, r, 8) for r in range(8)] == [(r * 8192, (r + 1) * 8192) for r in range(8)]
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


@pytest.mark.parametrize("stem", ["mnist_lat15", "tcvae_c3_n3_l60_fk21", "multiscale_c3_l60"])
def test_dp_recipe_cpu_gloo(stem):
    """SUM of shard gradients + BatchNorm statistics over all ranks == the single-process golden step."""
    _run_world("cpu", stem)


@pytest.mark.gpu
@pytest.mark.parametrize("stem", ["mnist_lat15", "tcvae_c3_n3_l60_fk21", "multiscale_c3_l60"])
def test_dp_hip_path_two_ranks(stem):
    """DataParallelTrainStep (HIP kernels, SyncBN hook, gradient all-reduce) on two ranks sharing the GPU
    reproduces the golden 3-step trajectory of the global batch."""
    _run_world("gpu", stem)


@pytest.mark.gpu
def test_collective_rccl_single_rank():
    """The RCCL calls DataParallelTrainStep makes (in-place all-gather of workspace rows, all-reduce SUM,
    broadcast) are accepted by torch.distributed's nccl backend on GPU tensors (world 1: one GPU here)."""
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "from vae_amd.parallel import Collective\n"
        "c = Collective()\n"
        "work = torch.arange(64, dtype=torch.float32, device='cuda')\n"
        "rows = work[16:48].view(1, 32)\n"
        "c.all_gather_rows(rows)\n"
        "g = torch.ones(1000, device='cuda')\n"
        "c.all_reduce_sum(g)\n"
        "dist.broadcast(g, src=0)\n"
        "torch.cuda.synchronize()\n"
        "assert torch.equal(work.cpu(), torch.arange(64.)) and float(g.sum()) == 1000.0\n"
        "dist.destroy_process_group()\n"
        "print('RCCL-OK')\n")
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and "RCCL-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
