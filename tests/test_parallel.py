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


@pytest.mark.parametrize("bad_rank", ["0", "1"])
def test_native_init_failure_is_agreed_on_by_every_rank(bad_rank):
    """One rank that cannot open / join RCCL must not desynchronise the others (vae_amd.parallel.native_init): both
    ranks fall back (required=False) or raise (required=True) together, and the next collective still pairs up."""
    _run_world("native_fail", bad_rank, timeout=120)


@pytest.mark.gpu
@pytest.mark.parametrize("stem,mode", [("mnist_lat15", "gpu"), ("tcvae_c3_n3_l60_fk21", "gpu"), ("multiscale_c3_l60", "gpu"),
                                       ("tcvae_c3_n3_l60_fk21", "gpu_layers"), ("multiscale_c3_l60", "gpu_layers")])
def test_dp_hip_path_two_ranks(stem, mode):
    """DataParallelTrainStep (HIP kernels, SyncBN hook, gradient all-reduce) on two ranks sharing the GPU
    reproduces the golden 3-step trajectory of the global batch.  The 1-D families run twice: on the whole-model
    seq1d path (statistics rows of the ranks exchanged between the phases) and on the per-layer program."""
    _run_world(mode, stem)


@pytest.mark.gpu
def test_dp_seq1d_three_uneven_ranks():
    """Three ranks with shards of 6 / 5 / 5 samples on the whole-model 1-D path: the backward prologue takes the
    global count per channel from the ranks' rows, not from world x its own batch."""
    _run_world("gpu", "multiscale_c3_l60", world=3)


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


@pytest.mark.gpu
@pytest.mark.parametrize("single_comm", [False, True])
def test_native_rccl_step_single_rank(single_comm):
    """The library's own RCCL path (vae_dp_*: SyncBN all-gather + gradient all-reduce enqueued from C on the caller's
    streams) at world 1, the most one GPU allows (RCCL refuses two ranks on one device): the exchanges run through
    the C-ABI, the data-parallel step makes no callback into Python, equals the single-GPU fused step bit for bit
    (an all-reduce over one rank is the identity), and can be captured into one hipGraph, collectives included.
    single_comm: VAE_DP_SINGLE_COMM=1 -- one communicator, every collective on the compute stream, one gradient bucket."""
    code = (
        "import os, sys, copy, ctypes, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "import vae_amd\n"
        "from vae_amd import _lib as L\n"
        "from vae_amd.parallel import DataParallelTrainStep, native_init\n"
        "from vae_amd.train import FusedTrainStep\n"
        "assert native_init() == 1 and L.lib().vae_dp_world() == 1\n"
        "work = torch.arange(64, dtype=torch.float32, device='cuda')\n"
        "L.check(L.lib().vae_dp_all_gather_rows(L.ptr(work[16:48]), 32, L.stream_ptr()), 'gather')\n"
        "g = torch.ones(1000, device='cuda')\n"
        "L.check(L.lib().vae_dp_all_reduce_sum(L.ptr(g), 1000, L.stream_ptr()), 'reduce')\n"
        "torch.cuda.synchronize()\n"
        "assert torch.equal(work.cpu(), torch.arange(64.)) and float(g.sum()) == 1000.0\n"
        "B = 64\n"
        "torch.manual_seed(5)\n"
        "ref_m = vae_amd.mnist_vae.VariationalAutoencoder(15).cuda().train()\n"
        "xs = [torch.rand(B, 1, 28, 28, device='cuda') for _ in range(3)]\n"
        "es = [torch.randn(B, 15, device='cuda') for _ in range(3)]\n"
        "def run(make):\n"
        "    m = copy.deepcopy(ref_m)\n"
        "    st = make(m, vae_amd.optim.Adam(m.parameters(), lr=1e-3))\n"
        "    st.inject_eps = True\n"
        "    losses = []\n"
        "    for x, e in zip(xs, es):\n"
        "        st.eps.copy_(e)\n"
        "        losses.append(float(st.step(x).item()))\n"
        "    torch.cuda.synchronize()\n"
        "    return st, losses, st.flat.clone()\n"
        "_, l0, p0 = run(lambda m, o: FusedTrainStep(m, o, B, 'cuda', use_graph=False, allow_seq1d=False))\n"
        "for graph in (False, True):\n"
        "    st, l1, p1 = run(lambda m, o: DataParallelTrainStep(m, o, B, 'cuda', use_graph=graph))\n"
        "    assert st.native and st.hook_calls == 0\n"
        f"    assert st.single_comm == {single_comm!r} and (st.comm_stream is None) == {single_comm!r}\n"
        f"    assert L.lib().vae_dp_single_comm() == {int(single_comm)}\n"
        "    assert (st.graph is not None) == graph\n"
        "    assert l1 == l0, (graph, l0, l1)\n"
        "    assert torch.equal(p0, p1), (graph, float((p0 - p1).abs().max()))\n"
        "L.check(L.lib().vae_dp_shutdown(), 'shutdown')\n"
        "assert L.lib().vae_dp_world() == 0\n"
        "dist.destroy_process_group()\n"
        "print('NATIVE-OK')\n")
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               VAE_DP_SINGLE_COMM="1" if single_comm else "0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "NATIVE-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
