"""Worker of the data-parallel tests (tests/test_parallel.py starts WORLD_SIZE copies of it).

mode "cpu": gloo, CPU tensors.  Each rank evaluates the ORACLE on its batch shard with BatchNorm
    statistics summed over the ranks, all-reduces the gradients with vae_amd.parallel.Collective and
    compares with the golden single-process gradients at the global batch: the data-parallel recipe
    (SUM of gradients + synchronised BatchNorm) and the collective helpers, without a GPU.
mode "gpu": gloo with GPU tensors staged through the host (two ranks share the one GPU of the test box,
    which RCCL refuses).  Each rank runs vae_amd.parallel.DataParallelTrainStep (the HIP path) on its
    shard for 3 steps and compares loss, gradients and the parameter trajectory with the golden vectors.
Prints "DP-OK <rank>" on success.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import GOLDEN_CASES, assert_grads_close, assert_params_close, load_golden, sub, truth64  # noqa: E402


def sync_bn(O, coll):
    """oracle._bn with the batch statistics taken over all ranks (autograd-aware all-reduce)."""
    from torch.distributed.nn.functional import all_reduce

    def _bn(sd, prefix, x, training, new_stats):
        assert training
        w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
        rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
        dims = [0] + list(range(2, x.dim()))
        n = all_reduce(torch.tensor(float(x.numel() // x.shape[1])), group=coll.group)
        mean = all_reduce(x.sum(dims), group=coll.group) / n
        shape = [1, -1] + [1] * (x.dim() - 2)
        var = all_reduce(((x - mean.view(shape)) ** 2).sum(dims), group=coll.group) / n
        with torch.no_grad():
            new_stats[prefix + ".running_mean"] = (1 - O.BN_MOMENTUM) * rm + O.BN_MOMENTUM * mean
            new_stats[prefix + ".running_var"] = (1 - O.BN_MOMENTUM) * rv + O.BN_MOMENTUM * var * n / (n - 1)
        return (x - mean.view(shape)) / torch.sqrt(var.view(shape) + O.BN_EPS) * w.view(shape) + b.view(shape)
    return _bn


def run_cpu(stem, family, kw):
    from oracle import vae_oracle as O
    from vae_amd.parallel import Collective, shard_bounds
    coll = Collective()
    g = load_golden(stem)
    sd = sub(g, "sd/")
    lo, hi = shard_bounds(g["x"].shape[0], coll.rank, coll.world)
    assert hi > lo
    # helper: all_gather_rows fills the other ranks' rows in place
    rows = torch.zeros(coll.world, 5)
    rows[coll.rank] = torch.arange(5.0) + 10 * coll.rank
    coll.all_gather_rows(rows)
    assert torch.equal(rows, torch.stack([torch.arange(5.0) + 10 * r for r in range(coll.world)]))
    O._bn = sync_bn(O, coll)
    ref = O.loss_and_grads(family, sd, g["x"][lo:hi], g["eps0"][lo:hi], **kw)
    keys = sorted(ref["grads"])
    flat = torch.cat([ref["grads"][k].reshape(-1) for k in keys])
    coll.all_reduce_sum(flat)
    loss = ref["loss"].detach().reshape(1).clone()
    coll.all_reduce_sum(loss)
    want = g["loss0"].item()
    assert abs(loss.item() - want) <= 2e-3 * abs(want) + 1e-2, (loss.item(), want)
    got, off = {}, 0
    for k in keys:
        n = ref["grads"][k].numel()
        got[k] = flat[off:off + n].view(ref["grads"][k].shape)
        off += n
    assert_grads_close(got, sub(g, "grad/"), truth=truth64(family, sd, g["x"], g["eps0"], **kw))
    # BatchNorm running statistics after the step equal the single-process ones
    for k, v in ref["new_stats"].items():
        assert torch.allclose(v, g["sd1/" + k], rtol=1e-4, atol=1e-5), k


def run_gpu(stem, family, kw):
    import vae_amd
    from test_cpu_host import _build
    from vae_amd.parallel import DataParallelTrainStep, shard_bounds
    dev = torch.device("cuda", 0)
    g = load_golden(stem)
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_bounds(g["x"].shape[0], rank, world)
    m = _build(stem)
    m.load_state_dict(sub(g, "sd/"))
    m.to(dev).train()
    opt = vae_amd.optim.Adam(m.parameters(), lr=1e-3)
    # mode "gpu": the path the step picks by itself (1-D families: the whole-model seq1d path with its rank-row
    # exchange between the phases); "gpu_layers": the per-layer program
    st = DataParallelTrainStep(m, opt, hi - lo, dev, allow_seq1d=(MODE != "gpu_layers"))
    assert bool(getattr(st.prog, "seq1d", False)) == (family != "mnist" and MODE != "gpu_layers"), (family, MODE)
    st.inject_eps = True
    x = g["x"][lo:hi].to(dev)
    for step in range(3):
        st.eps.copy_(g[f"eps{step}"][lo:hi])
        st.step(x)
        want = g[f"loss{step}"].item()
        got = st.global_loss()
        assert abs(got - want) <= 2e-3 * abs(want) + 1e-2, (step, got, want)
        if step == 0:
            grads = {k: p.grad.clone().cpu() for k, p in m.named_parameters()}
            assert_grads_close(grads, sub(g, "grad/"), truth=truth64(family, sub(g, "sd/"), g["x"], g["eps0"], **kw))
    torch.cuda.synchronize()
    assert_params_close({k: v.cpu() for k, v in m.state_dict().items()}, sub(g, "sd3/"), sub(g, "grad/"), steps=3, lr=1e-3)


def run_native_fail(bad_rank):
    """native_init when ONE rank cannot open / join RCCL (its vae_dp_unique_id is replaced by a failing stub): every
    rank must come out of it with the same answer and at the same point of the collective sequence -- checked by the
    broadcast that follows, which would pair with a mismatched collective (or hang) otherwise."""
    from vae_amd import _lib as L
    from vae_amd.parallel import native_init
    lib = L.lib()
    rank = dist.get_rank()
    if rank == int(bad_rank):
        lib.vae_dp_unique_id = lambda path, ident: -1
    assert native_init(required=False) == 0
    t = torch.arange(1000.0) * (1 if rank == 0 else -1)
    dist.broadcast(t, src=0)
    assert torch.equal(t, torch.arange(1000.0))
    try:
        native_init(required=True)
    except L.VaeAmdError as e:
        assert ("vae_dp_unique_id failed" in str(e)) == (rank == int(bad_rank)), str(e)
    else:
        raise AssertionError("native_init(required=True) must raise on every rank")
    t = torch.full((7,), float(rank))
    dist.all_reduce(t)
    assert torch.equal(t, torch.full((7,), float(sum(range(dist.get_world_size())))))
    assert lib.vae_dp_world() == 0


MODE = "gpu"


def main():
    global MODE
    mode, stem = sys.argv[1], sys.argv[2]
    MODE = mode
    dist.init_process_group("gloo")
    try:
        if mode == "native_fail":
            run_native_fail(stem)
        else:
            _, family, kw = next(c for c in GOLDEN_CASES if c[0] == stem)
            (run_cpu if mode == "cpu" else run_gpu)(stem, family, kw)
        dist.barrier()
        print(f"DP-OK {dist.get_rank()}", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
