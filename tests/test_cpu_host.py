"""CPU-only checks of the host side: the C-ABI library loads and exports every declared
symbol, modules mirror the reference's constructor/state_dict contract, and the product
refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden, sub


def test_library_exports_every_declared_symbol():
    from vae_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with make -C vae_amd/csrc"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "vae_amd.h")).read()
    declared = set(re.findall(r"\b(vae_[a-z_]+)\s*\(", header)) - {"vae_op"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert ctypes.sizeof(_lib.VaeOp) == 192  # must match struct vae_op


def _build(stem):
    import vae_amd
    from vae_amd.Decoders import LongShort_TCVAE_Decoder
    from vae_amd.Encoders import LongShort_TCVAE_Encoder
    if stem.startswith("tcvae"):
        C, N, Lw, fk, slope, lat = {"tcvae_c3_n3_l60_fk21": (3, 3, 60, 21, 0.2, 15),
                                    "tcvae_c3_n4_l60_fk15": (3, 4, 60, 15, 0.2, 15),
                                    "tcvae_c2_n2_l30_relu": (2, 2, 30, None, 0.0, 4)}[stem]
        return vae_amd.VariationalAutoencoder(input_size=C, hidden_size=30, num_layers=N, latent_dims=lat,
                                              v_encoder=LongShort_TCVAE_Encoder, v_decoder=LongShort_TCVAE_Decoder,
                                              L=Lw, slope=slope, first_kernel=fk)
    if stem.startswith("multiscale"):
        return vae_amd.multiscale_vae.vae(3, 60, 15)
    return vae_amd.mnist_vae.VariationalAutoencoder(15)


SEEDS = {"tcvae_c3_n3_l60_fk21": 11, "tcvae_c3_n4_l60_fk15": 12, "tcvae_c2_n2_l30_relu": 13,
         "multiscale_c3_l60": 14, "mnist_lat15": 15}


def test_state_dict_keys_shapes_and_seeded_init_match_reference(golden_case):
    """Same constructor call under the same torch seed gives the reference's state_dict: same
    keys, shapes AND values (parameter creation order and init distributions are mirrored)."""
    stem, family, kw, g = golden_case
    torch.manual_seed(SEEDS[stem])
    m = _build(stem)
    want = sub(g, "sd/")
    got = m.state_dict()
    assert list(got.keys()) == list(want.keys())
    for k, v in want.items():
        assert got[k].shape == v.shape, k
        torch.testing.assert_close(got[k].float(), v.float(), rtol=0, atol=0, msg=lambda s, k=k: f"{k}: {s}")
    m.load_state_dict(want)  # strict load works


def test_lin_size_and_too_short_window():
    from vae_amd.Encoders import TCVAE_Encoder, lin_size
    assert lin_size(60, 3) * 3 == 126 and lin_size(60, 3, 21) * 3 == 90
    with pytest.raises(ValueError):
        lin_size(8, 5)
    with pytest.raises(ValueError):
        TCVAE_Encoder(3, 5, 4, L=8)


def test_no_cpu_fallback():
    import vae_amd
    m = _build("tcvae_c2_n2_l30_relu")
    with pytest.raises(vae_amd.VaeAmdError):
        m(torch.zeros(2, 2, 30))
    with pytest.raises(vae_amd.VaeAmdError):
        vae_amd.train.criterion(torch.zeros(2, 2), torch.zeros(2, 2), torch.zeros(2, 3), torch.zeros(2, 3))


def test_program_layout_is_consistent():
    """Lowering a model on CPU (no launch): every op's refs fall inside the declared spaces."""
    from vae_amd import _lib as L
    m = _build("mnist_lat15")
    params = [p for _, p in m.named_parameters()]
    m._param_off, off = {}, 0
    for p in params:
        m._param_off[id(p)] = off
        off += (p.numel() + 3) // 4 * 4
    m._state_off, soff = {}, 0
    for b_ in m._float_buffers():
        m._state_off[id(b_)] = soff
        soff += (b_.numel() + 3) // 4 * 4
    from vae_amd.program import ProgramBuilder
    b = ProgramBuilder(32, m._param_off, m._state_off)
    m.emit_program(b, True)
    prog = b.build()
    assert prog.n_ops == 16 and prog.work_size > 0
    sizes = {L.SP_WORK: prog.work_size, L.SP_PARAM: off, L.SP_GRAD: off, L.SP_STATE: soff}
    for op in prog.ops:
        for r in op.r:
            if r >= 0 and (r >> 56) in sizes:
                assert (r & ((1 << 56) - 1)) < sizes[r >> 56]


def test_forward_kernel_routing_of_the_mnist_layers():
    """vae_conv_forward_variant is a host-only query (no launch): the kernel family each MNIST conv layer runs on,
    as DESIGN.md's kernel table states it -- 300 thin.h streaming, 400 mid_conv_k, 200 MFMA gather tiles,
    500 scat_convT_k.  A routing change that silently sends a layer back to the chunked gather shows up here."""
    import ctypes
    from vae_amd import _lib as L
    from vae_amd.program import ProgramBuilder
    m = _build("mnist_lat15")
    m._param_off, off = {}, 0
    for _, p in m.named_parameters():
        m._param_off[id(p)] = off
        off += (p.numel() + 3) // 4 * 4
    m._state_off, soff = {}, 0
    for b_ in m._float_buffers():
        m._state_off[id(b_)] = soff
        soff += (b_.numel() + 3) // 4 * 4
    for B in (32, 4096):
        b = ProgramBuilder(B, m._param_off, m._state_off)
        m.emit_program(b, True)
        got = [(op.kind, L.lib().vae_conv_forward_variant(ctypes.byref(op), B))
               for op in b.build().ops if op.kind in (L.OP_CONV, L.OP_CONVT)]
        assert got == [(L.OP_CONV, 300), (L.OP_CONV, 400), (L.OP_CONV, 200),
                       (L.OP_CONVT, 500), (L.OP_CONVT, 500), (L.OP_CONVT, 300)], (B, got)


def test_gradient_buckets_split_at_the_reparameterisation():
    """DataParallelTrainStep's two gradient buckets (vae_run_backward_range): decoder + loss ops first, then the
    encoder; the flat-buffer ranges partition the parameters at the first decoder parameter."""
    from vae_amd.program import ProgramBuilder
    for stem in ("mnist_lat15", "multiscale_c3_l60", "tcvae_c3_n3_l60_fk21"):
        m = _build(stem)
        m._param_off, off = {}, 0
        names = []
        for n, p in m.named_parameters():
            m._param_off[id(p)] = off
            names.append((n, off))
            off += (p.numel() + 3) // 4 * 4
        m._state_off, soff = {}, 0
        for b_ in m._float_buffers():
            m._state_off[id(b_)] = soff
            soff += (b_.numel() + 3) // 4 * 4
        b = ProgramBuilder(32, m._param_off, m._state_off)
        m.emit_program(b, True)
        prog = b.build()
        buckets = prog.grad_buckets()
        assert buckets is not None, stem
        (b0, e0, lo0, hi0), (b1, e1, lo1, hi1) = buckets
        assert e0 == prog.n_ops and b0 == e1 and b1 == 0          # backward order, every op exactly once
        first_dec = min(o for n, o in names if n.startswith("decoder"))
        assert lo0 == first_dec and hi0 is None and (lo1, hi1) == (0, first_dec), (stem, buckets, first_dec)


def test_c_abi_rejects_bad_programs_before_any_launch():
    """Host-side validation of the executor (no GPU needed: every check precedes the first HIP call):
    null arguments, batch out of range, unknown op kind, inconsistent conv geometry, references outside the
    bound address spaces.  Each returns VAE_EINVAL (-1) and leaves a message in vae_last_error()."""
    import ctypes
    from vae_amd import _lib as L
    lib = L.lib()
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    fake = (ctypes.c_float * 64)()                      # host memory: only its ADDRESS is used, nothing is launched
    bases = (vp * 4)(ctypes.addressof(fake), ctypes.addressof(fake), ctypes.addressof(fake), ctypes.addressof(fake))
    sizes = (i64 * 4)(64, 64, 64, 64)

    def run(ops, n, B=8, n_bases=4):
        rc = lib.vae_run_forward(ops, n, bases, sizes, n_bases, B, 1, None)
        return rc, lib.vae_last_error().decode()

    ops = (L.VaeOp * 1)()
    assert run(ops, 1, B=0) == (-1, "batch 0 out of range")
    assert run(ops, 1, n_bases=2)[0] == -1 and "n_bases" in lib.vae_last_error().decode()
    ops[0].kind = 99
    rc, msg = run(ops, 1)
    assert rc == -1 and "unknown kind 99" in msg
    # conv whose output size does not follow from input, kernel, stride and padding
    ops[0].kind = L.OP_CONV
    for j, v in enumerate((1, 28, 28, 8, 13, 14, 3, 3, 2, 2, 1, 1)):
        ops[0].d[j] = v
    rc, msg = run(ops, 1)
    assert rc == -1 and "inconsistent" in msg, msg
    # a Linear whose input extent leaves the WORK space (64 floats bound, 8 x 100 needed)
    ops[0] = L.VaeOp()
    ops[0].kind = L.OP_LINEAR
    for j in range(12):
        ops[0].r[j] = -1
    ops[0].r[0], ops[0].r[1], ops[0].r[4] = 0, 0, (L.SP_PARAM << 56)
    for j, v in enumerate((100, 4, 100, 4, 100, 4)):
        ops[0].d[j] = v
    rc, msg = run(ops, 1)
    assert rc == -1 and "exceeds space" in msg, msg
    # backward over an op range outside the program
    rc = lib.vae_run_backward_range(ops, 1, bases, sizes, 4, 8, None, 0, 2, 1)
    assert rc == -1 and "op range" in lib.vae_last_error().decode()


def test_device_window_loader_has_no_cpu_path():
    """DeviceSlidingWindow keeps the reference Dataset semantics on the host side (len, .dataset) but its gather is
    a HIP kernel: on a CPU tensor it raises instead of falling back."""
    import pytest
    import vae_amd
    from vae_amd.train import DeviceSlidingWindow
    data = torch.arange(3 * 100, dtype=torch.float32).view(3, 100)
    loader = DeviceSlidingWindow(data, 60, batch_size=16, device="cpu")
    assert len(loader.dataset) == 40 and len(loader) == 3
    assert torch.equal(loader.dataset[5], data[:, 5:65])          # the reference Dataset (train.py:25-28)
    with pytest.raises(vae_amd.VaeAmdError):
        loader.gather(torch.arange(4))
    with pytest.raises(ValueError):
        DeviceSlidingWindow(data[:, :60], 60, device="cpu")       # T must exceed L


def test_data_parallel_entry_points_fail_cleanly_without_a_communicator():
    """vae_dp_* before vae_dp_init: no communicator, every call answers with an error code and a message (host-only:
    nothing here touches a GPU or opens librccl)."""
    import ctypes
    from vae_amd import _lib as L
    lib = L.lib()
    assert lib.vae_dp_world() == 0
    assert lib.vae_dp_enable(0) == 0                       # removing the hook is always fine
    assert lib.vae_dp_enable(1) != 0 and b"vae_dp_init" in lib.vae_last_error()
    buf = (ctypes.c_float * 8)()
    assert lib.vae_dp_all_reduce_sum(buf, 8, None) != 0 and b"communicator" in lib.vae_last_error()
    assert lib.vae_dp_all_reduce_sum(None, 8, None) != 0
    assert lib.vae_dp_all_gather_rows(None, 4, None) != 0
    assert lib.vae_dp_init(None, None, 0, 1) != 0          # null id
    ident = (ctypes.c_ubyte * L.DP_ID_BYTES)()
    assert lib.vae_dp_init(None, ident, 3, 2) != 0         # rank outside the world
    assert lib.vae_dp_shutdown() == 0                      # nothing to destroy


def test_backward_step_rejects_missing_optimiser_state():
    import ctypes
    from vae_amd import _lib as L
    lib = L.lib()
    ops = (L.VaeOp * 1)()
    bases = (ctypes.c_void_p * 4)()
    sizes = (ctypes.c_int64 * 4)()
    rc = lib.vae_run_backward_step(ops, 0, bases, sizes, 4, 8, None, None, None, None, None, None, 0, None)
    assert rc != 0 and b"optimiser state" in lib.vae_last_error()
