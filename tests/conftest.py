import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# (file stem, family, oracle kwargs)
GOLDEN_CASES = [
    ("tcvae_c3_n3_l60_fk21", "tcvae", dict(num_layers=3, slope=0.2)),
    ("tcvae_c3_n4_l60_fk15", "tcvae", dict(num_layers=4, slope=0.2)),
    ("tcvae_c2_n2_l30_relu", "tcvae", dict(num_layers=2, slope=0.0)),
    ("multiscale_c3_l60", "multiscale", dict()),
    ("mnist_lat15", "mnist", dict()),
]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# How often the tolerance escape hatches below were used in this session (printed in the terminal summary, so a
# regression that moves many elements through a hatch is visible even while the tests stay green).
HATCH = {"grad_elements": 0, "grad_fp64_hatch": 0, "grad_kink_hatch": 0, "param_elements": 0, "param_stragglers": 0}


# Session ceilings: a regression that pushes many elements through a hatch FAILS the session instead of printing.
# (last full GPU run: 9 + 35 of 1,268,177 gradient elements = 7e-6 and 2.8e-5)
HATCH_CEILING = 1e-4


def hatch_over_ceiling():
    n = HATCH["grad_elements"]
    return [name for name in ("grad_fp64_hatch", "grad_kink_hatch") if n and HATCH[name] > HATCH_CEILING * n]


def pytest_sessionfinish(session, exitstatus):
    over = hatch_over_ceiling()
    if over and session.exitstatus == 0:
        session.exitstatus = 1


def pytest_terminal_summary(terminalreporter):
    over = hatch_over_ceiling()
    if over:
        terminalreporter.write_line(
            "FAILED tolerance-hatch ceiling: %s exceeded %g x %d gradient elements (%s)"
            % (", ".join(over), HATCH_CEILING, HATCH["grad_elements"],
               ", ".join("%s=%d" % (k, HATCH[k]) for k in over)), red=True)
    if HATCH["grad_elements"] or HATCH["param_elements"]:
        terminalreporter.write_line(
            "tolerance hatches: of %d gradient elements %d were accepted through the fp64-truth hatch and %d as "
            "ReLU-kink flips; %d of %d parameter elements were Adam stragglers (<= 1%% per tensor allowed)"
            % (HATCH["grad_elements"], HATCH["grad_fp64_hatch"], HATCH["grad_kink_hatch"], HATCH["param_stragglers"],
               HATCH["param_elements"]))


def load_golden(stem):
    import numpy as np
    import torch

    z = np.load(os.path.join(GOLDEN, stem + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a.copy()) if a.ndim else torch.tensor(a.item(), dtype=torch.int64 if a.dtype.kind == "i" else torch.float32)
    return out


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(params=GOLDEN_CASES, ids=[c[0] for c in GOLDEN_CASES])
def golden_case(request):
    stem, family, kw = request.param
    return stem, family, kw, load_golden(stem)


def assert_grads_close(got, want, rtol=1e-3, atol_rel=2e-5, truth=None, kinks=True, kink_exempt=()):
    """Gradient comparison used by every parity test: rtol 1e-3 (BASELINE.json north_star)
    plus an absolute floor of atol_rel x the largest gradient magnitude in the whole model.
    The floor is needed because some gradients are pure cancellation noise in fp32 (e.g.
    the bias of a conv whose outputs are all positive before BatchNorm: true gradient 0).

    truth (optional): the same gradients from the oracle evaluated in float64.  An element that
    misses the fp32 reference by more than the tolerance is still accepted when it is at least as
    close to the float64 truth as a small multiple of the fp32 reference's OWN rounding error
    (ill-conditioned sums over raw data of magnitude ~100: two fp32 evaluation orders differ).

    kinks=False: the model has no activation kink (LeakyReLU slope 1.0), so the ReLU-kink hatch is closed.
    kink_exempt: name prefixes of tensors that sit behind no activation kink (their gradient cannot change when
    a pre-activation flips sides): the kink hatch is closed for them too."""
    import torch

    assert set(got) == set(want), (sorted(set(got) ^ set(want)))
    gmax = max(float(v.abs().max()) for v in want.values())
    for k, v in want.items():
        a, v = got[k].detach().float().cpu(), v.float()
        bad = (a - v).abs() > atol_rel * gmax + rtol * v.abs()
        HATCH["grad_elements"] += bad.numel()
        if bad.any() and truth is not None:
            t = truth[k].double()
            ref_err = (v.double() - t).abs()
            missed = int(bad.sum())
            bad &= (a.double() - t).abs() > 8 * ref_err + atol_rel * gmax
            HATCH["grad_fp64_hatch"] += missed - int(bad.sum())
            if missed - int(bad.sum()):
                print(f"[hatch] grad {k}: {missed - int(bad.sum())}/{bad.numel()} elements accepted via the fp64 truth")
            if bad.any() and kinks and not any(k.startswith(p) for p in kink_exempt):
                # Activation kinks: at B = 4096 a few of the ~3e6 pre-activations lie within fp32 rounding of 0, and
                # two fp32 evaluation orders put them on different sides of the ReLU.  Each flip moves the weight
                # gradient of its layer by one sample's contribution (a rank-one term, ~1/B of the tensor).  Such a
                # tensor is accepted when, as a whole, it is within half the rtol of the fp64 truth in relative L2.
                # ... or when it is as close to the fp64 truth as the fp32 reference itself is (within 3x, in L2): the
                # reference's own flips land on other elements than ours.
                err, ref_l2 = float((a.double() - t).norm()), float((v.double() - t).norm())
                rel = err / float(t.norm().clamp_min(1e-30))
                if rel <= 0.5 * rtol or err <= 3.0 * ref_l2 + atol_rel * gmax:
                    HATCH["grad_kink_hatch"] += int(bad.sum())
                    print(f"[hatch] grad {k}: {int(bad.sum())}/{bad.numel()} elements accepted as ReLU-kink flips "
                          f"(tensor L2 error vs fp64 truth {err:.3e} = {rel:.2e} relative; fp32 reference's own {ref_l2:.3e})")
                    bad[:] = False
        assert not bad.any(), (f"grad {k}: {int(bad.sum())}/{bad.numel()} elements off, max abs diff "
                               f"{float((a - v).abs().max()):.4e} (allowed {atol_rel * gmax:.3e} + {rtol}*|ref|)")


def assert_params_close(got, want, grads0, steps, lr, rtol=2e-4, atol=2e-5, noise_rel=2e-5):
    """Parameters after `steps` Adam steps.  Adam divides each gradient element by its own
    running magnitude, so an absolute gradient error e on an element of size |g| moves that
    parameter by about lr*e/|g| per step (a pure-noise gradient, see assert_grads_close, takes
    +-lr steps of arbitrary sign).  Per-element tolerance therefore is
        atol + rtol*|want| + steps*lr*min(1, 10*noise/|g0|),   noise = noise_rel x global max |g0|
    which is ~2e-5 for well-conditioned elements and at most 2.5*steps*lr for noise elements (two trajectories stepping in opposite directions)."""
    import torch

    gmax = max(float(v.abs().max()) for v in grads0.values())
    for k, v in want.items():
        a = got[k].detach().cpu()
        if not v.dtype.is_floating_point:
            assert int(a) == int(v), k
            continue
        a = a.float()
        tol = max(atol, 0.1 * steps * lr) + rtol * v.abs()  # a wrong update direction would show as >= lr
        if k in grads0:
            tol = tol + 2.5 * steps * lr * torch.clamp(10 * noise_rel * gmax / grads0[k].abs().clamp_min(1e-30), max=1.0)
        elif "running_" in k:  # running stats see the +-lr drift of noisy biases upstream
            tol = tol + steps * lr
        bad = (a - v).abs() > tol
        # activation kinks (LeakyReLU/ReLU at 0) make a handful of elements chaotic over several
        # steps at tiny batch sizes: allow <=1% stragglers, never beyond the opposite-direction cap
        worst = float((a - v).abs().max())
        HATCH["param_elements"] += bad.numel()
        HATCH["param_stragglers"] += int(bad.sum())
        if bad.any():
            print(f"[hatch] param {k}: {int(bad.sum())}/{bad.numel()} stragglers, worst {worst:.3e}")
        assert bad.float().mean() <= 0.01 and worst <= 2.5 * steps * lr + atol, \
            f"param {k}: {int(bad.sum())}/{bad.numel()} elements off, max diff {worst:.3e}"


def truth64(family, sd, x, eps, **kw):
    """Gradients of the oracle evaluated in float64 (see assert_grads_close)."""
    from oracle import vae_oracle as O
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    return O.loss_and_grads(family, sd64, x.double(), eps.double(), **kw)["grads"]
