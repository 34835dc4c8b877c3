"""Bit-reproducible accumulation (ssp_set_deterministic / SSP_DETERMINISTIC=1, csrc/det.hip.h): the same pair step from the same
state, run three times on fresh engines, must produce the SAME BITS - scalars, every gradient, the parameters and Adam moments
after the optimizer step, the BatchNorm running statistics.  The reference has no such switch (train4.py never enables
torch.use_deterministic_algorithms); the default mode of this library keeps plain floating-point atomics, whose commit order
varies from run to run (~1e-6 relative on gradients, budgeted by the parity tests)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

ARCH = "SuperPointNet_gauss2_ssmall"   # all three heads: detector, descriptor, segmentation


def _run(algo, B, H, W, deterministic, second_step=True, reps=3):
    from semantic_superpoint_amd import lib as L, synth
    dev = torch.device("cuda:0")
    L.set_deterministic(deterministic)
    try:
        sd = synth.default_init_state_dict(L.layer_table(ARCH), seed=3)
        sample = synth.make_pair(B, H, W, dev, seed=41, semantic=True)
        outs = []
        for _ in range(reps):
            e = L.Engine(ARCH, B, H, W, dev)
            e.set_conv_algo(algo)
            e.load_state_dict(sd)
            e.zero_grad()
            sc = e.pair_step(sample, indices=None, seed=5, train=True).clone()   # device-sampled matches: same seed, same points
            sc2 = sc
            if second_step:
                e.adam_step(1e-3)
                sc2 = e.pair_step(sample, indices=None, seed=6, train=True).clone()  # a second step on the moved weights
            torch.cuda.synchronize()
            outs.append({"scalars": sc.cpu(), "scalars2": sc2.cpu(), "grads": e.grads.cpu().clone(), "params": e.params.cpu().clone(),
                         "adam_m": e.adam_m.cpu().clone(), "adam_v": e.adam_v.cpu().clone(), "bn_running": e.bn_running.cpu().clone()})
            del e
        return outs
    finally:
        L.set_deterministic(False)


@pytest.mark.parametrize("algo", [1, 12])
def test_pair_step_is_bit_identical_across_runs_in_deterministic_mode(algo):
    """fp32 path (Winograd kernels, fused BatchNorm-backward reductions) and bf16 path, at a size where every accumulator is hit
    by many blocks (B = 4, 240x320: 1200 tiles per 3x3 launch, 1000 matches x 100 non-matches per image)."""
    outs = _run(algo, 4, 240, 320, True)
    for k in outs[0]:
        for rep in (1, 2):
            same = torch.equal(outs[0][k], outs[rep][k])
            if not same:
                d = (outs[0][k].double() - outs[rep][k].double()).abs()
                raise AssertionError("algo %d: %s differs between runs 0 and %d: %d elements, max |diff| %.3e"
                                     % (algo, k, rep, int((d > 0).sum()), float(d.max())))
    assert torch.isfinite(outs[0]["grads"]).all() and float(outs[0]["grads"].abs().max()) > 0


def test_deterministic_mode_agrees_with_the_default_mode():
    """The quantised accumulation changes no result beyond the run-to-run noise of the default mode: scalars 1e-6, gradients
    1e-5 of the largest one (the same budget the parity tests give the atomics).  One step, no optimizer step in between: Adam's
    first update is lr * sign(g) wherever |g| is small, which turns a 1e-7 difference of a gradient into 2e-3 of a weight."""
    from semantic_superpoint_amd import lib as L
    a = _run(1, 2, 120, 160, True, second_step=False, reps=1)[0]
    b = _run(1, 2, 120, 160, False, second_step=False, reps=1)[0]
    assert float((a["scalars"] - b["scalars"]).abs().max()) <= 1e-6 * max(1.0, float(b["scalars"].abs().max()))
    layout, n_params = L.param_layout(ARCH, 133)
    worst = 0.0
    for name, shape, off in layout:
        n = 1
        for d in shape:
            n *= d
        ga, gb = a["grads"][off:off + n], b["grads"][off:off + n]
        m = float(gb.abs().max())
        if m < 1e-6:   # (mathematically zero: a conv bias in front of a BatchNorm)
            continue
        r = float((ga - gb).abs().max()) / m
        worst = max(worst, r)
        assert r <= 1e-5, (name, r)
    print("deterministic vs default accumulation: worst per-tensor max |diff| / max |g| = %.2e" % worst)
