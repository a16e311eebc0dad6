"""GPU suite (-m gpu): the same checks as the CPU emulator suite, on libmapnet_hip.so through the C
ABI on a real MI355X, at small sizes against the oracle / torch fp64 and at BASELINE.json's full
sizes through size-independent properties (adjoint identities, finite decreasing loss)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest
import torch

import checks

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from geomapnet_amd import _binding
    assert torch.cuda.is_available()
    b = _binding.hip()
    assert b.backend_name == "hip"
    return b


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [
    (2, 9, 11, 64, 64, 3, 1, 1), (2, 9, 11, 64, 128, 3, 2, 1), (1, 8, 10, 64, 128, 1, 2, 0), (3, 5, 6, 128, 192, 3, 1, 1),
    (2, 64, 86, 64, 64, 3, 1, 1),      # layer1 geometry at 256x341
    (2, 32, 43, 128, 256, 3, 2, 1),    # layer3.0.conv1
    (4, 8, 11, 512, 512, 3, 1, 1),     # layer4
])
def test_conv_forward(lib, dtype, shape):
    checks.check_conv_fwd(lib, DEV, dtype, *shape)


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [
    (2, 9, 11, 64, 64, 3, 1, 1), (2, 9, 11, 64, 128, 3, 2, 1), (2, 8, 10, 64, 128, 1, 2, 0), (2, 7, 9, 64, 128, 1, 2, 0),
    (2, 64, 86, 64, 128, 3, 2, 1), (2, 16, 22, 256, 512, 1, 2, 0), (4, 8, 11, 512, 512, 3, 1, 1),
])
def test_conv_data_gradient(lib, dtype, shape):
    checks.check_conv_dgrad(lib, DEV, dtype, *shape)


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
@pytest.mark.parametrize("shape,mode", [
    ((2, 8, 11, 64, 128, 3, 2, 1), "plain"), ((2, 9, 10, 64, 128, 3, 2, 1), "out_gate"), ((1, 8, 12, 64, 128, 3, 2, 1), "res_gate"),
    ((2, 8, 11, 64, 128, 1, 2, 0), "inplace"), ((2, 9, 11, 64, 64, 3, 1, 1), "out_gate"),
    ((3, 64, 86, 64, 128, 3, 2, 1), "plain"),      # layer2.0.conv1 geometry
    ((3, 32, 43, 128, 256, 3, 2, 1), "out_gate"),  # layer3.0.conv1 (odd width 43)
    ((3, 32, 43, 128, 256, 1, 2, 0), "inplace"),   # layer3.0.downsample
    ((24, 16, 22, 256, 256, 3, 1, 1), "out_gate"),  # 8448 rows x N = 256: the 288x256 one-round configuration
])
@pytest.mark.parametrize("parity", [1, 0])
def test_conv_data_gradient_op(lib, dtype, shape, mode, parity):
    checks.check_conv_dgrad_op(lib, DEV, dtype, *shape, parity=parity, mode=mode)


@pytest.mark.parametrize("wgs", [0, 7, 64])
@pytest.mark.parametrize("mode", ["plain", "res_gate", "out_gate"])
def test_conv_halo_pp(lib, mode, wgs):
    """halo_pp.h, the persistent two-wave-group kernel that is layer1's only forward / data-gradient path in the fp16 mode, as
    an operator on the hardware: layer1 geometry (rows of 86 pixels, several images), ragged small shapes, every epilogue
    variant, several persistent-workgroup counts (0 = the launcher's choice; 7 and 64 = tiles per workgroup that do / do not
    divide evenly), forward with BatchNorm sums and data gradient"""
    for (B, H, W) in ((2, 64, 86), (3, 9, 11), (1, 33, 70)):
        checks.check_conv_halo(lib, DEV, B, H, W, Cout=64, dgrad=False, mode="plain", seed=H * 100 + W, pp_wgs=wgs)
        checks.check_conv_halo(lib, DEV, B, H, W, Cout=64, dgrad=True, mode=mode, seed=H * 100 + W + 1, pp_wgs=wgs)


def test_layer1_h2_convolution_with_the_weights_in_registers(lib):
    """halo_h2.h (round 6; layer1's forward convolutions in the fp16x2 / fp16x2m modes): one persistent 4-wave workgroup per CU keeps
    the hi and lo halves of the weights in registers.  Layer1 geometry (rows of 86 pixels), ragged small shapes, and a size with 256
    concurrent workgroups repeated as a race screen of its double-buffered halo DMA"""
    for (B, H, W) in ((2, 64, 86), (3, 9, 11), (1, 33, 70)):
        checks.check_conv_halo_h2(lib, DEV, B, H, W, seed=H * 100 + W)
    for rep in range(4):
        checks.check_conv_halo_h2(lib, DEV, 24, 64, 86, seed=600 + rep)


def test_conv_halo_pp_race_screen(lib):
    """the same kernel at a size with hundreds of concurrent persistent workgroups, repeated: its DMA / two-group hand-over has
    no other detector on the hardware (the emulator runs workgroups one at a time)"""
    for rep in range(4):
        checks.check_conv_halo(lib, DEV, 24, 64, 86, Cout=64, dgrad=False, mode="plain", seed=300 + rep)
        checks.check_conv_halo(lib, DEV, 24, 64, 86, Cout=64, dgrad=True, mode="out_gate", seed=400 + rep)
        checks.check_conv_halo(lib, DEV, 24, 64, 86, Cout=64, dgrad=True, mode="res_gate", seed=500 + rep, pp_wgs=100)


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
@pytest.mark.parametrize("shape,blocks", [
    ((2, 9, 11, 64, 64, 3, 1, 1), 8), ((3, 9, 11, 64, 128, 3, 2, 1), 8), ((2, 8, 10, 64, 128, 1, 2, 0), 1),
    ((5, 5, 6, 128, 128, 3, 1, 1), 40), ((2, 64, 86, 64, 64, 3, 1, 1), 1024), ((4, 8, 11, 512, 512, 3, 1, 1), 1024),
    ((3, 7, 9, 128, 128, 3, 1, 1), 40), ((7, 16, 22, 256, 256, 3, 1, 1), 1024),
])
def test_conv_weight_gradient(lib, dtype, shape, blocks):
    checks.check_conv_wgrad(lib, DEV, dtype, *shape, target_blocks=blocks)


@pytest.mark.parametrize("shape", [(2, 9, 11, 64, 64, 3, 1, 1), (3, 20, 22, 128, 64, 3, 1, 1), (2, 6, 7, 72, 80, 3, 1, 1),
                                   (6, 64, 86, 64, 64, 3, 1, 1), (12, 32, 43, 128, 128, 3, 1, 1), (24, 16, 22, 256, 256, 3, 1, 1),
                                   (48, 8, 11, 512, 512, 3, 1, 1)])
@pytest.mark.parametrize("dtype", [1, 2, 3])
def test_fused_weight_gradient_through_workspace(lib, dtype, shape):
    """wgrad_fused.h (fp16 kernel / fp32x3 kernel / h2 kernel) with partial tiles stored to a (NaN-filled) workspace and added
    up by the reduce kernel, as the plan runs it (layer geometries included); bit-identical between two launches when one
    reduction group covers the columns"""
    if dtype == 3 and shape[3] % 32:  # h2 tensors hold whole 32-channel groups
        shape = shape[:3] + (96, 160) + shape[5:]
    checks.check_conv_wgrad(lib, DEV, dtype, *shape, ws=True)


@pytest.mark.parametrize("dtype", [0, 1, 2])
@pytest.mark.parametrize("hw", [(20, 27), (21, 26), (256, 341)])
def test_stem_conv(lib, dtype, hw):
    checks.check_stem(lib, DEV, dtype, 2, *hw)


@pytest.mark.parametrize("shape", [(6, 128, 128), (37, 256, 200), (65, 128, 70), (192, 512, 2048)])
def test_pose_head_dense_layer(lib, shape):
    """csrc/dense.h: the head's fc layer forward / data gradient / weight + bias gradient in 32 x 32 tiles and the pose
    regressors' weight gradient, ragged rows and feature counts, against torch fp64"""
    checks.check_dense(lib, DEV, *shape)


@pytest.mark.parametrize("shape", [(1, 20, 27), (2, 33, 70), (3, 256, 341)])
def test_stem_backward_two_launch_form(lib, shape):
    """csrc/stem_bwd.h against the maxpool_bwd -> bn_bwd -> wgrad chain it replaces, identical fp16 tensors"""
    checks.check_stem_bwd(lib, DEV, *shape)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("M,C,kw", [(300, 64, dict()), (77, 128, dict(with_res=False)), (130, 512, dict(relu=False, with_res=False)),
                                    (2 * 64 * 86, 64, dict())])
def test_batchnorm(lib, dtype, M, C, kw):
    checks.check_bn(lib, DEV, dtype, M, C, **kw)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("hw,ties", [((8, 11), False), ((9, 10), True), ((128, 171), True)])
def test_maxpool(lib, dtype, hw, ties):
    checks.check_maxpool(lib, DEV, dtype, 2, hw[0], hw[1], 64, ties=ties)


def test_criteria_against_reference_golden(lib, golden_dir):
    checks.check_criterion_golden(lib, DEV, golden_dir)


def test_calc_vos_against_reference_golden(lib, golden_dir):
    checks.check_calc_vos_golden(lib, DEV, golden_dir)


@pytest.mark.parametrize("max_norm", [0.0, 5.0])
def test_fused_adam(lib, max_norm):
    checks.check_adam(lib, DEV, n=1000003, max_norm=max_norm)


# ---- whole training step vs the oracle, identical synthetic batches and weights -----------------------
def test_mapnet_train_step_fp32_parity_small(lib):
    checks.check_train_step(lib, DEV, "fp32", mode="mapnet", N=2, H=64, W=85, steps=2, loss_rtol=1e-4, pose_atol=2e-3)


@pytest.mark.parametrize("mask", ["0", "3"])
def test_mapnet_train_step_fp32_parity_stem_variants(lib, monkeypatch, mask):
    """MN_FUSE_STEM: 0 = separate BatchNorm / max-pool passes, 3 = fused forward AND max-pool gradient gathered inside
    the BatchNorm backward (the default, 1, is what every other test runs)"""
    monkeypatch.setenv("MN_FUSE_STEM", mask)
    checks.check_train_step(lib, DEV, "fp32", mode="mapnet", N=2, H=64, W=85, steps=1, loss_rtol=1e-4, pose_atol=2e-3)


def test_mapnet_train_step_fp32_parity_one_weight_gradient_fork_per_block(lib, monkeypatch):
    """MN_EARLY_FORK=0: one weight-gradient fork per block instead of the deferred schedule"""
    monkeypatch.setenv("MN_EARLY_FORK", "0")
    checks.check_train_step(lib, DEV, "fp32", mode="mapnet", N=2, H=64, W=85, steps=2, loss_rtol=1e-4, pose_atol=2e-3)


@pytest.mark.parametrize("sched", ["0", "1", "2"])
@pytest.mark.parametrize("dtype", ["fp16x2m", "fp16x2", "fp16"])
def test_mapnet_train_step_under_every_weight_gradient_schedule(lib, monkeypatch, dtype, sched):
    """MN_WGRAD_SCHED (read per plan): one fork per block / a fork as soon as d(conv output) exists / deferred to the next
    BatchNorm-backward pass.  The defaults differ by mode since round 4 (fp16 0, fp16x2 1, fp32 2): every mode must be parity-green
    under every order -- the side stream only reads tensors that live until the stage's join."""
    monkeypatch.setenv("MN_WGRAD_SCHED", sched)
    if dtype in ("fp16x2", "fp16x2m"):
        checks.check_train_step(lib, DEV, dtype, mode="mapnet", N=2, H=64, W=85, steps=2, loss_rtol=1e-4, pose_atol=2e-3)
    else:
        # (fp16: 1.5x what the mode measures at this shape, gradients included -- checks.FP16_SMALL)
        checks.check_train_step(lib, DEV, "fp16", mode="mapnet", N=2, H=64, W=85, steps=2, **checks.fp16_small_gates(2, 64, 85))


@pytest.mark.parametrize("tail", ["0", "3"])
@pytest.mark.parametrize("dtype", ["fp16x2m", "fp16"])
def test_mapnet_train_step_with_layer1_weight_gradients_beside_the_stem_backward(lib, monkeypatch, dtype, tail):
    """MN_WGRAD_TAIL (read per plan; default 1, what every other test runs): the weight gradients of the first k blocks of layer1 are
    forked in front of the stem's backward launches instead of beside their own block's data gradients.  Only the order of launches on
    the side stream changes: gradients and the updated parameters must stay on the oracle's under k = 0 (round 5's order) and k = 3."""
    monkeypatch.setenv("MN_WGRAD_TAIL", tail)
    if dtype == "fp16x2m":
        checks.check_train_step(lib, DEV, dtype, mode="mapnet", N=2, H=64, W=85, steps=2, loss_rtol=1e-4, pose_atol=2e-3)
    else:
        checks.check_train_step(lib, DEV, "fp16", mode="mapnet", N=2, H=64, W=85, steps=2, **checks.fp16_small_gates(2, 64, 85))


def test_mapnet_train_step_fp32_parity_full_resolution(lib):
    """(N=2, T=3, 3, 256, 341): north-star tolerances 1e-4 on loss (relative, |loss| > 1) and 1e-3 on pose"""
    checks.check_train_step(lib, DEV, "fp32", mode="mapnet", N=2, H=256, W=341, steps=1, loss_rtol=1e-4, pose_atol=1e-3,
                            pose_abs=1e-3)


def test_mapnet_train_step_fp32x3_parity_full_resolution(lib):
    """the parity mode on the fast matrix pipe (fp32 tensors, f16x3 forward / bf16x3 backward split-operand contractions):
    north-star tolerances as written, 1e-4 on loss (relative, |loss| > 1) and 1e-3 on pose (max abs), gradients to the
    fp32 build's 2e-2 per tensor"""
    checks.check_train_step(lib, DEV, "fp32x3", mode="mapnet", N=2, H=256, W=341, steps=1, loss_rtol=1e-4, pose_atol=1e-3,
                            pose_abs=1e-3)


def test_mapnet_train_step_fp16x2_parity_full_resolution(lib):
    """the fp16-pair mode (h2 conv operands split once by their producers, three fp16 MFMAs per product on DMA-fed operands,
    fp32 everything else): north-star tolerances as written, gradients to the fp32 build's 2e-2 per tensor"""
    checks.check_train_step(lib, DEV, "fp16x2", mode="mapnet", N=2, H=256, W=341, steps=1, loss_rtol=1e-4, pose_atol=1e-3,
                            pose_abs=1e-3)


def test_mapnet_train_step_fp16x2m_parity_full_resolution(lib):
    """fp16x2m (round 5): the fp16x2 forward pass (its loss and poses, bit for bit) + the fp16 mode's single-MFMA backward pass on
    plain fp16 copies, gates and BatchNorm statistics from the exact forward values: north-star tolerances as written, gradients to
    the fp32-class bar of 2e-2 per tensor"""
    checks.check_train_step(lib, DEV, "fp16x2m", mode="mapnet", N=2, H=256, W=341, steps=1, loss_rtol=1e-4, pose_atol=1e-3,
                            pose_abs=1e-3)


def test_mapnet_train_step_fp16x2m_two_steps_small(lib):
    checks.check_train_step(lib, DEV, "fp16x2m", mode="mapnet", N=2, H=64, W=85, steps=2, loss_rtol=1e-4, pose_atol=2e-3)


def test_mapnet_online_train_step_fp16x2m_parity_clip_and_nan_filter(lib):
    checks.check_train_step(lib, DEV, "fp16x2m", mode="mapnet++", N=2, H=64, W=85, steps=1, max_grad_norm=5.0, lr=1e-5, wd=0.0,
                            filter_nans=True, grad_l2_rtol=None)


def test_posenet_train_step_fp16x2m(lib):
    checks.check_train_step(lib, DEV, "fp16x2m", mode="posenet", N=5, H=64, W=85, steps=1)


@pytest.mark.parametrize("shape", [
    (2, 9, 11, 64, 64, 3, 1, 1), (2, 9, 11, 64, 128, 3, 2, 1), (1, 8, 10, 64, 128, 1, 2, 0), (3, 5, 6, 128, 256, 3, 1, 1),
    (7, 16, 22, 256, 256, 3, 1, 1), (4, 32, 43, 128, 128, 3, 1, 1), (5, 8, 11, 512, 512, 3, 1, 1), (1, 64, 86, 64, 64, 3, 1, 1),
])
def test_conv_forward_with_fp8_cross_terms(lib, shape):
    """h2q operands (the experimental fp16x2q mode's forward convolutions): hi*hi on the fp16 pipe + both cross terms of a K-step in one
    v_mfma_scale_f32_32x32x64_f8f6f4 from the fp8 planes, against torch fp64 on exactly the values the planes stand for -- this pins
    the instruction's operand layout and scale semantics as probed (tools/probes/mfma_scale_probe*.hip) on real layer geometries"""
    checks.check_conv_fwd_h2q(lib, DEV, *shape)


def test_mapnet_train_step_fp16x2q_fp8_cross_terms(lib):
    """the experimental mode's whole step at full resolution: poses inside the north-star bar (at a 2x margin instead of fp16x2m's
    60x), loss inside 1e-3 relative at this 6-image batch (7e-6 at the benchmark batch), gradients inside 15 % per tensor"""
    checks.check_train_step(lib, DEV, "fp16x2q", mode="mapnet", N=2, H=256, W=341, steps=1, loss_rtol=1e-3, pose_atol=1e-3,
                            pose_abs=1e-3, grad_l2_rtol=0.15)


def test_mapnet_train_step_fp16x2_two_steps_small(lib):
    checks.check_train_step(lib, DEV, "fp16x2", mode="mapnet", N=2, H=64, W=85, steps=2, loss_rtol=1e-4, pose_atol=2e-3)


def test_mapnet_online_train_step_fp16x2_parity_clip_and_nan_filter(lib):
    checks.check_train_step(lib, DEV, "fp16x2", mode="mapnet++", N=2, H=64, W=85, steps=1, max_grad_norm=5.0, lr=1e-5, wd=0.0,
                            filter_nans=True, grad_l2_rtol=None)


def test_posenet_train_step_fp16x2(lib):
    checks.check_train_step(lib, DEV, "fp16x2", mode="posenet", N=5, H=64, W=85, steps=1)


def test_mapnet_train_step_fp32x3_two_steps_small(lib):
    checks.check_train_step(lib, DEV, "fp32x3", mode="mapnet", N=2, H=64, W=85, steps=2, loss_rtol=1e-4, pose_atol=2e-3)


def test_mapnet_online_train_step_fp32x3_parity_clip_and_nan_filter(lib):
    checks.check_train_step(lib, DEV, "fp32x3", mode="mapnet++", N=2, H=64, W=85, steps=1, max_grad_norm=5.0, lr=1e-5, wd=0.0,
                            filter_nans=True, grad_l2_rtol=None)


def test_fp16_loss_trajectory_tracks_the_parity_mode(lib):
    """50 Adam steps on one fixed full-resolution batch (8 windows x T=3, 256x341), fp16 against fp32x3, both on the GPU:
    the two loss curves stay within 25 % of the curve's total descent of each other at every step"""
    l16, l32, gap = checks.check_loss_trajectory(lib, DEV, N=8, H=256, W=341, steps=50)
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "loss_trajectory_fp16_vs_fp32x3.json"), "w") as f:
            json.dump({"fp16": l16, "fp32x3": l32, "max_gap_over_descent": gap}, f)
    print("trajectory gap / descent:", gap)


def test_fp16x2m_loss_trajectory_tracks_fp16x2(lib, monkeypatch):
    """the same 50 steps in fp16x2m against fp16x2 under MN_DETERMINISTIC=1 (no summation-order noise: the curves differ by the
    backward pass's arithmetic only): within the envelope the fp16 mode is held to; fp32x3 against fp16x2 -- two fp32-CLASS evaluations
    of the same steps -- is recorded beside it as the scale of what any change of arithmetic does to a trajectory"""
    monkeypatch.setenv("MN_DETERMINISTIC", "1")
    curves, gap, other = checks.check_loss_trajectory(lib, DEV, N=8, H=256, W=341, steps=50, modes=("fp16x2m", "fp16x2"),
                                                      extra=("fp32x3",))
    import json
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "loss_trajectory_fp16x2m_vs_fp16x2.json"), "w") as f:
            json.dump({"curves": curves, "fp16x2m_vs_fp16x2_max_gap_over_descent": gap, "vs_fp16x2": other}, f)
    print("trajectory gap / descent: fp16x2m vs fp16x2", gap, "| fp32x3 vs fp16x2", other)


def test_posenet_train_step_fp32_parity(lib):
    checks.check_train_step(lib, DEV, "fp32", mode="posenet", N=5, H=96, W=128, steps=1)


def test_mapnet_online_train_step_fp32_parity_clip_and_nan_filter(lib):
    checks.check_train_step(lib, DEV, "fp32", mode="mapnet++", N=2, H=64, W=85, steps=1, max_grad_norm=5.0, lr=1e-5, wd=0.0,
                            filter_nans=True, grad_l2_rtol=None)


def test_mapnet_online_gradients_fp32(lib):
    checks.check_train_step(lib, DEV, "fp32", mode="mapnet++", N=2, H=64, W=85, steps=1, lr=1e-5, wd=0.0)


def test_mapnet_gps_train_step_fp32(lib):
    checks.check_train_step(lib, DEV, "fp32", mode="mapnet++", N=2, H=64, W=85, steps=1, gps=True, lr=1e-5, wd=0.0)


def test_mapnet_train_step_fp16_close(lib):
    """fp16 build, two windows (6 images) at full resolution: BatchNorm statistics over so few samples make the storage
    rounding weigh more than at 64 windows (FP16_ENVELOPE).  Held to 1.5x what the mode measures on MI355X at this shape
    (checks.FP16_SMALL, profiles/r06/c3_fp16_deviation_at_the_suite_shapes.txt: loss 6.9e-4 relative, pose 9.6e-3 max abs, worst-tensor
    gradient 0.355) -- until round 6 the gates were 1e-2 / 3e-2 and no gradient check, i.e. a 3x regression passed."""
    checks.check_train_step(lib, DEV, "fp16", mode="mapnet", N=2, H=256, W=341, steps=1, **checks.fp16_small_gates(2, 256, 341))


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_nan_filter_with_a_nan_cotangent(lib, dtype):
    """filter_hook (models/posenet.py:28-34,50-51) fed a real NaN d(pred) through head_bwd_*: d(input), d(weight), d(bias)
    vs the oracle's autograd + hooks; without the filter NaN reaches the same parameters"""
    checks.check_nan_filter(lib, DEV, dtype, N=2, H=64, W=85)


@pytest.mark.parametrize("dtype", ["fp32", "fp16x2"])
def test_dropout_on_the_device_with_the_oracle_applying_the_same_mask(lib, dtype):
    """models/posenet.py:68-69 with droprate = 0.5 (what every shipped config asks for): device Philox mask, oracle fed the same mask"""
    checks.check_dropout(lib, DEV, dtype, N=2, H=64, W=85)


def test_fp16_trains_to_the_accuracy_of_the_parity_mode():
    """BASELINE's metric names "median t/q err": a learnable synthetic scene (data.RenderedFrames) trained for 1280 steps through
    scripts/train.py -> scripts/eval.py in fp16 and in the parity mode (fp16x2m), FIVE seeds each, under MN_DETERMINISTIC=1 -- every
    step is bit-reproducible, so a (mode, seed) pair always trains to the same numbers and the two modes differ by their arithmetic
    only, not by the summation order of atomics (rounds 3-4 compared single irreproducible runs with a 4x margin).  Asserted on the
    MEANS over the seeds: both modes learn (below 0.4 of what predicting the mean training pose scores: 1.00 / 35.6 deg) and fp16 stays
    within 1.3x of the parity mode on both numbers.  The margin is 1.5 sigma of the ratio of two five-seed means (single seeds spread
    with a standard deviation of ~30 % in either mode; recorded run, profiles/r05/accuracy_deterministic_five_seeds.json: fp16
    0.216 / 10.1 deg, fp16x2m 0.205 / 10.7 deg, fp16x2 0.253 / 13.2 deg)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import accuracy_eval
    seeds = (7, 8, 9, 10, 11)
    res, mean = {}, {}
    for d in ("fp16", "fp16x2m"):
        res[d] = []
        for sd in seeds:
            r, base = accuracy_eval.train_and_eval(d, 40, 512, 128, 64, 85, 16, 1e-3, seed=sd, deterministic=True)
            res[d].append(dict(r, seed=sd))
        mean[d] = {k: sum(r[k] for r in res[d]) / len(seeds) for k in ("median_t", "median_q")}
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "accuracy_fp16_vs_fp16x2m_five_seeds.json"), "w") as f:
            json.dump({"runs": res, "means": mean, "baseline_predict_mean": base}, f)
    for d in mean:
        assert mean[d]["median_t"] < 0.4 * base["median_t"] and mean[d]["median_q"] < 0.4 * base["median_q"], (d, mean[d], base)
    assert mean["fp16"]["median_t"] <= 1.3 * mean["fp16x2m"]["median_t"], mean
    assert mean["fp16"]["median_q"] <= 1.3 * mean["fp16x2m"]["median_q"], mean


def test_fp16_overflow_skips_the_step_and_lowers_the_scale(lib):
    checks.check_overflow_skip(lib, DEV, N=2, H=64, W=85, more=3)


@pytest.mark.parametrize("dtype_name", ["fp16x2", "fp16x2m"])
def test_parity_mode_overflow_skips_the_step_and_the_recovery_step_meets_the_bar(lib, dtype_name):
    """the parity modes' gradients pass through fp16 halves under a loss scale: an overflowed step is skipped, the scale backs off,
    and the first applied step afterwards is still the oracle's first step to 1e-4 / 1e-3"""
    checks.check_overflow_skip(lib, DEV, N=2, H=64, W=85, more=3, dtype_name=dtype_name)


@pytest.mark.parametrize("dtype_name", ["fp16", "fp16x2m"])
def test_overflow_bookkeeping_acts_on_completed_attempts_not_on_polls(lib, dtype_name):
    checks.check_overflow_progress_accounting(lib, DEV, dtype_name, N=2, H=64, W=85)


# ---- BASELINE.json configurations at FULL size against the oracle, fp32 (north-star bar asserted) and fp16 (recorded) --
def _record_parity(rec):
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_full_size.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    print("full-size parity:", json.dumps(rec))


def test_full_size_parity_configs2_mapnet_64_windows(lib):
    """BASELINE configs[2], the benchmarked configuration: (64, T=3, 3, 256, 341), MapNetCriterion, Adam"""
    _record_parity(checks.check_full_size_parity(lib, DEV, "mapnet", 64))


def test_full_size_parity_configs1_posenet_batch_64(lib):
    """BASELINE configs[1]: PoseNet, batch 64, 256x341, absolute-pose loss only"""
    _record_parity(checks.check_full_size_parity(lib, DEV, "posenet", 64))


def test_full_size_parity_configs4_mapnet_online_64_windows(lib):
    """BASELINE configs[4] shape per GPU: MapNet++ (64 windows x 2T = 384 images), lr 1e-5, clip 5, NaN filter"""
    import psutil
    if psutil.virtual_memory().available < 200e9:
        pytest.skip("the oracle's autograd graph of 384 full-size images needs ~100 GB of host memory")
    _record_parity(checks.check_full_size_parity(lib, DEV, "mapnet++", 64, max_grad_norm=5.0, lr=1e-5, wd=0.0, filter_nans=True))


def test_mapnet_staged_step_fp32_parity_with_rccl(lib, monkeypatch):
    """the data-parallel form of the step on one GPU: forward / 4 backward stages / optimiser as separate captured
    segments with a (single-rank) RCCL all-reduce of every gradient bucket in between"""
    import torch.distributed as dist
    import geomapnet_amd.train as T
    monkeypatch.setattr(T, "_FORCE_STAGED", True)
    started = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29613", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        started = True
    try:
        import geomapnet_amd.dp as dp
        # the parity mode of round 5 first (fp16x2m: loss scale, stage joins of its fp16 backward launches), unprofiled
        checks.check_train_step(lib, DEV, "fp16x2m", mode="mapnet", N=2, H=64, W=85, steps=2, loss_rtol=1e-4, pose_atol=2e-3)
        dp.set_profiling(True)
        checks.check_train_step(lib, DEV, "fp32", mode="mapnet", N=2, H=64, W=85, steps=2, loss_rtol=1e-4, pose_atol=2e-3)
        # the measurement hooks bench.py --gpus N reports: exposed communication per step and the per-bucket timeline
        assert len(dp.exposed_comm_ms()) == 2
        tl = dp.bucket_timeline_ms()
        assert len(tl) == 2 and sorted(tl[0]) == [0, 1, 2, 3]
        for st in (3, 2, 1, 0):
            ready, passed = tl[0][st]
            assert 0.0 <= ready <= passed
        assert tl[0][3][0] <= tl[0][2][0] <= tl[0][1][0] <= tl[0][0][0]  # buckets become ready in backward order
        dp.set_profiling(False)
    finally:
        if started:
            dist.destroy_process_group()


def test_eval_forward_parity(lib):
    checks.check_eval_forward(lib, DEV, "fp32", B=3, H=128, W=171)
    checks.check_eval_forward(lib, DEV, "fp16x2m", B=3, H=128, W=171)  # (the scripts' default dtype: the split-operand forward pass)
    checks.check_eval_forward(lib, DEV, "fp16", B=3, H=128, W=171, atol=2.5e-3)  # (1.5x the measured 1.66e-3 of the pose scale; was 3e-2)


def test_checkpoint_interop_and_resume(lib):
    checks.check_checkpoint_interop(lib, DEV, H=64, W=85)


def test_uint8_input_pipeline(lib):
    checks.check_u8_input(lib, DEV, N=2, H=128, W=171)
    checks.check_u8_input(lib, DEV, N=2, H=128, W=171, dtype_name="fp16x2m")


@pytest.mark.parametrize("dtype,u8", [("fp16", False), ("fp16x2m", True)])
def test_device_feed_prefetch_cannot_race_the_step(lib, dtype, u8):
    """DeviceFeed (geomapnet_amd/feed.py; Trainer's input path): batch k+1 is copied from pinned host memory on a copy stream while
    step k runs.  Eight distinct 24-image full-resolution batches through two rotating staging buffers against synchronous copies,
    bit for bit under MN_DETERMINISTIC=1 (losses of every step and the final parameters)"""
    checks.check_device_feed(lib, DEV, dtype, N=8, H=256, W=341, batches=8, u8=u8)


def test_staged_step_with_stand_in_collectives_and_deferred_buckets_computes_the_same_bits(lib):
    """the one-GPU rehearsal of the 8-GPU step (MN_DP_STANDIN: the library's occupancy stand-in on a communication stream where each
    bucket's all-reduce is issued; MN_DP_DEFER=1: buckets 2..0 after the last backward stage) is a change of SCHEDULE: the staged step
    fed by DeviceFeed must compute the bits of the plain fused step"""
    checks.check_device_feed(lib, DEV, "fp16", N=4, H=128, W=171, batches=4,
                             staged_env={"MN_FORCE_STAGED": "1", "MN_DP_STANDIN": "16,256,200,40", "MN_DP_DEFER": "1"})


def test_eval_flow_and_metric(lib):
    """scripts/eval.py flow on synthetic windows: median / mean translation and rotation error"""
    checks.check_eval_flow(lib, DEV, "fp32", L=8, T=3, H=128, W=171)
    checks.check_eval_flow(lib, DEV, "fp16", L=8, T=3, H=128, W=171, rtol=4.6e-3, check_q=False)  # (1.5x the measured 3.05e-3; was 3e-2)


# ---- BASELINE full size: size-independent properties -----------------------------------------------------
@pytest.mark.parametrize("dtype", [1, 0])
def test_conv_adjoint_identity_full_size(lib, dtype):
    """<conv(x, w), gy> = <x, dgrad(gy, w)> = <w, wgrad(gy, x)> at B=192, layer2 geometry (32x43, 128 ch)"""
    from geomapnet_amd._binding import ptr
    td = checks.TD[dtype]
    B, H, W, Ci, Co, k = 192, 32, 43, 128, 128, 3
    g, Ho, Wo = checks.fwd_geom(B, H, W, Ci, Co, k, 1, 1)
    gd, _, _ = checks.dgrad_geom(B, H, W, Ci, Co, k, 1, 1)
    gen = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(B, H, W, Ci, device=DEV, generator=gen).to(td)
    w = (torch.randn(Co, k, k, Ci, device=DEV, generator=gen) * 0.05).to(td)
    gy = torch.randn(B, Ho, Wo, Co, device=DEV, generator=gen).to(td)
    wt = w.permute(3, 1, 2, 0).contiguous()
    y = torch.zeros(B, Ho, Wo, Co, dtype=td, device=DEV)
    gx = torch.zeros(B, H, W, Ci, dtype=td, device=DEV)
    gw = torch.zeros(Co, k * k * Ci, device=DEV)
    one = C.c_float(1.0)
    lib.check(lib.op_igemm(dtype, C.byref(g), ptr(x), ptr(w), ptr(y), Co, None, None, 0, None, None, one, ptr(checks.zero_page("cuda")), None))
    lib.check(lib.op_igemm(dtype, C.byref(gd), ptr(gy), ptr(wt), ptr(gx), Ci, None, None, 0, None, None, one, ptr(checks.zero_page("cuda")), None))
    lib.check(lib.op_wgrad(dtype, C.byref(g), ptr(gy), Co, ptr(x), ptr(gw), k * k * Ci, None, one, 1024, ptr(checks.zero_page("cuda")), None))
    torch.cuda.synchronize()
    a = (y.double() * gy.double()).sum().item()
    b = (x.double() * gx.double()).sum().item()
    c = (w.double().reshape(Co, -1) * gw.double()).sum().item()
    scale = (y.double().norm() * gy.double().norm()).item()
    tol = 3e-3 if dtype == 1 else 3e-6
    assert abs(a - b) <= tol * scale and abs(a - c) <= tol * scale, (a, b, c, scale)


def test_full_size_training_is_finite_and_learns(lib):
    """BASELINE configs[2] shape (64 windows x T=3, 256x341, fp16): 6 steps on one fixed batch"""
    import geomapnet_amd as G
    G.set_compute_dtype("fp16")
    torch.manual_seed(7)
    net = G.MapNet(G.PoseNet(G.resnet34(), droprate=0.0, pretrained=False)).cuda()
    crit = G.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True).cuda()
    opt = G.Optimizer([{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}],
                      "adam", base_lr=1e-4, weight_decay=5e-4)
    net.train()
    import oracle
    x, t = oracle.make_batch("mapnet", 64, 256, 341, seed=7)
    x, t = x.cuda(), t.cuda()
    losses = []
    for _ in range(6):
        l, out = G.step_feedfwd(x, net, True, t, crit, opt, True)
        assert np.isfinite(l) and torch.isfinite(out).all()
        losses.append(l)
    assert out.shape == (64, 3, 6)
    assert losses[-1] < losses[0], losses
    sd = net.state_dict()
    assert all(torch.isfinite(v).all() for v in sd.values() if v.dtype == torch.float32)
    assert int(sd["mapnet.feature_extractor.bn1.num_batches_tracked"]) == 6


@pytest.mark.parametrize("mode,N,T", [("mapnet", 1, 2), ("mapnet", 3, 5), ("mapnet", 2, 7), ("online", 1, 2), ("online", 3, 4),
                                      ("gps", 2, 5), ("posenet", 1, 1), ("mapnet", 70, 4), ("online", 33, 3), ("mapnet", 512, 3)])
def test_criteria_vs_oracle_other_window_lengths(lib, mode, N, T):
    checks.check_criterion_vs_oracle(lib, DEV, mode, N, T)


def test_pose_graph_golden(lib, golden_dir):
    checks.check_pgo_golden(lib, DEV, golden_dir)


@pytest.mark.parametrize("N,fc,sig", [(7, False, (1.0, 1.0, 1.0, 1.0)), (7, True, (0.5, 2.0, 20.0, 20.0)), (2, False, (1.0, 1.0, 1.0, 1.0)),
                                      (12, True, (1.0, 1.0, 2.0, 2.0)), (3, True, (1.0, 1.0, 1.0, 1.0))])
def test_pose_graph_vs_oracle(lib, N, fc, sig):
    checks.check_pgo_vs_oracle(lib, DEV, W=16, N=N, fc=fc, sig=sig)


@pytest.mark.parametrize("fc", [False, True])
def test_pose_graph_properties_at_eval_set_scale(lib, fc):
    checks.check_pgo_properties(lib, DEV, W=4096, N=7, fc=fc)


def test_forced_288x256_configuration():
    """the 12-wave 288x256 tile on small ragged problems against torch fp64 (tests/forced_config_cases.py), in a
    process of its own because the configuration knob is read once"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MN_IGEMM_CONFIG="12")
    subprocess.run([sys.executable, os.path.join(here, "forced_config_cases.py"), "hip"], check=True, env=env, timeout=600)


def test_train_and_eval_command_lines(tmp_path):
    """scripts/train.py -> checkpoint -> scripts/eval.py (plain and --pose_graph) on the product library: the reference's
    command lines and Trainer loop, synthetic frames.  Kept last in this file."""
    import configparser
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    import train as train_script
    import eval as eval_script

    def config(name, **over):
        s = configparser.ConfigParser()
        s.read(os.path.join(root, "scripts", "configs", name))
        for k, v in over.items():
            s["hyperparameters" if k in ("skip", "steps") else "training"][k] = str(v)
        fn = str(tmp_path / name)
        with open(fn, "w") as f:
            s.write(f)
        return fn

    cfg = config("synthetic_mapnet.ini", n_epochs=2, batch_size=8, snapshot=1, val_freq=1, skip=2)
    args = train_script.build_parser().parse_args(
        ["--model", "mapnet", "--config_file", cfg, "--learn_beta", "--learn_gamma", "--dtype", "fp16", "--synthetic_length",
         "32", "--synthetic_val_length", "8", "--height", "64", "--width", "85", "--num_workers", "0", "--logdir",
         str(tmp_path / "logs")])
    lines = []
    tr = train_script.run(args, log=lines.append)
    # (the validation loss itself is not asserted: eval-mode BatchNorm on a randomly initialised network after a handful
    # of steps produces huge activations, checks.check_eval_flow)
    assert sum(l.startswith("Val ") and "val_loss" in l for l in lines) == 2
    ck = torch.load(tr.final_checkpoint, weights_only=False)
    assert ck["epoch"] == 2 and ck["optim_state_dict"]["state"][0]["step"] == 8
    assert all(torch.isfinite(v).all() for v in ck["model_state_dict"].values() if v.dtype == torch.float32)

    eargs = eval_script.build_parser().parse_args(
        ["--model", "mapnet", "--config_file", cfg, "--weights", tr.final_checkpoint, "--dtype", "fp32", "--synthetic_length",
         "16", "--height", "64", "--width", "85", "--val"])
    summary, pred, targ = eval_script.run(eargs, log=lines.append)
    assert pred.shape == (16, 7) and np.isfinite(pred).all() and np.isfinite(summary["median_t"])
    pcfg = config("synthetic_pose_graph.ini", skip=1, steps=5)
    eargs = eval_script.build_parser().parse_args(
        ["--model", "mapnet", "--config_file", pcfg, "--weights", tr.final_checkpoint, "--dtype", "fp32", "--synthetic_length",
         "16", "--height", "64", "--width", "85", "--pose_graph"])
    summary, pred, targ = eval_script.run(eargs, log=lines.append)
    assert pred.shape == (16, 7) and np.isfinite(pred).all()
    np.testing.assert_allclose(np.linalg.norm(pred[:, 3:], axis=1), 1.0, atol=1e-9)


def _run_forced(env, default_path):
    """kernel variants selected by knobs that are read once run in a process of their own.  default_path=True: the knobs
    only pin what the product runs anyway (race screens of the default kernels at layer geometries) -- a failure FAILS the
    suite.  default_path=False: an off-by-default variant, parity-checked in the emulator; a failure is reported as xfail
    with the output's tail: the variant is not ready, the product path (the rest of this file) has not lost parity."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    knobs = {k: v for k, v in env.items() if k.startswith("MN_")}
    report = pytest.fail if default_path else pytest.xfail
    try:
        r = subprocess.run([sys.executable, os.path.join(here, "forced_config_cases.py"), "hip"], env=env, timeout=300,
                           capture_output=True, text=True)
    except subprocess.TimeoutExpired:
        report("kernel variant %s did not finish in 300 s" % knobs)
    if r.returncode != 0:
        report("kernel variant %s failed on the GPU:\n%s" % (knobs, (r.stdout + r.stderr)[-1500:]))


def test_weight_gradient_with_assembly_transpose_reads():
    """the plain-GEMM fp16 weight-gradient kernel (MN_WGRAD_FUSED=0 routes the 3x3 layers to it) with its transpose reads
    issued from inline assembly and hand-placed waits, against torch fp64 at small and layer-sized shapes, repeated."""
    # (the kernel the stride-2 / 1x1 layers run by default)
    _run_forced(dict(os.environ, MN_WGRAD_CASES="1", MN_WGRAD_FUSED="0"), True)


def test_fused_weight_gradient_race_screen():
    """wgrad_fused.h (default for the 3x3 stride-1 layers) at layer geometries with hundreds of concurrent workgroups,
    repeated, in a process of its own"""
    _run_forced(dict(os.environ, MN_WGRAD_CASES="1", MN_WGRAD_FUSED="1"), True)


def test_chunk_resident_a_kernel_race_screen():
    """igemm_halo.h (the default for the 3x3 stride-1 convolutions of layers 2-4) against torch fp64, plus layer2-4
    geometries repeated, in a process of its own: both tile shapes"""
    _run_forced(dict(os.environ, MN_IGEMM_CONFIG="12", MN_IGEMM_HALO="1"), True)  # the 256-column shape (layer3)
    env = dict(os.environ, MN_IGEMM_HALO="2", MN_HALO384="0")                      # the 128-column shape (layers 2 and 4), 288 rows
    env.pop("MN_IGEMM_CONFIG", None)
    _run_forced(env, True)
    _run_forced(dict(env, MN_HALO384="2"), True)                                  # ... its 8-wave 384-row form
    _run_forced(dict(env, MN_HALO_A1="2"), True)                                  # ... and layer2's two-workgroup single-image shape


def test_chunk_resident_a_kernel_h2_race_screen():
    """igemm_halo.h with h2 operands (the 3x3 stride-1 convolutions of layers 2-4 in the fp16x2 mode): both tile shapes against
    torch fp64, layer geometries repeated"""
    _run_forced(dict(os.environ, MN_H2_CASES="1", MN_H2_HALO256="0", MN_HALO_A1="0"), True)
    _run_forced(dict(os.environ, MN_H2_CASES="1", MN_H2_HALO256="0", MN_HALO_A1="2"), True)
    _run_forced(dict(os.environ, MN_H2_CASES="1", MN_H2_HALO256="1"), True)


@pytest.mark.parametrize("dtype_name", ["fp16", "fp32", "fp32x3", "fp16x2", "fp16x2m"])
def test_deterministic_mode_is_bit_reproducible(lib, dtype_name):
    """MN_DETERMINISTIC=1: three MapNet training steps (clipping on) twice from the same state -> identical bits; the
    default mode's atomics only differ from it by summation order"""
    diff = checks.check_deterministic(lib, DEV, dtype_name, N=2, H=64, W=85, steps=3)
    assert diff < 1e-3


def test_deterministic_mode_benchmark_shape(lib):
    """... at the benchmark's tensor shapes (192 images of 256 x 341: layer1's 4608 accumulator rows, 8-wave fused weight
    gradients, persistent layer1 kernels)"""
    diff = checks.check_deterministic(lib, DEV, "fp16", N=64, H=256, W=341, steps=2)
    assert diff < 1e-3


def test_second_step_fp32_with_a_smooth_update(lib):
    """Two steps with Adam's epsilon at 1.0 (update ~ lr * m, smooth in the gradient): the SECOND step -- which starts from
    the first step's moments, bias corrections, repacked weights and running statistics -- is held to loss 2e-3 / pose 3e-2
    relative to the pose scale and the parameters' total displacement over both steps to 20 % of the oracle's.  Why not
    tighter: the first-step GRADIENTS of two correct fp32 evaluations differ by ~1 % wherever one ReLU input within 1e-5 of
    zero takes the other sign (tools/gate_margin.py: the fp32 oracle flips one gate against the fp64 oracle in this very
    batch; tools/grad_accuracy.py: the HIP build's gradients agree with fp64 to 1e-5 above the flipped gates and to 1e-2
    below), and the network amplifies that ~30x per step (measured here: 7e-4 / 5.8e-2 absolute at step 2).  (With eps = 1e-8 a later step can only be compared loosely: the oracle
    itself drifts by 1.5e-2 / 5e-2 at step 3 from a one-ulp weight perturbation, profiles/r02/oracle_multi_step_sensitivity.txt.)"""
    rep = checks.check_train_step(lib, DEV, "fp32", mode="mapnet", N=2, H=64, W=85, steps=2, lr=1e-3, adam_eps=1.0,
                                  loss_rtol=2e-3, pose_atol=3e-2, grad_l2_rtol=None)
    assert rep[-1][0] == "displacement_rel_l2" and rep[-1][1] < 0.2, rep[-1]


@pytest.mark.parametrize("method,kw,max_norm", [
    ("sgd", {}, 0.0), ("sgd", {"momentum": 0.9}, 5.0), ("sgd", {"momentum": 0.9, "dampening": 0.1}, 0.0),
    ("sgd", {"momentum": 0.8, "nesterov": True}, 0.0),
    ("rmsprop", {}, 0.0), ("rmsprop", {"momentum": 0.5, "alpha": 0.9}, 5.0),
])
def test_fused_sgd_rmsprop(lib, method, kw, max_norm):
    checks.check_sgd_rmsprop(lib, DEV, method, n=1000003, max_norm=max_norm, **kw)


@pytest.mark.parametrize("method,kw", [
    ("sgd", {"momentum": 0.9, "lr_decay": 0.1, "lr_stepvalues": [3, 6]}),
    ("rmsprop", {"momentum": 0.5}),
])
def test_training_step_with_sgd_and_rmsprop(lib, method, kw):
    """the reference wrapper's other two methods (common/optimizer.py:16-26) through step_feedfwd, vs the oracle"""
    checks.check_train_other_optimizers(lib, DEV, method, N=2, H=64, W=85, steps=2, **kw)
