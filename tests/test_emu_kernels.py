"""CPU suite: every kernel of geomapnet_amd/csrc executed through the SIMT emulator build and
checked against torch fp64 / the oracle / the golden vectors (tests/checks.py)."""
import pytest

import checks
import emu_lib

DEV = "cpu"


@pytest.fixture(scope="module")
def lib():
    return emu_lib.load()


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [
    (2, 9, 11, 64, 64, 3, 1, 1),     # layer1-like, BN=64 tile
    (2, 9, 11, 64, 128, 3, 2, 1),    # stride-2 3x3
    (1, 8, 10, 64, 128, 1, 2, 0),    # 1x1 projection
    (3, 5, 6, 128, 192, 3, 1, 1),    # N not a multiple of the 128 tile
])
def test_conv_forward(lib, dtype, shape):
    checks.check_conv_fwd(lib, DEV, dtype, *shape)


@pytest.mark.parametrize("shape", [
    (2, 9, 11, 64, 64, 3, 1, 1),     # layer1-like (chunk-resident 64-column shape)
    (2, 9, 11, 64, 128, 3, 2, 1),    # stride-2 3x3 (generic kernel)
    (1, 8, 10, 64, 128, 1, 2, 0),    # 1x1 projection
    (3, 5, 6, 128, 256, 3, 1, 1),    # chunk-resident 128-column shapes
])
def test_conv_forward_with_fp8_cross_terms(lib, shape):
    """h2q operands (round 5, the fp16x2q mode's forward convolutions): one scaled fp8 MFMA per K-step for both cross terms"""
    checks.check_conv_fwd_h2q(lib, DEV, *shape)


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [
    (2, 9, 11, 64, 64, 3, 1, 1),
    (2, 9, 11, 64, 128, 3, 2, 1),    # stride-2 data gradient (div = 2 gather)
    (2, 8, 10, 64, 128, 1, 2, 0),
    (2, 7, 9, 64, 128, 1, 2, 0),     # odd input size
])
def test_conv_data_gradient(lib, dtype, shape):
    checks.check_conv_dgrad(lib, DEV, dtype, *shape)


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
@pytest.mark.parametrize("shape,blocks", [
    ((2, 9, 11, 64, 64, 3, 1, 1), 8),
    ((3, 9, 11, 64, 128, 3, 2, 1), 8),
    ((2, 8, 10, 64, 128, 1, 2, 0), 1),
    ((5, 5, 6, 128, 128, 3, 1, 1), 40),   # many splits of the reduction
    ((3, 7, 9, 128, 128, 3, 1, 1), 40),   # table-driven gather, ragged last split (M = 189)
])
def test_conv_weight_gradient(lib, dtype, shape, blocks):
    checks.check_conv_wgrad(lib, DEV, dtype, *shape, target_blocks=blocks)


@pytest.mark.parametrize("dtype", [0, 1, 2])
@pytest.mark.parametrize("hw", [(20, 27), (21, 26)])
def test_stem_conv(lib, dtype, hw):
    checks.check_stem(lib, DEV, dtype, 2, *hw)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("M,C,kw", [
    (300, 64, dict()),
    (77, 128, dict(with_res=False)),
    (130, 512, dict(relu=False, with_res=False)),
])
def test_batchnorm(lib, dtype, M, C, kw):
    checks.check_bn(lib, DEV, dtype, M, C, **kw)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("hw,ties", [((8, 11), False), ((9, 10), True)])
def test_maxpool(lib, dtype, hw, ties):
    checks.check_maxpool(lib, DEV, dtype, 2, hw[0], hw[1], 64, ties=ties)


def test_criteria_against_reference_golden(lib, golden_dir):
    checks.check_criterion_golden(lib, DEV, golden_dir)


def test_calc_vos_against_reference_golden(lib, golden_dir):
    checks.check_calc_vos_golden(lib, DEV, golden_dir)


@pytest.mark.parametrize("max_norm", [0.0, 5.0])
def test_fused_adam(lib, max_norm):
    checks.check_adam(lib, DEV, max_norm=max_norm)


@pytest.mark.parametrize("method,kw,max_norm", [
    ("sgd", {}, 0.0), ("sgd", {"momentum": 0.9}, 5.0), ("sgd", {"momentum": 0.9, "dampening": 0.1}, 0.0),
    ("sgd", {"momentum": 0.8, "nesterov": True}, 0.0),
    ("rmsprop", {}, 0.0), ("rmsprop", {"momentum": 0.5, "alpha": 0.9}, 5.0),
])
def test_fused_sgd_rmsprop(lib, method, kw, max_norm):
    checks.check_sgd_rmsprop(lib, DEV, method, max_norm=max_norm, **kw)


def test_pose_graph_golden(lib, golden_dir):
    checks.check_pgo_golden(lib, DEV, golden_dir)


@pytest.mark.parametrize("N,fc,sig", [(7, False, (1.0, 1.0, 1.0, 1.0)), (7, True, (0.5, 2.0, 20.0, 20.0)), (2, False, (1.0, 1.0, 1.0, 1.0)),
                                      (12, True, (1.0, 1.0, 2.0, 2.0))])
def test_pose_graph_vs_oracle(lib, N, fc, sig):
    checks.check_pgo_vs_oracle(lib, DEV, W=3, N=N, fc=fc, sig=sig)


def test_pose_graph_properties(lib):
    checks.check_pgo_properties(lib, DEV, W=24, N=7, fc=True)


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
@pytest.mark.parametrize("shape,mode", [
    ((2, 8, 11, 64, 128, 3, 2, 1), "plain"),      # odd width: the parity classes have 6 and 5 columns
    ((2, 9, 10, 64, 128, 3, 2, 1), "out_gate"),   # odd height
    ((1, 8, 12, 64, 128, 3, 2, 1), "res_gate"),
    ((2, 8, 11, 64, 128, 1, 2, 0), "inplace"),    # 1x1 projection: one class, accumulated in place
    ((2, 9, 11, 64, 64, 3, 1, 1), "out_gate"),    # stride 1: generic form
])
@pytest.mark.parametrize("parity", [1, 0])
def test_conv_data_gradient_op(lib, dtype, shape, mode, parity):
    checks.check_conv_dgrad_op(lib, DEV, dtype, *shape, parity=parity, mode=mode)


def _random_cases(seed, n):
    import numpy as np
    rng = np.random.default_rng(seed)
    return [(int(rng.integers(1, 3)), int(rng.integers(3, 37)), int(rng.integers(3, 37))) for _ in range(n)]


@pytest.mark.parametrize("case", _random_cases(2024, 5))
def test_conv_halo_random_geometry(lib, case):
    """seeded random image sizes (ragged tiles, images smaller than one tile), forward with statistics and data
    gradient with the gated identity path"""
    B, H, W = case
    checks.check_conv_halo(lib, DEV, B, H, W, Cout=64, dgrad=False, mode="plain", seed=H * 100 + W)
    checks.check_conv_halo(lib, DEV, B, H, W, Cout=64, dgrad=True, mode="out_gate", seed=H * 100 + W + 1)


@pytest.mark.parametrize("case", _random_cases(77, 4))
def test_conv_data_gradient_parity_random_geometry(lib, case):
    """stride-2 data gradient by parity classes vs the generic form's reference on seeded random image sizes"""
    B, H, W = case
    H, W = max(H, 4), max(W, 4)
    checks.check_conv_dgrad_op(lib, DEV, 1, B, H, W, 64, 128, 3, 2, 1, parity=1, mode="out_gate", seed=H * 100 + W)
    checks.check_conv_dgrad_op(lib, DEV, 0, B, H, W, 64, 128, 1, 2, 0, parity=1, mode="inplace", seed=H * 100 + W + 1)


@pytest.mark.parametrize("shape", [(2, 9, 11, 64, 64, 3, 1, 1), (3, 20, 22, 128, 64, 3, 1, 1), (40, 3, 5, 64, 128, 3, 1, 1),
                                   (2, 6, 7, 72, 80, 3, 1, 1), (2, 8, 11, 256, 256, 3, 1, 1)])
@pytest.mark.parametrize("dtype", [1, 2, 3])
def test_fused_weight_gradient_through_workspace(lib, dtype, shape):
    """wgrad_fused.h (fp16 kernel; fp32x3 kernel: fp32 tensors split into bf16 halves at the LDS write) with partial tiles
    stored to a (NaN-filled) workspace and added up by the reduce kernel, as the plan runs it; bit-identical between two
    launches when a single reduction group covers the columns"""
    if dtype == 3 and shape[3] % 32:  # h2 tensors hold whole 32-channel groups: 96 / 160 channels = 1.5 / 2.5 tiles of 64
        shape = shape[:3] + (96, 160) + shape[5:]
    checks.check_conv_wgrad(lib, DEV, dtype, *shape, ws=True)


@pytest.mark.parametrize("shape", [(6, 128, 128), (37, 256, 200), (65, 128, 70)])
def test_pose_head_dense_layer(lib, shape):
    """csrc/dense.h: the head's fc layer forward / data gradient / weight + bias gradient in 32 x 32 tiles and the pose
    regressors' weight gradient, ragged rows and feature counts, against torch fp64"""
    checks.check_dense(lib, DEV, *shape)


@pytest.mark.parametrize("shape", [(1, 20, 27), (2, 33, 70)])
def test_stem_backward_two_launch_form(lib, shape, monkeypatch):
    """csrc/stem_bwd.h against the maxpool_bwd -> bn_bwd -> wgrad chain it replaces, identical fp16 tensors; three
    persistent workgroups so every workgroup walks several tiles"""
    monkeypatch.setenv("MN_STEM_WGS", "3")
    checks.check_stem_bwd(lib, DEV, *shape)


def test_forced_288x256_configuration():
    """the 12-wave 288x256 tile (packed tap masks, joint A/B DMA passes, odd wave-row count) on small ragged problems,
    in a process of its own because the configuration knob is read once"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MN_IGEMM_CONFIG="12")
    subprocess.run([sys.executable, os.path.join(here, "forced_config_cases.py"), "emu"], check=True, env=env, timeout=900)


def test_weight_gradient_with_assembly_transpose_reads():
    """wgrad_dma_kernel (the plain-GEMM fp16 weight gradient, transpose reads issued from inline assembly; since round 2 the
    3x3 stride-1 layers go to wgrad_fused.h, so MN_WGRAD_FUSED=0 routes the cases here): the emulator executes the
    kernel's address arithmetic (lane base + immediate offsets)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MN_WGRAD_CASES="1", MN_WGRAD_FUSED="0")
    subprocess.run([sys.executable, os.path.join(here, "forced_config_cases.py"), "emu"], check=True, env=env, timeout=900)


@pytest.mark.parametrize("mode,N,T", [("mapnet", 1, 2), ("mapnet", 3, 5), ("mapnet", 2, 7), ("online", 1, 2), ("online", 3, 4),
                                      ("gps", 2, 5), ("posenet", 1, 1), ("mapnet", 70, 4), ("online", 33, 3)])
def test_criteria_vs_oracle_other_window_lengths(lib, mode, N, T):
    checks.check_criterion_vs_oracle(lib, DEV, mode, N, T)


def test_experimental_chunk_resident_a_kernel():
    """igemm_halo.h (MN_IGEMM_HALO=1|2; 2 is the default since round 2): the 288x256 tile with the A operand staged once per 64-channel chunk
    and taps as row shifts into that image, against torch fp64 (forward with BatchNorm sums, data gradient with residual
    and gates; one to four chunks, ragged tiles, tiles spanning several images)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MN_IGEMM_CONFIG="12", MN_IGEMM_HALO="1")  # the 256-column shape (layer3)
    subprocess.run([sys.executable, os.path.join(here, "forced_config_cases.py"), "emu"], check=True, env=env, timeout=900)
    env = dict(os.environ, MN_IGEMM_HALO="2", MN_HALO384="0")        # the 128-column shape (layers 2 and 4), 288-row tiles
    env.pop("MN_IGEMM_CONFIG", None)
    subprocess.run([sys.executable, os.path.join(here, "forced_config_cases.py"), "emu"], check=True, env=env, timeout=900)
    env["MN_HALO384"] = "2"                                          # ... and the 8-wave 384-row tile
    subprocess.run([sys.executable, os.path.join(here, "forced_config_cases.py"), "emu"], check=True, env=env, timeout=900)
    env["MN_HALO_A1"] = "2"                                          # ... and layer2's two-workgroup single-image shape
    subprocess.run([sys.executable, os.path.join(here, "forced_config_cases.py"), "emu"], check=True, env=env, timeout=900)


@pytest.mark.parametrize("force256", ["0", "1"])
def test_chunk_resident_a_kernel_h2(force256):
    """igemm_halo.h with h2 operands (dtype 3: fp16-pair tensors, three MFMAs per product, fp32 output): the 128-column shape
    and, forced onto small problems, the 256-column shape -- forward with BatchNorm sums, data gradients with residual / gates"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MN_H2_CASES="1", MN_H2_HALO256=force256, MN_HALO_A1="0")
    subprocess.run([sys.executable, os.path.join(here, "forced_config_cases.py"), "emu"], check=True, env=env, timeout=1500)
    if force256 == "0":  # the 128-column launches again through layer2's two-workgroup single-image shape
        subprocess.run([sys.executable, os.path.join(here, "forced_config_cases.py"), "emu"], check=True, env=dict(env, MN_HALO_A1="2"),
                       timeout=1500)


def test_occupancy_stand_in_streams_its_bytes_and_lasts_its_time(lib):
    """mn_op_occupy (csrc/rehearsal.h), the one-GPU stand-in for an RCCL ring step: dst += src over the whole range, resident for at
    least the requested time"""
    import ctypes as C
    import time

    import torch
    from geomapnet_amd._binding import ptr
    src = torch.arange(4096, dtype=torch.float32)
    dst = torch.ones(4096, dtype=torch.float32)
    t0 = time.perf_counter()
    lib.check(lib.op_occupy(2, 64, C.c_float(20000.0), ptr(src), ptr(dst), C.c_int64(4096 * 4), 32, None))
    dt = time.perf_counter() - t0
    assert torch.equal(dst, src + 1.0)
    assert dt >= 0.02, dt
    assert lib.op_occupy(0, 64, C.c_float(1.0), None, None, C.c_int64(0), 0, None) != 0  # argument check


@pytest.mark.parametrize("shape", [(2, 9, 11), (1, 16, 32), (3, 7, 19)])
def test_layer1_h2_convolution_with_the_weights_in_registers(lib, shape):
    """halo_h2.h (round 6): ragged tiles, several tiles per workgroup (the emulated chip has 4 CUs), statistics"""
    checks.check_conv_halo_h2(lib, DEV, *shape)
