"""TEST INFRASTRUCTURE: convolution parity cases run in a process of their own with a tile configuration forced
through MN_IGEMM_CONFIG / MN_IGEMM_HALO / MN_WGRAD_FUSED (the library reads such knobs once).  `python forced_config_cases.py emu|hip`.
Configuration 12 = the 12-wave 288x256 tile, which the dispatcher picks by itself only for grids that fill most of
the chip (layer3 at 192 images); here it runs on small ragged problems against torch fp64."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main(backend):
    import checks
    if backend == "emu":
        import emu_lib
        lib, dev = emu_lib.load(), "cpu"
    else:
        from geomapnet_amd import _binding
        lib, dev = _binding.hip(), "cuda"
    if os.environ.get("MN_WGRAD_CASES") == "1":  # (MN_WGRAD_FUSED=0/1)
        # fp16 weight gradients of stride-1 "same" convolutions with the transpose reads issued from inline assembly
        # (wgrad_dma_kernel<..., ASMRD>): all four tile shapes, several splits, ragged last split
        for shape, blocks in (((2, 9, 11, 64, 64, 3, 1, 1), 8), ((3, 7, 9, 128, 128, 3, 1, 1), 40),
                              ((2, 9, 11, 64, 128, 1, 1, 0), 4), ((2, 9, 11, 64, 64, 1, 1, 0), 1),
                              ((4, 16, 22, 128, 256, 3, 1, 1), 64)):
            checks.check_conv_wgrad(lib, dev, 1, *shape, target_blocks=blocks)
        if backend != "emu":  # layer geometries at sizes with hundreds of concurrent workgroups, repeated (race screen)
            for rep in range(3):
                for shape in ((2, 64, 86, 64, 64, 3, 1, 1), (7, 16, 22, 256, 256, 3, 1, 1), (4, 8, 11, 512, 512, 3, 1, 1),
                              (6, 32, 43, 128, 128, 3, 1, 1)):
                    checks.check_conv_wgrad(lib, dev, 1, *shape, target_blocks=1024, seed=2 + rep)
                    if os.environ.get("MN_WGRAD_FUSED") == "1":  # the fp32x3 form of the fused kernel (wgrad_fused_x3_kernel)
                        checks.check_conv_wgrad(lib, dev, 2, *shape, seed=12 + rep, ws=True)
        print("forced-config cases ok")
        return
    if os.environ.get("MN_H2_CASES") == "1":
        # h2 operands (dtype 3) through igemm_halo.h: the 128-column shape takes every 3x3 stride-1 launch it covers; with
        # MN_H2_HALO256=1 the 256-column shape (layer3 at 192 images) runs on small ragged problems
        if os.environ.get("MN_H2_HALO256") == "1":
            shapes = ((4, 9, 11, 64, 256, 3, 1, 1), (3, 16, 22, 256, 256, 3, 1, 1), (2, 13, 31, 192, 256, 3, 1, 1))
            dg = ((3, 16, 22, 256, 256, 3, 1, 1), (5, 8, 11, 256, 128, 3, 1, 1))
        else:
            shapes = ((3, 9, 11, 64, 128, 3, 1, 1), (2, 12, 43, 128, 128, 3, 1, 1), (2, 7, 47, 64, 384, 3, 1, 1), (40, 3, 5, 64, 128, 3, 1, 1))
            dg = ((2, 12, 43, 128, 128, 3, 1, 1), (9, 8, 11, 512, 512, 3, 1, 1))
        # layer1 geometry (64 -> 64 channels: the generic 128x64 kernel): wide rows, ragged tiles
        for shape in ((2, 9, 11, 64, 64, 3, 1, 1), (1, 7, 87, 64, 64, 3, 1, 1), (40, 3, 5, 64, 64, 3, 1, 1)):
            checks.check_conv_fwd(lib, dev, 3, *shape)
        checks.check_conv_dgrad_op(lib, dev, 3, 2, 9, 43, 64, 64, 3, 1, 1, parity=1, mode="out_gate")
        for shape in shapes:
            checks.check_conv_fwd(lib, dev, 3, *shape)
        for mode in ("plain", "out_gate", "res_gate"):
            checks.check_conv_dgrad_op(lib, dev, 3, *dg[0], parity=1, mode=mode)
        checks.check_conv_dgrad_op(lib, dev, 3, *dg[1], parity=1, mode="out_gate")
        if backend != "emu":  # layer geometries with hundreds of concurrent workgroups, repeated (race screen)
            for rep in range(3):
                checks.check_conv_fwd(lib, dev, 3, 48, 32, 43, 128, 128, 3, 1, 1, seed=rep)
                checks.check_conv_fwd(lib, dev, 3, 96, 16, 22, 256, 256, 3, 1, 1, seed=5 + rep)
                checks.check_conv_fwd(lib, dev, 3, 96, 8, 11, 512, 512, 3, 1, 1, seed=10 + rep)
                checks.check_conv_dgrad_op(lib, dev, 3, 96, 16, 22, 256, 256, 3, 1, 1, parity=1, mode="res_gate", seed=20 + rep)
        print("forced-config cases ok")
        return
    if os.environ.get("MN_IGEMM_HALO") == "2" and os.environ.get("MN_IGEMM_CONFIG") is None:
        # the 128-column shape of igemm_halo.h (no forced tile configuration: it takes every launch it covers):
        # N = 128 / 384 / 512, rows up to the 47-pixel limit, one to eight chunks
        for shape in ((3, 9, 11, 64, 128, 3, 1, 1), (2, 12, 43, 128, 128, 3, 1, 1), (2, 7, 47, 64, 384, 3, 1, 1),
                      (9, 8, 11, 512, 512, 3, 1, 1), (40, 3, 5, 64, 128, 3, 1, 1)):
            checks.check_conv_fwd(lib, dev, 1, *shape)
        for mode in ("plain", "out_gate", "res_gate"):
            checks.check_conv_dgrad_op(lib, dev, 1, 2, 12, 43, 128, 128, 3, 1, 1, parity=1, mode=mode)
        checks.check_conv_dgrad_op(lib, dev, 1, 9, 8, 11, 512, 512, 3, 1, 1, parity=1, mode="res_gate")
        if backend != "emu":  # layer2 / layer4 geometries at a size that fills the chip, repeated (race screen)
            for rep in range(3):
                checks.check_conv_fwd(lib, dev, 1, 48, 32, 43, 128, 128, 3, 1, 1, seed=rep)
                checks.check_conv_fwd(lib, dev, 1, 192, 8, 11, 512, 512, 3, 1, 1, seed=10 + rep)
                checks.check_conv_dgrad_op(lib, dev, 1, 48, 32, 43, 128, 128, 3, 1, 1, parity=1, mode="res_gate", seed=20 + rep)
        print("forced-config cases ok")
        return
    assert os.environ.get("MN_IGEMM_CONFIG") == "12"
    if os.environ.get("MN_IGEMM_HALO") == "1":
        # igemm_halo.h (A operand staged once per 64-channel chunk): fp16 3x3 stride-1 convolutions, one to four chunks,
        # ragged last tile, tiles that start inside an image and span several, widths up to the 31-pixel limit
        for shape in ((4, 9, 11, 64, 256, 3, 1, 1), (3, 16, 22, 256, 256, 3, 1, 1), (7, 8, 11, 128, 512, 3, 1, 1),
                      (2, 13, 31, 192, 256, 3, 1, 1), (40, 3, 5, 64, 256, 3, 1, 1)):
            checks.check_conv_fwd(lib, dev, 1, *shape)
        for mode in ("plain", "out_gate", "res_gate"):
            checks.check_conv_dgrad_op(lib, dev, 1, 3, 16, 22, 256, 256, 3, 1, 1, parity=1, mode=mode)
        checks.check_conv_dgrad_op(lib, dev, 1, 5, 8, 11, 256, 128, 3, 1, 1, parity=1, mode="out_gate")
        if backend != "emu":  # layer3 / layer4 geometries with hundreds of concurrent workgroups, repeated (race screen)
            for rep in range(3):
                checks.check_conv_fwd(lib, dev, 1, 96, 16, 22, 256, 256, 3, 1, 1, seed=rep)
                checks.check_conv_fwd(lib, dev, 1, 96, 8, 11, 512, 512, 3, 1, 1, seed=10 + rep)
                checks.check_conv_dgrad_op(lib, dev, 1, 96, 16, 22, 256, 256, 3, 1, 1, parity=1, mode="res_gate", seed=20 + rep)
        print("forced-config cases ok")
        return
    for dtype in (0, 1):
        # (B, H, W, Cin, Cout, k, stride, pad): 396 / 663 / 198 rows = 1.4 / 2.3 / 0.7 tiles of 288 rows
        checks.check_conv_fwd(lib, dev, dtype, 4, 9, 11, 64, 256, 3, 1, 1)
        checks.check_conv_fwd(lib, dev, dtype, 3, 13, 17, 128, 256, 1, 1, 0)
        checks.check_conv_fwd(lib, dev, dtype, 2, 18, 22, 64, 256, 3, 2, 1)
        checks.check_conv_dgrad_op(lib, dev, dtype, 4, 9, 11, 256, 256, 3, 1, 1, parity=1, mode="out_gate")
        checks.check_conv_dgrad_op(lib, dev, dtype, 2, 9, 11, 256, 256, 3, 1, 1, parity=1, mode="res_gate")
        checks.check_conv_dgrad_op(lib, dev, dtype, 3, 16, 22, 256, 256, 3, 2, 1, parity=1, mode="plain")
        checks.check_conv_dgrad_op(lib, dev, dtype, 3, 16, 22, 256, 256, 1, 2, 0, parity=1, mode="inplace")
    print("forced-config cases ok")


if __name__ == "__main__":
    main(sys.argv[1])
