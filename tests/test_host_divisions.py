"""The divisions the round-5 kernels no longer do: restated in Python from orbslam2_amd/csrc/orbfe_common.hpp (xcd_map_magic_host /
xcd_map_of_magic) and orbfe_api.hip (cell_aux: ceil(2^16 / d) multipliers for lane / d), checked exhaustively over the ranges the
kernels can meet.  The GPU parity tests cover the kernels that use them; this pins the arithmetic the host tables rely on."""
import itertools


def xcd_split_log2(n_units):
    return 0 if n_units >= 5 else 1 if n_units >= 3 else 2 if n_units == 2 else 3


def magic_host(blocks_per_unit, n_units):  # xcd_map_magic_host
    lg = xcd_split_log2(n_units)
    per_xcd = (blocks_per_unit + (1 << lg) - 1) >> lg
    side = 8 >> lg
    rounds = (n_units + side - 1) // side
    if per_xcd <= 1 or per_xcd * rounds * per_xcd >= 1 << 32:
        return 0, per_xcd, rounds
    return ((1 << 32) + per_xcd - 1) // per_xcd, per_xcd, rounds


def test_block_map_magic_is_exact_for_every_block_of_the_grid():
    # FAST at KITTI geometry: 381 workgroups per image, 128 images; describe: 128 per image; stereo: 128 per pair; plus odd shapes
    for bpu, n in itertools.product((1, 2, 3, 7, 128, 381, 382, 1000, 4099, 65535), (1, 2, 3, 4, 5, 8, 64, 128, 1000)):
        magic, per_xcd, rounds = magic_host(bpu, n)
        if magic == 0:
            assert per_xcd <= 1 or per_xcd * rounds * per_xcd >= 1 << 32  # the kernel divides itself
            continue
        assert magic < 1 << 32
        top = per_xcd * rounds  # jb = blockIdx.x >> 3 stays below this (xcd_grid)
        step = max(1, top // 20011)
        for jb in itertools.chain(range(0, top, step), range(max(0, top - 3 * per_xcd), top), range(min(top, 3 * per_xcd))):
            assert (jb * magic) >> 32 == jb // per_xcd, (bpu, n, jb)


def test_lane_division_multipliers_are_exact():
    # cell_aux: lane / ng (ng <= 16 groups per row) and lane / cpr (cpr <= 5 chunks per row); any divisor up to 64 holds
    for d in range(1, 65):
        m = (65536 + d - 1) // d
        assert m <= 65536  # 17 bits in the table word
        for lane in range(64):
            assert (lane * m) >> 16 == lane // d, (d, lane)


def test_half_precision_thresholds():
    # orbfe_api.hip half_bits(): integers 1 .. 255 as IEEE half precision
    import numpy as np
    for t in range(0, 256):
        if t == 0:
            bits = 0
        else:
            e = t.bit_length() - 1
            bits = ((e + 15) << 10) | ((t << (10 - e)) & 0x3ff)
        assert bits == int(np.float16(t).view(np.uint16)), t
