"""Round-5 entry points: the context's own copy of the rBRIEF test table (orbfe_set_pattern: ORBextractor::pattern,
src/ORBextractor.cc:442-444 -- the table `north_star` has rank 0 broadcast) and the build id."""
import numpy as np
import pytest

from oracle import oracle as O
from orbslam2_amd import synth

pytestmark = pytest.mark.gpu

CFG = dict(width=480, height=320, nfeatures=600, fx=400.0, fy=400.0, cx=240.0, cy=160.0, bf=160.0)


def _same_frame(out, kl, dl, kr, dr, ur, dp):
    assert out["kps_left"].tobytes() == kl.tobytes() and out["kps_right"].tobytes() == kr.tobytes()
    assert np.array_equal(out["desc_left"], dl) and np.array_equal(out["desc_right"], dr)
    assert out["u_right"].tobytes() == ur.tobytes() and out["depth"].tobytes() == dp.tobytes()


def _oracle_frame(left, right, pattern=None):
    exl, exr = O.Extractor(nfeatures=CFG["nfeatures"]), O.Extractor(nfeatures=CFG["nfeatures"])
    if pattern is not None:
        exl.set_pattern(pattern); exr.set_pattern(pattern)
    kl, dl = exl.extract(left)
    kr, dr = exr.extract(right)
    ur, dp, m = O.stereo_matches(exl, exr, kl, dl, kr, dr, CFG["bf"], CFG["fx"])
    return kl, dl, kr, dr, ur, dp, m


def test_pattern_defaults_to_the_compiled_table_and_follows_set_pattern():
    from orbslam2_amd import api, dist as D
    left, right = synth.stereo_pair(CFG["width"], CFG["height"], seed=41)
    ctx = api.Context(**CFG)
    compiled = D.compiled_pattern()
    assert np.array_equal(ctx.pattern(), compiled)
    assert np.array_equal(O.Extractor().pattern(), compiled)
    ref0 = _oracle_frame(left, right)
    _same_frame(ctx.stereo_frame(left, right), *ref0[:6])
    # a permuted table (what a rank receives from dist.broadcast_pattern): descriptors, and through them the stereo matches,
    # follow it bit for bit; keypoints do not depend on it
    perm = np.random.default_rng(7).permutation(256)
    pat = D.broadcast_pattern(compiled[perm], "cpu")
    ctx.set_pattern(pat)
    assert np.array_equal(ctx.pattern(), pat)
    ref1 = _oracle_frame(left, right, pat)
    assert ref1[0].tobytes() == ref0[0].tobytes() and not np.array_equal(ref1[1], ref0[1])
    out1 = ctx.stereo_frame(left, right)
    _same_frame(out1, *ref1[:6])
    b0 = np.unpackbits(ref0[1], axis=1, bitorder="little")
    assert np.array_equal(np.unpackbits(out1["desc_left"], axis=1, bitorder="little"), b0[:, perm])
    # a table of our own: random points inside the reach the descriptor stage supports, incl. the extreme ones
    rng = np.random.default_rng(8)
    own = rng.integers(-13, 14, (256, 4)).astype(np.int32)
    own[0] = (13, 13, -13, -13); own[1] = (18, 0, 0, -18); own[2] = (-18, 4, 4, 18)
    ctx.set_pattern(own)
    ref2 = _oracle_frame(left, right, own)
    _same_frame(ctx.stereo_frame(left, right), *ref2[:6])
    assert ref2[6] > 0
    # and back
    ctx.set_pattern(compiled)
    _same_frame(ctx.stereo_frame(left, right), *ref0[:6])
    # a point that could rotate out of the staged patch is refused, and the table in use stays
    bad = compiled.copy(); bad[5] = (14, 13, 0, 0)
    with pytest.raises(api.OrbfeError) as e:
        ctx.set_pattern(bad)
    assert e.value.code == -5
    assert np.array_equal(ctx.pattern(), compiled)
    ctx.close()


def test_pattern_applies_to_the_batched_path_and_other_patch_sizes():
    import torch
    from orbslam2_amd import api, dist as D
    pat = D.compiled_pattern()[np.random.default_rng(9).permutation(256)]
    pairs = [synth.stereo_pair(CFG["width"], CFG["height"], seed=50 + i) for i in range(3)]
    dev = torch.from_numpy(np.stack([im for p in pairs for im in p])).cuda()
    ctx = api.Context(max_images=6, **CFG)
    ctx.set_pattern(pat)
    ctx.enqueue_stereo(dev.data_ptr(), 3)
    ctx.synchronize()
    for i, (left, right) in enumerate(pairs):
        kl, dl, kr, dr, ur, dp, _ = _oracle_frame(left, right, pat)
        got = ctx.fetch_image(2 * i, stereo=True)
        assert got["kps"].tobytes() == kl.tobytes() and np.array_equal(got["desc"], dl)
        assert got["u_right"].tobytes() == ur.tobytes() and got["depth"].tobytes() == dp.tobytes()
        assert np.array_equal(ctx.fetch_image(2 * i + 1)["desc"], dr)
    ctx.close()
    # describe_generic_kernel (half_patch_size != 15) reads the same table
    cfg = dict(CFG, patch_size=25, half_patch_size=12)
    ctx = api.Context(**cfg)
    ctx.set_pattern(pat)
    left = pairs[0][0]
    ex = O.Extractor(nfeatures=CFG["nfeatures"], patch_size=25, half_patch_size=12)
    ex.set_pattern(pat)
    k, d = ex.extract(left)
    gk, gd = ctx.extract(left)
    assert len(k) > 100 and gk.tobytes() == k.tobytes() and np.array_equal(gd, d)
    ctx.close()


def test_build_id_is_a_sha256_and_stable():
    from orbslam2_amd import api
    a, b = api.build_id(), api.build_id()
    assert a == b and len(a) == 64 and int(a, 16) >= 0


def test_packed_fetches_on_two_streams_share_one_staging_block_safely():
    """orbfe_fetch_batch_packed stages through ONE device block per context (round-4 advisor finding): a second fetch queued on another
    stream while the first one's copy is in flight must wait for it, not refill the block under it."""
    import torch
    from orbslam2_amd import api
    pairs = [synth.stereo_pair(CFG["width"], CFG["height"], seed=70 + i) for i in range(4)]
    dev = torch.from_numpy(np.stack([im for p in pairs for im in p])).cuda()
    ctx = api.Context(max_images=8, **CFG)
    ctx.enqueue_stereo(dev.data_ptr(), 4)
    ctx.synchronize()
    ref = [ctx.fetch_image(i, stereo=i % 2 == 0) for i in range(8)]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(20):
        la, lb = ctx.packed_layout(8, api.PACK_STEREO), ctx.packed_layout(8, api.PACK_STEREO | api.PACK_LEFT_ONLY)
        ba, bb = torch.zeros(la.bytes, dtype=torch.uint8).pin_memory(), torch.zeros(lb.bytes, dtype=torch.uint8).pin_memory()
        ctx.fetch_batch_packed(8, api.PACK_STEREO, ba.data_ptr(), la.bytes, sa.cuda_stream)
        ctx.fetch_batch_packed(8, api.PACK_STEREO | api.PACK_LEFT_ONLY, bb.data_ptr(), lb.bytes, sb.cuda_stream)  # no host wait in between
        sa.synchronize(); sb.synchronize()
        for o in range(la.n_images_out):
            got = ctx.expand_packed(ba.numpy(), la, o)
            assert got["kps"].tobytes() == ref[o]["kps"].tobytes() and np.array_equal(got["desc"], ref[o]["desc"]), (rep, o)
        for o in range(lb.n_images_out):
            got = ctx.expand_packed(bb.numpy(), lb, o)
            assert got["kps"].tobytes() == ref[2 * o]["kps"].tobytes() and got["u_right"].tobytes() == ref[2 * o]["u_right"].tobytes(), (rep, o)
    ctx.close()


def test_pack_direct_checks_the_pinned_range_and_accepts_a_block_inside_it():
    """ORBFE_PACK_DIRECT stores lay.bytes from the caller's pointer on, so the library asks the runtime for the pinned range behind the
    pointer (hipMemGetAddressRange) and refuses a block that would end outside it.  Exercised here: blocks that DO fit -- at the start of
    a pinned allocation and at an offset inside a larger one -- are accepted and filled correctly; pageable memory is refused.  (The
    refusal of a too-short range is not provoked on purpose: where the runtime cannot report the range the store would be real.)"""
    import torch
    from orbslam2_amd import api
    left, right = synth.stereo_pair(CFG["width"], CFG["height"], seed=90)
    dev = torch.from_numpy(np.stack([left, right])).cuda()
    ctx = api.Context(max_images=2, **CFG)
    ctx.enqueue_stereo(dev.data_ptr(), 1)
    ctx.synchronize()
    lay = ctx.packed_layout(2, api.PACK_STEREO)
    ref = ctx.fetch_image(0, stereo=True)
    blk = torch.zeros(lay.bytes + 8192, dtype=torch.uint8).pin_memory()
    for off in (0, 4096):
        blk.zero_()
        ctx.fetch_batch_packed(2, api.PACK_STEREO | api.PACK_DIRECT, blk.data_ptr() + off, lay.bytes, 0)
        ctx.synchronize()
        got = ctx.expand_packed(blk.numpy()[off:off + lay.bytes], lay, 0)
        assert got["kps"].tobytes() == ref["kps"].tobytes() and np.array_equal(got["desc"], ref["desc"])
        assert not blk.numpy()[off + lay.bytes:].any() and not blk.numpy()[:off].any()  # nothing outside the block was written
    pageable = np.zeros(lay.bytes, np.uint8)
    with pytest.raises(api.OrbfeError):
        ctx.fetch_batch_packed(2, api.PACK_STEREO | api.PACK_DIRECT, pageable.ctypes.data, lay.bytes, 0)
    ctx.close()
