"""Entry points and guards added in round 3 (ABI 4), through the C ABI on the GPU: orbfe_assign_features_to_grid, orbfe_stereo_batch,
orbfe_get_camera / orbfe_device_count / orbfe_vocab_bytes, orbfe_set_profiling_interval, the device-slot count check, and a caller
stream that is destroyed between the enqueue and the fetch (the library must not keep its handle)."""
import ctypes as C

import numpy as np
import pytest

from oracle import literal_matchers as LM
from oracle import oracle as O
from orbslam2_amd import synth

pytestmark = pytest.mark.gpu
W, H, NF = 480, 320, 600
KW = dict(width=W, height=H, nfeatures=NF, fx=400.0, fy=400.0, cx=W / 2, cy=H / 2, bf=160.0)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _grid(ctx, view, n):
    ctx.L.orbfe_assign_features_to_grid.restype = C.c_int
    ctx.L.orbfe_assign_features_to_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    off = np.zeros(64 * 48 + 1, np.int32); idx = np.zeros(max(n, 1), np.int32)
    rc = ctx.L.orbfe_assign_features_to_grid(ctx.h, C.byref(view), _p(off), _p(idx))
    return rc, off, idx


def test_assign_features_to_grid_upload_resident_and_fractional_bounds():
    from orbslam2_amd import api
    ctx = api.Context(**KW)
    left, right = synth.stereo_pair(W, H, seed=61)
    out = ctx.stereo_frame(left, right)
    k, d = out["kps_left"], out["desc_left"]
    cam = (400.0, 400.0, W / 2, H / 2, 160.0, 0.4)
    for bounds in ((0.0, float(W), 0.0, float(H)), (-7.35, W + 9.6, -4.2, H + 5.75)):  # whole-number and distorted-camera bounds
        lit = LM.Frame(k, None, None, bounds, cam, np.ones(8, np.float32))
        for slot in (None, 0):  # uploaded arrays / the frame where the extraction left it in HBM
            rc, off, idx = _grid(ctx, ctx._view(k, None, d, bounds, device_slot=slot), len(k))
            assert rc == 0
            got = [[idx[off[i * 48 + j]: off[i * 48 + j + 1]].tolist() for j in range(48)] for i in range(64)]
            assert got == lit.mGrid, (bounds, slot)
            assert off[-1] == sum(len(c) for col in lit.mGrid for c in col)
    rc, off, _ = _grid(ctx, ctx._view(k[:0], None, d[:0], (0.0, float(W), 0.0, float(H))), 0)
    assert rc == 0 and not off.any()
    ctx.close()


def test_resident_view_of_another_frame_is_refused():
    from orbslam2_amd import api
    ctx = api.Context(**KW)
    left, right = synth.stereo_pair(W, H, seed=62)
    out = ctx.stereo_frame(left, right)
    k, d = out["kps_left"], out["desc_left"]
    bounds = (0.0, float(W), 0.0, float(H))
    ok = ctx.features_in_area(ctx._view(k, None, d, bounds, device_slot=0), 200.0, 150.0, 30.0)
    assert len(ok) > 0
    with pytest.raises(api.OrbfeError):  # a view with one keypoint less than the slot holds: some other frame
        ctx.features_in_area(ctx._view(k[:-1], None, d[:-1], bounds, device_slot=0), 200.0, 150.0, 30.0)
    ctx.extract(left)  # now the latest call filled ONE slot
    with pytest.raises(api.OrbfeError):
        ctx.features_in_area(ctx._view(out["kps_right"], None, out["desc_right"], bounds, device_slot=1), 200.0, 150.0, 30.0)
    ctx.close()


def test_stereo_batch_equals_the_enqueue_path_and_getters():
    import torch
    from orbslam2_amd import api
    P = 5
    ctx = api.Context(max_images=2 * P, **KW)
    L = ctx.L
    L.orbfe_device_count.restype = C.c_int
    assert L.orbfe_device_count() >= 1
    cam = np.zeros(5, np.float32)
    L.orbfe_get_camera.restype = C.c_int; L.orbfe_get_camera.argtypes = [C.c_void_p, C.c_void_p]
    assert L.orbfe_get_camera(ctx.h, _p(cam)) == 0 and cam.tolist() == [400.0, 400.0, W / 2, H / 2, 160.0]
    L.orbfe_vocab_bytes.restype = C.c_longlong; L.orbfe_vocab_bytes.argtypes = [C.c_void_p]
    assert L.orbfe_vocab_bytes(ctx.h) == 0
    from orbslam2_amd import bow as B
    blob = B.build_vocabulary(np.random.default_rng(1).integers(0, 256, (3000, 32), dtype=np.uint8), k=10, levels=3)
    B.vocab_load(ctx, blob)
    assert 0 < L.orbfe_vocab_bytes(ctx.h) <= len(blob)
    host = np.stack([np.stack(synth.stereo_pair(W, H, seed=80 + i)) for i in range(P)]).reshape(2 * P, H, W)
    cap = ctx.capacity
    kps = np.zeros((2 * P, cap), api.KP_DTYPE); desc = np.zeros((2 * P, cap, 32), np.uint8); cnt = np.zeros(2 * P, np.int32)
    ur = np.zeros((2 * P, cap), np.float32); dp = np.zeros((2 * P, cap), np.float32)
    L.orbfe_stereo_batch.restype = C.c_int
    L.orbfe_stereo_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5
    assert L.orbfe_stereo_batch(ctx.h, _p(host), P, _p(kps), _p(desc), _p(cnt), _p(ur), _p(dp)) == 0
    d_images = torch.from_numpy(host).cuda()
    ctx.enqueue_stereo(d_images.data_ptr(), P)
    ctx.synchronize()
    for i in range(2 * P):
        ref = ctx.fetch_image(i, stereo=(i % 2 == 0))
        n = cnt[i]
        assert n == len(ref["kps"]) and np.array_equal(kps[i, :n], ref["kps"]) and np.array_equal(desc[i, :n], ref["desc"])
        if i % 2 == 0:
            assert np.array_equal(ur[i, :n], ref["u_right"]) and np.array_equal(dp[i, :n], ref["depth"])
    exl, exr = O.Extractor(nfeatures=NF), O.Extractor(nfeatures=NF)  # and the last pair against the oracle
    kl, dl = exl.extract(host[2 * P - 2]); kr, dr = exr.extract(host[2 * P - 1])
    uro, dpo, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, 160.0, 400.0)
    assert np.array_equal(kps[2 * P - 2, :cnt[2 * P - 2]], kl.astype(api.KP_DTYPE)) and np.array_equal(ur[2 * P - 2, :len(kl)], uro)
    assert L.orbfe_stereo_batch(ctx.h, _p(host), P + 1, _p(kps), _p(desc), _p(cnt), _p(ur), _p(dp)) == -4  # ORBFE_ERR_CAPACITY
    ctx.close()


def test_callers_stream_may_die_between_enqueue_and_fetch():
    """A caller may enqueue on its own stream, synchronise it and destroy it before it fetches or matches: the library waits on an
    event of its own, recorded on that stream at enqueue time, never on the stream handle."""
    import torch
    from orbslam2_amd import api
    ctx = api.Context(**KW)
    left, right = synth.stereo_pair(W, H, seed=63)
    d_images = torch.from_numpy(np.stack([left, right])).cuda()
    ref = ctx.stereo_frame(left, right)
    for _ in range(3):
        s = torch.cuda.Stream()
        ctx.enqueue_stereo(d_images.data_ptr(), 1, s.cuda_stream)
        s.synchronize()
        del s  # torch returns / destroys the stream; the context must not touch it again
        got = ctx.fetch_image(0, stereo=True)
        assert np.array_equal(got["kps"], ref["kps_left"]) and np.array_equal(got["u_right"], ref["u_right"])
        hits = ctx.features_in_area(ctx._view(got["kps"], None, got["desc"], (0.0, float(W), 0.0, float(H)), device_slot=0), 240.0, 160.0, 40.0)
        assert len(hits) > 0
    # and without the caller's synchronise: the blocking fetch itself waits for the enqueued work
    s = torch.cuda.Stream()
    ctx.enqueue_stereo(d_images.data_ptr(), 1, s.cuda_stream)
    got = ctx.fetch_image(0, stereo=True)
    assert np.array_equal(got["desc"], ref["desc_left"])
    ctx.close()


def test_profiling_interval_samples_every_nth_call():
    import torch
    from orbslam2_amd import api
    ctx = api.Context(**KW)
    left, right = synth.stereo_pair(W, H, seed=64)
    d_images = torch.from_numpy(np.stack([left, right])).cuda()
    ctx.set_profiling(1)
    ctx.set_profiling_interval(4)
    for _ in range(10):
        ctx.enqueue_stereo(d_images.data_ptr(), 1)
    ms, calls = ctx.stage_times(reset=True)
    assert calls == 3 and ms["fast"] > 0 and ms["describe"] > 0  # calls 0, 4, 8
    ctx.set_profiling_interval(1)
    for _ in range(5):
        ctx.enqueue_stereo(d_images.data_ptr(), 1)
    assert ctx.stage_times(reset=True)[1] == 5
    ctx.close()
