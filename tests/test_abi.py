"""The C-ABI library loads without a GPU, exports every symbol include/orbfe.h declares, validates its
arguments, and fails loudly (ORBFE_ERR_NO_DEVICE) instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import pytest

from orbslam2_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "orbfe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(orbfe_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    lib = api.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "liborbfe.so does not export %s" % n
    assert sorted(api.EXPORTS) == names, "api.EXPORTS out of sync with include/orbfe.h"
    assert lib.orbfe_abi_version() == 6  # 2: orbfe_frame_view.device_slot_plus1; 3: .keyframe; 4: orbfe_get_camera, orbfe_assign_features_to_grid; 5: packed fetch, orbfe_enqueue_rgbd; 6: orbfe_build_id, orbfe_set_pattern, orbfe_get_pattern
    bid = lib.orbfe_build_id().decode()
    assert len(bid) == 64 and all(c in "0123456789abcdef" for c in bid), bid


def test_struct_layouts():
    assert C.sizeof(api.Params) == 17 * 4
    assert C.sizeof(api.FrameView) == 4 + 4 + 3 * 8 + 4 * 4 + 4 + 4  # n, pad, 3 pointers, 4 floats, device_slot_plus1, keyframe
    assert api.KP_DTYPE.itemsize == 28


def test_invalid_params_rejected():
    lib = api.load()
    h = C.c_void_p()
    bad = api.Params(2000, 1.2, 0, 20, 7, 31, 15, 19, 1, 1, 0, 0, 1, 0, 640, 480, 2)  # nlevels = 0
    assert lib.orbfe_create(C.byref(bad), C.byref(h)) == api.ERR_INVALID
    assert b"invalid" in lib.orbfe_last_error(None)
    assert lib.orbfe_create(None, C.byref(h)) == api.ERR_INVALID
    # the rotated test pattern reaches 18 px from a keypoint: an edge threshold below the reference's 19 would let the descriptor
    # stage read outside the level image
    shallow = api.Params(2000, 1.2, 8, 20, 7, 19, 9, 15, 1, 1, 0, 0, 1, 0, 640, 480, 2)  # half_patch 9, edge 15
    assert lib.orbfe_create(C.byref(shallow), C.byref(h)) == api.ERR_INVALID


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked tests")
    with pytest.raises(api.OrbfeError) as e:
        api.Context(width=640, height=480)
    assert e.value.code == api.ERR_NO_DEVICE
    assert "no CPU path" in str(e.value)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "orbslam2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liborb_oracle" not in text and "from oracle" not in text and "import oracle" not in text \
                    and "orb_oracle.h" not in text, os.path.join(dirpath, f)


def test_status_returning_entry_points_are_function_try_blocks():
    """No C++ exception may cross the C ABI: every multi-line `extern "C" int` definition of the library is a function-try-block
    closed by ORBFE_CATCH (orbfe_host.h), the PNG ones by their own handlers."""
    import glob
    import re
    csrc = os.path.join(ROOT, "orbslam2_amd", "csrc")
    checked = 0
    for path in glob.glob(os.path.join(csrc, "*.hip")):
        lines = open(path).read().split("\n")
        for i, line in enumerate(lines):
            if not line.startswith('extern "C" int ') or line.rstrip().endswith(("}", ";")):
                continue
            j = i
            while not re.match(r"^(try )?\{$", lines[j]):
                j += 1
                assert j < i + 12, (path, i + 1)
            assert lines[j] == "try {", "%s:%d: entry point without a function-try-block" % (os.path.basename(path), i + 1)
            k = j + 1
            while not lines[k].startswith("}"):
                k += 1
            assert lines[k].startswith("} ORBFE_CATCH("), "%s:%d" % (os.path.basename(path), k + 1)
            checked += 1
    assert checked >= 60


def test_product_three_maxima_selection_equals_the_oracles_running_top_three():
    """orbfe_three_maxima needs no device: the product selects the three largest (size, -index) keys, the oracle keeps the reference's
    shifting if-chain (src/ORBmatcher.cc:1597-1638) -- two formulations, checked against each other on random, tied and empty bins."""
    import ctypes as C
    import numpy as np
    from oracle import oracle as O
    from orbslam2_amd import api
    lib = api.load()
    lib.orbfe_three_maxima.restype = C.c_int
    lib.orbfe_three_maxima.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 3
    rng = np.random.default_rng(3)
    cases = [np.zeros(30, np.int32), np.full(30, 5, np.int32), np.array([0] * 12 + [9] + [0] * 17, np.int32)]
    for _ in range(400):
        h = rng.integers(0, rng.integers(1, 40), 30).astype(np.int32)
        if rng.random() < 0.5:
            h[rng.integers(0, 30, 6)] = h.max()      # ties for the lead
        if rng.random() < 0.3:
            h[rng.random(30) < 0.8] = 0              # mostly empty (only bins 0..12 are ever filled by the matchers)
        cases.append(h)
    for h in cases:
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        assert lib.orbfe_three_maxima(h.ctypes.data_as(C.c_void_p), len(h), C.byref(a), C.byref(b), C.byref(c)) == 0
        assert (a.value, b.value, c.value) == tuple(O.three_maxima(h.tolist())), h.tolist()
