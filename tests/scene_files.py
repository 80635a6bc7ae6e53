"""Scene files shared by the C++ test drivers (tests/compat_stub/compat_selftest.cpp, tests/asan/resolve_harness.cpp): the
keypoint-level scene of tests/test_matchers.py plus the extras the drivers need, written as raw little-endian arrays."""
import numpy as np

from oracle import oracle as O
from tests.test_matchers import BF, CAM, CX, CY, FX, FY, H, LOG_SF, NL, W, _scene


def _bow_nodes(n, rng, shared):
    """Synthetic fBow2 maps: keypoint i of both frames mostly falls into the same vocabulary node."""
    node = (np.arange(n) * 7919 % 61).astype(np.uint32)
    node2 = node.copy()
    flip = rng.random(n) > shared
    node2[flip] = rng.integers(0, 61, int(flip.sum()))

    def csr(nd):
        nodes = np.unique(nd)
        off = [0]; feat = []
        for v in nodes:
            idx = np.nonzero(nd == v)[0]
            feat.extend(idx.tolist()); off.append(len(feat))
        return nodes.astype(np.uint32), np.array(off, np.int32), np.array(feat, np.int32)
    return csr(node), csr(node2)


def write_scene(d, seed=77):
    """Writes the scene into directory d (a pathlib.Path) and returns everything the expectations need."""
    s = _scene(seed, n_last=1100, n_distract=450)
    rng = np.random.default_rng(seed + 1)
    m, n = len(s["pos"]), len(s["k"])
    valid3 = s["valid"].copy()                      # 0: no map point, 1: usable, 2: present but outlier / bad
    valid3[(valid3 == 1) & (rng.random(m) < 0.1)] = 2
    usable = (valid3 == 1).astype(np.int32)
    dist0 = np.linalg.norm(s["pos"], axis=1).astype(np.float32)
    max_d = (dist0 * s["sf"][s["octave"]]).astype(np.float32)
    min_d = (max_d / s["sf"][NL - 1]).astype(np.float32)
    normal = (s["pos"] / dist0[:, None] + rng.normal(0, 0.4, (m, 3))).astype(np.float32)
    normal = (normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32)
    tp = O.is_in_frustum(s["T_cur"], CAM, s["bounds"], s["pos"], normal, max_d, min_d, 0.5, LOG_SF, NL)
    found = (rng.random(m) < 0.15).astype(np.uint8)
    k1 = s["k"].copy(); k1["octave"] = np.where(np.arange(n) % 4 == 0, 1, 0)
    k2 = k1.copy(); k2["x"] += rng.normal(5, 2, n).astype(np.float32); k2["y"] += rng.normal(-3, 2, n).astype(np.float32)
    k2["octave"] = np.where(np.arange(n) % 7 == 0, 2, 0)
    d2 = s["d"] ^ np.packbits(rng.random((n, 256)) < 0.05, axis=1, bitorder="little")
    kf_fv, f_fv = _bow_nodes(n, rng, 0.9)
    bow_valid = rng.choice([0, 1, 1, 1, 2], n).astype(np.int32)
    fuse_kf_obs = np.where(rng.random(n) < 0.4, rng.integers(0, 4, n), -1).astype(np.int32)
    files = {"cam.f32": np.array([FX, FY, CX, CY, BF, W, H], np.float32), "bounds.f32": np.array(s["bounds"], np.float32),
             "cur_k.bin": s["k"], "cur_d.bin": s["d"], "cur_ur.bin": s["ur"], "cur_has_obs.bin": s["cur_has_obs"],
             "last_pos.bin": s["pos"], "last_oct.bin": s["octave"], "last_ang.bin": s["angle"], "last_desc.bin": s["desc_last"],
             "last_valid.bin": valid3, "last_obs.bin": s["obs"], "T_last.bin": s["T_last"], "T_cur.bin": s["T_cur"],
             "normal.bin": normal, "max_d.bin": max_d, "min_d.bin": min_d, "tp.bin": tp, "already_found.bin": found,
             "init_k1.bin": k1, "init_d1.bin": s["d"], "init_k2.bin": k2, "init_d2.bin": d2,
             "bow_kf_valid.bin": bow_valid, "bow_kf_nodes.bin": kf_fv[0], "bow_kf_off.bin": kf_fv[1], "bow_kf_feat.bin": kf_fv[2],
             "bow_f_nodes.bin": f_fv[0], "bow_f_off.bin": f_fv[1], "bow_f_feat.bin": f_fv[2], "fuse_kf_obs.bin": fuse_kf_obs}
    from orbslam2_amd import synth
    files["rendered_image.bin"] = synth.stereo_pair(W, H, seed=seed + 5)[0]  # a full rendered image (not a keypoint-level scene) for the driver's resident-frame section
    for name, a in files.items():
        np.ascontiguousarray(a).tofile(d / name)
    return dict(s=s, m=m, n=n, valid3=valid3, usable=usable, max_d=max_d, min_d=min_d, normal=normal, tp=tp, found=found, k1=k1, k2=k2, d2=d2,
                kf_fv=kf_fv, f_fv=f_fv, bow_valid=bow_valid, fuse_kf_obs=fuse_kf_obs)
