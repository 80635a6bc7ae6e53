"""How many FAST cells of the benchmark images need the second attempt (minThFAST): a cell whose best candidate scores below iniThFAST
(or that has no candidate at all) ran cv::FAST twice (src/ORBextractor.cc:803-810)."""
import sys, math, numpy as np
sys.path.insert(0, '.')
from orbslam2_amd import api, synth
W, H = 1241, 376
def stats(name, img_l, img_r):
    ctx = api.Context(width=W, height=H, nfeatures=2000)
    ctx.stereo_frame(img_l, img_r)
    tot_cells = tot_second = tot_empty = 0
    for lv in range(8):
        xs, ys, sc = ctx.fetch_candidates(0, lv)
        w, h = ctx.level_size(lv)
        bw, bh = (w - 16) - 16, (h - 16) - 16
        ncols, nrows = bw // 30, bh // 30
        wc, hc = math.ceil(bw / ncols), math.ceil(bh / nrows)
        xs, ys, sc = np.array(xs), np.array(ys), np.array(sc)
        cj, ci = np.minimum(xs // wc, ncols - 1), np.minimum(ys // hc, nrows - 1)
        best = np.zeros((nrows, ncols), np.int32)
        np.maximum.at(best, (ci, cj), sc)
        second = int((best < 20).sum()); empty = int((best == 0).sum())
        tot_cells += nrows * ncols; tot_second += second; tot_empty += empty
        print("%s level %d: %3d x %2d cells, second attempt in %3d (%4.1f %%), of which without any corner %d; candidates %d" % (name, lv, ncols, nrows, second, 100.0 * second / (nrows * ncols), empty, len(xs)))
    print("%s: %d of %d cells take the second attempt (%.1f %%), %d of them find nothing" % (name, tot_second, tot_cells, 100.0 * tot_second / tot_cells, tot_empty))
    ctx.close()
l, r = synth.stereo_pair(W, H, seed=1234)
stats("synthetic", l, r)
try:  # tools may use the oracle (it only resizes the photograph here)
    sys.path.insert(0, 'tests')
    import natural
    left, right, d, nf = natural.pair("china_kitti")
    stats("photograph", left, right)
except Exception as e:
    print("photograph: skipped (%s)" % e)
