import sys, numpy as np
sys.path.insert(0,'.')
from orbslam2_amd import api, synth
l,r=synth.stereo_pair(1241,376,seed=1234)
ctx=api.Context(width=1241,height=376,nfeatures=2000)
ctx.stereo_frame(l,r)
for lv in range(8):
    xs,ys,sc=ctx.fetch_candidates(0,lv)
    w,h=ctx.level_size(lv)
    # cells of 30 px approx
    cx=(np.array(xs)//30); cy=(np.array(ys)//30)
    key=cy*100+cx
    u,c=np.unique(key,return_counts=True)
    print(lv,w,h,len(xs),'cells',len(u),'max/cell',c.max() if len(c) else 0,'mean',c.mean() if len(c) else 0)
