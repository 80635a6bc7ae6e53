#!/usr/bin/env python3
"""One-off: large frames (4K, 1440p, 1080p; other level counts / scale factors), stereo frame HIP vs oracle bit for bit.
Exercises the run-time tile pitch of fast_cell_kernel, the generic quadtree kernel and geometries whose fused pyramid tail does not fit.
  python3 tools/big_check.py   (on the GPU box)"""
import sys, numpy as np
sys.path.insert(0,'.')
from orbslam2_amd import api, synth
from oracle import oracle as O
for (w,h,nf,nl,sf) in [(3840,2160,5000,8,1.2),(2560,1440,3000,10,1.15),(1920,1080,4000,6,1.3)]:
    l,r=synth.stereo_pair(w,h,seed=4,n_rect=int(6000),n_disc=3000)
    kw=dict(nfeatures=nf,nlevels=nl,scale_factor=sf)
    fx,bf=0.7*w,0.2*w
    ctx=api.Context(width=w,height=h,fx=fx,fy=fx,cx=w/2,cy=h/2,bf=bf,**kw)
    out=ctx.stereo_frame(l,r)
    exl,exr=O.Extractor(**kw),O.Extractor(**kw)
    kl,dl=exl.extract(l); kr,dr=exr.extract(r)
    ur,dp,_=O.stereo_matches(exl,exr,kl,dl,kr,dr,bf,fx)
    ok=(np.array_equal(out["kps_left"],kl.astype(api.KP_DTYPE)) and np.array_equal(out["kps_right"],kr.astype(api.KP_DTYPE)) and np.array_equal(out["desc_left"],dl) and np.array_equal(out["u_right"],ur) and np.array_equal(out["depth"],dp))
    print(w,h,nf,nl,sf,len(kl),'OK' if ok else 'MISMATCH', 'quadtree kernel', ctx.L.orbfe_quadtree_kernel(ctx.h))
    ctx.close()
