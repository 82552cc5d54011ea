import sys,time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, orb_slam3_amd as A
from orb_slam3_amd import synth
A.settle_interpreter()
for (w,h,nf,nb) in ((752,480,1000,64),(752,480,1000,256),(1920,1080,5000,64)):
    src=[synth.scene(w,h,40+i) for i in range(8)]
    ex=A.ORBextractor(nf,1.2,8,20,7,w,h,max_batch=nb)
    batch=[src[i%8] for i in range(nb)]
    for _ in range(2): ex.extract_batch(batch,(0,0),copy=False)
    t0=time.perf_counter(); reps=int(os.environ.get('REPS','20'))
    for _ in range(reps): ex.extract_batch(batch,(0,0),copy=False)
    dt=time.perf_counter()-t0
    print(w,h,nf,nb, reps*nb/dt,'frames/s (list of images in, list of per-frame views out)')
    stack=np.stack(batch)
    for _ in range(2): ex.extract_batch_arrays(stack,(0,0))
    t0=time.perf_counter()
    for _ in range(reps): k,d,n,m=ex.extract_batch_arrays(stack,(0,0))
    dt=time.perf_counter()-t0
    print(w,h,nf,nb, reps*nb/dt,'frames/s (stacked array in, whole arrays out)', int(n.sum()/nb),'keypoints per frame')
    del ex
