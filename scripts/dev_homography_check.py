import numpy as np, sys
sys.path.insert(0,'.')
import sift_amd as S
from oracle import oracle as O
ctx=S.Context(0)
rng=np.random.default_rng(1)
Ht=np.array([[1.1,0.05,30],[-0.03,0.95,-12],[1e-5,-2e-5,1]])
for n,out in ((500,0.0),(500,0.5),(2000,0.7),(5,0.0),(4,0.0),(50,0.9),(3,0),(10000,0.6)):
    src=(rng.random((n,2))*[1920,1080]).astype(np.float32)
    p=np.c_[src,np.ones(n)]@Ht.T; dst=(p[:,:2]/p[:,2:]+rng.normal(0,0.3,(n,2))).astype(np.float32)
    k=int(n*out); dst[:k]=(rng.random((k,2))*[1920,1080]).astype(np.float32)
    H,m,i=ctx.find_homography(src,dst)
    Ho,mo,io=O.find_homography(src,dst)
    same = (H is None and Ho is None) or (H is not None and Ho is not None and np.array_equal(H.view(np.uint64),Ho.view(np.uint64)))
    print(n,out,i, 'H bit-equal', same, 'mask equal', np.array_equal(m,mo), {k:io[k] for k in io})
import time
for n,out in ((300,0.3),(3000,0.5),(3000,0.8),(20000,0.5)):
    src=(rng.random((n,2))*[1920,1080]).astype(np.float32)
    p=np.c_[src,np.ones(n)]@Ht.T; dst=(p[:,:2]/p[:,2:]+rng.normal(0,0.3,(n,2))).astype(np.float32)
    k=int(n*out); dst[:k]=(rng.random((k,2))*[1920,1080]).astype(np.float32)
    ctx.find_homography(src,dst)
    t=time.perf_counter(); 
    for _ in range(10): H,m,i=ctx.find_homography(src,dst)
    tg=(time.perf_counter()-t)/10
    t=time.perf_counter(); O.find_homography(src,dst); tc=time.perf_counter()-t
    print('n',n,'outliers',out,'iters',i['iterations'],'scored',i['hypotheses_scored'],'gpu ms %.3f cpu ms %.3f'%(tg*1e3,tc*1e3))
