import sys, time
sys.path.insert(0, '.')
import numpy as np
from PIL import Image
import sift_amd as S
from oracle import oracle as O

rgb = np.array(Image.open('tests/golden/Lenna.png').convert('RGB'))
bgr = rgb[:, :, ::-1].copy()
g = O.bgr_to_gray(bgr)
orc = O.OracleSift(g); orc.exec(threads=8)
ctx = S.Context(0)
b = ctx.batch(512, 512, 2)
b.upload(0, g); b.upload(1, bgr)
b.run(2); cnt = b.counts(2); print(cnt)
ok = True
for o in range(orc.octaves):
    for l in range(6):
        a = b.level(0, o, l, S.LEVEL_GAUSS); r = orc.gauss(o, l)
        a1 = b.level(1, o, l, S.LEVEL_GAUSS)
        if not np.array_equal(a, r) or not np.array_equal(a1, r):
            print('gauss mismatch', o, l, np.abs(a-r).max(), (a!=r).sum()); ok=False
    for l in range(5):
        a = b.level(0, o, l, S.LEVEL_DOG); r = orc.dog(o, l)
        if not np.array_equal(a, r):
            print('dog mismatch', o, l, np.abs(a-r).max(), (a!=r).sum()); ok=False
print('pyramid bit-exact:', ok)
c = b.candidates(0); c = c[np.lexsort((c[:,3],c[:,2],c[:,1],c[:,0]))]
oc = orc.candidates(); oc = np.stack([oc['oct'],oc['img'],oc['i'],oc['j']],1)
print('cands', c.shape, oc.shape, np.array_equal(c, oc))
kp, desc = b.download(0)
okp = orc.keypoints(); od = orc.descriptors()
print('n', len(kp), len(okp))
if len(kp)==len(okp):
    for f in ['x','y','size','angle','response','octave']:
        d = (kp[f]!=okp[f]).sum(); print(f, 'mismatch', d, np.abs(kp[f].astype(np.float64)-okp[f]).max())
    dd = desc.astype(np.int32)-od.astype(np.int32)
    print('desc mismatched elems', (dd!=0).sum(), 'rows', (np.abs(dd).max(1)>0).sum(), 'maxabs', np.abs(dd).max())
# knn
rgb2 = np.array(Image.open('tests/golden/LennaRotated.png').convert('RGB'))
g2 = O.bgr_to_gray(rgb2[:,:,::-1].copy())
orc2 = O.OracleSift(g2); orc2.exec(threads=8)
q = orc.descriptors().astype(np.uint8); t = orc2.descriptors().astype(np.uint8)
res, acc = ctx.knn2(q, t)
i1,d1,i2,d2,oacc = O.knn2_u8(q, t, threads=8)
print('knn i1', (res['i1']!=i1).sum(), 'd1', (res['d1']!=d1).sum(), 'i2', (res['i2']!=i2).sum(), 'd2', (res['d2']!=d2).sum(), 'acc', (acc!=oacc).sum(), acc.sum())
# bigger knn
tr = S.synth_descriptors(5, 20000); qq = S.synth_queries(6, tr, 3000)
t0=time.time(); res, acc = ctx.knn2(qq, tr); print('gpu knn', time.time()-t0)
i1,d1,i2,d2,oacc = O.knn2_u8(qq, tr, threads=8)
print('knn2 i1', (res['i1']!=i1).sum(), 'd1', (res['d1']!=d1).sum(), 'i2', (res['i2']!=i2).sum(), 'd2', (res['d2']!=d2).sum(), 'acc', (acc!=oacc).sum(), acc.sum())
print(b.run_profiled(2))
