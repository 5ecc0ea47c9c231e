import sys, numpy as np
sys.path.insert(0,'.')
from amico_amd import _capi, get_context, synthetic as S
from oracle import oracle
dirs=S.fibonacci_hemisphere(500); ht=S.build_htable(dirs)
s1=S.make_scheme(1,((1000.0,64),),seed=3)
K=S.freewater_kernels(s1,dirs)
y,d=S.freewater_signals(100000,K,ht,s1,seed=2)
ctx=get_context(); lut=_capi.upload_freewater(ctx,K,ht)
for lam2 in (1e-3, 1e-6):
    got=_capi.freewater_fit(ctx,lut,y,d,0.0,lam2,False)[0]
    st=ctx.last_stats()
    ref=oracle.freewater_fit(y[:20000],d[:20000],K,ht,lambda2=lam2,nthreads=64)['estimates']
    print('lam2',lam2,'max diff',np.abs(got[:20000]-ref).max(), st)
avg=S.directional_average_scheme(S.make_sandi_scheme()); Ks,Rs,d_in,d_isos=S.sandi_kernels(avg)
ys=S.sandi_signals(100000,Ks,avg,seed=2); ls=_capi.upload_sandi(ctx,Ks,Rs,d_in,d_isos)
for lam2 in (5e-3, 2e-6):
    got=_capi.sandi_fit(ctx,ls,ys,0.0,lam2)[0]; st=ctx.last_stats()
    ref=oracle.sandi_fit(ys[:20000],Ks,Rs,d_in,d_isos,lambda2=lam2,nthreads=64)['estimates']
    print('sandi lam2',lam2,'max rel diff',(np.abs(got[:20000]-ref)/np.maximum(np.abs(ref),1)).max(), st)
