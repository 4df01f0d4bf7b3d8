import sys; sys.path.insert(0,'/root/repo')
import torch
from dream_gnn_amd import ops, synth
dev=torch.device('cuda:0')
gen=torch.Generator(device=dev).manual_seed(4)
def t(fn):
    for _ in range(5): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(30): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/30
for n_dst,n_src,E in ((6250,100_000,10_000_000),(12500,50_000,10_000_000),(3125,100_000,10_000_000),(25000,100_000,10_000_000)):
    dst=torch.randint(0,n_dst,(E,),generator=gen,device=dev,dtype=torch.int32)
    src=torch.randint(0,n_src,(E,),generator=gen,device=dev,dtype=torch.int32)
    g=ops.CSRGraph(dst,src,n_dst,n_src)
    X=torch.randn(n_src,128,device=dev); ss=torch.rand(n_src,device=dev); ds=torch.rand(n_dst,device=dev)
    y=torch.empty(n_dst,128,device=dev)
    ms_sl=t(lambda: g.spmm(X,ss,ds,out=y))
    ref=y.clone()
    sp=ops._SplitSliced(g.indptr,g.eid,g._S.src,n_dst,n_src)
    ms_sp=t(lambda: sp.spmm(X,ss,ds,y,None))
    print("%6d rows x %6d sources, degree %5d: sliced %.4f ms   split(256-edge virtual rows, %d) %.4f ms   max|d| %.2e" % (n_dst,n_src,E//n_dst,ms_sl,sp.n_virtual,ms_sp,float((y-ref).abs().max()/ref.abs().max())))
