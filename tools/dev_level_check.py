import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from infernos_amd import _lib, ops
BF=torch.bfloat16
dev=_lib.require_device('cuda:0')
def run(B,T,taps,acc0,reps=1):
    c=32
    g=torch.Generator().manual_seed(B*13+T+sum(taps))
    x=torch.randn(B,T,c,generator=g).to(BF).to(dev)
    prev=torch.randn(B,T,c,generator=g).to(BF).to(dev)
    blocks=[]
    for k in taps:
        convs=[((torch.randn(c,c,k,generator=g)/(c*k)**0.5).to(BF).float(), torch.randn(c,generator=g)*0.1) for _ in range(6)]
        ws,nu,bias=ops.w_chain_pack(convs,dev); blocks.append((k,ws,nu,bias))
    ref=prev.clone()
    for j,(k,ws,nu,bias) in enumerate(blocks):
        ops.resblock_chain(x,ws,nu,bias,ref,nbatch=B,t=T,c=c,taps=k,scale=1/3,accumulate=(j>0 or acc0))
    for rep in range(reps):
        out=prev.clone()
        ops.resblock_level(x,[(k,ws,bias) for k,ws,nu,bias in blocks],out,nbatch=B,t=T,c=c,scale=1/3,accumulate=acc0)
        torch.cuda.synchronize()
        ne=(out.view(torch.int16)!=ref.view(torch.int16))
        bad=torch.nonzero(ne)
        print('B=%d T=%d taps=%s acc0=%s rep %d: bad elems %d, ref nan %d out nan %d'%(B,T,taps,acc0,rep,bad.size(0),int(ref.float().isnan().sum()),int(out.float().isnan().sum())))
        for b_,t_,c_ in bad[:24].tolist():
            print('    batch %d row %d (tile row %d: wave %d rb %d fr %d) ch %d: out %g ref %g prev %g'%(b_,t_,t_+64,(t_+64)//112,((t_+64)%112)//16,(t_+64)%16,c_,float(out[b_,t_,c_]),float(ref[b_,t_,c_]),float(prev[b_,t_,c_])))
run(256,768,(3,7,11),True,reps=3)
run(64,768,(3,7,11),True,reps=2)
