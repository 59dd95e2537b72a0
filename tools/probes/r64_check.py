import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from self_forcing_amd import ops
def ref(q,k,v):
    qf,kf,vf=q.float().cpu(),k.float().cpu(),v.float().cpu()
    s=torch.einsum("bqhd,bkhd->bhqk",qf,kf)/128**0.5
    return torch.einsum("bhqk,bkhd->bqhd",torch.softmax(s,-1),vf)
for (B,H,Lq,Lk) in [(1,1,64,64),(1,1,64,128),(1,2,100,200),(1,1,256,1000),(2,3,130,24),(1,12,700,1561)]:
    g=torch.Generator().manual_seed(Lq+Lk)
    q=torch.randn(B,Lq,H,128,generator=g).to(torch.bfloat16); k=torch.randn(B,Lk,H,128,generator=g).to(torch.bfloat16); v=torch.randn(B,Lk,H,128,generator=g).to(torch.bfloat16)
    o=ops.attention(q.cuda(),k.cuda(),v.cuda()); torch.cuda.synchronize()
    r=ref(q,k,v); err=((o.float().cpu()-r).norm()/r.norm()).item()
    print((B,H,Lq,Lk),"rel err %.4e"%err, "nan" if torch.isnan(o.float()).any() else "", flush=True)
