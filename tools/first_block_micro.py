"""block_one's backward at the bench shape (B = 4, 96^3, bf16): norm statistics + weight gradient with the norm backward on load (two launches' worth
of passes) against the one-pass form (dycon_first_block_bwd)."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd import ops
dev="cuda:0"; B,S,C=4,96,16; V=S**3
x=torch.randn(B,S,S,S,1,device=dev).bfloat16(); z=torch.randn(B,S,S,S,C,device=dev).bfloat16(); gy=torch.randn(B,S,S,S,C,device=dev).bfloat16()
gamma,beta=torch.ones(C,device=dev),torch.zeros(C,device=dev)
_,stats=ops.norm_fwd(z,B,V,C,16,gamma,beta,True)
gw,gb,dg,db=torch.empty(C,1,3,3,3,device=dev),torch.empty(C,device=dev),torch.empty(C,device=dev),torch.empty(C,device=dev)
def t(fn,reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps*1e3
def two():
    ws,ab=ops.norm_bwd_stats(z,gy,stats,B,V,C,16,gamma,beta,True,dg,db)
    ops.conv1_wgrad_normbwd(x,z,gy,stats,ab,B,16,gw,gb,gamma,beta,True)
def one(): ops.first_block_bwd(x,z,gy,stats,B,16,gw,gb,gamma,beta,True,dg,db)
print("two-step %.1f us   one-pass %.1f us" % (t(two), t(one)))
