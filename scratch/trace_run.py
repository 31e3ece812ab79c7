import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, ctypes
os.environ["DRNMF_TRACE"]="1"
from drnmf_amd import layers, ops, _capi
from oracle import drnmf_oracle as O
B,T,F,r,K=64,3,513,1000,25
P=O.synth_problem(B,T,F,r,seed=1)
p=dict(input_dim=F,hidden_dim=2*r,output_dim=F,mask_value=-1.,maxseq=T,K_layers=K,W=P["W"],alph=400.,lam1=1.,params_untied=["log_D","log_alph"],params_trainable=["log_D"])
m=layers.build_unfolded_snmf(p)
x=torch.from_numpy(P["X"]).cuda()
for i in range(3):
    h=m.cell.call(x,mask_value=-1.)
torch.cuda.synchronize()
ws=m.cell._ws[(B,T)]
tr=ws[:8*8*8].view(torch.int64).cpu().numpy().reshape(-1,8)
for row in tr[:7]:
    d=row[:5]-row[0]
    print("issue_loads=%d  loads_landed=%d  mfma_done=%d  end=%d  (cycles of s_memtime)"%(d[1],d[2],d[3],d[4]))
