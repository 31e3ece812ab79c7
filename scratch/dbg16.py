import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_parity as TP
from oracle import drnmf_oracle as O
dev = torch.device('cuda:0')
seq = sys.argv[1]          # e.g. "hf" = half then float
B,T,F,r,K = 3,1,16,8,2
P, alt, labels, N = TP._problem(B, T, F, r, K)
lay, u = O.maps_factored(alt, labels, K), O.u_scalars(alt)
emu = O.cell_forward_factored(P["X"], lay, u, P["log_h0"], operand_dtype=np.float16)
ex = O.cell_forward_factored(P["X"], lay, u, P["log_h0"])
out = []
for c in seq:
    h, _, _ = TP._run_cell(dev, P, alt, labels, N, K, operand_f16=(c == 'h'))
    out.append('%s:%.2e' % (c, np.abs(h - (emu if c == 'h' else ex)).max()))
print(seq, ' '.join(out))
