import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gsmvi_amd
from oracle import gsm_oracle as orc
eng = gsmvi_amd.get_engine()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
st = orc.make_update_state(D, 8, 1)
S = eng.asarray(st["S0"]); R = eng.empty(D, D); f = eng.new_flag()
for _ in range(12): eng.potrf(S, out=R, flag=f)
torch.cuda.synchronize()
