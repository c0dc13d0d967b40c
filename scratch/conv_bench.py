import sys, ctypes as C
sys.path.insert(0,'kidney-diffusion_amd')
from imagen_pytorch import _engine as E
import torch; torch.cuda.init(); torch.zeros(1,device='cuda')
lib=E.load()
f=C.CDLL(str(E._LIB_PATH.parents[2]/"scratch"/"conv_x"/"libkd_conv_bench.so")).kd_conv_bench; f.restype=C.c_int
shapes=[(16,128,128,128,128,3),(16,64,64,256,256,3),(16,16,16,1024,1024,3),(16,32,32,512,512,3)]
variants=[int(v) for v in sys.argv[1].split(',')] if len(sys.argv)>1 else list(range(10))
for (B,H,W,Ci,Co,K) in shapes:
    flop=2.0*B*H*W*Ci*Co*K*K
    res=[]
    for v in variants:
        us=C.c_float(); cs=C.c_float()
        rc=f(B,H,W,Ci,Co,K,1,K//2,v,5,C.byref(us),C.byref(cs))
        res.append(f"v{v}:{flop/us.value/1e6:6.1f}TF({cs.value:.3f})")
    print((B,H,W,Ci,Co,K), ' '.join(res), flush=True)
