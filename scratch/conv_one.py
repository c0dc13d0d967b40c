import sys, ctypes as C
sys.path.insert(0,'kidney-diffusion_amd')
from imagen_pytorch import _engine as E
import torch; torch.zeros(1,device='cuda')
lib=E.load(); f=C.CDLL(str(E._LIB_PATH.parents[2]/"scratch"/"conv_x"/"libkd_conv_bench.so")).kd_conv_bench; f.restype=C.c_int
v=int(sys.argv[1]); shape=[int(a) for a in sys.argv[2].split(',')]
B,H,W,Ci,Co,K=shape
us=C.c_float(); cs=C.c_float()
rc=f(B,H,W,Ci,Co,K,1,K//2,v,10,C.byref(us),C.byref(cs))
if rc: print('ERR', lib.kd_last_error())
print(v, shape, 2.0*B*H*W*Ci*Co*K*K/us.value/1e6 if us.value else 0, 'TF', us.value,'us', cs.value)
