"""Per-op profile of one forward of the headline UNet in plan order (label, macs, mfma_macs, avg us)."""
import sys, ctypes as C, torch
sys.path.insert(0,'.'); sys.path.insert(0,'kidney-diffusion_amd')
import bench
from imagen_pytorch import _engine as E
lib=E.load(); dev=torch.device('cuda:0')
B=int(sys.argv[1]) if len(sys.argv)>1 else 16
u=bench.build_unet(0)
import os
if 'KD_W43' in os.environ: u.wino43_min_cin=int(os.environ['KD_W43'])
h=u.engine(B, 256, dev, with_text=False)
x,lowres,ln,cond=bench.synthetic_inputs(B, dev)
t=torch.full((B,),0.3,device=dev); tl=torch.full((B,),-1.0,device=dev); out=torch.empty_like(x)
E.check(lib.kd_unet_forward(h,E.ptr(x),E.ptr(lowres),E.ptr(cond),E.ptr(t),E.ptr(tl),None,None,E.ptr(out),E.current_stream()))
buf=C.create_string_buffer(1<<20)
E.check(lib.kd_unet_profile(h,5,buf,len(buf),E.current_stream()))
print(buf.value.decode())
