"""Per-op profile of one forward of an ultra-res UNet (train_ultra_res.py:29-60) at a given batch:
   python scratch/dump_ops_stage.py <stage 1|2|3> <batch>"""
import sys, ctypes as C, torch
sys.path.insert(0,'.'); sys.path.insert(0,'kidney-diffusion_amd')
import bench
import imagen_pytorch as ip
from imagen_pytorch import _engine as E
lib=E.load(); dev=torch.device('cuda:0')
stage=int(sys.argv[1]); B=int(sys.argv[2]); S={1:64,2:256,3:1024}[stage]
torch.manual_seed(0)
with torch.device('meta'):
    u=ip.Unet(**bench.ULTRA_UNETS[stage], lowres_cond=stage>1, cond_on_text=False, text_embed_dim=None)
u=u.to_empty(device=dev)
with torch.no_grad():
    for p in u.parameters(): p.normal_(0, 0.02)
h=u.engine(B, S, dev, with_text=False)
x=torch.randn(B,3,S,S,device=dev); lr=torch.randn(B,3,S,S,device=dev) if stage>1 else None; cond=torch.rand(B,3,S,S,device=dev)
t=torch.full((B,),0.3,device=dev); tl=torch.full((B,),-1.0,device=dev) if stage>1 else None; out=torch.empty_like(x)
E.check(lib.kd_unet_forward(h,E.ptr(x),E.ptr(lr),E.ptr(cond),E.ptr(t),E.ptr(tl),None,None,E.ptr(out),E.current_stream()))
buf=C.create_string_buffer(1<<21)
E.check(lib.kd_unet_profile(h,5,buf,len(buf),E.current_stream()))
print(buf.value.decode())
