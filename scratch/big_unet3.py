# unet3 (train_ultra_res.py:51-60) at full dims, 1024x1024, B=1: does the plan fit and run?
import sys, time, torch, ctypes as C
sys.path.insert(0,'kidney-diffusion_amd')
import imagen_pytorch as ip
from imagen_pytorch import _engine as E
dev=torch.device('cuda:0'); lib=E.load()
B=int(sys.argv[1]) if len(sys.argv)>1 else 1
S=int(sys.argv[2]) if len(sys.argv)>2 else 1024
with torch.device('meta'):
    u=ip.Unet(dim=128, dim_mults=(1,2,4,8), num_resnet_blocks=(2,4,6,8), memory_efficient=True, layer_attns=False,
              layer_cross_attns=(False,False,False,True), init_conv_to_final_conv_residual=True, cond_images_channels=3,
              lowres_cond=True, cond_on_text=False, text_embed_dim=None)
u=u.to_empty(device=dev)
for p in u.parameters(): torch.nn.init.normal_(p, std=0.02)
t0=time.time(); h=u.engine(B,S,dev,with_text=False); print('plan', time.time()-t0,'s', 'hbm GB', lib.kd_unet_hbm_bytes(h)/1e9, 'GMAC/sample', lib.kd_unet_macs(h)/1e9/B, 'launches', lib.kd_unet_num_launches(h))
x=torch.randn(B,3,S,S,device=dev); lr=torch.randn(B,3,S,S,device=dev); cond=torch.rand(B,3,S,S,device=dev)
t=torch.full((B,),0.3,device=dev); tl=torch.full((B,),-1.0,device=dev); out=torch.empty_like(x)
for it in range(3):
    torch.cuda.synchronize(); t0=time.time()
    E.check(lib.kd_unet_forward(h,E.ptr(x),E.ptr(lr),E.ptr(cond),E.ptr(t),E.ptr(tl),None,None,E.ptr(out),E.current_stream()))
    torch.cuda.synchronize(); dt=time.time()-t0
    print('fwd', dt*1e3,'ms', 2*lib.kd_unet_macs(h)/dt/1e12,'TF/s', float(out.abs().mean()))
