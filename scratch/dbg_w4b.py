import sys, torch
sys.path[:0]=['.','kidney-diffusion_amd','tests']
import helpers as H
import test_unet_gpu as TU
dev=torch.device('cuda:0')
for name in (sys.argv[1:] or ["ultra3"]):
    B,S=16,128
    ou=H.oracle_unet(name, lowres_cond=True, seed=23).eval()
    x,lr,cond,t,tl=TU._inputs(name,B,S,True,seed=9)
    with torch.no_grad(): ref=ou(x,t,lowres_cond_img=lr,lowres_noise_times=tl,cond_images=cond)
    dv=lambda v: None if v is None else v.to(dev)
    for algo in (1,4,0):
        pu=H.product_unet_like(ou).to(dev); pu.conv_algo=algo
        got=pu(dv(x),dv(t),lowres_cond_img=dv(lr),lowres_noise_times=dv(tl),cond_images=dv(cond))
        print(name, algo, 'vs oracle', H.rel_l2(got,ref), 'finite', bool(torch.isfinite(got).all()), float(got.abs().max()), float(ref.abs().max()), flush=True)
