import sys, torch
sys.path[:0]=['.','kidney-diffusion_amd','tests']
import helpers as H
from oracle import imagen_ref as R
import imagen_pytorch as ip
dev=torch.device('cuda:0')
base=dict(H.UNET_KW['ultra3'])
def run(nrb, cross=None, B=16, S=128):
    kw=dict(base); kw['num_resnet_blocks']=nrb
    if cross is not None: kw['layer_cross_attns']=cross
    ou=H.randomize_(R.Unet(**kw, lowres_cond=True, cond_on_text=False, text_embed_dim=None), 23).eval()
    g=torch.Generator().manual_seed(9)
    x=torch.randn(B,3,S,S,generator=g); lr=torch.randn(B,3,S,S,generator=g); cond=torch.rand(B,3,S,S,generator=g)
    t=torch.randn(B,generator=g); tl=torch.full((B,),-1.3)
    outs={}
    for algo in (4,1):
        pu=H.product_unet_like(ou).to(dev); pu.conv_algo=algo
        outs[algo]=pu(x.to(dev),t.to(dev),lowres_cond_img=lr.to(dev),lowres_noise_times=tl.to(dev),cond_images=cond.to(dev)).cpu()
    print(nrb, cross, B, S, 'rel', H.rel_l2(outs[4],outs[1]), flush=True)
run((2,2,2,2)); run((2,4,2,2)); run((2,2,6,2)); run((2,2,2,8)); run((3,2,2,2)); run((2,4,6,8)); run((2,2,2,2),(False,False,True,True))
