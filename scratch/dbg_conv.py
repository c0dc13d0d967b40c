import sys, torch, torch.nn.functional as F
sys.path.insert(0,'kidney-diffusion_amd')
from imagen_pytorch import _engine as E
lib=E.load(); dev=torch.device('cuda:0')
def run(B,H,W,Cin,Cout,K,stride,pad):
    g=torch.Generator().manual_seed(1)
    x=torch.randn(B,Cin,H,W,generator=g); w=torch.randn(Cout,Cin,K,K,generator=g)*(Cin*K*K)**-0.5; b=torch.zeros(Cout)
    ref=F.conv2d(x,w,b,stride=stride,padding=pad)
    xd=x.permute(0,2,3,1).contiguous().to(dev); Ho,Wo=ref.shape[-2:]
    y=torch.full((B,Ho,Wo,Cout),float('nan'),device=dev)
    E.check(lib.kd_conv2d_nhwc(E.ptr(xd),E.ptr(w.to(dev)),E.ptr(b.to(dev)),E.ptr(y),B,H,W,Cin,Cout,K,K,stride,pad,0,E.current_stream()))
    got=y.permute(0,3,1,2).cpu()
    d=(got-ref).abs()
    print((B,H,W,Cin,Cout,K), 'rel', float((got-ref).norm()/ref.norm()))
    # error by output channel block of 8 and by flattened pixel index mod 8
    e_n = d.mean(dim=(0,2,3)); print(' per-n(first 64):', [round(float(v),3) for v in e_n[:64:4]])
    dm = d.permute(0,2,3,1).reshape(-1,Cout).mean(1); print(' per-m(first 32):', [round(float(v),3) for v in dm[:32]])
    ratio = (got/ref).permute(0,2,3,1).reshape(-1,Cout)
    print(' ratio sample', ratio[0,:8].tolist())
run(1,8,16,32,32,1,1,0)
run(1,8,16,32,64,1,1,0)
run(1,8,16,32,128,1,1,0)
run(1,8,16,64,32,1,1,0)
run(1,8,16,8,32,1,1,0)
