import sys, ctypes as C, torch
sys.path.insert(0,'.'); sys.path.insert(0,'kidney-diffusion_amd')
import bench
from imagen_pytorch import _engine as E
lib=E.load(); dev=torch.device('cuda:0')
B=int(sys.argv[1]) if len(sys.argv)>1 else 16
u=bench.build_unet(0)
h=u.engine(B, 256, dev, with_text=False)
x,lowres,ln,cond=bench.synthetic_inputs(B, dev)
t=torch.full((B,),0.3,device=dev); tl=torch.full((B,),-1.0,device=dev); out=torch.empty_like(x)
E.check(lib.kd_unet_forward(h,E.ptr(x),E.ptr(lowres),E.ptr(cond),E.ptr(t),E.ptr(tl),None,None,E.ptr(out),E.current_stream()))
buf=C.create_string_buffer(1<<20)
E.check(lib.kd_unet_profile(h,3,buf,len(buf),E.current_stream()))
rows=[l.split(',') for l in buf.value.decode().strip().split('\n')[1:]]
tot=sum(float(r[3]) for r in rows)
print('total us',tot)
agg={}
for r in rows:
    k=r[1]; a=agg.setdefault(k,[0,0.0,0]); a[0]+=1; a[1]+=float(r[3]); a[2]+=int(r[2])
for k,(n,us,macs) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:200]:
    tf = 2*macs/us/1e6 if us>0 else 0
    print(f"{us/tot*100:6.2f}% n={n:3d} us={us:9.1f} TF/s={tf:7.1f}  {k}")
