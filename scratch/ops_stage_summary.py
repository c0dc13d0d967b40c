"""Per-op profile of one forward of an ultra-res UNet (train_ultra_res.py:29-60) at a given batch, summed by label:
   python scratch/ops_stage_summary.py <stage 1|2|3> <batch> [top]"""
import sys, ctypes as C, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'kidney-diffusion_amd'))
import bench
import imagen_pytorch as ip
from imagen_pytorch import _engine as E
lib = E.load(); dev = torch.device('cuda:0')
stage = int(sys.argv[1]); B = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
S = {1: 64, 2: 256, 3: 1024}[stage]
with torch.device('meta'):
    u = ip.Unet(**bench.ULTRA_UNETS[stage], lowres_cond=stage > 1, cond_on_text=False, text_embed_dim=None)
u = u.to_empty(device=dev)
with torch.no_grad():
    for p in u.parameters(): p.normal_(0, 0.02)
h = u.engine(B, S, dev, with_text=False)
x = torch.randn(B, 3, S, S, device=dev); lr = torch.randn(B, 3, S, S, device=dev) if stage > 1 else None; cond = torch.rand(B, 3, S, S, device=dev)
t = torch.full((B,), 0.3, device=dev); tl = torch.full((B,), -1.0, device=dev) if stage > 1 else None; out = torch.empty_like(x)
E.check(lib.kd_unet_forward(h, E.ptr(x), E.ptr(lr), E.ptr(cond), E.ptr(t), E.ptr(tl), None, None, E.ptr(out), E.current_stream()))
buf = C.create_string_buffer(1 << 21)
E.check(lib.kd_unet_profile(h, 5, buf, len(buf), E.current_stream()))
rows = [l.split(',') for l in buf.value.decode().strip().split('\n')[1:]]
tot = sum(float(r[3]) for r in rows); agg = {}
for r in rows:
    a = agg.setdefault(r[1], [0, 0.0]); a[0] += 1; a[1] += float(r[3])
print(f"stage {stage} batch {B}: {len(rows)} launches, {tot / 1e3:.3f} ms (sum of per-op times), cond launches {lib.kd_unet_num_cond_launches(h)}")
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{us:9.1f} us  x{n:<3d} {us / n:8.1f} us each  {k}")
