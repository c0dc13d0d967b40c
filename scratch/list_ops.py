# Per-launch profile of the headline UNet (batch 16, 256x256): python scratch/list_ops.py [min_us]
import sys, ctypes as C, torch
sys.path.insert(0, "."); sys.path.insert(0, "kidney-diffusion_amd")
import bench
from imagen_pytorch import _engine as E
lib = E.load(); dev = torch.device("cuda:0")
u = bench.build_unet(0).to(dev)
h = u.engine(16, 256, dev, with_text=False)
x, lowres, ln, cond = bench.synthetic_inputs(16, dev)
t = torch.full((16,), 0.3, device=dev); tl = torch.full((16,), -1.0, device=dev)
out = torch.empty_like(x)
E.check(lib.kd_unet_forward(h, E.ptr(x), E.ptr(lowres), E.ptr(cond), E.ptr(t), E.ptr(tl), None, None, E.ptr(out), E.current_stream()))
buf = C.create_string_buffer(1 << 21)
E.check(lib.kd_unet_profile(h, 3, buf, len(buf), E.current_stream()))
rows = [l.split(",") for l in buf.value.decode().strip().split("\n")[1:]]
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
agg = {}
for r in rows:
    a = agg.setdefault(r[1], [0, 0.0, 0]); a[0] += 1; a[1] += float(r[3]); a[2] += int(r[2])
tot = sum(v[1] for v in agg.values())
print("total us", round(tot, 1))
for k, (n, us, macs) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if us >= thr and "wino" not in k:
        print(f"{us / tot * 100:5.2f}% n={n:2d} us={us:8.1f} TF/s={2 * macs / us / 1e6:6.1f}  {k}")
