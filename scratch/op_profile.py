"""Per-op device times of one forward of the headline plan (kd_unet_profile), sorted: where the step's time goes by launch.
  python scratch/op_profile.py [top_n]"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "kidney-diffusion_amd"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from imagen_pytorch import _engine as E  # noqa: E402

top = int(sys.argv[1]) if len(sys.argv) > 1 else 60
w43 = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # Unet.wino43_min_cin (0 = default)
x3 = int(sys.argv[3]) if len(sys.argv) > 3 else 0    # Unet.gemm_bf16x3 (0 = default, -1 = fp32 MFMA position GEMMs)
dev = torch.device("cuda:0")
lib = E.load()
lin = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # Unet.x3_linear (0 = default, -1 = token GEMMs on conv_buf_kernel, n = K >= n)
unet = bench.build_unet(0).to(dev)
unet.wino43_min_cin = w43
unet.gemm_bf16x3 = x3
unet.x3_linear = lin
x, lowres, _, cond = bench.synthetic_inputs(bench.BATCH, dev)
t = torch.zeros(bench.BATCH, device=dev)
with torch.no_grad():
    unet(x, t, lowres_cond_img=lowres, lowres_noise_times=t, cond_images=cond)
h = unet.engine(bench.BATCH, bench.SIZE, dev, with_text=False)
buf = C.create_string_buffer(1 << 20)
E.check(lib.kd_unet_profile(h, 5, buf, len(buf), E.current_stream()))
rows = [l.split(",") for l in buf.value.decode().strip().split("\n")[1:]]
rows = [(float(r[3]), r[1], int(r[2]), int(r[4])) for r in rows]
tot = sum(r[0] for r in rows)
print(f"{len(rows)} launches, {tot / 1e3:.3f} ms")
agg = {}
for us, label, macs, mfma in rows:
    a = agg.setdefault(label, [0, 0.0, macs, mfma])
    a[0] += 1
    a[1] += us
for label, (n, us, macs, mfma) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    tf = 2.0 * (mfma or macs) / (us / n) / 1e6 if (mfma or macs) else 0.0
    print(f"{us:9.1f} us  x{n:<3d} {us / n:8.1f} us each  {tf:7.1f} TF issued  {label}")
