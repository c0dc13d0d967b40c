# Forward-pass time of the reference's other UNet configurations at full size (random weights):
#   unet1u  train_ultra_res.py:29-36  base 64x64, 3 cond channels         (BASELINE configs[0]/[3] stage 1)
#   unet1c  train.py:30-41            base 64x64 seg-cond (text dim 3, 4 cond channels)  (configs[1])
#   unet2   train_ultra_res.py:39-48  SR 64->256                           (configs[2], the headline)
#   unet3   train_ultra_res.py:51-60  SR 256->1024                         (configs[3] stage 3)
# usage: python scratch/fwd_configs.py <name> <batch> [size] [wino43_min_cin] [gemm_bf16x3] [x3_linear]
import sys, time, ctypes as C, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / 'kidney-diffusion_amd'))
import imagen_pytorch as ip
from imagen_pytorch import _engine as E

F, T = False, True
CFG = {
    'unet1u': (dict(dim=256, dim_mults=(1, 2, 4, 8), num_resnet_blocks=3, layer_attns=(F, T, T, T),
                    layer_cross_attns=(F, T, T, T), cond_images_channels=3, lowres_cond=False, cond_on_text=False,
                    text_embed_dim=None), 64),
    'unet1c': (dict(dim=256, dim_mults=(1, 2, 3, 4), cond_dim=512, text_embed_dim=3, num_resnet_blocks=3,
                    layer_attns=(F, T, T, T), layer_cross_attns=(F, T, T, T), cond_images_channels=4,
                    lowres_cond=False, cond_on_text=True), 64),
    'unet2': (dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=2, memory_efficient=True,
                   layer_attns=(F, F, F, T), layer_cross_attns=(F, F, T, T), init_conv_to_final_conv_residual=True,
                   cond_images_channels=3, lowres_cond=True, cond_on_text=False, text_embed_dim=None), 256),
    'unet3': (dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 6, 8), memory_efficient=True,
                   layer_attns=False, layer_cross_attns=(F, F, F, T), init_conv_to_final_conv_residual=True,
                   cond_images_channels=3, lowres_cond=True, cond_on_text=False, text_embed_dim=None), 1024),
}
name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kw, S = CFG[name]
if len(sys.argv) > 3:
    S = int(sys.argv[3])
dev = torch.device('cuda:0'); lib = E.load()
with torch.device('meta'):
    u = ip.Unet(**kw)
u = u.to_empty(device=dev)
for p in u.parameters():
    torch.nn.init.normal_(p, std=0.02)
if len(sys.argv) > 4:
    u.wino43_min_cin = int(sys.argv[4])
if len(sys.argv) > 5:
    u.gemm_bf16x3 = int(sys.argv[5])
if len(sys.argv) > 6:
    u.x3_linear = int(sys.argv[6])
with_text = bool(kw.get('cond_on_text'))
t0 = time.time(); h = u.engine(B, S, dev, with_text=with_text)
print(name, 'B', B, 'S', S, 'plan %.1f s' % (time.time() - t0), 'hbm GB %.1f' % (lib.kd_unet_hbm_bytes(h) / 1e9),
      'GMAC/sample %.1f' % (lib.kd_unet_macs(h) / 1e9 / B), 'issued GMAC/sample %.1f' % (lib.kd_unet_mfma_macs(h) / 1e9 / B),
      'launches', lib.kd_unet_num_launches(h))
cc = kw.get('cond_images_channels', 0)
x = torch.randn(B, 3, S, S, device=dev)
lr = torch.randn(B, 3, S, S, device=dev) if kw['lowres_cond'] else None
cond = torch.rand(B, cc, S, S, device=dev) if cc else None
t = torch.full((B,), 0.3, device=dev)
tl = torch.full((B,), -1.0, device=dev) if kw['lowres_cond'] else None
tok = hid = None
if with_text:
    te = torch.tensor([0.0, 0.5, 0.2], device=dev).reshape(1, 1, 3).repeat(B, 1, 1)   # sample_cond.py:37
    tok, hid = u.text_cond(h, te, None, False, dev)
out = torch.empty_like(x)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    E.check(lib.kd_unet_forward(h, E.ptr(x), E.ptr(lr), E.ptr(cond), E.ptr(t), E.ptr(tl), E.ptr(tok), E.ptr(hid),
                                E.ptr(out), E.current_stream()))
    torch.cuda.synchronize(); dt = time.time() - t0
    print('fwd %.2f ms' % (dt * 1e3), '%.1f TF/s algorithmic' % (2 * lib.kd_unet_macs(h) / dt / 1e12),
          '%.1f issued' % (2 * lib.kd_unet_mfma_macs(h) / dt / 1e12), float(out.abs().mean()))
buf = C.create_string_buffer(1 << 21)
E.check(lib.kd_unet_profile(h, 2, buf, len(buf), E.current_stream()))
rows = [l.split(',') for l in buf.value.decode().strip().split('\n')[1:]]
tot = sum(float(r[3]) for r in rows); agg = {}
for r in rows:
    a = agg.setdefault(r[1], [0, 0.0, 0]); a[0] += 1; a[1] += float(r[3]); a[2] += int(r[2])
print('profile total us', tot)
for k, (n, us, macs) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{us / tot * 100:6.2f}% n={n:3d} us={us:9.1f} TF/s={2 * macs / us / 1e6 if us else 0:7.1f}  {k}")
