# size of the conditioning-table row of each plan (kd_unet_hbm_bytes grows by T x cond_bytes at the first sampling call)
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'kidney-diffusion_amd')
import bench
import imagen_pytorch as ip
from imagen_pytorch import _engine as E
lib = E.load(); dev = torch.device('cuda:0'); T = 16
for stage, B in ((1, 1), (2, 1), (3, 1), (2, 16), (1, 8)):
    sizes = (64, 256, 1024)
    with torch.device('meta'):
        unets = tuple(ip.Unet(**bench.ULTRA_UNETS[i]) if i == stage else ip.NullUnet() for i in (1, 2, 3))
    for i, u in enumerate(unets):
        if i != stage - 1: u.lowres_cond = i > 0
    im = ip.Imagen(unets=unets, image_sizes=sizes, timesteps=(T, T, T), pred_objectives=("noise",) * 3,
                   random_crop_sizes=(None, None, 256), condition_on_text=False).to_empty(device=dev)
    for p in im.parameters(): torch.nn.init.normal_(p, std=0.02)
    S = sizes[stage - 1]
    low = torch.rand(B, 3, sizes[stage - 2], sizes[stage - 2]).to(dev) if stage > 1 else None
    cond = torch.rand(B, 3, S, S).to(dev)
    u = im.unets[stage - 1]
    h = u.engine(B, S, dev, with_text=False)
    b0 = lib.kd_unet_hbm_bytes(h)
    im.sample(batch_size=B, cond_images=cond, start_image_or_video=low, start_at_unet_number=stage, stop_at_unet_number=stage, use_tqdm=False, device=dev, seed=1)
    b1 = lib.kd_unet_hbm_bytes(h)
    print(f"stage {stage} batch {B}: table row {(b1 - b0) / T / 1e6:.2f} MB  (T=1024 -> {(b1 - b0) / T * 1024 / 1e9:.2f} GB)")
    del im, u
