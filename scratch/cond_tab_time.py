# Step time of a batch-1 stage with the conditioning table on / off:  python scratch/cond_tab_time.py <stage> [T]
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'kidney-diffusion_amd')
import bench
import imagen_pytorch as ip
dev = torch.device('cuda:0')
stage = int(sys.argv[1]); T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
sizes = (64, 256, 1024)
with torch.device('meta'):
    unets = tuple(ip.Unet(**bench.ULTRA_UNETS[i]) if i == stage else ip.NullUnet() for i in (1, 2, 3))
for i, u in enumerate(unets):
    if i != stage - 1:
        u.lowres_cond = i > 0
im = ip.Imagen(unets=unets, image_sizes=sizes, timesteps=(T, T, T), pred_objectives=("noise",) * 3,
               random_crop_sizes=(None, None, 256), condition_on_text=False).to_empty(device=dev)
for p in im.parameters():
    torch.nn.init.normal_(p, std=0.02)
S = sizes[stage - 1]
g = torch.Generator().manual_seed(0)
low = torch.rand(1, 3, sizes[stage - 2], sizes[stage - 2], generator=g).to(dev) if stage > 1 else None
cond = torch.rand(1, 3, 1024, 1024, generator=g).to(dev)
for table in (0, -1, 0, -1):
    im.cond_table = table
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        out = im.sample(batch_size=1, cond_images=cond, start_image_or_video=low, start_at_unet_number=stage,
                        stop_at_unet_number=stage, use_tqdm=False, device=dev, seed=it)
        torch.cuda.synchronize(); dt = time.time() - t0
    print(f"stage {stage} T={T} cond_table={table}: {dt * 1e3 / T:.3f} ms/step (third call)", float(out.mean()))
