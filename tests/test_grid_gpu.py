"""Host-side pieces of the ultra-res driver that run on the DEVICE in production (the zoomed image of the level
above lives in HBM, so conditioning images, fallback crops and the mag-2 tissue mask never touch the host):
same results as on the CPU, bit for bit where only data movement is involved.
  cond images   sample_ultra_res.py:356-400      tissue mask / patch filter   :317-352
  inpaint tensors with fallback crops :128-174   canvas stitch                :430-446"""
import pytest
import torch

from ultra_res import grid as G
from ultra_res import pipeline as P

pytestmark = pytest.mark.gpu


def test_cond_images_tissue_mask_and_stitch_on_device_equal_cpu(device):
    g = torch.Generator().manual_seed(5)
    zoomed = torch.rand(1, 3, 1024, 1024, generator=g)
    geom = G.grid_geometry(1024, 1, 0.25)
    pos = [(0, 0), (3, 5), (7, 7), (2, 6)]
    for v2 in (False, True):
        a = G.cond_images_for_grid(zoomed, geom, pos, centre_crop_channels=v2)
        b = G.cond_images_for_grid(zoomed.to(device), geom, pos, centre_crop_channels=v2)
        assert b.is_cuda and torch.equal(a, b.cpu())
    # mag-2 numbers on a stand-in of the mag-1 canvas: HSV threshold, 5x5 erode, 51x51 dilate, footprint filter
    img = torch.rand(1, 3, 1280, 1280, generator=g) * 0.06 + 0.92
    img[0, :, 200:500, 300:900] = torch.tensor([0.75, 0.35, 0.8])[:, None, None]
    img[0, :, 900:903, 100:103] = torch.tensor([0.7, 0.3, 0.8])[:, None, None]
    m_cpu = G.tissue_mask(img)
    m_gpu = G.tissue_mask(img.to(device))
    assert m_gpu.is_cuda and torch.equal(m_cpu, m_gpu.cpu()) and 0 < int(m_cpu.sum()) < m_cpu.numel()
    geom2 = G.grid_geometry(1280, 2, 0.25)
    assert G.tissue_patch_positions(m_cpu, geom2) == G.tissue_patch_positions(m_gpu, geom2)
    # inpaint assembly with a filtered-out neighbour (bilinear fallback crop of the conditioning image) and stitch
    S = 64
    cond = torch.rand(3, 1024, 1024, generator=g)
    done = {(0, 1): torch.rand(3, S, S, generator=g)}
    kw = dict(size=S, overlap=0.25, orientation=-1, num_patches_width=3, patch_width=166)
    ip_c, im_c = G.assemble_inpaint((1, 1), [(0, 1), (1, 1)], done, cond_image=cond, **kw)
    ip_g, im_g = G.assemble_inpaint((1, 1), [(0, 1), (1, 1)], {k: v.to(device) for k, v in done.items()},
                                    cond_image=cond.to(device), **kw)
    assert torch.equal(im_c, im_g.cpu()) and torch.allclose(ip_c, ip_g.cpu(), atol=1e-6)   # bilinear: fp32 rounding
    assert torch.equal(ip_c[:, :16, 16:], ip_g.cpu()[:, :16, 16:])                          # copied strips: exact
    patches = [torch.rand(3, 1024, 1024, generator=g) for _ in range(2)]
    sub = G.GridGeometry(geom.patch_width, geom.patch_dist, 2, geom.out_patch_dist, 1024 + geom.out_patch_dist)
    c_cpu = G.stitch_canvas(patches, [(0, 0), (0, 1)], sub, background=zoomed)
    c_gpu = G.stitch_canvas([p.to(device) for p in patches], [(0, 0), (0, 1)], sub, background=zoomed.to(device))
    assert torch.equal(c_cpu[0, :, :1024, :1792], c_gpu.cpu()[0, :, :1024, :1792])          # pasted patches: exact
    assert torch.allclose(c_cpu, c_gpu.cpu(), atol=1e-6)                                     # bilinear background


def test_one_magnification_level_on_the_engine(device):
    """generate_high_res_image (sample_ultra_res.py:414-448) over the HIP engine with every tensor in HBM: a 2x2
    corner of the mag-1 grid at reduced model dims, pipelined stages, v2 conditioning (6 channels)."""
    import helpers as H
    import imagen_pytorch as ip
    from oracle import sampler_ref as RS  # noqa: F401  (weights only)
    from ultra_res import distributed as D

    kw1 = dict(H.UNET_KW["ultra1"], cond_images_channels=6)
    kw2 = dict(H.UNET_KW["ultra2"], cond_images_channels=6)
    kw3 = dict(H.UNET_KW["ultra3"], cond_images_channels=6)
    torch.manual_seed(3)
    pim = ip.Imagen([ip.Unet(**kw1), ip.Unet(**kw2), ip.Unet(**kw3)], image_sizes=(64, 256, 1024), timesteps=(2, 2, 2),
                    pred_objectives=("noise", "noise", "noise"), condition_on_text=False)
    with torch.no_grad():
        for u in pim.unets:
            u.final_conv.weight.normal_(0, 0.02)
    pim = pim.to(device)
    zoomed = torch.rand(1, 3, 1024, 1024, device=device)
    fn = D.imagen_sample_fn(lambda stage: pim, 1, device, seed=11)
    pos = [(0, 0), (0, 1), (1, 0), (1, 1)]
    canvas = P.generate_high_res_image(fn, zoomed, 1, overlap=0.25, version="v2", device=device, patch_pos=pos)
    assert canvas.is_cuda and canvas.shape == (1, 3, 6400, 6400) and torch.isfinite(canvas).all()
    # the generated region is sampler output in [0,1], patch (1,1)'s overlap strips were pasted from its neighbours,
    # the rest of the canvas is the bilinearly enlarged background
    assert canvas[0, :, :1792, :1792].min() >= 0 and canvas[0, :, :1792, :1792].max() <= 1
    again = P.generate_high_res_image(fn, zoomed, 1, overlap=0.25, version="v2", device=device, patch_pos=pos)
    assert torch.equal(canvas, again)   # seeded per patch: reproducible
    bg = torch.nn.functional.interpolate(zoomed, size=(6400, 6400), mode="bilinear", align_corners=False)
    assert torch.equal(canvas[0, :, 2000:, 2000:], bg[0, :, 2000:, 2000:])
