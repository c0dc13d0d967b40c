"""Generates tests/golden/geometry.json: crop windows the reference's host geometry must produce, from a
restatement of the THIRD-PARTY formulas the reference calls (none of this repo's product code is imported):

  * torchvision.transforms.functional.center_crop (torchvision 0.15, the version of the reference's
    requirements.txt:106 `torchvision==0.15.1+cu118`):
        crop_top  = int(round((image_height - crop_height) / 2.0))
        crop_left = int(round((image_width  - crop_width ) / 2.0))
    (Python round = half to even), after zero padding an image smaller than the crop by
        [(cw - w) // 2, (ch - h) // 2, (cw - w + 1) // 2, (ch - h + 1) // 2]   (left, top, right, bottom);
  * the reference's own integer geometry: get_patch_width (sample_ultra_res.py:273-280),
    patch_dist / num_patches_width (:305-314), shift (:360-370), canvas (:430-431).

Call sites pinned: transforms.CenterCrop(PATCH_SIZE) :391, CenterCrop(patch_width) :393 (version v2) and :419
(--ignore_unet_1).  Run:  python tests/golden/make_geometry_golden.py
"""
import json
import math
from pathlib import Path

PATCH_SIZE = 1024
KIDNEY = [40000, 6500, 1024]   # ultra_res_patient_dataset.py:18
AIRS = [10000, 3328, 1024]     # ultra_res_airs.py:23


def tv_center_crop_window(h, w, crop):
    """(pad_left, pad_top, pad_right, pad_bottom, top, left) of torchvision's center_crop to crop x crop."""
    pl = pt = pr = pb = 0
    if crop > w or crop > h:
        pl = (crop - w) // 2 if crop > w else 0
        pt = (crop - h) // 2 if crop > h else 0
        pr = (crop - w + 1) // 2 if crop > w else 0
        pb = (crop - h + 1) // 2 if crop > h else 0
        h, w = h + pt + pb, w + pl + pr
        if crop == w and crop == h:
            return pl, pt, pr, pb, 0, 0
    top = int(round((h - crop) / 2.0))
    left = int(round((w - crop) / 2.0))
    return pl, pt, pr, pb, top, left


def level(sizes, mag, zoomed_width, overlap, airs=False):
    pw = int(sizes[mag] * PATCH_SIZE / sizes[mag - 1])
    dist = int(pw * (1 - overlap))
    n = 1 + math.ceil((zoomed_width - pw) / dist)
    if airs:
        n = max(1, n - 1)
    out_dist = int(PATCH_SIZE * (1 - overlap))
    crop1 = tv_center_crop_window(zoomed_width, zoomed_width, PATCH_SIZE)    # :391
    crop2 = tv_center_crop_window(PATCH_SIZE, PATCH_SIZE, pw)                # :393 / :419
    patches = []
    for (i, j) in [(0, 0), (1, 2), (n - 1, n - 1), (n // 2, n // 3)]:
        cy, cx = i * dist + pw // 2, j * dist + pw // 2
        patches.append({"pos": [i, j], "shift": [zoomed_width // 2 - cy, zoomed_width // 2 - cx]})
    return {"patch_width": pw, "patch_dist": dist, "num_patches_width": n, "out_patch_dist": out_dist,
            "canvas_width": PATCH_SIZE + (n - 1) * out_dist, "zoomed_width": zoomed_width,
            "crop_to_patch_size": list(crop1), "crop_to_patch_width": list(crop2), "patches": patches}


def main():
    g = {
        "center_crop_offsets": [{"size": s, "crop": c, "window": list(tv_center_crop_window(s, s, c))}
                                for s, c in [(1024, 166), (1024, 161), (1024, 340), (1024, 1024), (6400, 1024),
                                             (2560, 1024), (1000, 1024), (1023, 1024), (7, 4), (8, 3), (9, 4), (5, 2)]],
        "kidney_mag1": level(KIDNEY, 1, 1024, 0.25),
        "kidney_mag2": level(KIDNEY, 2, 6400, 0.25),
        "airs_mag1": level(AIRS, 1, 1024, 0.25, airs=True),
    }
    assert g["kidney_mag2"]["crop_to_patch_width"][4:] == [432, 432]   # 863 / 2 = 431.5 -> 432 (half to even)
    Path(__file__).with_name("geometry.json").write_text(json.dumps(g, indent=1) + "\n")


if __name__ == "__main__":
    main()
