#!/usr/bin/env python3
"""Generates tests/golden/natural_crops.npz: eight grayscale crops of the natural images the reference ships
(images/result.jpg — a screenshot of the running system with saturated highlights and long straight edges,
obj/pineapple/pineapple.jpg — a textured photograph, depth_estimate/assets/zoedepth-teaser.png — photographs next to smooth depth
maps) in the shapes the reference's configurations produce (640x480, 752x480, 600x350, 512x512, 1024x1024), each with the output of
the REFERENCE'S OWN src/ORBextractor.cc on it (oracle/_ref/libref_orbextractor.so, compiled where it lies; its five OpenCV algorithm
calls are the oracle's isolated primitives).  What the synthetic generator lacks — clipped whites, flat areas with JPEG blocking,
text, thin lines — is what these crops bring.  Run in the build container (needs /root/reference and PIL)."""
import hashlib
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

REF = "/root/reference"
# name: (file, left, top, width, height, nfeatures)
CROPS = {
    "result_640x480": ("images/result.jpg", 120, 200, 640, 480, 1000),
    "result_752x480": ("images/result.jpg", 1500, 300, 752, 480, 1200),
    "result_600x350": ("images/result.jpg", 900, 60, 600, 350, 1000),
    "pineapple_640x480": ("obj/pineapple/pineapple.jpg", 200, 260, 640, 480, 1000),
    "pineapple_1024x1024": ("obj/pineapple/pineapple.jpg", 0, 0, 1024, 1024, 2000),
    "teaser_752x480": ("depth_estimate/assets/zoedepth-teaser.png", 40, 150, 752, 480, 1200),
    "teaser_600x350": ("depth_estimate/assets/zoedepth-teaser.png", 2300, 200, 600, 350, 1000),
    "teaser_512x512": ("depth_estimate/assets/zoedepth-teaser.png", 3300, 120, 512, 512, 1000),
}

out = {}
for name, (rel, x, y, w, h, nf) in CROPS.items():
    im = Image.open(os.path.join(REF, rel)).convert("RGB").convert("L")     # ITU-R 601 luma, as PIL defines it
    img = np.ascontiguousarray(np.asarray(im, np.uint8)[y:y + h, x:x + w])
    assert img.shape == (h, w), (name, img.shape)
    kps, desc, mono = po.RefExtractor(nf, 1.2, 8, 20, 7).extract(img, (0, 1000))
    okps, odesc, omono = po.OracleExtractor(nf, 1.2, 8, 20, 7).extract(img, (0, 1000))
    assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc), name
    out[name + "_img"], out[name + "_nf"] = img, nf
    out[name + "_kps"], out[name + "_desc"], out[name + "_mono"] = kps, desc, mono
    sat = float((img >= 250).mean())
    print(f"{name}: {len(kps)} keypoints, {100 * sat:.1f} % saturated, mean {img.mean():.0f}, sha {hashlib.sha256(img.tobytes()).hexdigest()[:12]}")
np.savez_compressed(os.path.join(HERE, "natural_crops.npz"), **out)
print("natural_crops.npz:", os.path.getsize(os.path.join(HERE, "natural_crops.npz")) // 1024, "KiB")
