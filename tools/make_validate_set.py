"""Writes the image set tools/validate_opencv.cpp runs on: the natural crops of tests/golden/natural_crops.npz (photographs /
screenshots: saturated highlights, text, JPEG blocking, dark low-contrast areas) plus frames of the synthetic stream in the shapes the
reference's YAML files produce.   python tools/make_validate_set.py out.bin [--synthetic N]
Format: "ORBXVS01", int32 count, then per image int32 rows, cols, name length, the name, rows x cols bytes."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    nsyn = int(sys.argv[sys.argv.index("--synthetic") + 1]) if "--synthetic" in sys.argv else 2
    from orb_slam3_modified_amd import synth
    imgs = []
    nat = np.load(os.path.join(ROOT, "tests", "golden", "natural_crops.npz"))
    for k in sorted(nat.files):
        if k.endswith("_img"):
            imgs.append((k[:-4], np.ascontiguousarray(nat[k])))
    for rows, cols in ((480, 640), (480, 752), (350, 600), (512, 512)):
        for i, f in enumerate(synth.make_stream(nsyn, rows, cols)):
            imgs.append((f"synth_{cols}x{rows}_{i}", f))
    with open(out, "wb") as f:
        f.write(b"ORBXVS01" + struct.pack("<i", len(imgs)))
        for name, im in imgs:
            assert im.dtype == np.uint8 and im.ndim == 2
            nb = name.encode()
            f.write(struct.pack("<iii", im.shape[0], im.shape[1], len(nb)) + nb + im.tobytes())
    print(f"{out}: {len(imgs)} images")


if __name__ == "__main__":
    main()
