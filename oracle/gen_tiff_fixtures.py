"""TEST INFRASTRUCTURE (oracle side).  Golden GeoTIFF tiles for the tile reader, written by libtiff 4.7.1 through
Pillow (the codec family cv2.imread uses in the reference, dataset/Dataset.py:664-728): small tiles with the sample types of the
archive (float32 backscatter with NaN no-data, uint8 masks, int32 / float32 DEM, uint16) in every compression / predictor libtiff offers
for them.  `expected.npz` holds the arrays that went in.

    python oracle/gen_tiff_fixtures.py          # rewrites tests/golden/tiff/
"""
import os

import numpy as np
from PIL import Image, features

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "tiff")
H, W = 56, 40


def arrays():
    rng = np.random.default_rng(20260929)
    sar = (rng.gamma(4.0, 0.25, (H, W)) * 0.09).astype(np.float32)
    sar[3, 4] = np.nan
    sar[10:20, 5:30] = 0.02                                   # a calm water body: long runs for the codecs
    sar[40:, :8] = np.nan                                     # swath edge
    mask = np.zeros((H, W), np.uint8)
    mask[10:20, 5:30] = 1
    mask[22:30, 5:12] = 2
    mask[rng.random((H, W)) < 0.02] = 3
    dem32 = (np.add.outer(np.arange(H), np.arange(W)) * 7 - 120).astype(np.int32)     # Pillow writes signed integers as 32 bit
    demf = (200 + 150 * np.sin(np.linspace(0, 3, H))[:, None] * np.cos(np.linspace(0, 6, W))[None]).astype(np.float32)
    u16 = rng.integers(0, 65535, (H, W)).astype(np.uint16)
    return {"sar": sar, "mask": mask, "dem32": dem32, "demf": demf, "u16": u16}


def main():
    assert features.check("libtiff"), "Pillow without libtiff"
    os.makedirs(OUT, exist_ok=True)
    arrs = arrays()
    names = []
    for key, a in arrs.items():
        flt = a.dtype.kind == "f"
        for comp in (None, "tiff_lzw", "tiff_adobe_deflate", "packbits"):
            for pred in ((1, 3) if flt else (1, 2)):
                if pred != 1 and comp in (None, "packbits"):
                    continue
                name = f"{key}_{comp or 'none'}_p{pred}.tif"
                im = Image.fromarray(a, mode="F") if flt else Image.fromarray(a)
                im.save(os.path.join(OUT, name), compression=comp, **({"tiffinfo": {317: pred}} if pred != 1 else {}))
                back = np.array(Image.open(os.path.join(OUT, name)))
                assert np.array_equal(back.astype(a.dtype), a, equal_nan=True), name
                names.append(name)
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **arrs)
    print(len(names), "fixtures, libtiff", features.version("libtiff"), "->", OUT)


if __name__ == "__main__":
    main()
