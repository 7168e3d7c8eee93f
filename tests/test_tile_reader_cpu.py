"""SURVEY.md §8(f) N4, host half: the native GeoTIFF tile reader (kurosiwo_amd/csrc/tile_reader.hip behind ksmi_tiff_* and
ksmi_tile_batch_read) against tiles written by libtiff (tests/golden/tiff/, made by oracle/gen_tiff_fixtures.py through Pillow), the
codec family behind the reference's cv2.imread(path, IMREAD_ANYDEPTH) (dataset/Dataset.py:664-728).  Byte / integer / bit-pattern
work: the bar is bit-exact, NaN no-data included.  Host-only: runs without a GPU."""
import glob
import itertools
import os

import numpy as np
import pytest

from kurosiwo_amd import _lib, geotiff as G

TIFF_DIR = os.path.join(os.path.dirname(__file__), "golden", "tiff")


def _same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


def test_every_libtiff_fixture_decodes_bit_exactly():
    want = np.load(os.path.join(TIFF_DIR, "expected.npz"))
    files = sorted(glob.glob(os.path.join(TIFF_DIR, "*.tif")))
    assert len(files) == 30
    seen = set()
    for f in files:
        key = os.path.basename(f).split("_")[0]
        got, meta = G.read(f)
        assert np.array_equal(got, want[key], equal_nan=True) and got.dtype == want[key].dtype, f
        if got.dtype == np.float32:
            assert _same(got, want[key]), f                               # NaN payloads too
        as_f32, _ = G.read(f, dtype=np.float32)                           # what cv2 hands the Dataset
        assert np.array_equal(as_f32, want[key].astype(np.float32), equal_nan=True), f
        seen.add((meta["compression"], meta["predictor"], str(meta["dtype"])))
    assert {c for c, _, _ in seen} == {1, 5, 8, 32773} and {p for _, p, _ in seen} == {1, 2, 3}


def test_batch_reader_fills_one_staging_buffer():
    import torch
    want = np.load(os.path.join(TIFF_DIR, "expected.npz"))
    files = sorted(glob.glob(os.path.join(TIFF_DIR, "sar_*.tif"))) + sorted(glob.glob(os.path.join(TIFF_DIR, "mask_*.tif")))
    H, W = want["sar"].shape
    buf = torch.empty((len(files), H, W), dtype=torch.float32)
    for threads in (1, 3, 16):
        buf.fill_(-7.0)
        G.read_batch(files, H, W, out=buf, threads=threads)
        for i, f in enumerate(files):
            key = os.path.basename(f).split("_")[0]
            assert np.array_equal(buf[i].numpy(), want[key].astype(np.float32), equal_nan=True), (f, threads)
    assert G.read_batch([], H, W).shape == (0, H, W)
    with pytest.raises(_lib.KsmiError, match="expected 1 x 224 x 224"):
        G.read_batch(files[:2], 224, 224)
    with pytest.raises(_lib.KsmiError, match="cannot open"):
        G.read_batch(files[:1] + ["/nonexistent/MS1_IVV.tif"], H, W)
    with pytest.raises(ValueError):
        G.read_batch(files, H, W, out=torch.empty((1, H, W)))


def test_layouts_libtiff_did_not_write():
    """tiles, BigTIFF, big-endian files, planar multi-band, geo tags: written by the numpy writer of kurosiwo_amd/geotiff.py, read by
    the native reader and (where Pillow can) by libtiff as the independent side"""
    want = np.load(os.path.join(TIFF_DIR, "expected.npz"))
    try:
        from PIL import Image
    except ImportError:
        Image = None
    import tempfile
    checked_by_libtiff = 0
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "t.tif")
        for comp, tile, be, big in itertools.product((None, "lzw", "deflate", "packbits"), (None, (16, 32)), (False, True), (False, True)):
            for key in ("sar", "mask", "dem16", "u16"):
                a = want["dem32"].astype(np.int16) if key == "dem16" else want[key]      # int16: the SRTM-style DEM of the archive
                pred = 1 if comp in (None, "packbits") else (3 if a.dtype.kind == "f" else 2)
                G.write(p, a, compression=comp, predictor=pred, tile=tile, big_endian=be, bigtiff=big, rows_per_strip=None if tile else 7,
                        nodata=float("nan") if a.dtype.kind == "f" else 0, pixel_scale=(10, 10), origin=(500000.0, 4.2e6))
                got, meta = G.read(p)
                assert _same(got, a), (key, comp, tile, be, big)
                assert meta["pixel_scale"] == (10.0, 10.0) and meta["origin"] == (500000.0, 4.2e6)
                assert (meta["nodata"] != meta["nodata"]) if a.dtype.kind == "f" else meta["nodata"] == 0.0
                assert (meta["tiled"], meta["big_endian"], meta["bigtiff"]) == (int(bool(tile)), int(be), int(big))
                # Pillow: no big-endian BigTIFF; its big-endian path is reliable for the unsigned types only (it swaps libtiff's
                # already-native floats once more and leaves signed 16-bit samples unswapped)
                if Image is not None and (not be or (not big and key in ("mask", "u16"))):
                    back = np.array(Image.open(p))
                    assert _same(back.astype(a.dtype), a), ("libtiff", key, comp, tile, be, big)
                    checked_by_libtiff += 1
        bands = np.stack([want["sar"], want["sar"] * 2, want["demf"]])
        for planar, (comp, pred) in itertools.product((False, True), ((None, 1), ("lzw", 3), ("deflate", 3), ("packbits", 1))):
            G.write(p, bands, compression=comp, predictor=pred, planar=planar, tile=(16, 16))
            got, _ = G.read(p)
            assert _same(got, bands), (planar, comp)
        f64 = want["demf"].astype(np.float64)
        G.write(p, f64, compression="deflate", predictor=3)
        assert _same(G.read(p)[0], f64)
    assert Image is None or checked_by_libtiff == 4 * 2 * (2 * 4 + 1 * 2)


def test_damaged_files_are_errors_not_crashes(tmp_path):
    src = os.path.join(TIFF_DIR, "sar_tiff_lzw_p3.tif")
    blob = open(src, "rb").read()
    p = str(tmp_path / "x.tif")
    for cut in (0, 3, 7, 100, len(blob) // 2, len(blob) - 5):
        open(p, "wb").write(blob[:cut])
        with pytest.raises(_lib.KsmiError):
            G.read(p)
    open(p, "wb").write(b"II*\0" + b"\xff" * 64)
    with pytest.raises(_lib.KsmiError):
        G.read(p)
    rng = np.random.default_rng(1)
    for _ in range(200):                                                  # random byte flips: any outcome but a crash
        b = bytearray(blob)
        for k in rng.integers(0, len(b), 4):
            b[k] = int(rng.integers(0, 256))
        open(p, "wb").write(bytes(b))
        try:
            G.read(p)
        except _lib.KsmiError:
            pass
