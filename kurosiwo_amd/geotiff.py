"""GeoTIFF tiles of the Kuro Siwo archive (host half).

Reading is native (kurosiwo_amd/csrc/tile_reader.hip behind ksmi_tiff_* / ksmi_tile_batch_read): it replaces the reference's
cv2.imread(path, cv2.IMREAD_ANYDEPTH) (dataset/Dataset.py:664-728) and rioxarray.open_rasterio for the DEM (:730-737).
`write` is a small pure-numpy TIFF writer: it exists to build synthetic archives and test fixtures (tools/make_synthetic_archive.py),
the product path never writes tiles."""
import ctypes as C
import struct
import zlib

import numpy as np

from . import _lib

_NATIVE = {(1, 8): np.uint8, (1, 16): np.uint16, (1, 32): np.uint32, (1, 64): np.uint64, (2, 8): np.int8, (2, 16): np.int16,
           (2, 32): np.int32, (2, 64): np.int64, (3, 32): np.float32, (3, 64): np.float64}


def _info_dict(ti):
    d = {k: int(getattr(ti, k)) for k in ("width", "height", "bands", "bits", "sample_format", "compression", "predictor", "tiled",
                                          "big_endian", "bigtiff")}
    d["dtype"] = np.dtype(_NATIVE[(d["sample_format"], d["bits"])])
    d["pixel_scale"] = (ti.pixel_scale[0], ti.pixel_scale[1]) if ti.has_geo & 1 else None
    d["origin"] = (ti.origin[0] - ti.tie_pixel[0] * ti.pixel_scale[0], ti.origin[1] + ti.tie_pixel[1] * ti.pixel_scale[1]) if ti.has_geo == 3 else None
    d["nodata"] = float(ti.nodata) if ti.has_nodata else None
    return d


def info(path):
    ti = _lib.TiffInfo()
    _lib.check(_lib.load().ksmi_tiff_info_read(str(path).encode(), C.byref(ti)), "tiff_info")
    return _info_dict(ti)


def read(path, dtype=None, squeeze=True):
    """The first image of `path` as [bands, H, W] ([H, W] for one band when `squeeze`) in the file's own sample type, or as float32
    with dtype=np.float32 (what cv2 returns for the archive's float tiles).  Returns (array, info)."""
    lib = _lib.load()
    ti = _lib.TiffInfo()
    _lib.check(lib.ksmi_tiff_info_read(str(path).encode(), C.byref(ti)), "tiff_info")
    meta = _info_dict(ti)
    shape = (meta["bands"], meta["height"], meta["width"])
    if dtype is not None and np.dtype(dtype) == np.float32:
        out = np.empty(shape, np.float32)
        _lib.check(lib.ksmi_tiff_read_f32(str(path).encode(), out.ctypes.data, out.size, None), "tiff_read")
    else:
        out = np.empty(shape, meta["dtype"])
        _lib.check(lib.ksmi_tiff_read_native(str(path).encode(), out.ctypes.data, out.size, None), "tiff_read")
        if dtype is not None:
            out = out.astype(dtype)
    return (out[0] if squeeze and shape[0] == 1 else out), meta


def read_batch(paths, H, W, out=None, threads=8, bands=1):
    """n H x W tiles of `bands` bands -> float32 [n, H, W] ([n, bands, H, W] for bands > 1), decoded by `threads` native threads into
    `out` (a torch tensor or numpy array, e.g. one pinned staging buffer) or a new numpy array."""
    lib = _lib.load()
    n = len(paths)
    if out is None:
        out = np.empty((n, H, W) if bands == 1 else (n, bands, H, W), np.float32)
    if hasattr(out, "data_ptr"):
        import torch
        if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() < n * bands * H * W or out.device.type != "cpu":
            raise ValueError("read_batch: `out` must be a contiguous float32 CPU tensor of at least n*H*W elements")
        ptr = out.data_ptr()
    else:
        if out.dtype != np.float32 or not out.flags["C_CONTIGUOUS"] or out.size < n * bands * H * W:
            raise ValueError("read_batch: `out` must be a C-contiguous float32 array of at least n*H*W elements")
        ptr = out.ctypes.data
    arr = (C.c_char_p * max(n, 1))(*[str(p).encode() for p in paths])
    _lib.check(lib.ksmi_tile_batch_read_bands(arr, n, ptr, int(bands), H, W, int(threads)), "tile_batch_read")
    return out


def fill_nodata(tiles, threads=1):
    """In place: every NaN of each [H, W] tile of `tiles` ([..., H, W] contiguous float32, numpy or CPU torch) takes the value of the
    nearest valid pixel (rioxarray interpolate_na(method="nearest"), dataset/Dataset.py:733-735)."""
    H, W = tiles.shape[-2:]
    if hasattr(tiles, "data_ptr"):
        import torch
        if tiles.dtype != torch.float32 or not tiles.is_contiguous() or tiles.device.type != "cpu":
            raise ValueError("fill_nodata: contiguous float32 CPU tensor")
        ptr, n = tiles.data_ptr(), tiles.numel() // (H * W)
    else:
        if tiles.dtype != np.float32 or not tiles.flags["C_CONTIGUOUS"]:
            raise ValueError("fill_nodata: C-contiguous float32 array")
        ptr, n = tiles.ctypes.data, tiles.size // (H * W)
    _lib.check(_lib.load().ksmi_tiles_fill_nodata(ptr, n, H, W, int(threads)), "tiles_fill_nodata")
    return tiles


# ------------------------------------------------------------------------------------------------ writer (fixtures, synthetic archives)
def _lzw_encode(data):
    """TIFF 6.0 §13 LZW: MSB-first, 9..12 bits, early change"""
    out = bytearray()
    acc, nacc = 0, 0

    def put(code, bits):
        nonlocal acc, nacc
        acc = (acc << bits) | code
        nacc += bits
        while nacc >= 8:
            out.append((acc >> (nacc - 8)) & 0xFF)
            nacc -= 8
        acc &= (1 << nacc) - 1
    table = {bytes([i]): i for i in range(256)}
    nxt, bits = 258, 9
    put(256, bits)
    w = b""
    for byte in data:
        wc = w + bytes([byte])
        if wc in table:
            w = wc
            continue
        put(table[w], bits)
        table[wc] = nxt
        nxt += 1
        if nxt == (1 << bits) and bits < 12:
            bits += 1
        elif nxt == 4094:
            put(256, bits)
            table = {bytes([i]): i for i in range(256)}
            nxt, bits = 258, 9
        w = bytes([byte])
    if w:
        put(table[w], bits)
        nxt += 1
        if nxt == (1 << bits) and bits < 12:
            bits += 1
    put(257, bits)
    if nacc:
        out.append((acc << (8 - nacc)) & 0xFF)
    return bytes(out)


def _packbits_encode(data):
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        run = 1
        while i + run < n and run < 128 and data[i + run] == data[i]:
            run += 1
        if run >= 2:
            out += bytes([257 - run, data[i]])
            i += run
            continue
        j = i + 1
        while j < n and j - i < 128 and not (j + 1 < n and data[j] == data[j + 1]):
            j += 1
        out.append(j - i - 1)
        out += data[i:j]
        i = j
    return bytes(out)


_COMPRESSION = {None: 1, "none": 1, "lzw": 5, "deflate": 8, "adobe_deflate": 8, "old_deflate": 32946, "packbits": 32773}


def write(path, array, compression=None, predictor=1, tile=None, rows_per_strip=None, big_endian=False, bigtiff=False, planar=False,
          nodata=None, pixel_scale=None, origin=None):
    """array: [H, W] or [bands, H, W].  tile=(th, tw) writes tiles (multiples of 16), else strips of `rows_per_strip` rows."""
    a = np.asarray(array)
    if a.ndim == 2:
        a = a[None]
    Bn, H, W = a.shape
    kind = {"u": 1, "i": 2, "f": 3}[a.dtype.kind]
    bps = a.dtype.itemsize
    comp = _COMPRESSION[compression]
    if predictor != 1 and comp in (1, 32773):
        raise ValueError("a predictor needs LZW or Deflate (libtiff ignores the tag for the other codecs)")
    if (predictor == 3) != (a.dtype.kind == "f") and predictor != 1:
        raise ValueError("predictor 2 is for integer samples, predictor 3 for floating point")
    spp = 1 if planar else Bn
    planes = [a[b:b + 1] for b in range(Bn)] if planar else [a]

    def encode_rows(block):
        """block [spp, rows, cols] -> bytes of a chunk"""
        px = np.ascontiguousarray(np.moveaxis(block, 0, -1))                       # [rows, cols, spp]
        rows, cols = px.shape[:2]
        if predictor == 3:
            be = px.astype(px.dtype.newbyteorder(">")).view(np.uint8).reshape(rows, cols * spp, bps)
            planes_ = np.ascontiguousarray(np.moveaxis(be, 2, 1)).reshape(rows, bps * cols * spp)      # byte planes, MSB first
            d = planes_.copy()
            d[:, spp:] = planes_[:, spp:] - planes_[:, :-spp]
            raw = d.tobytes()
        else:
            v = px.reshape(rows, cols * spp)
            if predictor == 2:
                u = v.view(np.dtype(f"u{bps}"))
                d = u.copy()
                d[:, spp:] = u[:, spp:] - u[:, :-spp]
                v = d
            raw = v.astype(v.dtype.newbyteorder(">" if big_endian else "<")).tobytes()
        if comp == 1:
            return raw
        if comp == 5:
            return _lzw_encode(raw)
        if comp == 32773:
            rb = len(raw) // rows                                                   # PackBits rows are packed separately (TIFF 6.0 §9)
            return b"".join(_packbits_encode(raw[i * rb:(i + 1) * rb]) for i in range(rows))
        return zlib.compress(raw, 6)
    chunks = []
    for pl in planes:
        if tile:
            th, tw = tile
            for y0 in range(0, H, th):
                for x0 in range(0, W, tw):
                    blk = np.zeros((pl.shape[0], th, tw), a.dtype)
                    sub = pl[:, y0:y0 + th, x0:x0 + tw]
                    blk[:, :sub.shape[1], :sub.shape[2]] = sub
                    chunks.append(encode_rows(blk))
        else:
            rps = rows_per_strip or H
            for y0 in range(0, H, rps):
                chunks.append(encode_rows(pl[:, y0:y0 + rps]))
    E = ">" if big_endian else "<"
    osz = 8 if bigtiff else 4
    head = (b"MM" if big_endian else b"II") + (struct.pack(E + "HHHQ", 43, 8, 0, 16) if bigtiff else struct.pack(E + "HI", 42, 8))
    data_pos = len(head)
    offsets, pos = [], data_pos
    for c in chunks:
        offsets.append(pos)
        pos += len(c) + (len(c) & 1)
    LONG = 16 if bigtiff else 4
    tags = [(256, 4, [W]), (257, 4, [H]), (258, 3, [bps * 8] * Bn), (259, 3, [comp]), (262, 3, [1]), (277, 3, [Bn]),
            (284, 3, [2 if planar else 1]), (339, 3, [kind] * Bn)]
    if tile:
        tags += [(322, 4, [tile[1]]), (323, 4, [tile[0]]), (324, LONG, offsets), (325, LONG, [len(c) for c in chunks])]
    else:
        tags += [(273, LONG, offsets), (278, 4, [rows_per_strip or H]), (279, LONG, [len(c) for c in chunks])]
    if predictor != 1:
        tags.append((317, 3, [predictor]))
    if pixel_scale is not None:
        tags.append((33550, 12, [float(pixel_scale[0]), float(pixel_scale[1]), 0.0]))
    if origin is not None:
        tags.append((33922, 12, [0.0, 0.0, 0.0, float(origin[0]), float(origin[1]), 0.0]))
    if nodata is not None:
        tags.append((42113, 2, list((("nan" if nodata != nodata else repr(float(nodata))) + "\0").encode())))
    tags.sort()
    fmt = {2: "B", 3: "H", 4: "I", 12: "d", 16: "Q"}
    ifd_pos = pos
    nent = len(tags)
    esz = 20 if bigtiff else 12
    extra_pos = ifd_pos + (8 if bigtiff else 2) + nent * esz + osz
    ifd = struct.pack(E + ("Q" if bigtiff else "H"), nent)
    extra = b""
    for tag, typ, vals in tags:
        payload = struct.pack(E + fmt[typ] * len(vals), *vals)
        ifd += struct.pack(E + "HH" + ("Q" if bigtiff else "I"), tag, typ, len(vals))
        if len(payload) <= osz:
            ifd += payload.ljust(osz, b"\0")
        else:
            ifd += struct.pack(E + ("Q" if bigtiff else "I"), extra_pos + len(extra))
            extra += payload + (b"\0" if len(payload) & 1 else b"")
    ifd += struct.pack(E + ("Q" if bigtiff else "I"), 0)
    head = (b"MM" if big_endian else b"II") + (struct.pack(E + "HHHQ", 43, 8, 0, ifd_pos) if bigtiff else struct.pack(E + "HI", 42, ifd_pos))
    with open(path, "wb") as f:
        f.write(head)
        for c in chunks:
            f.write(c)
            if len(c) & 1:
                f.write(b"\0")
        f.write(ifd)
        f.write(extra)
