// SURVEY.md §8(f) N4, host half: reader for the archive's GeoTIFF tiles (MS1_IVV/IVH, SL1_*, SL2_*, MK0_MLU/MNA/DEM).
// The reference decodes them one by one in DataLoader workers with cv2.imread(path, IMREAD_ANYDEPTH) (dataset/Dataset.py:664-728)
// and rioxarray for the DEM (:730-737); here a batch of tiles is decoded by a small thread pool straight into one (pinned) fp32
// buffer [n][H][W] that goes to the GPU in a single copy, where clamp / nan_to_num / Normalize happen inside the first convolution
// (ksmi_conv_first_forward_raw).  Host-only translation unit: no device code.
//
// TIFF 6.0 + BigTIFF, both byte orders, strips and tiles, chunky and planar; compression none / LZW (5) / Deflate (8, 32946) /
// PackBits (32773); predictor 1 / 2 (horizontal differencing) / 3 (floating point); 8/16/32/64-bit unsigned, signed, IEEE samples.
// GeoTIFF tags read: ModelPixelScale (33550), ModelTiepoint (33922), GDAL_NODATA (42113).
#include <zlib.h>

#include <atomic>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ksmi.h"
#include "errors.h"

namespace {

struct Reader {
  const uint8_t* p;
  size_t n;
  bool be;       // file is big-endian
  bool big;      // BigTIFF
  bool ok(uint64_t off, uint64_t len) const { return off <= n && len <= n - off; }
  uint16_t u16(uint64_t o) const { return be ? (uint16_t)(p[o] << 8 | p[o + 1]) : (uint16_t)(p[o] | p[o + 1] << 8); }
  uint32_t u32(uint64_t o) const {
    return be ? ((uint32_t)p[o] << 24 | (uint32_t)p[o + 1] << 16 | (uint32_t)p[o + 2] << 8 | p[o + 3])
              : ((uint32_t)p[o + 3] << 24 | (uint32_t)p[o + 2] << 16 | (uint32_t)p[o + 1] << 8 | p[o]);
  }
  uint64_t u64(uint64_t o) const { return be ? ((uint64_t)u32(o) << 32 | u32(o + 4)) : ((uint64_t)u32(o + 4) << 32 | u32(o)); }
};

int type_size(int t) {
  switch (t) {
    case 1: case 2: case 6: case 7: return 1;
    case 3: case 8: return 2;
    case 4: case 9: case 11: case 13: return 4;
    case 5: case 10: case 12: case 16: case 17: case 18: return 8;
    default: return 0;
  }
}

struct Entry {
  int tag = 0, type = 0;
  uint64_t count = 0, data = 0;   // data = byte position of the values inside the file
};

struct Layout {
  ksmi_tiff_info info{};
  int planar = 1, photometric = 1;
  uint32_t rows_per_strip = 0, tile_w = 0, tile_h = 0;
  std::vector<uint64_t> offsets, counts;
};

bool entry_uint(const Reader& r, const Entry& e, uint64_t i, uint64_t* v) {
  const int ts = type_size(e.type);
  if (!ts || i >= e.count || !r.ok(e.data + i * ts, ts)) return false;
  const uint64_t o = e.data + i * ts;
  switch (e.type) {
    case 1: case 6: case 7: *v = r.p[o]; return true;
    case 3: case 8: *v = r.u16(o); return true;
    case 4: case 9: case 13: *v = r.u32(o); return true;
    case 16: case 17: case 18: *v = r.u64(o); return true;
    default: return false;
  }
}

bool entry_double(const Reader& r, const Entry& e, uint64_t i, double* v) {
  if (e.type == 12) {
    if (i >= e.count || !r.ok(e.data + i * 8, 8)) return false;
    const uint64_t b = r.u64(e.data + i * 8);
    memcpy(v, &b, 8);
    return true;
  }
  if (e.type == 11) {
    if (i >= e.count || !r.ok(e.data + i * 4, 4)) return false;
    const uint32_t b = r.u32(e.data + i * 4);
    float f;
    memcpy(&f, &b, 4);
    *v = f;
    return true;
  }
  uint64_t u;
  if (!entry_uint(r, e, i, &u)) return false;
  *v = (double)u;
  return true;
}

// first image file directory -> layout; returns an error text or nullptr
const char* parse(const Reader& r0, Layout* L) {
  Reader r = r0;
  if (r.n < 8) return "shorter than a TIFF header";
  if (r.p[0] == 'I' && r.p[1] == 'I') r.be = false;
  else if (r.p[0] == 'M' && r.p[1] == 'M') r.be = true;
  else return "not a TIFF file (byte-order mark)";
  const int magic = r.u16(2);
  uint64_t ifd;
  if (magic == 42) { r.big = false; ifd = r.u32(4); }
  else if (magic == 43) {
    if (r.n < 16 || r.u16(4) != 8) return "BigTIFF with an offset size other than 8";
    r.big = true; ifd = r.u64(8);
  } else return "not a TIFF file (magic)";
  const int esz = r.big ? 20 : 12, csz = r.big ? 8 : 2;
  if (!r.ok(ifd, csz)) return "image file directory outside the file";
  const uint64_t nent = r.big ? r.u64(ifd) : r.u16(ifd);
  if (nent > 4096 || !r.ok(ifd + csz, nent * esz)) return "image file directory outside the file";
  ksmi_tiff_info& I = L->info;
  I.bands = 1; I.bits = 1; I.sample_format = 1; I.compression = 1; I.predictor = 1;
  I.big_endian = r.be; I.bigtiff = r.big;
  Entry off{}, cnt{};
  bool has_off = false, has_cnt = false;
  for (uint64_t k = 0; k < nent; ++k) {
    const uint64_t o = ifd + csz + k * esz;
    Entry e;
    e.tag = r.u16(o); e.type = r.u16(o + 2);
    e.count = r.big ? r.u64(o + 4) : r.u32(o + 4);
    const int ts = type_size(e.type);
    if (!ts) continue;                               // unknown field type: skip the field (TIFF 6.0 §2)
    if (e.count > r.n / (uint64_t)ts) return "a field is larger than the file";       // (also keeps count * size from wrapping)
    const uint64_t bytes = e.count * ts, inl = r.big ? 8 : 4, vpos = o + (r.big ? 12 : 8);
    e.data = bytes <= inl ? vpos : (r.big ? r.u64(vpos) : r.u32(vpos));
    if (!r.ok(e.data, bytes)) return "a field points outside the file";
    uint64_t v = 0;
    double d = 0;
    switch (e.tag) {
      case 256: if (entry_uint(r, e, 0, &v)) I.width = (int32_t)v; break;
      case 257: if (entry_uint(r, e, 0, &v)) I.height = (int32_t)v; break;
      case 258:
        if (entry_uint(r, e, 0, &v)) I.bits = (int32_t)v;
        for (uint64_t i = 1; i < e.count; ++i) { uint64_t w; if (entry_uint(r, e, i, &w) && w != v) return "bands of different bit depth"; }
        break;
      case 259: if (entry_uint(r, e, 0, &v)) I.compression = (int32_t)v; break;
      case 262: if (entry_uint(r, e, 0, &v)) L->photometric = (int)v; break;
      case 266: if (entry_uint(r, e, 0, &v) && v != 1) return "FillOrder 2 is not supported"; break;
      case 273: case 324: off = e; has_off = true; break;
      case 277: if (entry_uint(r, e, 0, &v)) I.bands = (int32_t)v; break;
      case 278: if (entry_uint(r, e, 0, &v)) L->rows_per_strip = (uint32_t)(v > 0xffffffffu ? 0xffffffffu : v); break;
      case 279: case 325: cnt = e; has_cnt = true; break;
      case 284: if (entry_uint(r, e, 0, &v)) L->planar = (int)v; break;
      case 317: if (entry_uint(r, e, 0, &v)) I.predictor = (int32_t)v; break;
      case 322: if (entry_uint(r, e, 0, &v)) L->tile_w = (uint32_t)v; break;
      case 323: if (entry_uint(r, e, 0, &v)) L->tile_h = (uint32_t)v; break;
      case 339:
        if (entry_uint(r, e, 0, &v)) I.sample_format = (int32_t)v;
        for (uint64_t i = 1; i < e.count; ++i) { uint64_t w; if (entry_uint(r, e, i, &w) && w != v) return "bands of different sample format"; }
        break;
      case 33550:
        if (e.count >= 2 && entry_double(r, e, 0, &d)) { I.pixel_scale[0] = d; entry_double(r, e, 1, &I.pixel_scale[1]); I.has_geo |= 1; }
        break;
      case 33922:
        if (e.count >= 6) {
          double t[6];
          for (int i = 0; i < 6; ++i) entry_double(r, e, i, &t[i]);
          I.origin[0] = t[3]; I.origin[1] = t[4];          // model position of raster point (t[0], t[1]); GDAL writes (0, 0)
          I.tie_pixel[0] = t[0]; I.tie_pixel[1] = t[1];
          I.has_geo |= 2;
        }
        break;
      case 42113: {
        char buf[64] = {0};
        const size_t m = bytes < 63 ? (size_t)bytes : 63;   // `bytes` is the length r.ok() validated above
        memcpy(buf, r.p + e.data, m);
        char* end = nullptr;
        const double nd = strtod(buf, &end);                // "nan" parses to NaN
        if (end != buf) { I.nodata = nd; I.has_nodata = 1; }
        break;
      }
      default: break;
    }
  }
  if (I.width <= 0 || I.height <= 0) return "no image size";
  if (I.bands < 1 || I.bands > 64) return "unsupported band count";
  if (I.sample_format == 4) I.sample_format = 1;     // "undefined" data: unsigned by convention
  if (I.sample_format < 1 || I.sample_format > 3) return "unsupported SampleFormat";
  if (!(I.bits == 8 || I.bits == 16 || I.bits == 32 || I.bits == 64)) return "unsupported BitsPerSample (8, 16, 32, 64)";
  if (I.sample_format == 3 && I.bits < 32) return "unsupported floating-point width";
  if (L->photometric == 6 || L->photometric == 3) return "palette / YCbCr images are not tiles of this archive";
  if (!(I.compression == 1 || I.compression == 5 || I.compression == 8 || I.compression == 32946 || I.compression == 32773))
    return "unsupported Compression (none, LZW, Deflate, PackBits)";
  if (I.predictor < 1 || I.predictor > 3) return "unsupported Predictor";
  if (I.compression == 1 || I.compression == 32773) I.predictor = 1;    // libtiff: the predictor belongs to the LZW / Deflate codecs only
  if (I.predictor == 3 && I.sample_format != 3) return "floating-point predictor on integer samples";
  if (I.predictor == 2 && I.sample_format == 3) return "horizontal differencing on floating-point samples";
  if (L->planar != 1 && L->planar != 2) return "bad PlanarConfiguration";
  if (!has_off || !has_cnt) return "no strip / tile offsets";
  I.tiled = L->tile_w != 0;
  uint64_t per_plane;
  if (I.tiled) {
    if (!L->tile_h) return "TileWidth without TileLength";
    per_plane = (uint64_t)((I.width + L->tile_w - 1) / L->tile_w) * ((I.height + L->tile_h - 1) / L->tile_h);
  } else {
    if (L->rows_per_strip == 0 || L->rows_per_strip > (uint32_t)I.height) L->rows_per_strip = (uint32_t)I.height;
    per_plane = (I.height + L->rows_per_strip - 1) / L->rows_per_strip;
  }
  const uint64_t chunks = per_plane * (L->planar == 2 ? I.bands : 1);
  if (off.count < chunks || cnt.count < chunks) return "fewer strips / tiles than the image needs";
  L->offsets.resize(chunks); L->counts.resize(chunks);
  for (uint64_t i = 0; i < chunks; ++i) {
    if (!entry_uint(r, off, i, &L->offsets[i]) || !entry_uint(r, cnt, i, &L->counts[i])) return "bad strip / tile table";
    if (!r.ok(L->offsets[i], L->counts[i])) return "a strip / tile lies outside the file";
  }
  return nullptr;
}

// ---- decoders ---------------------------------------------------------------------------------
// TIFF LZW (TIFF 6.0 §13): MSB-first codes of 9..12 bits, Clear = 256, EOI = 257, code width grows one code early.
// A table entry is (where in the OUTPUT its string was first written, length): a code is expanded by a forward copy from earlier
// output instead of a walk along a prefix chain, and the entry a code creates (previous string + first byte of this one) is simply
// the previous string's position with one more byte, because this string is written right behind it.
int64_t lzw_decode(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap) {
  struct Ent { uint32_t pos, len; };
  Ent tab[4096];
  int next = 258, bits = 9;
  int64_t prev_pos = -1;          // start of the previous code's string in dst (-1: none since the last Clear)
  uint32_t prev_len = 0;
  uint64_t acc = 0;
  int have = 0;
  int64_t sp = 0, dp = 0;
  for (;;) {
    if (have < bits) {
      if (sp + 4 <= n) { acc = (acc << 32) | ((uint32_t)src[sp] << 24 | (uint32_t)src[sp + 1] << 16 | (uint32_t)src[sp + 2] << 8 | src[sp + 3]); sp += 4; have += 32; }
      else {
        while (have < bits && sp < n) { acc = (acc << 8) | src[sp++]; have += 8; }
        if (have < bits) break;                               // ran out without EOI: what was decoded stands
      }
    }
    const int code = (int)((acc >> (have - bits)) & ((1u << bits) - 1));
    have -= bits;
    if (code == 257) break;
    if (code == 256) { next = 258; bits = 9; prev_pos = -1; continue; }
    uint32_t len;
    if (code < 256) {
      if (dp >= cap) break;
      dst[dp] = (uint8_t)code;
      len = 1;
    } else {
      int64_t from;
      if (prev_pos < 0) return -1;
      if (code < next) { from = tab[code].pos; len = tab[code].len; }
      else if (code == next) { from = prev_pos; len = prev_len + 1; }
      else return -1;
      if (dp + len > cap) {                                   // the strip is full: copy what fits, the rest is padding
        for (int64_t k = 0; dp + k < cap; ++k) dst[dp + k] = dst[from + k];
        dp = cap;
        break;
      }
      const uint8_t* f = dst + from;
      uint8_t* t = dst + dp;
      if (dp - from >= 8 && len >= 8) {
        uint32_t k = 0;
        for (; k + 8 <= len; k += 8) memcpy(t + k, f + k, 8);
        for (; k < len; ++k) t[k] = f[k];
      } else {
        for (uint32_t k = 0; k < len; ++k) t[k] = f[k];       // may overlap its own output (the KwKwK case): forward, byte by byte
      }
    }
    if (prev_pos >= 0 && next < 4096) {
      tab[next] = {(uint32_t)prev_pos, prev_len + 1};
      ++next;
      if (next == (1 << bits) - 1 && bits < 12) ++bits;
    }
    prev_pos = dp; prev_len = len;
    dp += len;
  }
  return dp;
}

int64_t packbits_decode(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap) {
  int64_t sp = 0, dp = 0;
  while (sp < n && dp < cap) {
    const int8_t h = (int8_t)src[sp++];
    if (h >= 0) {
      int64_t m = (int64_t)h + 1;
      if (sp + m > n) m = n - sp;
      if (dp + m > cap) m = cap - dp;
      memcpy(dst + dp, src + sp, (size_t)m);
      sp += (int64_t)h + 1; dp += m;
    } else if (h != -128) {
      if (sp >= n) break;
      int64_t m = 1 - (int64_t)h;
      if (dp + m > cap) m = cap - dp;
      memset(dst + dp, src[sp++], (size_t)m);
      dp += m;
    }
  }
  return dp;
}

inline void swap_bytes(uint8_t* p, int64_t count, int size) {
  for (int64_t i = 0; i < count; ++i, p += size)
    for (int a = 0, b = size - 1; a < b; ++a, --b) { const uint8_t t = p[a]; p[a] = p[b]; p[b] = t; }
}

template <typename T>
void undo_hdiff(uint8_t* row, int64_t samples, int stride) {
  T* v = (T*)row;
  if (stride == 1) {                                   // running sum in a register (no store-to-load chain through memory)
    T acc = v[0];
    for (int64_t i = 1; i < samples; ++i) { acc = (T)(acc + v[i]); v[i] = acc; }
    return;
  }
  for (int64_t i = stride; i < samples; ++i) v[i] = (T)(v[i] + v[i - stride]);
}

// libtiff's floating-point predictor (Adobe Photoshop TIFF Technical Note 3): bytes differenced across the row with stride = bands,
// then the row holds byte planes, most significant first
void undo_fpred(uint8_t* row, uint8_t* tmp, int64_t samples, int stride, int bps) {
  const int64_t rb = samples * bps;
  if (stride == 1) {
    uint8_t acc = row[0];
    for (int64_t i = 1; i < rb; ++i) { acc = (uint8_t)(acc + row[i]); row[i] = acc; }
  } else {
    for (int64_t i = (int64_t)stride; i < rb; ++i) row[i] = (uint8_t)(row[i] + row[i - stride]);
  }
  memcpy(tmp, row, (size_t)rb);
  if (bps == 4) {
    const uint8_t *p0 = tmp, *p1 = tmp + samples, *p2 = tmp + 2 * samples, *p3 = tmp + 3 * samples;
    uint32_t* o = (uint32_t*)row;
    for (int64_t i = 0; i < samples; ++i) o[i] = (uint32_t)p0[i] << 24 | (uint32_t)p1[i] << 16 | (uint32_t)p2[i] << 8 | p3[i];
    return;
  }
  for (int64_t i = 0; i < samples; ++i)
    for (int b = 0; b < bps; ++b) row[i * bps + (bps - 1 - b)] = tmp[(int64_t)b * samples + i];     // little-endian host
}

template <typename T>
inline float as_f32(const uint8_t* p) { T v; memcpy(&v, p, sizeof(T)); return (float)v; }

inline float sample_f32(const uint8_t* p, int bits, int fmt) {
  switch (fmt * 100 + bits) {
    case 108: return (float)*p;
    case 116: return as_f32<uint16_t>(p);
    case 132: return as_f32<uint32_t>(p);
    case 164: return as_f32<uint64_t>(p);
    case 208: return (float)*(const int8_t*)p;
    case 216: return as_f32<int16_t>(p);
    case 232: return as_f32<int32_t>(p);
    case 264: return as_f32<int64_t>(p);
    case 332: return as_f32<float>(p);
    default: return as_f32<double>(p);
  }
}

struct FileBytes {
  std::vector<uint8_t> b;
  const char* load(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return "cannot open";
    if (fseek(f, 0, SEEK_END)) { fclose(f); return "cannot seek"; }
    const long n = ftell(f);
    if (n < 0) { fclose(f); return "cannot tell the size"; }
    rewind(f);
    b.resize((size_t)n);
    const size_t got = n ? fread(b.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n ? nullptr : "short read";
  }
};

// decode the first image of `path`: either into fp32 [bands][H][W] (out_f32) or into the file's sample type, host byte order,
// [bands][H][W] (out_raw)
std::string decode_file_(const char* path, float* out_f32, uint8_t* out_raw, int64_t cap_elems, ksmi_tiff_info* info_out,
                         int want_h, int want_w, int want_bands) {
  FileBytes fb;
  if (const char* e = fb.load(path)) return std::string(e) + ": " + path;
  Reader r{fb.b.data(), fb.b.size(), false, false};
  Layout L;
  if (const char* e = parse(r, &L)) return std::string(e) + ": " + path;
  const ksmi_tiff_info& I = L.info;
  if (info_out) *info_out = I;
  if (!out_f32 && !out_raw) return "";
  if ((want_h && I.height != want_h) || (want_w && I.width != want_w) || (want_bands && I.bands != want_bands))
    return "tile of " + std::to_string(I.bands) + " x " + std::to_string(I.height) + " x " + std::to_string(I.width) + ", expected " +
           std::to_string(want_bands) + " x " + std::to_string(want_h) + " x " + std::to_string(want_w) + ": " + path;
  const int64_t H = I.height, W = I.width, Bn = I.bands;
  if (H > (1 << 20) || W > (1 << 20)) return std::string("implausible image size: ") + path;     // (Bn <= 64: the product below cannot wrap)
  if (cap_elems < Bn * H * W) return std::string("output buffer too small: ") + path;
  const int bps = I.bits / 8;
  const int spp = L.planar == 1 ? (int)Bn : 1;                 // samples per pixel inside one chunk
  const int64_t cw = I.tiled ? L.tile_w : W, ch = I.tiled ? L.tile_h : L.rows_per_strip;
  const int64_t across = I.tiled ? (W + cw - 1) / cw : 1, down = (H + ch - 1) / ch;
  if (cw <= 0 || ch <= 0 || cw > (1 << 20) || ch > (1 << 20) || cw * ch * spp * bps > ((int64_t)1 << 31))
    return std::string("implausible strip / tile size: ") + path;
  std::vector<uint8_t> buf((size_t)(cw * ch * spp * bps)), tmp((size_t)(cw * spp * bps));
  const int planes = L.planar == 2 ? (int)Bn : 1;
  for (int pl = 0; pl < planes; ++pl)
    for (int64_t cy = 0; cy < down; ++cy)
      for (int64_t cx = 0; cx < across; ++cx) {
        const uint64_t idx = ((uint64_t)pl * down + cy) * across + cx;
        const int64_t rows = I.tiled ? ch : (H - cy * ch < ch ? H - cy * ch : ch);     // strips: the last one is short; tiles: padded
        const int64_t need = rows * cw * spp * bps;
        const uint8_t* src = r.p + L.offsets[idx];
        const int64_t n = (int64_t)L.counts[idx];
        int64_t got;
        switch (I.compression) {
          case 1: got = n < need ? n : need; memcpy(buf.data(), src, (size_t)got); break;
          case 5: got = lzw_decode(src, n, buf.data(), need); break;
          case 32773: got = packbits_decode(src, n, buf.data(), need); break;
          default: {
            uLongf dl = (uLongf)need;
            const int zr = uncompress(buf.data(), &dl, src, (uLong)n);
            got = (zr == Z_OK || zr == Z_BUF_ERROR) ? (int64_t)dl : -1;
          }
        }
        if (got < need) return "strip / tile " + std::to_string(idx) + " decodes to " + std::to_string(got) + " of " + std::to_string(need) + " bytes: " + path;
        const int64_t samples = cw * spp;
        for (int64_t y = 0; y < rows; ++y) {
          uint8_t* row = buf.data() + y * samples * bps;
          if (I.predictor == 3) undo_fpred(row, tmp.data(), samples, spp, bps);
          else {
            if (I.big_endian && bps > 1) swap_bytes(row, samples, bps);
            if (I.predictor == 2) {
              if (bps == 1) undo_hdiff<uint8_t>(row, samples, spp);
              else if (bps == 2) undo_hdiff<uint16_t>(row, samples, spp);
              else if (bps == 4) undo_hdiff<uint32_t>(row, samples, spp);
              else undo_hdiff<uint64_t>(row, samples, spp);
            }
          }
          const int64_t oy = cy * ch + y;
          if (oy >= H) break;
          const int64_t x0 = cx * cw, xs = W - x0 < cw ? W - x0 : cw;
          for (int s = 0; s < spp; ++s) {
            const int64_t band = L.planar == 2 ? pl : s;
            const int64_t o = (band * H + oy) * W + x0;
            if (out_f32 && spp == 1 && I.sample_format == 3 && I.bits == 32) memcpy(out_f32 + o, row, (size_t)xs * 4);
            else if (out_f32 && spp == 1 && I.sample_format == 1 && I.bits == 8)
              for (int64_t x = 0; x < xs; ++x) out_f32[o + x] = (float)row[x];
            else if (out_f32)
              for (int64_t x = 0; x < xs; ++x) out_f32[o + x] = sample_f32(row + (x * spp + s) * bps, I.bits, I.sample_format);
            else
              for (int64_t x = 0; x < xs; ++x) memcpy(out_raw + (o + x) * bps, row + (x * spp + s) * bps, (size_t)bps);
          }
        }
      }
  return "";
}

std::string decode_file(const char* path, float* out_f32, uint8_t* out_raw, int64_t cap_elems, ksmi_tiff_info* info_out, int want_h,
                        int want_w, int want_bands) {
  try {
    return decode_file_(path, out_f32, out_raw, cap_elems, info_out, want_h, want_w, want_bands);
  } catch (const std::exception& e) {                       // allocation failure on a damaged header: an error, not a crash
    return std::string(e.what()) + ": " + path;
  }
}

// every NaN of a tile takes the value of the nearest valid pixel (exact Euclidean distance on the pixel grid; ties: the smaller
// row offset, rows above before rows below, left before right).  Per row the nearest valid column to the left / right of every x
// is tabulated once; a NaN pixel then scans rows outward until the row offset alone exceeds the best distance.
void fill_nodata_tile(float* a, int H, int W, std::vector<int>& left, std::vector<int>& right, std::vector<float>& src) {
  bool any = false, all = true;
  for (int64_t i = 0; i < (int64_t)H * W; ++i) { const bool bad = a[i] != a[i]; any |= bad; all &= bad; }
  if (!any || all) return;
  src.assign(a, a + (int64_t)H * W);
  left.resize((size_t)H * W); right.resize((size_t)H * W);
  for (int y = 0; y < H; ++y) {
    const float* r = src.data() + (int64_t)y * W;
    int last = -1;
    for (int x = 0; x < W; ++x) { if (r[x] == r[x]) last = x; left[(size_t)y * W + x] = last; }
    last = -1;
    for (int x = W - 1; x >= 0; --x) { if (r[x] == r[x]) last = x; right[(size_t)y * W + x] = last; }
  }
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      if (src[(int64_t)y * W + x] == src[(int64_t)y * W + x]) continue;
      int64_t best = INT64_MAX;
      float val = 0.f;
      for (int d = 0; d < H; ++d) {
        if ((int64_t)d * d >= best) break;
        for (int sgn = 0; sgn < (d ? 2 : 1); ++sgn) {
          const int yy = sgn ? y + d : y - d;
          if (yy < 0 || yy >= H) continue;
          const int l = left[(size_t)yy * W + x], r = right[(size_t)yy * W + x];
          if (l >= 0) { const int64_t q = (int64_t)d * d + (int64_t)(x - l) * (x - l); if (q < best) { best = q; val = src[(int64_t)yy * W + l]; } }
          if (r >= 0) { const int64_t q = (int64_t)d * d + (int64_t)(r - x) * (r - x); if (q < best) { best = q; val = src[(int64_t)yy * W + r]; } }
        }
      }
      a[(int64_t)y * W + x] = val;
    }
}

}  // namespace

extern "C" {

int ksmi_tiff_info_read(const char* path, ksmi_tiff_info* info) {
  if (!path || !info) return ksmi_fail(KSMI_E_ARG, "tiff_info: null argument");
  const std::string e = decode_file(path, nullptr, nullptr, 0, info, 0, 0, 0);
  return e.empty() ? 0 : ksmi_fail(KSMI_E_ARG, e.c_str());
}

int ksmi_tiff_read_f32(const char* path, float* out, int64_t cap_elems, ksmi_tiff_info* info) {
  if (!path || !out) return ksmi_fail(KSMI_E_ARG, "tiff_read: null argument");
  const std::string e = decode_file(path, out, nullptr, cap_elems, info, 0, 0, 0);
  return e.empty() ? 0 : ksmi_fail(KSMI_E_ARG, e.c_str());
}

int ksmi_tiff_read_native(const char* path, void* out, int64_t cap_elems, ksmi_tiff_info* info) {
  if (!path || !out) return ksmi_fail(KSMI_E_ARG, "tiff_read: null argument");
  const std::string e = decode_file(path, nullptr, (uint8_t*)out, cap_elems, info, 0, 0, 0);
  return e.empty() ? 0 : ksmi_fail(KSMI_E_ARG, e.c_str());
}

int ksmi_tile_batch_read(const char* const* paths, int n, float* out, int H, int W, int threads) {
  return ksmi_tile_batch_read_bands(paths, n, out, 1, H, W, threads);
}

int ksmi_tile_batch_read_bands(const char* const* paths, int n, float* out, int bands, int H, int W, int threads) {
  if (n < 0 || (n && (!paths || !out)) || H <= 0 || W <= 0 || bands < 1) return ksmi_fail(KSMI_E_ARG, "tile_batch_read: bad argument");
  if (threads < 1) threads = 1;
  if (threads > n) threads = n > 0 ? n : 1;
  std::atomic<int> next{0}, failed{0};
  std::string first_error;
  std::atomic_flag lock = ATOMIC_FLAG_INIT;
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n || failed.load()) return;
      const std::string e = paths[i] ? decode_file(paths[i], out + (int64_t)i * bands * H * W, nullptr, (int64_t)bands * H * W, nullptr, H, W, bands)
                                     : "null path";
      if (!e.empty()) {
        while (lock.test_and_set()) {}
        if (!failed.exchange(1)) first_error = e;
        lock.clear();
      }
    }
  };
  if (threads == 1) work();
  else {
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(work);
    for (auto& t : pool) t.join();
  }
  return failed.load() ? ksmi_fail(KSMI_E_ARG, first_error.c_str()) : 0;
}

int ksmi_tiles_fill_nodata(float* tiles, int n, int H, int W, int threads) {
  if (n < 0 || (n && !tiles) || H <= 0 || W <= 0) return ksmi_fail(KSMI_E_ARG, "tiles_fill_nodata: bad argument");
  if (threads < 1) threads = 1;
  if (threads > n) threads = n > 0 ? n : 1;
  std::atomic<int> next{0};
  auto work = [&]() {
    std::vector<int> left, right;
    std::vector<float> src;
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) return;
      fill_nodata_tile(tiles + (int64_t)i * H * W, H, W, left, right, src);
    }
  };
  try {
    if (threads == 1) work();
    else {
      std::vector<std::thread> pool;
      for (int t = 0; t < threads; ++t) pool.emplace_back(work);
      for (auto& t : pool) t.join();
    }
  } catch (const std::exception& e) {
    return ksmi_fail(KSMI_E_ARG, e.what());
  }
  return 0;
}

}  // extern "C"
