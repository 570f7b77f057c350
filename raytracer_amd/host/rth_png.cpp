// Minimal PNG codec on top of zlib. The reference uses LodePNG (src/image/lodepng.*) only to
// turn a file into 8-bit RGBA and back (src/image/PNG.cpp:125-165); what has to match is the
// pixels, not the file bytes. Colour handling around it (÷255, sRGB <-> linear with the
// reference's double-precision pow) is restated from src/image/PNG.cpp:7-30.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "rth_scene.h"
#include "rth_threads.h"

namespace rth {
namespace {

uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
void put32(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

int paeth(int a, int b, int c) {
  int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}

// src/image/PNG.cpp:14-18 — float argument, double constants, result narrowed to float
float gammaToLinear(float channel) {
  if (channel < 0.04045) return channel / 12.92;
  return std::pow((channel + 0.055) / 1.055, 2.4);
}
// src/image/PNG.cpp:7-12
float linearToGamma(float channel) {
  if (channel < 0.0031308) return 12.92 * channel;
  return 1.055 * std::pow(channel, 1 / 2.4) - 0.055;
}

}  // namespace

bool decodePNG8(const std::string &filename, int &w, int &h, std::vector<uint8_t> &rgba8, std::string &error) {
  FILE *f = std::fopen(filename.c_str(), "rb");
  if (!f) { error = "failed to open file for reading: " + filename; return false; }
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t got;
  while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
  std::fclose(f);
  static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) { error = "not a PNG file: " + filename; return false; }
  size_t pos = 8;
  int bitDepth = 0, colorType = 0, interlace = 0;
  std::vector<uint8_t> idat, palette, trns;
  bool seenIHDR = false;
  while (pos + 12 <= file.size()) {
    uint32_t len = be32(&file[pos]);
    if (pos + 12 + (size_t)len > file.size()) { error = "truncated PNG chunk"; return false; }
    const uint8_t *type = &file[pos + 4];
    const uint8_t *data = &file[pos + 8];
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) { error = "bad IHDR"; return false; }
      w = (int)be32(data); h = (int)be32(data + 4);
      bitDepth = data[8]; colorType = data[9]; interlace = data[12];
      // Only the combinations of the PNG specification's table: anything else would shift by a negative or
      // oversized amount, or divide by zero, below. LodePNG reports a decoder error for them too.
      const bool depthOk = colorType == 0   ? (bitDepth == 1 || bitDepth == 2 || bitDepth == 4 || bitDepth == 8 || bitDepth == 16)
                           : colorType == 3 ? (bitDepth == 1 || bitDepth == 2 || bitDepth == 4 || bitDepth == 8)
                           : (colorType == 2 || colorType == 4 || colorType == 6) ? (bitDepth == 8 || bitDepth == 16)
                                                                                  : false;
      if (!depthOk) { error = "bad PNG colour type / bit depth"; return false; }
      if (data[10] != 0 || data[11] != 0) { error = "bad PNG compression or filter method"; return false; }
      if (w <= 0 || h <= 0 || (uint64_t)w * (uint64_t)h > ((uint64_t)1 << 28)) { error = "bad PNG size"; return false; }
      seenIHDR = true;
    } else if (!std::memcmp(type, "PLTE", 4)) palette.assign(data, data + len);
    else if (!std::memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
    else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
    else if (!std::memcmp(type, "IEND", 4)) break;
    pos += 12 + (size_t)len;
  }
  if (!seenIHDR || w <= 0 || h <= 0) { error = "PNG without a valid IHDR"; return false; }
  if (interlace != 0 && interlace != 1) { error = "bad PNG interlace method"; return false; }
  int channels;
  switch (colorType) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: error = "bad PNG colour type"; return false;
  }
  const int bitsPerPixel = channels * bitDepth;
  const int bpp = std::max(1, bitsPerPixel / 8);
  auto rowBytesOf = [&](int pixels) { return ((size_t)pixels * bitsPerPixel + 7) / 8; };
  // sub-images: the whole image, or the seven Adam7 passes (x start, y start, x step, y step)
  struct Pass { int x0, y0, dx, dy; };
  static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  static const Pass whole[1] = {{0, 0, 1, 1}};
  const Pass *passes = interlace ? adam7 : whole;
  const int numPasses = interlace ? 7 : 1;
  size_t rawSize = 0;
  for (int p = 0; p < numPasses; ++p) {
    int pw = (w - passes[p].x0 + passes[p].dx - 1) / passes[p].dx, ph = (h - passes[p].y0 + passes[p].dy - 1) / passes[p].dy;
    if (pw > 0 && ph > 0) rawSize += (rowBytesOf(pw) + 1) * (size_t)ph;
  }
  std::vector<uint8_t> raw(rawSize);
  uLongf rawLen = (uLongf)raw.size();
  int zr = uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size());
  if (zr != Z_OK || rawLen != raw.size()) { error = "PNG inflate failed"; return false; }
  rgba8.assign((size_t)w * h * 4, 255);
  auto sample = [&](const uint8_t *row, int idx) -> unsigned {  // idx-th sample of the row, full bit depth
    if (bitDepth == 8) return row[idx];
    if (bitDepth == 16) return (unsigned)row[idx * 2] << 8 | row[idx * 2 + 1];
    int bitPos = idx * bitDepth;
    return (row[bitPos >> 3] >> (8 - bitDepth - (bitPos & 7))) & ((1u << bitDepth) - 1);
  };
  const unsigned maxv = (1u << bitDepth) - 1;
  auto to8 = [&](unsigned v) -> uint8_t {
    if (bitDepth == 8) return (uint8_t)v;
    if (bitDepth == 16) return (uint8_t)(v >> 8);
    return (uint8_t)((v * 255u) / maxv);
  };
  auto emit = [&](const uint8_t *row, int x, uint8_t *o) {  // pixel x of an unfiltered row -> 8-bit RGBA
    if (colorType == 0) {
      unsigned g = sample(row, x);
      o[0] = o[1] = o[2] = to8(g);
      if (trns.size() >= 2 && g == ((unsigned)trns[0] << 8 | trns[1])) o[3] = 0;
    } else if (colorType == 2) {
      unsigned r = sample(row, x * 3), g = sample(row, x * 3 + 1), b = sample(row, x * 3 + 2);
      o[0] = to8(r); o[1] = to8(g); o[2] = to8(b);
      if (trns.size() >= 6 && r == ((unsigned)trns[0] << 8 | trns[1]) && g == ((unsigned)trns[2] << 8 | trns[3]) &&
          b == ((unsigned)trns[4] << 8 | trns[5])) o[3] = 0;
    } else if (colorType == 3) {
      unsigned i = sample(row, x);
      if ((size_t)i * 3 + 2 < palette.size()) { o[0] = palette[i * 3]; o[1] = palette[i * 3 + 1]; o[2] = palette[i * 3 + 2]; }
      else { o[0] = o[1] = o[2] = 0; }
      if (i < trns.size()) o[3] = trns[i];
    } else if (colorType == 4) {
      o[0] = o[1] = o[2] = to8(sample(row, x * 2)); o[3] = to8(sample(row, x * 2 + 1));
    } else {
      o[0] = to8(sample(row, x * 4)); o[1] = to8(sample(row, x * 4 + 1));
      o[2] = to8(sample(row, x * 4 + 2)); o[3] = to8(sample(row, x * 4 + 3));
    }
  };
  size_t at = 0;
  for (int p = 0; p < numPasses; ++p) {
    const Pass &ps = passes[p];
    const int pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
    if (pw <= 0 || ph <= 0) continue;
    const size_t rowBytes = rowBytesOf(pw);
    std::vector<uint8_t> prev(rowBytes, 0), cur(rowBytes);
    for (int y = 0; y < ph; ++y) {
      const uint8_t *row = &raw[at];
      at += rowBytes + 1;
      const int ft = row[0];
      for (size_t i = 0; i < rowBytes; ++i) {  // undo the row filter
        int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
        int x = row[1 + i];
        switch (ft) {
          case 0: break;
          case 1: x += a; break;
          case 2: x += b; break;
          case 3: x += (a + b) >> 1; break;
          case 4: x += paeth(a, b, c); break;
          default: error = "bad PNG filter type"; return false;
        }
        cur[i] = (uint8_t)x;
      }
      const int oy = ps.y0 + y * ps.dy;
      for (int x = 0; x < pw; ++x) emit(cur.data(), x, &rgba8[((size_t)oy * w + ps.x0 + (size_t)x * ps.dx) * 4]);
      prev.swap(cur);
    }
  }
  return true;
}

bool readPNG(const std::string &filename, Image &out, std::string &error) {
  std::vector<uint8_t> bytes;
  if (!decodePNG8(filename, out.width, out.height, bytes, error)) return false;
  out.rgba.resize(bytes.size());
  // src/image/PNG.cpp:137-144: byte / 255.0f, then toLinear on r,g,b (alpha untouched)
  // a byte has 256 values: the same expressions, evaluated once per value instead of once per texel
  float linear[256], plain[256];
  for (int v = 0; v < 256; ++v) {
    plain[v] = (uint8_t)v / 255.0f;
    linear[v] = gammaToLinear((uint8_t)v / 255.0f);
  }
  for (size_t i = 0; i < bytes.size(); i += 4) {
    out.rgba[i + 0] = linear[bytes[i + 0]];
    out.rgba[i + 1] = linear[bytes[i + 1]];
    out.rgba[i + 2] = linear[bytes[i + 2]];
    out.rgba[i + 3] = plain[bytes[i + 3]];
  }
  return true;
}

void toSRGB8(const float *rgba, size_t numPixels, uint8_t *rgba8) {
  // src/image/PNG.cpp:151-157: toSRGB on r,g,b; static_cast<unsigned char>(round(x * 255))
  // pixels are independent: any split over the host threads gives the same bytes
  parallelChunks((int64_t)numPixels, 1 << 14, [&](int64_t begin, int64_t end, int) {
    for (int64_t i = begin; i < end; ++i) {
      float r = linearToGamma(rgba[i * 4 + 0]), g = linearToGamma(rgba[i * 4 + 1]), b = linearToGamma(rgba[i * 4 + 2]);
      float a = rgba[i * 4 + 3];
      rgba8[i * 4 + 0] = static_cast<unsigned char>(std::round(r * 255));
      rgba8[i * 4 + 1] = static_cast<unsigned char>(std::round(g * 255));
      rgba8[i * 4 + 2] = static_cast<unsigned char>(std::round(b * 255));
      rgba8[i * 4 + 3] = static_cast<unsigned char>(std::round(a * 255));
    }
  });
}

// zlib stream of `raw` (what an IDAT chunk holds), deflated in independent pieces on the host threads: every piece is a raw
// deflate stream that ends on a byte boundary (Z_SYNC_FLUSH; the last one Z_FINISH), so their concatenation is one valid
// deflate stream; the Adler-32 of the whole is combined from the pieces'. A piece cannot refer back into the one before
// it, which costs a fraction of a percent of compression. Level 6, like the single-call path it replaces.
static bool deflateParallel(const std::vector<uint8_t> &raw, std::vector<uint8_t> &out, std::string &error) {
  const int64_t kPiece = 1 << 19;
  const int64_t pieces = std::max<int64_t>(1, ((int64_t)raw.size() + kPiece - 1) / kPiece);
  std::vector<std::vector<uint8_t>> part((size_t)pieces);
  std::vector<uLong> adler((size_t)pieces, 1);
  std::vector<int> failed((size_t)pieces, 0);
  parallelItems(pieces, [&](int64_t i, int) {
    const size_t begin = (size_t)(i * kPiece), len = std::min<size_t>((size_t)kPiece, raw.size() - begin);
    z_stream z{};
    if (deflateInit2(&z, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { failed[(size_t)i] = 1; return; }
    std::vector<uint8_t> &dst = part[(size_t)i];
    dst.resize(deflateBound(&z, (uLong)len) + 16);
    z.next_in = const_cast<Bytef *>(raw.data() + begin); z.avail_in = (uInt)len;
    z.next_out = dst.data(); z.avail_out = (uInt)dst.size();
    const int rc = deflate(&z, i + 1 == pieces ? Z_FINISH : Z_SYNC_FLUSH);
    if ((i + 1 == pieces ? rc != Z_STREAM_END : rc != Z_OK) || z.avail_in != 0) failed[(size_t)i] = 1;
    dst.resize(z.total_out);
    deflateEnd(&z);
    adler[(size_t)i] = adler32(adler32(0L, Z_NULL, 0), raw.data() + begin, (uInt)len);
  });
  for (int f : failed) if (f) { error = "PNG deflate failed"; return false; }
  out.clear();
  out.push_back(0x78); out.push_back(0x9c);  // zlib header: deflate, 32 KiB window, default level, no dictionary
  uLong sum = adler32(0L, Z_NULL, 0);
  for (int64_t i = 0; i < pieces; ++i) {
    out.insert(out.end(), part[(size_t)i].begin(), part[(size_t)i].end());
    const size_t begin = (size_t)(i * kPiece), len = std::min<size_t>((size_t)kPiece, raw.size() - begin);
    sum = adler32_combine(sum, adler[(size_t)i], (z_off_t)len);
  }
  put32(out, (uint32_t)sum);
  return true;
}

bool writePNG8(const std::string &filename, int w, int h, const uint8_t *rgba8, std::string &error) {
  std::vector<uint8_t> raw((size_t)h * ((size_t)w * 4 + 1));
  for (int y = 0; y < h; ++y) {
    uint8_t *row = &raw[(size_t)y * ((size_t)w * 4 + 1)];
    row[0] = 0;  // filter type None
    std::memcpy(row + 1, rgba8 + (size_t)y * w * 4, (size_t)w * 4);
  }
  std::vector<uint8_t> comp;
  if (!deflateParallel(raw, comp, error)) return false;
  std::vector<uint8_t> out = {137, 80, 78, 71, 13, 10, 26, 10};
  auto chunk = [&](const char *type, const std::vector<uint8_t> &data) {
    put32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, (uint32_t)crc32(0, &out[start], (uInt)(out.size() - start)));
  };
  std::vector<uint8_t> ihdr;
  put32(ihdr, (uint32_t)w); put32(ihdr, (uint32_t)h);
  ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  chunk("IHDR", ihdr);
  chunk("IDAT", comp);
  chunk("IEND", {});
  FILE *f = std::fopen(filename.c_str(), "wb");
  if (!f) { error = "failed to open file for writing: " + filename; return false; }
  bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
  std::fclose(f);
  if (!ok) error = "short write: " + filename;
  return ok;
}

}  // namespace rth
