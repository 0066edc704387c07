#include "DatasetIO.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace VDO_SLAM {

namespace {
bool read_file(const std::string& path, std::vector<unsigned char>& buf) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  if (n < 0) { std::fclose(f); return false; }
  buf.resize((size_t)n);
  const size_t got = n ? std::fread(buf.data(), 1, (size_t)n, f) : 0;
  std::fclose(f);
  return got == (size_t)n;
}
constexpr int kMaxImageDim = 1 << 15;       // header fields are untrusted: sizes beyond this are rejected before any allocation
unsigned be32(const unsigned char* p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }
}  // namespace

bool ReadOpticalFlow(const std::string& path, cv::Mat& flow) {
  std::vector<unsigned char> b;
  if (!read_file(path, b) || b.size() < 12) return false;
  float magic; int w, h;
  std::memcpy(&magic, b.data(), 4); std::memcpy(&w, b.data() + 4, 4); std::memcpy(&h, b.data() + 8, 4);
  if (magic != 202021.25f || w <= 0 || h <= 0 || w > kMaxImageDim || h > kMaxImageDim || b.size() < 12 + (size_t)w * h * 8) return false;      // "PIEH"
  flow.create(h, w, cv::CV_32FC2);
  std::memcpy(flow.data, b.data() + 12, (size_t)w * h * 8);
  return true;
}

bool LoadMask(const std::string& path, cv::Mat& mask) {
  std::vector<unsigned char> b;
  if (mask.empty() || !read_file(path, b)) return false;
  std::memset(mask.data, 0, mask.step * (size_t)mask.rows);
  const unsigned char *p = b.data(), *end = p + b.size();
  int row = 0;
  while (p < end && row < mask.rows) {
    // one text line = one image row; the first `cols` integers count (LoadMask reads exactly imMask.cols per non-empty line)
    const unsigned char* eol = (const unsigned char*)std::memchr(p, '\n', (size_t)(end - p));
    if (!eol) eol = end;
    int col = 0;
    bool any = false;
    const unsigned char* q = p;
    while (q < eol) {
      while (q < eol && (*q == ' ' || *q == '\t' || *q == '\r')) ++q;
      if (q >= eol) break;
      bool neg = false;
      if (*q == '-') { neg = true; ++q; }
      if (q >= eol || *q < '0' || *q > '9') { ++q; continue; }
      int v = 0;
      while (q < eol && *q >= '0' && *q <= '9') { if (v < 100000000) v = v * 10 + (*q - '0'); ++q; }      // (clamped: no signed overflow on a long digit run)
      any = true;
      if (col < mask.cols) mask.at<int32_t>(row, col) = neg ? -v : v;
      ++col;
    }
    if (any) ++row;                                   // empty lines are skipped like `if(!s.empty())`
    p = eol + 1;
  }
  return row > 0;
}

bool ReadPNG(const std::string& path, cv::Mat& img, bool as_float) {
  std::vector<unsigned char> b;
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (!read_file(path, b) || b.size() < 33 || std::memcmp(b.data(), sig, 8) != 0) return false;
  unsigned w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
  std::vector<unsigned char> z;
  size_t pos = 8;
  while (pos + 12 <= b.size()) {
    const unsigned len = be32(&b[pos]);
    const unsigned char* type = &b[pos + 4];
    if (pos + 12 + (size_t)len > b.size()) return false;
    const unsigned char* data = &b[pos + 8];
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len < 13) return false;
      w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
    } else if (!std::memcmp(type, "IDAT", 4)) {
      z.insert(z.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 6 ? 4 : ctype == 4 ? 2 : 0;
  if (!w || !h || w > (unsigned)kMaxImageDim || h > (unsigned)kMaxImageDim || !ch || interlace || (depth != 8 && depth != 16) || ctype == 4) return false;
  if (depth == 16 && ch != 1 && !as_float) return false;
  const size_t bpp = (size_t)ch * depth / 8, stride = (size_t)w * bpp;
  if (z.empty() || (stride + 1) * h > 1032 * z.size() + 64) return false;      // deflate expands at most ~1032x: the header lies about the size
  std::vector<unsigned char> raw((stride + 1) * h);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, z.data(), (uLong)z.size()) != Z_OK || out_len != raw.size()) return false;
  // undo the scanline filters in place (PNG spec 9.2)
  std::vector<unsigned char> prev(stride, 0);
  for (unsigned y = 0; y < h; ++y) {
    unsigned char* line = &raw[y * (stride + 1)];
    const int ft = line[0];
    unsigned char* cur = line + 1;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, up = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int pred = 0;
      switch (ft) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = up; break;
        case 3: pred = (a + up) / 2; break;
        case 4: { const int p = a + up - c, pa = std::abs(p - a), pb = std::abs(p - up), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? up : c); break; }
        default: return false;
      }
      cur[i] = (unsigned char)(cur[i] + pred);
    }
    std::memcpy(prev.data(), cur, stride);
  }
  if (as_float) {
    if (ch != 1) return false;
    img.create((int)h, (int)w, cv::CV_32FC1);
    for (unsigned y = 0; y < h; ++y) {
      const unsigned char* cur = &raw[y * (stride + 1) + 1];
      for (unsigned x = 0; x < w; ++x) img.at<float>((int)y, (int)x) = depth == 16 ? (float)((cur[2 * x] << 8) | cur[2 * x + 1]) : (float)cur[x];
    }
    return true;
  }
  img.create((int)h, (int)w, VDO_CV_MAKETYPE(cv::CV_8U, ch));
  for (unsigned y = 0; y < h; ++y) {
    const unsigned char* cur = &raw[y * (stride + 1) + 1];
    unsigned char* dst = img.data + (size_t)y * img.step;
    if (ch == 1) std::memcpy(dst, cur, w);
    else for (unsigned x = 0; x < w; ++x) {            // RGB(A) -> BGR(A), cv::imread's channel order
      dst[ch * x] = cur[ch * x + 2]; dst[ch * x + 1] = cur[ch * x + 1]; dst[ch * x + 2] = cur[ch * x];
      if (ch == 4) dst[4 * x + 3] = cur[4 * x + 3];
    }
  }
  return true;
}

}  // namespace VDO_SLAM

extern "C" {
// flat hooks for the tests: sizes are returned through dims[0..2] = rows, cols, channels; data is copied into `out` when it is not NULL
int host_io_read_flo(const char* path, int* dims, float* out) try {
  cv::Mat m;
  if (!VDO_SLAM::ReadOpticalFlow(path, m)) return -1;
  dims[0] = m.rows; dims[1] = m.cols; dims[2] = 2;
  if (out) std::memcpy(out, m.data, (size_t)m.rows * m.cols * 8);
  return 0;
} catch (...) { return -2; }
int host_io_load_mask(const char* path, int rows, int cols, int* out) try {
  cv::Mat m(rows, cols, cv::CV_32SC1, out);
  return VDO_SLAM::LoadMask(path, m) ? 0 : -1;
} catch (...) { return -2; }
int host_io_read_png(const char* path, int as_float, int* dims, void* out) try {
  cv::Mat m;
  if (!VDO_SLAM::ReadPNG(path, m, as_float != 0)) return -1;
  dims[0] = m.rows; dims[1] = m.cols; dims[2] = m.channels();
  if (out) std::memcpy(out, m.data, m.step * (size_t)m.rows);
  return 0;
} catch (...) { return -2; }
}
