// Host-side PNG writer for the projection stage's file tree (no GPU work in this file).
//
// FruitModel.get_outputs_for_projections writes two images per (super-cluster, camera, sub-cluster) job with
// torchvision.utils.save_image (crop_nerf/fruit_nerf/fruit_nerf.py:304,315): 8-bit RGB PNGs of a frame that is black except
// for the pixels whose rays hit the job's box, all three channels equal.  The merger reads them back with OpenCV
// (segmentation/merger.py:222,250): what has to agree is the decoded pixels, not the compressed bytes.
//
// cn_png_write_gray_rects writes a batch of such files from the compact form the projection kernels leave (one byte per
// pixel of the job's screen rectangle).  The rows above and below a rectangle are all zero; their deflate streams are
// assembled from cached pieces of 2^k rows and spliced in -- every piece ends with a full flush, so pieces concatenate -- and the
// Adler-32 of a run of zeros is closed-form, so a file costs the rectangle's rows, not the frame's.  The call holds no
// Python state: worker threads enter it through ctypes with the GIL released (a Python encoder of the same stream spends
// ~150 us per file under the GIL, which caps 16 threads at ~5 000 files/s; this one scales with the threads).
#include <sys/stat.h>
#include <zlib.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/cropnerf_hip.h"

namespace cn {
void set_error(const char* fmt, ...);
}

namespace {

using Bytes = std::vector<unsigned char>;

// raw deflate of `data`, ended by a full flush (byte-aligned, history reset): appended to `out`
bool deflate_piece(const unsigned char* data, size_t n, Bytes& out) {
  z_stream zs;
  std::memset(&zs, 0, sizeof zs);
  if (deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  const size_t bound = deflateBound(&zs, (uLong)n) + 16;
  const size_t at = out.size();
  out.resize(at + bound);
  zs.next_in = const_cast<unsigned char*>(data);
  zs.avail_in = (uInt)n;
  zs.next_out = out.data() + at;
  zs.avail_out = (uInt)bound;
  const int rc = deflate(&zs, Z_FULL_FLUSH);
  const bool ok = (rc == Z_OK || rc == Z_BUF_ERROR) && zs.avail_in == 0 && zs.avail_out > 0;
  out.resize(at + (bound - zs.avail_out));
  deflateEnd(&zs);
  return ok;
}

std::mutex g_band_mutex;
std::map<std::pair<size_t, size_t>, Bytes> g_zero_bands;  // (row bytes, 2^k rows) -> deflate piece of that many zero bytes

// `rows` all-zero rows as the concatenation of cached pieces of 2^k rows, largest first (one piece per set bit of `rows`):
// a frame has up to `height` different band heights above and as many below its rectangles, but only log2(height) pieces
bool zero_band(size_t row_bytes, size_t rows, Bytes& out) {
  for (int k = 30; k >= 0; --k) {
    const size_t n = (size_t)1 << k;
    if (!(rows & n)) continue;
    {
      std::lock_guard<std::mutex> lock(g_band_mutex);
      auto it = g_zero_bands.find({row_bytes, n});
      if (it != g_zero_bands.end()) {
        out.insert(out.end(), it->second.begin(), it->second.end());
        continue;
      }
    }
    Bytes zeros(row_bytes * n, 0), piece;
    if (!deflate_piece(zeros.data(), zeros.size(), piece)) return false;
    out.insert(out.end(), piece.begin(), piece.end());
    std::lock_guard<std::mutex> lock(g_band_mutex);
    g_zero_bands.emplace(std::make_pair(row_bytes, n), std::move(piece));
  }
  return true;
}

// Adler-32 after n more zero bytes: a unchanged, b += n * a (mod 65521)
uint32_t adler_zeros(uint32_t adler, uint64_t n) {
  const uint64_t a = adler & 0xFFFFu, b = adler >> 16;
  return (uint32_t)(((b + (n % 65521u) * a) % 65521u) << 16 | a);
}

void put_u32(Bytes& v, uint32_t x) {
  v.push_back((unsigned char)(x >> 24));
  v.push_back((unsigned char)(x >> 16));
  v.push_back((unsigned char)(x >> 8));
  v.push_back((unsigned char)x);
}

void put_chunk(Bytes& file, const char tag[4], const unsigned char* data, size_t n) {
  put_u32(file, (uint32_t)n);
  const size_t at = file.size();
  file.insert(file.end(), tag, tag + 4);
  file.insert(file.end(), data, data + n);
  put_u32(file, (uint32_t)crc32(0L, file.data() + at, (uInt)(n + 4)));
}

bool encode(const uint8_t* crop, int x0, int y0, int w, int h, int height, int width, Bytes& file, Bytes& mid, Bytes& idat) {
  const size_t row_bytes = 1 + 3 * (size_t)width;  // filter type 0 + RGB
  const size_t below = (size_t)(height - y0 - h);
  mid.assign((size_t)h * row_bytes, 0);
  for (int r = 0; r < h; ++r) {
    unsigned char* row = mid.data() + (size_t)r * row_bytes + 1 + 3 * (size_t)x0;
    const uint8_t* src = crop + (size_t)r * w;
    for (int c = 0; c < w; ++c) row[3 * c] = row[3 * c + 1] = row[3 * c + 2] = src[c];
  }
  uint32_t adler = adler_zeros(1u, row_bytes * (uint64_t)y0);
  if (!mid.empty()) adler = (uint32_t)adler32(adler, mid.data(), (uInt)mid.size());
  adler = adler_zeros(adler, row_bytes * (uint64_t)below);
  idat.clear();
  idat.push_back(0x78);
  idat.push_back(0x01);
  if (y0 > 0 && !zero_band(row_bytes, (size_t)y0, idat)) return false;
  if (h > 0 && !deflate_piece(mid.data(), mid.size(), idat)) return false;
  if (below > 0 && !zero_band(row_bytes, below, idat)) return false;
  static const unsigned char final_block[5] = {0x01, 0x00, 0x00, 0xFF, 0xFF};  // final, empty stored block
  idat.insert(idat.end(), final_block, final_block + 5);
  put_u32(idat, adler);
  static const unsigned char magic[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
  file.clear();
  file.insert(file.end(), magic, magic + 8);
  Bytes ihdr;
  put_u32(ihdr, (uint32_t)width);
  put_u32(ihdr, (uint32_t)height);
  const unsigned char tail[5] = {8, 2, 0, 0, 0};  // 8 bits, colour type 2 (RGB), deflate, adaptive filtering, no interlace
  ihdr.insert(ihdr.end(), tail, tail + 5);
  put_chunk(file, "IHDR", ihdr.data(), ihdr.size());
  put_chunk(file, "IDAT", idat.data(), idat.size());
  put_chunk(file, "IEND", nullptr, 0);
  return true;
}

// mkdir -p of the directory part of `path`
bool make_parent_dirs(const char* path) {
  std::string p(path);
  const size_t last = p.find_last_of('/');
  if (last == std::string::npos || last == 0) return true;
  p.resize(last);
  struct stat st;
  if (stat(p.c_str(), &st) == 0) return S_ISDIR(st.st_mode);
  for (size_t i = 1; i <= p.size(); ++i) {
    if (i != p.size() && p[i] != '/') continue;
    const std::string sub = p.substr(0, i);
    if (mkdir(sub.c_str(), 0777) != 0 && errno != EEXIST) return false;
  }
  return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

}  // namespace

extern "C" int cn_png_write_gray_rects(int32_t count, const char* const* paths, const uint8_t* values,
                                       const int64_t* value_offsets, const int32_t* rects, int32_t image_height,
                                       int32_t image_width, int32_t make_dirs) {
  if (count <= 0) return CN_OK;
  if (!paths || !value_offsets || !rects || image_height <= 0 || image_width <= 0) {
    cn::set_error("cn_png_write_gray_rects: bad argument");
    return CN_ERR_INVALID;
  }
  Bytes file, mid, idat;
  for (int32_t i = 0; i < count; ++i) {
    const int x0 = rects[4 * i], y0 = rects[4 * i + 1], w = rects[4 * i + 2], h = rects[4 * i + 3];
    const bool empty = w <= 0 || h <= 0;
    if (!empty && (x0 < 0 || y0 < 0 || x0 + w > image_width || y0 + h > image_height || !values)) {
      cn::set_error("cn_png_write_gray_rects: rectangle %d (%d, %d, %d, %d) outside the %d x %d frame", i, x0, y0, w, h,
                    image_height, image_width);
      return CN_ERR_INVALID;
    }
    if (!encode(empty ? nullptr : values + value_offsets[i], empty ? 0 : x0, empty ? 0 : y0, empty ? 0 : w, empty ? 0 : h,
                image_height, image_width, file, mid, idat)) {
      cn::set_error("cn_png_write_gray_rects: deflate failed for %s", paths[i]);
      return CN_ERR_LAUNCH;
    }
    if (make_dirs && !make_parent_dirs(paths[i])) {
      cn::set_error("cn_png_write_gray_rects: cannot create the directory of %s: %s", paths[i], std::strerror(errno));
      return CN_ERR_INVALID;
    }
    FILE* f = std::fopen(paths[i], "wb");
    if (!f) {
      cn::set_error("cn_png_write_gray_rects: cannot open %s: %s", paths[i], std::strerror(errno));
      return CN_ERR_INVALID;
    }
    const size_t wrote = std::fwrite(file.data(), 1, file.size(), f);
    const int closed = std::fclose(f);
    if (wrote != file.size() || closed != 0) {
      cn::set_error("cn_png_write_gray_rects: short write to %s", paths[i]);
      return CN_ERR_INVALID;
    }
  }
  return CN_OK;
}
