// Host-side TFRecord / tf.Example reader and writer (include/tg_io.h).  Plain C++: no HIP calls.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tg_io.h"
#include "tg_common.h"

namespace {

// ---------------------------------------------------------------- CRC-32C, slicing-by-8
struct CrcTables {
  uint32_t t[8][256];
  CrcTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
  }
};
const CrcTables kCrc;

uint32_t crc32c(const uint8_t* p, int64_t n) {
  uint32_t c = 0xFFFFFFFFu;
  while (n > 0 && (reinterpret_cast<uintptr_t>(p) & 7)) { c = kCrc.t[0][(c ^ *p++) & 0xff] ^ (c >> 8); --n; }
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    v ^= c;
    c = kCrc.t[7][v & 0xff] ^ kCrc.t[6][(v >> 8) & 0xff] ^ kCrc.t[5][(v >> 16) & 0xff] ^ kCrc.t[4][(v >> 24) & 0xff] ^
        kCrc.t[3][(v >> 32) & 0xff] ^ kCrc.t[2][(v >> 40) & 0xff] ^ kCrc.t[1][(v >> 48) & 0xff] ^ kCrc.t[0][(v >> 56) & 0xff];
    p += 8; n -= 8;
  }
  while (n-- > 0) c = kCrc.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

inline uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

// ---------------------------------------------------------------- protobuf wire format
struct Span { const uint8_t* p; int64_t n; };

bool read_varint(Span& s, uint64_t* v) {
  uint64_t r = 0;
  for (int shift = 0; shift < 64 && s.n > 0; shift += 7) {
    const uint8_t b = *s.p++; --s.n;
    r |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) { *v = r; return true; }
  }
  return false;
}

// next field of a message: number, wire type, and either the varint value or the length-delimited / fixed payload
bool next_field(Span& s, uint32_t* field, uint32_t* wt, uint64_t* val, Span* sub) {
  uint64_t key;
  if (!read_varint(s, &key)) return false;
  *field = (uint32_t)(key >> 3); *wt = (uint32_t)(key & 7);
  switch (*wt) {
    case 0: return read_varint(s, val);
    case 1: if (s.n < 8) return false; sub->p = s.p; sub->n = 8; s.p += 8; s.n -= 8; return true;
    case 5: if (s.n < 4) return false; sub->p = s.p; sub->n = 4; s.p += 4; s.n -= 4; return true;
    case 2: {
      uint64_t len;
      if (!read_varint(s, &len) || len > (uint64_t)s.n) return false;
      sub->p = s.p; sub->n = (int64_t)len; s.p += len; s.n -= (int64_t)len;
      return true;
    }
    default: return false;
  }
}

void put_varint(std::string& o, uint64_t v) {
  while (v >= 0x80) { o.push_back((char)(v | 0x80)); v >>= 7; }
  o.push_back((char)v);
}
void put_ld(std::string& o, uint32_t field, const void* p, size_t n) {
  put_varint(o, (field << 3) | 2);
  put_varint(o, n);
  o.append(static_cast<const char*>(p), n);
}
void put_ld(std::string& o, uint32_t field, const std::string& s) { put_ld(o, field, s.data(), s.size()); }

std::string feature_entry(const char* key, const std::string& feature) {
  std::string e;
  put_ld(e, 1, key, strlen(key));
  put_ld(e, 2, feature);
  std::string out;
  put_ld(out, 1, e);          // Features.feature map entry
  return out;
}
std::string bytes_feature(const void* p, size_t n) {
  std::string bl, f;
  put_ld(bl, 1, p, n);        // BytesList.value
  put_ld(f, 1, bl);           // Feature.bytes_list
  return f;
}
std::string int64_feature(int64_t v) {
  std::string packed, il, f;
  put_varint(packed, (uint64_t)v);
  put_ld(il, 1, packed);      // Int64List.value, packed
  put_ld(f, 3, il);           // Feature.int64_list
  return f;
}

int parse_example(const uint8_t* rec, int64_t len, const uint8_t** image, int64_t* image_len, int64_t* label, int64_t* height, int64_t* width) {
  bool have[4] = {false, false, false, false};
  Span ex{rec, len};
  uint32_t f, wt; uint64_t v = 0; Span feats{nullptr, 0};
  while (ex.n > 0) {
    if (!next_field(ex, &f, &wt, &v, &feats)) { tg::set_error("tf.Example: malformed message"); return TG_ERR_INVALID; }
    if (f != 1 || wt != 2) continue;                                    // Example.features
    while (feats.n > 0) {
      Span entry{nullptr, 0};
      if (!next_field(feats, &f, &wt, &v, &entry)) { tg::set_error("tf.Example: malformed Features"); return TG_ERR_INVALID; }
      if (f != 1 || wt != 2) continue;                                  // map entry
      Span key{nullptr, 0}, feat{nullptr, 0}, sub{nullptr, 0};
      while (entry.n > 0) {
        if (!next_field(entry, &f, &wt, &v, &sub)) { tg::set_error("tf.Example: malformed map entry"); return TG_ERR_INVALID; }
        if (f == 1 && wt == 2) key = sub;
        else if (f == 2 && wt == 2) feat = sub;
      }
      int which = -1;
      if (key.n == 5 && !memcmp(key.p, "image", 5)) which = 0;
      else if (key.n == 5 && !memcmp(key.p, "label", 5)) which = 1;
      else if (key.n == 6 && !memcmp(key.p, "height", 6)) which = 2;
      else if (key.n == 5 && !memcmp(key.p, "width", 5)) which = 3;
      if (which < 0) continue;
      while (feat.n > 0) {
        Span lst{nullptr, 0};
        if (!next_field(feat, &f, &wt, &v, &lst)) { tg::set_error("tf.Example: malformed Feature"); return TG_ERR_INVALID; }
        if (which == 0 && f == 1 && wt == 2) {                          // bytes_list
          while (lst.n > 0) {
            Span val{nullptr, 0};
            if (!next_field(lst, &f, &wt, &v, &val)) { tg::set_error("tf.Example: malformed BytesList"); return TG_ERR_INVALID; }
            if (f == 1 && wt == 2 && !have[0]) { *image = val.p; *image_len = val.n; have[0] = true; }
          }
        } else if (which > 0 && f == 3 && wt == 2) {                    // int64_list
          while (lst.n > 0) {
            Span val{nullptr, 0}; uint64_t x = 0; bool got = false;
            if (!next_field(lst, &f, &wt, &v, &val)) { tg::set_error("tf.Example: malformed Int64List"); return TG_ERR_INVALID; }
            if (f != 1) continue;
            if (wt == 2) got = read_varint(val, &x);                    // packed: first element
            else if (wt == 0) { x = v; got = true; }
            if (got && !have[which]) {
              have[which] = true;
              (which == 1 ? *label : which == 2 ? *height : *width) = (int64_t)x;
            }
          }
        }
      }
    }
  }
  static const char* names[4] = {"image", "label", "height", "width"};
  for (int i = 0; i < 4; ++i)
    if (!have[i]) { tg::set_error("tf.Example: feature '%s' is missing (FixedLenFeature without default)", names[i]); return TG_ERR_INVALID; }
  return TG_OK;
}

struct Dataset {
  int fd = -1;
  const uint8_t* base = nullptr;
  int64_t bytes = 0;
  std::vector<int64_t> off, len;
  int h = 0, w = 0, c = 0;
};

void destroy(Dataset* d) {
  if (d->base) munmap(const_cast<uint8_t*>(d->base), (size_t)d->bytes);
  if (d->fd >= 0) close(d->fd);
  delete d;
}

}  // namespace

extern "C" {

uint32_t tg_crc32c(const void* data, int64_t n) { return crc32c(static_cast<const uint8_t*>(data), n); }
uint32_t tg_crc32c_masked(const void* data, int64_t n) { return mask_crc(crc32c(static_cast<const uint8_t*>(data), n)); }

int tg_example_parse(const uint8_t* rec, int64_t len, const uint8_t** image, int64_t* image_len, int64_t* label, int64_t* height,
                     int64_t* width) {
  TG_REQUIRE(rec && len >= 0 && image && image_len && label && height && width, "example_parse: null argument");
  return parse_example(rec, len, image, image_len, label, height, width);
}

int tg_tfrecord_write(const char* path, const uint8_t* images, const int64_t* labels, int64_t n, int h, int w, int c, int append) {
  TG_REQUIRE(path && (n == 0 || (images && labels)) && n >= 0 && h > 0 && w > 0 && c > 0, "tfrecord_write: bad arguments");
  FILE* f = fopen(path, append ? "ab" : "wb");
  TG_REQUIRE(f != nullptr, "tfrecord_write: cannot open %s", path);
  const size_t px = (size_t)h * w * c;
  for (int64_t i = 0; i < n; ++i) {
    std::string feats = feature_entry("image", bytes_feature(images + i * px, px)) + feature_entry("label", int64_feature(labels[i])) +
                        feature_entry("height", int64_feature(h)) + feature_entry("width", int64_feature(w));
    std::string ex;
    put_ld(ex, 1, feats);                                              // Example.features
    const uint64_t len = ex.size();
    uint8_t head[12];
    memcpy(head, &len, 8);
    const uint32_t c1 = mask_crc(crc32c(head, 8)), c2 = mask_crc(crc32c(reinterpret_cast<const uint8_t*>(ex.data()), (int64_t)len));
    memcpy(head + 8, &c1, 4);
    if (fwrite(head, 1, 12, f) != 12 || fwrite(ex.data(), 1, len, f) != len || fwrite(&c2, 1, 4, f) != 4) {
      fclose(f);
      tg::set_error("tfrecord_write: short write to %s", path);
      return TG_ERR_INVALID;
    }
  }
  TG_REQUIRE(fclose(f) == 0, "tfrecord_write: close failed for %s", path);
  return TG_OK;
}

int tg_record_append(const char* path, const void* payload, int64_t len, int append) {
  TG_REQUIRE(path && (len == 0 || payload) && len >= 0, "record_append: bad arguments");
  FILE* f = fopen(path, append ? "ab" : "wb");
  TG_REQUIRE(f != nullptr, "record_append: cannot open %s", path);
  const uint64_t n = (uint64_t)len;
  uint8_t head[12];
  memcpy(head, &n, 8);
  const uint32_t c1 = mask_crc(crc32c(head, 8)), c2 = mask_crc(crc32c(static_cast<const uint8_t*>(payload), len));
  memcpy(head + 8, &c1, 4);
  const bool ok = fwrite(head, 1, 12, f) == 12 && (len == 0 || fwrite(payload, 1, (size_t)len, f) == (size_t)len) && fwrite(&c2, 1, 4, f) == 4;
  const bool closed = fclose(f) == 0;
  TG_REQUIRE(ok && closed, "record_append: short write to %s", path);
  return TG_OK;
}

int tg_ds_open(const char* path, void** handle) {
  TG_REQUIRE(path && handle, "ds_open: null argument");
  Dataset* d = new Dataset();
  d->fd = open(path, O_RDONLY);
  if (d->fd < 0) { delete d; tg::set_error("ds_open: cannot open %s", path); return TG_ERR_INVALID; }
  struct stat st;
  if (fstat(d->fd, &st) != 0) { destroy(d); tg::set_error("ds_open: fstat failed for %s", path); return TG_ERR_INVALID; }
  d->bytes = st.st_size;
  if (d->bytes > 0) {
    void* m = mmap(nullptr, (size_t)d->bytes, PROT_READ, MAP_PRIVATE, d->fd, 0);
    if (m == MAP_FAILED) { destroy(d); tg::set_error("ds_open: mmap failed for %s", path); return TG_ERR_INVALID; }
    d->base = static_cast<const uint8_t*>(m);
  }
  auto fail = [&](const char* what, int64_t pos) {
    tg::set_error("ds_open: %s at byte %lld of %s", what, (long long)pos, path);
    destroy(d);
    return TG_ERR_INVALID;
  };
  int64_t pos = 0;
  while (pos < d->bytes) {
    if (pos + 12 > d->bytes) return fail("truncated record header", pos);
    uint64_t len; uint32_t c;
    memcpy(&len, d->base + pos, 8);
    memcpy(&c, d->base + pos + 8, 4);
    if (c != mask_crc(crc32c(d->base + pos, 8))) return fail("length CRC mismatch", pos);
    if (len > (uint64_t)d->bytes || pos + 16 + (int64_t)len > d->bytes) return fail("truncated record payload", pos);
    memcpy(&c, d->base + pos + 12 + len, 4);
    if (c != mask_crc(crc32c(d->base + pos + 12, (int64_t)len))) return fail("payload CRC mismatch", pos);
    d->off.push_back(pos + 12);
    d->len.push_back((int64_t)len);
    pos += 16 + (int64_t)len;
  }
  if (!d->off.empty()) {
    const uint8_t* img = nullptr; int64_t il = 0, lab = 0, hh = 0, ww = 0;
    int rc = parse_example(d->base + d->off[0], d->len[0], &img, &il, &lab, &hh, &ww);
    if (rc != TG_OK) { destroy(d); return rc; }
    // height / width come from the file: bound them before multiplying (2^32 x 2^32 would wrap to 0 and divide by zero below)
    constexpr int64_t kMaxSide = 1 << 16;
    if (hh <= 0 || ww <= 0 || hh > kMaxSide || ww > kMaxSide || il <= 0 || hh * ww > il || il % (hh * ww) != 0)
      return fail("record 0: image bytes do not match height*width", d->off[0]);
    d->h = (int)hh; d->w = (int)ww; d->c = (int)(il / (hh * ww));
  }
  *handle = d;
  return TG_OK;
}

int64_t tg_ds_size(void* handle) { return handle ? (int64_t) static_cast<Dataset*>(handle)->off.size() : -1; }

int tg_ds_shape(void* handle, int* h, int* w, int* c) {
  TG_REQUIRE(handle && h && w && c, "ds_shape: null argument");
  Dataset* d = static_cast<Dataset*>(handle);
  TG_REQUIRE(!d->off.empty(), "ds_shape: empty dataset");
  *h = d->h; *w = d->w; *c = d->c;
  return TG_OK;
}

int tg_ds_record(void* handle, int64_t i, const uint8_t** payload, int64_t* len) {
  TG_REQUIRE(handle && payload && len, "ds_record: null argument");
  Dataset* d = static_cast<Dataset*>(handle);
  TG_REQUIRE(i >= 0 && i < (int64_t)d->off.size(), "ds_record: index %lld out of range [0,%lld)", (long long)i, (long long)d->off.size());
  *payload = d->base + d->off[i];
  *len = d->len[i];
  return TG_OK;
}

int tg_ds_gather(void* handle, const int64_t* idx, int64_t n, uint8_t* images, int32_t* labels, int n_threads) {
  TG_REQUIRE(handle && (n == 0 || (idx && images && labels)) && n >= 0, "ds_gather: bad arguments");
  Dataset* d = static_cast<Dataset*>(handle);
  const int64_t size = (int64_t)d->off.size();
  for (int64_t i = 0; i < n; ++i)
    TG_REQUIRE(idx[i] >= 0 && idx[i] < size, "ds_gather: index %lld out of range [0,%lld)", (long long)idx[i], (long long)size);
  const int64_t px = (int64_t)d->h * d->w * d->c;
  const int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
  std::vector<std::string> errs(nt);
  auto work = [&](int t) {
    for (int64_t i = t; i < n; i += nt) {
      const uint8_t* img = nullptr; int64_t il = 0, lab = 0, hh = 0, ww = 0;
      if (parse_example(d->base + d->off[idx[i]], d->len[idx[i]], &img, &il, &lab, &hh, &ww) != TG_OK) {
        errs[t] = std::string("record ") + std::to_string(idx[i]) + ": " + tg_last_error_string();
        return;
      }
      if (hh != d->h || ww != d->w || il != px) {
        errs[t] = "record " + std::to_string(idx[i]) + ": geometry " + std::to_string(hh) + "x" + std::to_string(ww) + " (" + std::to_string(il) +
                  " bytes) differs from record 0";
        return;
      }
      memcpy(images + i * px, img, (size_t)px);
      labels[i] = (int32_t)lab;
    }
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (const std::string& e : errs)
    if (!e.empty()) { tg::set_error("ds_gather: %s", e.c_str()); return TG_ERR_INVALID; }
  return TG_OK;
}

int tg_ds_close(void* handle) {
  if (handle) destroy(static_cast<Dataset*>(handle));
  return TG_OK;
}

}  // extern "C"
