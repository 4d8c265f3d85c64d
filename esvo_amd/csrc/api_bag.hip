// api_bag.hip — rosbag v2 ingest: host code, no kernels.
//
// Replaces the reading side of events_repacking_helper (events_repacking_helper/src/EventMessageEditor.cpp:104-119:
// rosbag::Bag::open + rosbag::View + MessageInstance::instantiate<dvs_msgs::EventArray>) for the ingest path: the bag's
// chunk records are walked in file order and every serialised dvs_msgs/EventArray message of the wanted topic is handed
// out as a byte range -- exactly what esvo_ts_push_event_array stages (13-byte wire records, widened on the device), so
// no std::vector<dvs_msgs::Event> is ever materialised.  The ROS bag format 2.0 is public (wiki.ros.org/Bags/Format/2.0):
//   "#ROSBAG V2.0\n", then records  <u32 header_len><fields: u32 len, "name=value">...<u32 data_len><data>
//   op 0x03 bag header, 0x05 chunk (compression none | bz2 | lz4, size), 0x07 connection (conn, topic; data: type, ...),
//   0x02 message data (conn, time), 0x04 index data, 0x06 chunk info.
// Chunks are decompressed with the system's libbz2 / liblz4 (dlopen; rosbag's lz4 chunks are LZ4 frames).
#include <dlfcn.h>

#include <cstdio>
#include <map>

#include "context.hpp"

namespace {
struct Field { const uint8_t* p; uint32_t n; };
typedef std::map<std::string, Field> Fields;

inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

bool parse_fields(const uint8_t* p, size_t n, Fields& out) {
  out.clear();
  size_t o = 0;
  while (o + 4 <= n) {
    const uint32_t len = rd32(p + o);
    o += 4;
    if (len == 0 || o + len > n) return false;
    const uint8_t* f = p + o;
    size_t eq = 0;
    while (eq < len && f[eq] != '=') ++eq;
    if (eq == len) return false;
    out[std::string(reinterpret_cast<const char*>(f), eq)] = Field{f + eq + 1, (uint32_t)(len - eq - 1)};
    o += len;
  }
  return o == n;
}
inline bool field_u8(const Fields& f, const char* k, uint8_t& v) {
  auto it = f.find(k);
  if (it == f.end() || it->second.n < 1) return false;
  v = it->second.p[0];
  return true;
}
inline bool field_u32(const Fields& f, const char* k, uint32_t& v) {
  auto it = f.find(k);
  if (it == f.end() || it->second.n < 4) return false;
  v = rd32(it->second.p);
  return true;
}
inline std::string field_str(const Fields& f, const char* k) {
  auto it = f.find(k);
  return it == f.end() ? std::string() : std::string(reinterpret_cast<const char*>(it->second.p), it->second.n);
}

typedef int (*bz2_fn)(char*, unsigned int*, char*, unsigned int, int, int);
typedef int (*lz4_fn)(const char*, char*, int, int);
typedef int (*lz4_dict_fn)(const char*, char*, int, int, const char*, int);

// LZ4 frame (magic 0x184D2204) -> bytes; blocks may be stored, independent or linked
bool lz4_frame_decode(const uint8_t* src, size_t n, std::vector<uint8_t>& dst, size_t expect, std::string& why) {
  static lz4_fn dec = nullptr;
  static lz4_dict_fn dec_dict = nullptr;
  if (!dec) {
    void* lib = dlopen("liblz4.so.1", RTLD_NOW);
    if (!lib) lib = dlopen("liblz4.so", RTLD_NOW);
    if (lib) {
      dec = reinterpret_cast<lz4_fn>(dlsym(lib, "LZ4_decompress_safe"));
      dec_dict = reinterpret_cast<lz4_dict_fn>(dlsym(lib, "LZ4_decompress_safe_usingDict"));
    }
    if (!dec || !dec_dict) { why = "liblz4 not available for an lz4-compressed chunk"; return false; }
  }
  if (n < 7 || rd32(src) != 0x184D2204u) { why = "lz4 chunk is not an LZ4 frame"; return false; }
  const uint8_t flg = src[4], bd = src[5];
  if ((flg >> 6) != 1) { why = "unsupported LZ4 frame version"; return false; }
  const bool independent = flg & 0x20, block_checksum = flg & 0x10, has_size = flg & 0x08, has_dict = flg & 0x01;
  static const size_t bmax_tab[8] = {0, 0, 0, 0, 64u << 10, 256u << 10, 1u << 20, 4u << 20};
  const size_t bmax = bmax_tab[(bd >> 4) & 7];
  if (!bmax) { why = "bad LZ4 block size"; return false; }
  size_t o = 6 + (has_size ? 8 : 0) + (has_dict ? 4 : 0) + 1;  // + header checksum byte
  dst.clear();
  dst.reserve(expect);
  while (o + 4 <= n) {
    const uint32_t bs = rd32(src + o);
    o += 4;
    if (bs == 0) break;  // end mark
    const uint32_t len = bs & 0x7fffffffu;
    if (o + len > n) { why = "truncated LZ4 block"; return false; }
    if (bs & 0x80000000u) {
      dst.insert(dst.end(), src + o, src + o + len);
    } else {
      const size_t at = dst.size();
      dst.resize(at + bmax);
      int got;
      if (independent || at == 0) got = dec(reinterpret_cast<const char*>(src + o), reinterpret_cast<char*>(dst.data() + at), (int)len, (int)bmax);
      else {
        const size_t dict = at < (64u << 10) ? at : (64u << 10);
        got = dec_dict(reinterpret_cast<const char*>(src + o), reinterpret_cast<char*>(dst.data() + at), (int)len, (int)bmax,
                       reinterpret_cast<const char*>(dst.data() + at - dict), (int)dict);
      }
      if (got < 0) { why = "LZ4 block does not decode"; return false; }
      dst.resize(at + (size_t)got);
    }
    o += len + (block_checksum ? 4 : 0);
  }
  return true;
}
bool bz2_decode(const uint8_t* src, size_t n, std::vector<uint8_t>& dst, size_t expect, std::string& why) {
  static bz2_fn dec = nullptr;
  if (!dec) {
    void* lib = dlopen("libbz2.so.1.0", RTLD_NOW);
    if (!lib) lib = dlopen("libbz2.so.1", RTLD_NOW);
    if (lib) dec = reinterpret_cast<bz2_fn>(dlsym(lib, "BZ2_bzBuffToBuffDecompress"));
    if (!dec) { why = "libbz2 not available for a bz2-compressed chunk"; return false; }
  }
  dst.resize(expect ? expect : 1);
  unsigned int out_len = (unsigned int)dst.size();
  const int rc = dec(reinterpret_cast<char*>(dst.data()), &out_len, const_cast<char*>(reinterpret_cast<const char*>(src)), (unsigned int)n, 0, 0);
  if (rc != 0) { why = "bz2 chunk does not decode"; return false; }
  dst.resize(out_len);
  return true;
}
}  // namespace

struct esvo_bag {
  FILE* f = nullptr;
  uint64_t file_size = 0;         // every length read from the file is bounded by what the file can still hold
  std::string err;
  std::map<uint32_t, std::pair<std::string, std::string>> conns;  // conn id -> (topic, type)
  std::vector<uint8_t> rec_header, rec_data, chunk;  // current record / decompressed chunk
  const uint8_t* cur = nullptr;   // walking position inside the chunk
  size_t cur_left = 0;
  std::string topic_out;
  uint32_t chunk_count = 0, conn_count = 0;
  uint64_t chunk_serial = 0;      // chunks loaded so far

  // read one top-level record: header fields + data (into rec_header / rec_data)
  int read_record(Fields& fields) {
    uint8_t len4[4];
    if (fread(len4, 1, 4, f) != 4) return 1;  // end of file
    const uint32_t hl = rd32(len4);
    if (hl > (64u << 20)) { err = "bag record header too large (corrupt file?)"; return -1; }
    rec_header.resize(hl);
    if (hl && fread(rec_header.data(), 1, hl, f) != hl) { err = "truncated bag record header"; return -1; }
    if (fread(len4, 1, 4, f) != 4) { err = "truncated bag record"; return -1; }
    const uint32_t dl = rd32(len4);
    {
      const long at = ftell(f);
      if (at < 0 || (uint64_t)at + dl > file_size) { err = "bag record data longer than the file (truncated or corrupt)"; return -1; }
    }
    rec_data.resize(dl);
    if (dl && fread(rec_data.data(), 1, dl, f) != dl) { err = "truncated bag record data"; return -1; }
    if (!parse_fields(rec_header.data(), hl, fields)) { err = "malformed bag record header"; return -1; }
    return 0;
  }
  void note_connection(const Fields& hdr, const uint8_t* data, size_t n) {
    uint32_t id = 0;
    if (!field_u32(hdr, "conn", id)) return;
    Fields ch;
    parse_fields(data, n, ch);
    conns[id] = std::make_pair(field_str(hdr, "topic"), field_str(ch, "type"));
  }
};

extern "C" {

int esvo_bag_open(const char* path, esvo_bag_handle* out) {
  esvo_context* h = nullptr;
  if (!path || !out) return ESVO_ERR_INVALID_ARG;
  FILE* f = fopen(path, "rb");
  if (!f) FAIL(ESVO_ERR_INVALID_ARG, std::string("cannot open bag file ") + path);
  char magic[13];
  if (fread(magic, 1, 13, f) != 13 || std::memcmp(magic, "#ROSBAG V2.0\n", 13) != 0) {
    fclose(f);
    FAIL(ESVO_ERR_UNSUPPORTED, "not a rosbag format 2.0 file");
  }
  esvo_bag* b = new esvo_bag();
  b->f = f;
  {
    const long at = ftell(f);
    fseek(f, 0, SEEK_END);
    const long end = ftell(f);
    fseek(f, at, SEEK_SET);
    b->file_size = end > 0 ? (uint64_t)end : 0;
  }
  Fields fields;
  uint8_t op = 0;
  if (b->read_record(fields) != 0 || !field_u8(fields, "op", op) || op != 0x03) {  // bag header record (padded to 4096 B)
    std::string e = b->err.empty() ? "bag header record missing" : b->err;
    fclose(f);
    delete b;
    FAIL(ESVO_ERR_UNSUPPORTED, e);
  }
  field_u32(fields, "chunk_count", b->chunk_count);
  field_u32(fields, "conn_count", b->conn_count);
  *out = b;
  return ESVO_OK;
}

int esvo_bag_close(esvo_bag_handle b) {
  if (!b) return ESVO_OK;
  if (b->f) fclose(b->f);
  delete b;
  return ESVO_OK;
}

const char* esvo_bag_last_error(esvo_bag_handle b) { return b ? b->err.c_str() : ""; }

static int bag_next_event_array(esvo_bag_handle b, const char* topic, const uint8_t** msg, size_t* n_bytes, uint64_t* stamp_ns,
                                const char** topic_out);
// no exception crosses the C boundary: an allocation a corrupt length field asked for becomes an error code
int esvo_bag_next_event_array(esvo_bag_handle b, const char* topic, const uint8_t** msg, size_t* n_bytes, uint64_t* stamp_ns,
                              const char** topic_out) {
  if (!b || !msg || !n_bytes) return ESVO_ERR_INVALID_ARG;
  try {
    return bag_next_event_array(b, topic, msg, n_bytes, stamp_ns, topic_out);
  } catch (const std::exception& e) {
    b->err = std::string("bag read failed: ") + e.what();
    return ESVO_ERR_UNSUPPORTED;
  }
}
static int bag_next_event_array(esvo_bag_handle b, const char* topic, const uint8_t** msg, size_t* n_bytes, uint64_t* stamp_ns,
                                const char** topic_out) {
  Fields fields;
  while (true) {
    // ---- records inside the current chunk ----
    while (b->cur_left >= 8) {
      const uint32_t hl = rd32(b->cur);
      if ((size_t)hl + 8 > b->cur_left) { b->err = "malformed record inside a chunk"; return ESVO_ERR_UNSUPPORTED; }
      const uint8_t* hp = b->cur + 4;
      const uint32_t dl = rd32(hp + hl);
      if ((size_t)hl + 8 + dl > b->cur_left) { b->err = "truncated record inside a chunk"; return ESVO_ERR_UNSUPPORTED; }
      const uint8_t* dp = hp + hl + 4;
      b->cur += (size_t)hl + 8 + dl;
      b->cur_left -= (size_t)hl + 8 + dl;
      if (!parse_fields(hp, hl, fields)) { b->err = "malformed record header inside a chunk"; return ESVO_ERR_UNSUPPORTED; }
      uint8_t op = 0;
      field_u8(fields, "op", op);
      if (op == 0x07) { b->note_connection(fields, dp, dl); continue; }
      if (op != 0x02) continue;
      uint32_t conn = 0;
      if (!field_u32(fields, "conn", conn)) continue;
      auto it = b->conns.find(conn);
      if (it == b->conns.end() || it->second.second != "dvs_msgs/EventArray") continue;
      if (topic && it->second.first != topic) continue;
      if (stamp_ns) {
        auto t = fields.find("time");
        *stamp_ns = (t != fields.end() && t->second.n >= 8) ? (uint64_t)rd32(t->second.p) * 1000000000ull + rd32(t->second.p + 4) : 0;
      }
      b->topic_out = it->second.first;
      if (topic_out) *topic_out = b->topic_out.c_str();
      *msg = dp;
      *n_bytes = dl;
      return ESVO_OK;
    }
    // ---- next top-level record ----
    const int rc = b->read_record(fields);
    if (rc > 0) return 1;  // end of bag
    if (rc < 0) return ESVO_ERR_UNSUPPORTED;
    uint8_t op = 0;
    field_u8(fields, "op", op);
    if (op == 0x07) { b->note_connection(fields, b->rec_data.data(), b->rec_data.size()); continue; }
    if (op != 0x05) continue;  // index data, chunk infos, ...
    const std::string comp = field_str(fields, "compression");
    uint32_t size = 0;
    field_u32(fields, "size", size);
    if (size > (1u << 30)) { b->err = "chunk claims more than 1 GiB of uncompressed data (corrupt file?)"; return ESVO_ERR_UNSUPPORTED; }
    if (comp == "none") {
      b->chunk.swap(b->rec_data);
    } else if (comp == "bz2") {
      if (!bz2_decode(b->rec_data.data(), b->rec_data.size(), b->chunk, size, b->err)) return ESVO_ERR_UNSUPPORTED;
    } else if (comp == "lz4") {
      if (!lz4_frame_decode(b->rec_data.data(), b->rec_data.size(), b->chunk, size, b->err)) return ESVO_ERR_UNSUPPORTED;
    } else {
      b->err = "unknown chunk compression '" + comp + "'";
      return ESVO_ERR_UNSUPPORTED;
    }
    if (size && b->chunk.size() != size) { b->err = "chunk size does not match its header"; return ESVO_ERR_UNSUPPORTED; }
    b->cur = b->chunk.data();
    b->cur_left = b->chunk.size();
    b->chunk_serial++;
  }
}

// stage every dvs_msgs/EventArray message of `topic` whose bag time stamp is below until_ns (0: the whole bag)
int esvo_ts_push_bag(esvo_handle h, int cam, esvo_bag_handle b, const char* topic, uint64_t until_ns, size_t* n_events) {
  if (!h || !b) return ESVO_ERR_INVALID_ARG;
  size_t total = 0;
  int status = ESVO_OK;
  while (true) {
    const uint8_t* msg = nullptr;
    size_t nb = 0;
    uint64_t stamp = 0;
    const uint64_t serial0 = b->chunk_serial;
    const uint8_t* cur0 = b->cur;
    const size_t left0 = b->cur_left;
    int rc = esvo_bag_next_event_array(b, topic, &msg, &nb, &stamp, nullptr);
    if (rc == 1) break;
    if (rc) { g_create_error = "bag: " + b->err; status = rc; break; }
    bool step_back = until_ns && stamp >= until_ns;  // not yet: the next call must find this message again
    if (!step_back) {
      size_t n = 0;
      rc = esvo_ts_push_event_array(h, cam, msg, nb, &n);
      // a refused message ("event ring full: render before staging more" is ordinary flow control) is not consumed either:
      // the caller renders and calls again, and no EventArray is lost
      if (rc) { status = rc; step_back = true; }
      else total += n;
    }
    if (step_back) {
      if (b->chunk_serial == serial0) { b->cur = cur0; b->cur_left = left0; }     // same chunk: rewind the walk
      else { b->cur = b->chunk.data(); b->cur_left = b->chunk.size(); }           // it is the first match of a new chunk
      break;
    }
  }
  if (n_events) *n_events = total;  // also on failure: what was staged before it
  return status;
}

}  // extern "C"
