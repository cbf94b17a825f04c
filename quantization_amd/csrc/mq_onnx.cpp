// ONNX initializer reader: a protobuf wire-format walk over ModelProto -> GraphProto -> TensorProto / NodeProto,
// on memory-mapped files.  Only the fields the analysis flow needs are decoded; everything else is skipped by
// wire type.  Field numbers are those of onnx.proto (ModelProto.graph = 7; GraphProto.node = 1,
// .initializer = 5; NodeProto.input = 1, .output = 2, .op_type = 4; TensorProto.dims = 1, .data_type = 2,
// .name = 8, .raw_data = 9, .external_data = 13, .data_location = 14; StringStringEntryProto.key = 1, .value = 2).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "mq_onnx.h"

namespace {

thread_local std::string g_error;

struct Mapping {
  const uint8_t* base = nullptr;
  size_t size = 0;
};

struct Span {
  const uint8_t* p;
  const uint8_t* end;
};

bool read_varint(Span& s, uint64_t* v) {
  uint64_t r = 0;
  for (int shift = 0; shift < 64 && s.p < s.end; shift += 7) {
    uint8_t b = *s.p++;
    r |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) {
      *v = r;
      return true;
    }
  }
  return false;
}

// Reads one field header; for length-delimited fields `sub` is the payload.  Returns false at the end / on error.
bool next_field(Span& s, uint32_t* field, uint32_t* wire, uint64_t* value, Span* sub, bool* ok) {
  *ok = true;
  if (s.p >= s.end) return false;
  uint64_t key;
  if (!read_varint(s, &key)) { *ok = false; return false; }
  *field = (uint32_t)(key >> 3);
  *wire = (uint32_t)(key & 7);
  switch (*wire) {
    case 0:
      if (!read_varint(s, value)) { *ok = false; return false; }
      return true;
    case 1:
      if (s.end - s.p < 8) { *ok = false; return false; }
      memcpy(value, s.p, 8);
      s.p += 8;
      return true;
    case 2: {
      uint64_t len;
      if (!read_varint(s, &len) || len > (uint64_t)(s.end - s.p)) { *ok = false; return false; }
      sub->p = s.p;
      sub->end = s.p + len;
      s.p += len;
      return true;
    }
    case 5:
      if (s.end - s.p < 4) { *ok = false; return false; }
      *value = 0;
      memcpy(value, s.p, 4);
      s.p += 4;
      return true;
    default:
      *ok = false;  // groups (3, 4) do not occur in onnx.proto
      return false;
  }
}

std::string str_of(const Span& s) { return std::string(reinterpret_cast<const char*>(s.p), (size_t)(s.end - s.p)); }

struct Tensor {
  std::string name;
  int32_t data_type = 0;
  std::vector<int64_t> dims;
  const uint8_t* raw = nullptr;
  uint64_t raw_len = 0;
  bool external = false;
  std::string location;
  uint64_t offset = 0;
  int64_t length = -1;
  bool compute = false;
};

struct Node {
  std::string op;
  std::vector<std::string> in, out;
};

}  // namespace

struct mq_onnx_model {
  std::string dir;
  Mapping file;
  std::map<std::string, Mapping> ext;
  std::vector<Tensor> tensors;
  std::unordered_map<std::string, int64_t> by_name;
};

namespace {

bool map_file(const std::string& path, Mapping* m) {
  int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) {
    g_error = "cannot open " + path;
    return false;
  }
  struct stat st;
  if (fstat(fd, &st) != 0) {
    close(fd);
    g_error = "cannot stat " + path;
    return false;
  }
  m->size = (size_t)st.st_size;
  m->base = nullptr;
  if (m->size) {
    void* p = mmap(nullptr, m->size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) {
      close(fd);
      g_error = "cannot map " + path;
      return false;
    }
    m->base = static_cast<const uint8_t*>(p);
  }
  close(fd);
  return true;
}

bool parse_tensor(Span s, Tensor* t) {
  uint32_t f, w;
  uint64_t v;
  Span sub{};
  bool ok;
  int32_t location = 0;
  while (next_field(s, &f, &w, &v, &sub, &ok)) {
    if (f == 1) {  // dims: repeated int64, packed or one varint per element
      if (w == 2) {
        Span d = sub;
        uint64_t x;
        while (d.p < d.end) {
          if (!read_varint(d, &x)) return false;
          t->dims.push_back((int64_t)x);
        }
      } else if (w == 0) {
        t->dims.push_back((int64_t)v);
      }
    } else if (f == 2 && w == 0) {
      t->data_type = (int32_t)v;
    } else if (f == 8 && w == 2) {
      t->name = str_of(sub);
    } else if (f == 9 && w == 2) {
      t->raw = sub.p;
      t->raw_len = (uint64_t)(sub.end - sub.p);
    } else if (f == 13 && w == 2) {  // StringStringEntryProto
      Span e = sub;
      uint32_t ef, ew;
      uint64_t ev;
      Span es{};
      bool eok;
      std::string key, val;
      while (next_field(e, &ef, &ew, &ev, &es, &eok)) {
        if (ef == 1 && ew == 2) key = str_of(es);
        if (ef == 2 && ew == 2) val = str_of(es);
      }
      if (!eok) return false;
      if (key == "location") t->location = val;
      else if (key == "offset") t->offset = strtoull(val.c_str(), nullptr, 10);
      else if (key == "length") t->length = (int64_t)strtoull(val.c_str(), nullptr, 10);
    } else if (f == 14 && w == 0) {
      location = (int32_t)v;
    }
  }
  if (!ok) return false;
  t->external = location == 1;
  return true;
}

bool parse_node(Span s, Node* n) {
  uint32_t f, w;
  uint64_t v;
  Span sub{};
  bool ok;
  while (next_field(s, &f, &w, &v, &sub, &ok)) {
    if (w != 2) continue;
    if (f == 1) n->in.push_back(str_of(sub));
    else if (f == 2) n->out.push_back(str_of(sub));
    else if (f == 4) n->op = str_of(sub);
  }
  return ok;
}

size_t elem_size(int32_t dt) {
  switch (dt) {
    case MQ_ONNX_FLOAT: case MQ_ONNX_INT32: case 12: return 4;
    case MQ_ONNX_UINT8: case MQ_ONNX_INT8: case 9: return 1;
    case 4: case 5: case MQ_ONNX_FLOAT16: case MQ_ONNX_BFLOAT16: return 2;
    case MQ_ONNX_INT64: case MQ_ONNX_DOUBLE: case 13: return 8;
    default: return 0;
  }
}

}  // namespace

extern "C" const char* mq_onnx_error(void) { return g_error.c_str(); }

extern "C" int mq_onnx_open(const char* model_path, mq_onnx_model** out) {
  if (out) *out = nullptr;
  if (!model_path || !out) {
    g_error = "null argument";
    return -1;
  }
  mq_onnx_model* m = new mq_onnx_model();
  std::string path(model_path);
  size_t slash = path.find_last_of('/');
  m->dir = slash == std::string::npos ? "." : path.substr(0, slash);
  if (!map_file(path, &m->file)) {
    delete m;
    return -1;
  }
  Span model{m->file.base, m->file.base + m->file.size};
  uint32_t f, w;
  uint64_t v;
  Span sub{};
  bool ok = true;
  std::vector<Node> nodes;
  bool have_graph = false;
  while (next_field(model, &f, &w, &v, &sub, &ok)) {
    if (f != 7 || w != 2) continue;  // ModelProto.graph
    have_graph = true;
    Span g = sub;
    uint32_t gf, gw;
    uint64_t gv;
    Span gs{};
    bool gok;
    while (next_field(g, &gf, &gw, &gv, &gs, &gok)) {
      if (gw != 2) continue;
      if (gf == 5) {
        Tensor t;
        if (!parse_tensor(gs, &t)) { gok = false; break; }
        m->tensors.push_back(std::move(t));
      } else if (gf == 1) {
        Node n;
        if (!parse_node(gs, &n)) { gok = false; break; }
        nodes.push_back(std::move(n));
      }
    }
    if (!gok) { ok = false; break; }
  }
  if (!ok || !have_graph) {
    g_error = ok ? "no graph in " + path : "malformed protobuf in " + path;
    mq_onnx_close(m);
    return -1;
  }
  for (size_t i = 0; i < m->tensors.size(); ++i) m->by_name.emplace(m->tensors[i].name, (int64_t)i);
  // find_compute_weights: input[1] of MatMul / Gemm, an initializer itself or behind one Transpose
  std::unordered_map<std::string, const Node*> producer;
  for (const Node& n : nodes)
    for (const std::string& o : n.out) producer[o] = &n;  // later nodes overwrite, as the script's dict does
  for (const Node& n : nodes) {
    if (n.op != "MatMul" && n.op != "Gemm") continue;
    if (n.in.size() < 2) continue;
    auto it = m->by_name.find(n.in[1]);
    if (it != m->by_name.end()) {
      m->tensors[(size_t)it->second].compute = true;
      continue;
    }
    auto pr = producer.find(n.in[1]);
    if (pr != producer.end() && pr->second->op == "Transpose" && !pr->second->in.empty()) {
      auto iw = m->by_name.find(pr->second->in[0]);
      if (iw != m->by_name.end()) m->tensors[(size_t)iw->second].compute = true;
    }
  }
  // external data: map each named file once, bound-check every tensor
  for (Tensor& t : m->tensors) {
    if (!t.external) continue;
    if (t.location.empty() || t.location.find("..") != std::string::npos || t.location[0] == '/') {
      g_error = "tensor " + t.name + ": bad external-data location";
      mq_onnx_close(m);
      return -1;
    }
    auto it = m->ext.find(t.location);
    if (it == m->ext.end()) {
      Mapping e;
      if (!map_file(m->dir + "/" + t.location, &e)) {
        mq_onnx_close(m);
        return -1;
      }
      it = m->ext.emplace(t.location, e).first;
    }
    const Mapping& e = it->second;
    uint64_t len = t.length >= 0 ? (uint64_t)t.length : (t.offset <= e.size ? e.size - t.offset : 0);
    if (t.offset > e.size || len > e.size - t.offset) {
      g_error = "tensor " + t.name + ": external data out of range";
      mq_onnx_close(m);
      return -1;
    }
    t.raw = e.base + t.offset;
    t.raw_len = len;
  }
  // raw_data must hold prod(dims) elements of the declared type (what reshape would check)
  for (const Tensor& t : m->tensors) {
    if (!t.raw) continue;
    size_t es = elem_size(t.data_type);
    uint64_t n = 1;
    for (int64_t d : t.dims) n *= (uint64_t)(d < 0 ? 0 : d);
    if (es && n * es != t.raw_len) {
      g_error = "tensor " + t.name + ": raw_data size does not match dims";
      mq_onnx_close(m);
      return -1;
    }
  }
  *out = m;
  return 0;
}

extern "C" void mq_onnx_close(mq_onnx_model* m) {
  if (!m) return;
  if (m->file.base) munmap(const_cast<uint8_t*>(m->file.base), m->file.size);
  for (auto& kv : m->ext)
    if (kv.second.base) munmap(const_cast<uint8_t*>(kv.second.base), kv.second.size);
  delete m;
}

extern "C" int64_t mq_onnx_num_initializers(const mq_onnx_model* m) { return m ? (int64_t)m->tensors.size() : -1; }

extern "C" int mq_onnx_initializer(const mq_onnx_model* m, int64_t index, mq_onnx_tensor* out) {
  if (!m || !out || index < 0 || index >= (int64_t)m->tensors.size()) {
    g_error = "bad initializer index";
    return -1;
  }
  const Tensor& t = m->tensors[(size_t)index];
  if (t.dims.size() > 8) {
    g_error = "tensor " + t.name + ": more than 8 dimensions";
    return -1;
  }
  memset(out, 0, sizeof(*out));
  out->name = t.name.c_str();
  out->data_type = t.data_type;
  out->ndims = (int32_t)t.dims.size();
  for (size_t i = 0; i < t.dims.size(); ++i) out->dims[i] = t.dims[i];
  out->data = t.raw;
  out->nbytes = t.raw_len;
  out->external = t.external ? 1 : 0;
  out->compute_weight = t.compute ? 1 : 0;
  return 0;
}

extern "C" int64_t mq_onnx_find(const mq_onnx_model* m, const char* name) {
  if (!m || !name) return -1;
  auto it = m->by_name.find(name);
  return it == m->by_name.end() ? -1 : it->second;
}
