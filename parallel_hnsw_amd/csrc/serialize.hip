// On-disk interchange with the Rust crate: serialize_hnsw / deserialize_hnsw
// (/root/reference/src/serialize.rs:33-209).  Host-side I/O only.
//
//   <dir>/meta               serde_json of HNSWMeta { layer_count, build_parameters }   :28-31,52-57
//   <dir>/comparator/        the Comparator's own files (user defined in the crate)     :59-64,144-147
//   <dir>/layer.meta.N       serde_json of LayerMeta { node_count, neighborhood_size }  :21-25,80-85
//   <dir>/layer.nodes.N      node_count raw native-endian usize VectorIds               :97-103
//   <dir>/layer.neighbors.N  node_count*neighborhood_size raw usize NodeIds, !0 = empty :114-121
// with N counted from the BOTTOM (layer_number = layer_count - i - 1, :67).  The comparator
// directory written here holds this library's store (meta.json + vectors.f32); a crate user
// substitutes their own Serializable comparator.  A missing comparator entry means
// "index not found" (:144-146).
#include <sys/stat.h>

#include <charconv>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "phnsw_internal.h"

// serde_json prints f32 with the shortest round-trip digits and always a ".0" on integers
static std::string json_f32(float f) {
  char buf[64];
  auto r = std::to_chars(buf, buf + sizeof(buf), f);
  std::string s(buf, r.ptr);
  size_t e = s.find('e');
  if (e != std::string::npos) {  // 1e-05 -> 1e-5
    std::string mant = s.substr(0, e), ex = s.substr(e + 1);
    bool neg = !ex.empty() && ex[0] == '-';
    if (!ex.empty() && (ex[0] == '-' || ex[0] == '+')) ex = ex.substr(1);
    while (ex.size() > 1 && ex[0] == '0') ex = ex.substr(1);
    return mant + "e" + (neg ? "-" : "") + ex;
  }
  if (s.find('.') == std::string::npos && s.find("inf") == std::string::npos && s.find("nan") == std::string::npos)
    s += ".0";
  return s;
}

static std::string json_sp(const phnsw_search_params &p) {
  std::ostringstream o;
  o << "{\"number_of_candidates\":" << p.number_of_candidates
    << ",\"upper_layer_candidate_count\":" << p.upper_layer_candidate_count << ",\"probe_depth\":" << p.probe_depth
    << "}";
  return o.str();
}

// field order = struct declaration order (parameters.rs:3-64)
static std::string json_bp(const phnsw_build_params &b) {
  std::ostringstream o;
  o << "{\"order\":" << b.order << ",\"zero_layer_neighborhood_size\":" << b.zero_layer_neighborhood_size
    << ",\"neighborhood_size\":" << b.neighborhood_size << ",\"optimization\":{\"promotion_threshold\":"
    << json_f32(b.optimization.promotion_threshold)
    << ",\"neighborhood_threshold\":" << json_f32(b.optimization.neighborhood_threshold)
    << ",\"recall_proportion\":" << json_f32(b.optimization.recall_proportion)
    << ",\"promotion_proportion\":" << json_f32(b.optimization.promotion_proportion)
    << ",\"search\":" << json_sp(b.optimization.search)
    << "},\"initial_partition_search\":" << json_sp(b.initial_partition_search) << "}";
  return o.str();
}

static int write_file(const std::string &path, const void *data, size_t bytes) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) {
    ph_set_error("cannot open %s for writing", path.c_str());
    return PHNSW_E_INVALID;
  }
  size_t w = bytes ? fwrite(data, 1, bytes, f) : 0;
  const int closed = fclose(f);  // a write error may only surface when the buffer is flushed
  if (w != bytes || closed != 0) {
    ph_set_error("short write to %s", path.c_str());
    return PHNSW_E_INVALID;
  }
  return 0;
}

// std::fs::create_dir_all  serialize.rs:41-47
static void create_dir_all(const std::string &dir) {
  for (size_t i = 1; i <= dir.size(); i++)
    if (i == dir.size() || dir[i] == '/') mkdir(dir.substr(0, i).c_str(), 0777);
}

static int read_file(const std::string &path, std::string &out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    ph_set_error("cannot open %s", path.c_str());
    return PHNSW_E_INVALID;
  }
  std::ostringstream ss;
  ss << f.rdbuf();
  out = ss.str();
  return 0;
}

static bool exists(const std::string &p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0;
}

// the value that follows "key": in a serde_json object (numbers only; keys here are unique
// within the sub-object that starts at `from`)
static bool json_number(const std::string &s, const std::string &key, size_t from, double *out, size_t *pos = nullptr) {
  size_t k = s.find("\"" + key + "\"", from);
  if (k == std::string::npos) return false;
  size_t c = s.find(':', k);
  if (c == std::string::npos) return false;
  c++;
  while (c < s.size() && (s[c] == ' ' || s[c] == '\n' || s[c] == '\t')) c++;
  char *end = nullptr;
  double v = strtod(s.c_str() + c, &end);
  if (end == s.c_str() + c) return false;
  *out = v;
  if (pos) *pos = (size_t)(end - s.c_str());
  return true;
}

static bool parse_sp(const std::string &s, size_t from, phnsw_search_params *p) {
  double a, b, c;
  if (!json_number(s, "number_of_candidates", from, &a) || !json_number(s, "upper_layer_candidate_count", from, &b) ||
      !json_number(s, "probe_depth", from, &c))
    return false;
  p->number_of_candidates = (uint64_t)a;
  p->upper_layer_candidate_count = (uint64_t)b;
  p->probe_depth = (uint64_t)c;
  return true;
}

extern "C" int phnsw_index_serialize(const phnsw_index *ix, const char *path) try {
  if (!ix || !path) {
    ph_set_error("phnsw_index_serialize: invalid argument");
    return PHNSW_E_INVALID;
  }
  const phnsw_store *s = ix->store;
  PH_HIP(hipSetDevice(s->device));
  std::string dir(path);
  create_dir_all(dir);
  if (!exists(dir)) {
    ph_set_error("cannot create directory %s", dir.c_str());
    return PHNSW_E_INVALID;
  }
  const size_t layer_count = ix->layers.size();
  std::string meta = "{\"layer_count\":" + std::to_string(layer_count) + ",\"build_parameters\":" + json_bp(ix->bp) + "}";
  int rc = write_file(dir + "/meta", meta.data(), meta.size());
  if (rc) return rc;
  if (layer_count > 0) {
    // layers[0].comparator.serialize(<dir>/comparator)  :59-64
    std::string cdir = dir + "/comparator";
    mkdir(cdir.c_str(), 0777);
    std::ostringstream cm;
    cm << "{\"n\":" << s->n << ",\"dim\":" << s->dim << ",\"metric\":" << s->metric
       << ",\"kind\":\"" << (s->codes ? "pq" : "f32") << "\"";
    if (s->codes) cm << ",\"m\":" << s->pq_m << ",\"ksub\":" << s->pq_ksub << ",\"dsub\":" << s->pq_dsub;
    cm << "}";
    std::string cms = cm.str();
    rc = write_file(cdir + "/meta.json", cms.data(), cms.size());
    if (rc) return rc;
    if (s->codes) {
      std::vector<uint8_t> codes((size_t)s->n * s->pq_m);
      std::vector<float> cb((size_t)s->pq_m * s->pq_ksub * s->pq_dsub);
      PH_HIP(hipMemcpy(codes.data(), s->codes, codes.size(), hipMemcpyDeviceToHost));
      PH_HIP(hipMemcpy(cb.data(), s->codebook, cb.size() * 4, hipMemcpyDeviceToHost));
      rc = write_file(cdir + "/codes.u8", codes.data(), codes.size());
      if (!rc) rc = write_file(cdir + "/codebook.f32", cb.data(), cb.size() * 4);
    } else {
      std::vector<float> rows((size_t)s->n * s->dim);
      PH_HIP(hipMemcpy2D(rows.data(), (size_t)s->dim * 4, s->rows, (size_t)s->ld * 4, (size_t)s->dim * 4, s->n,
                         hipMemcpyDeviceToHost));
      rc = write_file(cdir + "/vectors.f32", rows.data(), rows.size() * 4);
    }
    if (rc) return rc;
  }
  for (size_t i = 0; i < layer_count; i++) {
    const size_t layer_number = layer_count - i - 1;  // :67
    const PhLayerHost &L = ix->layers[i];
    std::string suffix = "." + std::to_string(layer_number);
    std::string lm = "{\"node_count\":" + std::to_string(L.n_nodes) + ",\"neighborhood_size\":" + std::to_string(L.W) + "}";
    rc = write_file(dir + "/layer.meta" + suffix, lm.data(), lm.size());
    if (rc) return rc;
    std::vector<uint32_t> n32(L.n_nodes), b32((size_t)L.n_nodes * L.W);
    PH_HIP(hipMemcpy(n32.data(), L.nodes, n32.size() * 4, hipMemcpyDeviceToHost));
    PH_HIP(hipMemcpy(b32.data(), L.neighbors, b32.size() * 4, hipMemcpyDeviceToHost));
    std::vector<uint64_t> n64(n32.begin(), n32.end()), b64(b32.size());
    for (size_t k = 0; k < b32.size(); k++) b64[k] = b32[k] == PH_EMPTY32 ? PHNSW_EMPTY : b32[k];
    rc = write_file(dir + "/layer.nodes" + suffix, n64.data(), n64.size() * 8);
    if (!rc) rc = write_file(dir + "/layer.neighbors" + suffix, b64.data(), b64.size() * 8);
    if (rc) return rc;
  }
  return 0;
} catch (...) { return ph_caught(); }

// deserialize_hnsw  serialize.rs:126-209 against an existing store (the C::Params role)
extern "C" int phnsw_index_deserialize(phnsw_store *s, const char *path, phnsw_index **out) try {
  if (!s || !path || !out) {
    ph_set_error("phnsw_index_deserialize: invalid argument");
    return PHNSW_E_INVALID;
  }
  std::string dir(path), meta;
  int rc = read_file(dir + "/meta", meta);
  if (rc) return rc;
  double lc = 0;
  if (!json_number(meta, "layer_count", 0, &lc) || lc < 0 || lc > PH_MAX_LAYERS) {
    ph_set_error("%s/meta: layer_count missing or out of range", path);
    return PHNSW_E_INVALID;
  }
  if (!exists(dir + "/comparator")) {  // SerializationError::IndexNotFound  :144-146
    ph_set_error("Index not found (no comparator entry under %s)", path);
    return PHNSW_E_INVALID;
  }
  phnsw_build_params bp;
  phnsw_default_build_params(&bp);
  double v;
  size_t opt = meta.find("\"optimization\""), ips = meta.find("\"initial_partition_search\"");
  if (json_number(meta, "order", 0, &v)) bp.order = (uint64_t)v;
  if (json_number(meta, "zero_layer_neighborhood_size", 0, &v)) bp.zero_layer_neighborhood_size = (uint64_t)v;
  // "neighborhood_size" is a suffix of the previous key: search for it preceded by a comma
  size_t ns = meta.find(",\"neighborhood_size\"");
  if (ns != std::string::npos && json_number(meta, "neighborhood_size", ns, &v)) bp.neighborhood_size = (uint64_t)v;
  if (opt != std::string::npos) {
    if (json_number(meta, "promotion_threshold", opt, &v)) bp.optimization.promotion_threshold = (float)v;
    if (json_number(meta, "neighborhood_threshold", opt, &v)) bp.optimization.neighborhood_threshold = (float)v;
    if (json_number(meta, "recall_proportion", opt, &v)) bp.optimization.recall_proportion = (float)v;
    if (json_number(meta, "promotion_proportion", opt, &v)) bp.optimization.promotion_proportion = (float)v;
    size_t se = meta.find("\"search\"", opt);
    if (se != std::string::npos) parse_sp(meta, se, &bp.optimization.search);
  }
  if (ips != std::string::npos) parse_sp(meta, ips, &bp.initial_partition_search);
  const size_t layer_count = (size_t)lc;
  std::vector<std::vector<uint64_t>> nodes(layer_count), nbrs(layer_count);
  std::vector<uint64_t> counts(layer_count), widths(layer_count);
  for (size_t i = 0; i < layer_count; i++) {
    const size_t layer_number = layer_count - i - 1;
    std::string suffix = "." + std::to_string(layer_number), lm, raw;
    rc = read_file(dir + "/layer.meta" + suffix, lm);
    if (rc) return rc;
    double nc, w;
    if (!json_number(lm, "node_count", 0, &nc) || !json_number(lm, "neighborhood_size", 0, &w)) {
      ph_set_error("%s/layer.meta%s: malformed", path, suffix.c_str());
      return PHNSW_E_INVALID;
    }
    // a corrupt or crafted meta file must not size an allocation: same limits as phnsw_index_from_layers
    if (!(nc >= 1.0 && nc < 2147483647.0) || !(w >= 1.0 && w <= 64.0) || nc != (double)(uint64_t)nc || w != (double)(uint64_t)w) {
      ph_set_error("%s/layer.meta%s: node_count %.17g / neighborhood_size %.17g out of range (1..2^31-2, 1..64)", path,
                   suffix.c_str(), nc, w);
      return PHNSW_E_INVALID;
    }
    counts[i] = (uint64_t)nc;
    widths[i] = (uint64_t)w;
    rc = read_file(dir + "/layer.nodes" + suffix, raw);
    if (rc) return rc;
    if (raw.size() != counts[i] * 8) {  // read_exact would fail  :177
      ph_set_error("%s/layer.nodes%s: expected %llu bytes, found %zu", path, suffix.c_str(),
                   (unsigned long long)(counts[i] * 8), raw.size());
      return PHNSW_E_INVALID;
    }
    nodes[i].resize(counts[i]);
    memcpy(nodes[i].data(), raw.data(), raw.size());
    rc = read_file(dir + "/layer.neighbors" + suffix, raw);
    if (rc) return rc;
    if (raw.size() != counts[i] * widths[i] * 8) {
      ph_set_error("%s/layer.neighbors%s: expected %llu bytes, found %zu", path, suffix.c_str(),
                   (unsigned long long)(counts[i] * widths[i] * 8), raw.size());
      return PHNSW_E_INVALID;
    }
    nbrs[i].resize(counts[i] * widths[i]);
    memcpy(nbrs[i].data(), raw.data(), raw.size());
  }
  if (layer_count == 0) {
    ph_set_error("%s holds an index without layers", path);
    return PHNSW_E_INVALID;
  }
  std::vector<const uint64_t *> pn(layer_count), pb(layer_count);
  for (size_t i = 0; i < layer_count; i++) {
    pn[i] = nodes[i].data();
    pb[i] = nbrs[i].data();
  }
  rc = phnsw_index_from_layers(s, (uint32_t)layer_count, counts.data(), widths.data(), pn.data(), pb.data(), out);
  if (rc) return rc;
  (*out)->bp = bp;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_index_build_params(const phnsw_index *ix, phnsw_build_params *bp) try {
  if (!ix || !bp) return PHNSW_E_INVALID;
  *bp = ix->bp;
  return 0;
} catch (...) { return ph_caught(); }
