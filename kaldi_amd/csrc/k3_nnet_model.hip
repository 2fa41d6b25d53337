// k3_nnet_model.hip -- host-only: Kaldi nnet3 model reader (text + binary) and layer fuser.  See k3_nnet_model.h.
#include "k3_nnet_model.h"
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <numeric>
#include <set>
#include <sstream>

namespace k3 {

int Scalar::as_int() const {
  if (from_binary4) { int32_t v; memcpy(&v, &raw32, 4); return v; }
  return (int)llround(num);
}
float Scalar::as_float() const {
  if (from_binary4) { float v; memcpy(&v, &raw32, 4); return v; }
  return (float)num;
}
const Field *RawComponent::get(const std::string &tag) const {
  auto it = fields.find(tag);
  return it == fields.end() ? nullptr : &it->second;
}

namespace {

struct Cursor {
  const char *p, *end;
  bool binary;
  std::string err;
  bool fail(const std::string &m) { if (err.empty()) err = m; return false; }
};

void SkipWs(Cursor &c) { while (c.p < c.end && isspace((unsigned char)*c.p)) c.p++; }

// base/io-funcs.cc ReadToken: text = whitespace separated; binary = token followed by one space
bool ReadToken(Cursor &c, std::string *tok) {
  if (!c.binary) SkipWs(c);
  const char *s = c.p;
  while (c.p < c.end && !isspace((unsigned char)*c.p)) c.p++;
  if (c.p == s) return c.fail("unexpected end of model file while reading a token");
  tok->assign(s, c.p - s);
  if (c.p < c.end) c.p++;   // consume the single separator
  return true;
}
bool PeekIsTag(Cursor &c) {
  if (!c.binary) SkipWs(c);
  return c.p < c.end && *c.p == '<';
}

bool ReadBinaryArray(Cursor &c, Field *f) {
  const bool dbl = (*c.p == 'D'), mat = (c.p[1] == 'M');
  c.p += 3;
  auto rd_i32 = [&](int *v) {
    if (c.p + 5 > c.end || *c.p != 4) return false;
    int32_t x; memcpy(&x, c.p + 1, 4); c.p += 5; *v = x; return true;
  };
  int r = 1, n = 0;
  if (mat) { if (!rd_i32(&r) || !rd_i32(&n)) return c.fail("bad binary matrix header"); }
  else { if (!rd_i32(&n)) return c.fail("bad binary vector header"); }
  const size_t count = (size_t)r * n, esz = dbl ? 8 : 4;
  if (c.p + count * esz > c.end) return c.fail("truncated binary array");
  f->is_array = true; f->rows = (n == 0 && !mat) ? 0 : r; f->cols = n; f->data.resize(count);
  if (dbl) for (size_t i = 0; i < count; i++) { double d; memcpy(&d, c.p + 8 * i, 8); f->data[i] = (float)d; }
  else if (count) memcpy(f->data.data(), c.p, count * 4);
  c.p += count * esz;
  return true;
}

bool ReadTextArray(Cursor &c, Field *f) {
  c.p++;  // '['
  std::vector<float> &d = f->data;
  int rows = 0, cur = 0, cols = -1;
  while (true) {
    while (c.p < c.end && (*c.p == ' ' || *c.p == '\t' || *c.p == '\r')) c.p++;
    if (c.p >= c.end) return c.fail("unterminated [ ] in text model");
    if (*c.p == '\n' || *c.p == ']') {
      if (cur > 0) { if (cols >= 0 && cur != cols) return c.fail("ragged matrix rows in text model"); cols = cur; rows++; cur = 0; }
      if (*c.p++ == ']') break;
      continue;
    }
    char *e; float v = strtof(c.p, &e);
    if (e == c.p) return c.fail(std::string("bad number in text array near '") + std::string(c.p, std::min<size_t>(12, c.end - c.p)) + "'");
    d.push_back(v); cur++; c.p = e;
  }
  f->is_array = true; f->rows = rows; f->cols = cols < 0 ? 0 : cols;
  return true;
}

// Everything following a <Tag> up to the next tag.
bool ReadValue(Cursor &c, const std::string &tag, Field *f) {
  if (c.binary) {
    while (c.p < c.end && *c.p != '<') {
      const unsigned char ch = (unsigned char)*c.p;
      if (tag == "<TimeOffsets>") {   // WriteIntegerVector, base/io-funcs-inl.h:198-211
        if (ch != 4 || c.p + 5 > c.end) return c.fail("bad <TimeOffsets>");
        int32_t n; memcpy(&n, c.p + 1, 4); c.p += 5;
        if (n < 0 || c.p + 4 * (size_t)n > c.end) return c.fail("bad <TimeOffsets> size");
        f->ints.resize(n); if (n) memcpy(f->ints.data(), c.p, 4 * (size_t)n); c.p += 4 * (size_t)n;
        return true;
      }
      if (ch == 4 && c.p + 5 <= c.end) { Scalar s; s.from_binary4 = true; memcpy(&s.raw32, c.p + 1, 4); f->scalars.push_back(s); c.p += 5; }
      else if (ch == 8 && c.p + 9 <= c.end) { Scalar s; s.from_binary8 = true; memcpy(&s.num, c.p + 1, 8); f->scalars.push_back(s); c.p += 9; }
      else if ((ch == 'T' || ch == 'F') && c.p + 1 < c.end && (c.p[1] == '<' || c.p[1] == ' ')) {
        Scalar s; s.num = (ch == 'T'); f->scalars.push_back(s); c.p++; if (*c.p == ' ') c.p++;
      } else if ((ch == 'F' || ch == 'D') && c.p + 3 <= c.end && (c.p[1] == 'M' || c.p[1] == 'V') && c.p[2] == ' ') {
        return ReadBinaryArray(c, f);
      } else {
        return c.fail("cannot parse binary value after " + tag + " (compressed matrices / unknown types are unsupported)");
      }
    }
    return true;
  }
  SkipWs(c);
  if (c.p < c.end && *c.p == '[') {
    if (!ReadTextArray(c, f)) return false;
    if (tag == "<TimeOffsets>") { for (float v : f->data) f->ints.push_back((int)lrintf(v)); }
    return true;
  }
  while (true) {
    SkipWs(c);
    if (c.p >= c.end || *c.p == '<') break;
    std::string t; if (!ReadToken(c, &t)) return false;
    Scalar s;
    if (t == "T" || t == "F") s.num = (t == "T");
    else { char *e; s.num = strtod(t.c_str(), &e); if (e == t.c_str()) return c.fail("bad scalar '" + t + "' after " + tag); }
    f->scalars.push_back(s);
  }
  return true;
}

bool ReadNnet3(Cursor &c, RawModel *m) {
  std::string tok;
  if (!ReadToken(c, &tok) || tok != "<Nnet3>") return c.fail("expected <Nnet3>, got '" + tok + "'");
  // config section: lines up to the first empty line (nnet-nnet.cc:602-612)
  // position at the start of the first config line: skip spaces and at most one newline
  while (c.p < c.end && (*c.p == ' ' || *c.p == '\r')) c.p++;
  if (c.p < c.end && *c.p == '\n') c.p++;
  while (c.p < c.end) {
    const char *nl = (const char *)memchr(c.p, '\n', c.end - c.p);
    if (!nl) return c.fail("unterminated config section");
    std::string line(c.p, nl - c.p); c.p = nl + 1;
    while (!line.empty() && isspace((unsigned char)line.back())) line.pop_back();
    if (line.empty()) { if (m->config_lines.empty()) continue; break; }
    m->config_lines.push_back(line);
  }
  if (!ReadToken(c, &tok) || tok != "<NumComponents>") return c.fail("expected <NumComponents>, got '" + tok + "'");
  Field nf; if (!ReadValue(c, tok, &nf) || nf.scalars.empty()) return c.fail("bad <NumComponents>");
  const int n = nf.scalars[0].as_int();
  if (n < 0 || n >= 100000) return c.fail("bad component count");
  for (int i = 0; i < n; i++) {
    RawComponent rc;
    if (!ReadToken(c, &tok) || tok != "<ComponentName>") return c.fail("expected <ComponentName>, got '" + tok + "'");
    if (!ReadToken(c, &rc.name) || !ReadToken(c, &tok)) return false;
    if (tok.size() < 3 || tok.front() != '<' || tok.back() != '>') return c.fail("bad component type token '" + tok + "'");
    rc.type = tok.substr(1, tok.size() - 2);
    const std::string closing = "</" + rc.type + ">";
    while (true) {
      if (!ReadToken(c, &tok)) return false;
      if (tok == closing) break;
      if (tok.empty() || tok[0] != '<') return c.fail("expected a <Tag> in component " + rc.name + ", got '" + tok + "'");
      Field f; if (!ReadValue(c, tok, &f)) return false;
      rc.fields[tok] = std::move(f);
    }
    m->components.push_back(std::move(rc));
  }
  if (!ReadToken(c, &tok) || tok != "</Nnet3>") return c.fail("expected </Nnet3>, got '" + tok + "'");
  return true;
}

}  // namespace

bool ReadModelFile(const std::string &path, RawModel *out, std::string *err) {
  std::ifstream is(path, std::ios::binary);
  if (!is) { *err = "cannot open " + path; return false; }
  std::string buf((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
  Cursor c{buf.data(), buf.data() + buf.size(), false, ""};
  if (buf.size() >= 2 && buf[0] == '\0' && buf[1] == 'B') { c.binary = true; c.p += 2; }
  // .mdl: <TransitionModel> ... </TransitionModel> then AmNnetSimple (nnet-nnet.cc:588-599, am-nnet-simple.cc:47-57)
  {
    Cursor probe = c; if (!probe.binary) SkipWs(probe);
    static const char kTm[] = "<TransitionModel>";
    if ((size_t)(probe.end - probe.p) > sizeof(kTm) && memcmp(probe.p, kTm, sizeof(kTm) - 1) == 0) {
      static const char kEnd[] = "</TransitionModel>";
      const char *e = std::search(probe.p, probe.end, kEnd, kEnd + sizeof(kEnd) - 1);
      if (e == probe.end) { *err = "unterminated <TransitionModel> in " + path; return false; }
      c.p = e + sizeof(kEnd) - 1;
      if (c.p < c.end && (*c.p == ' ' || *c.p == '\n')) c.p++;
      out->has_am = true;
    }
  }
  if (!ReadNnet3(c, out)) { *err = path + ": " + c.err; return false; }
  if (out->has_am) {
    std::string tok;
    while (PeekIsTag(c) && ReadToken(c, &tok)) {
      Field f; if (!ReadValue(c, tok, &f)) { *err = path + ": " + c.err; return false; }
      if (tok == "<LeftContext>" && !f.scalars.empty()) out->left_context = f.scalars[0].as_int();
      else if (tok == "<RightContext>" && !f.scalars.empty()) out->right_context = f.scalars[0].as_int();
      else if (tok == "<Priors>") out->priors = f.data;
    }
  }
  return true;
}

// ------------------------------------------------------------------------------ fuser ----
namespace {

struct Desc {   // nnet-descriptor.h subset
  enum Kind { kNode, kOffset, kAppend, kSum, kScale, kIvector } kind = kNode;      // kIvector = ReplaceIndex(ivector, t, 0)
  std::string node;
  int offset = 0;
  float scale = 1.0f;
  std::vector<Desc> args;
};

struct DescParser {
  const std::string &s; size_t i = 0; std::string err;
  explicit DescParser(const std::string &str) : s(str) {}
  void ws() { while (i < s.size() && isspace((unsigned char)s[i])) i++; }
  bool parse(Desc *d) {
    ws();
    size_t b = i;
    while (i < s.size() && (isalnum((unsigned char)s[i]) || s[i] == '_' || s[i] == '-' || s[i] == '.')) i++;
    std::string word = s.substr(b, i - b);
    if (word.empty()) { err = "bad descriptor '" + s + "'"; return false; }
    ws();
    if (i >= s.size() || s[i] != '(') { d->kind = Desc::kNode; d->node = word; return true; }
    i++;  // '('
    if (word == "Offset") {
      d->kind = Desc::kOffset; d->args.resize(1);
      if (!parse(&d->args[0]) || !expect(',')) return false;
      if (!integer(&d->offset)) return false;
      ws();
      if (i < s.size() && s[i] == ',') { err = "Offset(..., t, x) with an x offset is unsupported"; return false; }
    } else if (word == "Scale") {
      d->kind = Desc::kScale; d->args.resize(1);
      ws(); char *e; d->scale = strtof(s.c_str() + i, &e); if (e == s.c_str() + i) { err = "bad Scale() in '" + s + "'"; return false; }
      i = e - s.c_str();
      if (!expect(',') || !parse(&d->args[0])) return false;
    } else if (word == "Append" || word == "Sum") {
      d->kind = word == "Append" ? Desc::kAppend : Desc::kSum;
      while (true) {
        d->args.emplace_back();
        if (!parse(&d->args.back())) return false;
        ws();
        if (i < s.size() && s[i] == ',') { i++; continue; }
        break;
      }
    } else if (word == "ReplaceIndex") {      // only the recipes' ReplaceIndex(ivector, t, 0): "the i-vector of this chunk, whatever t" (nnet-descriptor.h:246-268)
      d->kind = Desc::kIvector; Desc inner; int zero = -1;
      if (!parse(&inner) || !expect(',')) return false;
      ws(); const bool is_t = i < s.size() && s[i] == 't'; if (is_t) i++;
      if (!is_t || !expect(',') || !integer(&zero) || zero != 0 || inner.kind != Desc::kNode || inner.node != "ivector") {
        err = "only ReplaceIndex(ivector, t, 0) is supported, got '" + s + "'";
        return false;
      }
      d->node = "ivector";
    } else {
      err = "descriptor function " + word + "() is unsupported (IfDefined/Failover/Round/Const need recurrent or multi-input models)";
      return false;
    }
    return expect(')');
  }
  bool expect(char ch) { ws(); if (i < s.size() && s[i] == ch) { i++; return true; } err = std::string("expected '") + ch + "' in descriptor '" + s + "'"; return false; }
  bool integer(int *v) {
    ws();
    char *e;
    long x = strtol(s.c_str() + i, &e, 10);
    if (e == s.c_str() + i) {
      err = "bad integer in descriptor '" + s + "'";
      return false;
    }
    i = e - s.c_str();
    *v = (int)x;
    return true;
  }
};

void CollectNodes(const Desc &d, std::vector<std::string> *out) {
  if (d.kind == Desc::kNode) out->push_back(d.node);
  for (const Desc &a : d.args) CollectNodes(a, out);
}

std::map<std::string, std::string> ParseKeyValues(const std::string &line, std::string *first) {
  // "component-node name=x component=y input=Append(a, b)" -> map; values run to the next " key=" boundary
  std::map<std::string, std::string> kv;
  size_t sp = line.find(' ');
  *first = line.substr(0, sp);
  size_t i = sp;
  while (i != std::string::npos && i < line.size()) {
    while (i < line.size() && isspace((unsigned char)line[i])) i++;
    size_t eq = line.find('=', i);
    if (eq == std::string::npos) break;
    std::string key = line.substr(i, eq - i);
    size_t j = eq + 1; int depth = 0; size_t vend = line.size();
    for (size_t k = j; k < line.size(); k++) {
      if (line[k] == '(') depth++;
      else if (line[k] == ')') depth--;
      else if (isspace((unsigned char)line[k]) && depth == 0) {
        size_t k2 = k; while (k2 < line.size() && isspace((unsigned char)line[k2])) k2++;
        size_t e2 = k2; while (e2 < line.size() && (isalnum((unsigned char)line[e2]) || line[e2] == '-' || line[e2] == '_')) e2++;
        if (e2 < line.size() && line[e2] == '=' && e2 > k2) { vend = k; break; }
      }
    }
    kv[key] = line.substr(j, vend - j);
    i = vend;
  }
  return kv;
}

bool IsAffineLike(const std::string &t) {
  return t == "AffineComponent" || t == "NaturalGradientAffineComponent" || t == "FixedAffineComponent" ||
         t == "LinearComponent" || t == "TdnnComponent";
}
bool IsIdentityAtTest(const std::string &t) {
  return t == "NoOpComponent" || t == "DropoutComponent" || t == "GeneralDropoutComponent" || t == "SpecAugmentTimeMaskComponent";
}

// test-mode BatchNorm scale/offset: ComputeDerived(), nnet-normalize-component.cc:209-247 (same precisions)
bool BatchNormScaleOffset(const RawComponent &c, std::vector<float> *scale, std::vector<float> *offset, std::string *err) {
  const Field *fd = c.get("<Dim>"), *fb = c.get("<BlockDim>"), *fe = c.get("<Epsilon>"), *ft = c.get("<TargetRms>"),
              *fc = c.get("<Count>"), *fm = c.get("<StatsMean>"), *fv = c.get("<StatsVar>");
  if (!fd || !fb || !fe || !ft || !fc || !fm || !fv || fd->scalars.empty() || fb->scalars.empty()) { *err = "BatchNormComponent " + c.name + ": missing fields"; return false; }
  const int dim = fd->scalars[0].as_int(), bdim = fb->scalars[0].as_int();
  const float eps = fe->scalars[0].as_float(), rms = ft->scalars[0].as_float();
  const double count = fc->scalars[0].from_binary4 ? fc->scalars[0].as_float() : fc->scalars[0].num;
  if (count <= 0.0) { *err = "Test mode set in BatchNormComponent " + c.name + ", but no stats."; return false; }
  if ((int)fm->data.size() != bdim || (int)fv->data.size() != bdim || bdim <= 0 || dim % bdim) { *err = "BatchNormComponent " + c.name + ": bad stats dims"; return false; }
  scale->resize(dim); offset->resize(dim);
  for (int i = 0; i < bdim; i++) {
    double ssum = (double)fm->data[i], ssq = (double)fv->data[i] + ssum * ssum;   // Read(): :605-610
    ssum *= count; ssq *= count;
    float off = (float)ssum; off *= (float)(-1.0 / count);                         // -mean
    float sc = (float)ssq; sc *= (float)(1.0 / count);
    sc = sc + (-1.0f) * off * off;
    if (sc < 0.0f) sc = 0.0f;
    sc += eps;
    sc = powf(sc, -0.5f);
    sc *= rms;
    off *= sc;
    for (int b = 0; b < dim / bdim; b++) { (*scale)[b * bdim + i] = sc; (*offset)[b * bdim + i] = off; }
  }
  return true;
}

}  // namespace

bool FuseModel(const RawModel &raw, FusedModel *fm, std::string *err) {
  std::map<std::string, const RawComponent *> comps;
  for (const RawComponent &c : raw.components) comps[c.name] = &c;
  fm->num_components = (int)raw.components.size();
  for (const RawComponent &c : raw.components)
    for (const char *tag : {"<LinearParams>", "<BiasParams>", "<Params>"})
      if (const Field *f = c.get(tag)) fm->num_params += (int64_t)f->data.size();

  struct CfgNode { std::string kind, name, component; Desc desc; };
  std::vector<CfgNode> cfg;
  std::map<std::string, int> consumers;
  for (const std::string &line : raw.config_lines) {
    std::string first; auto kv = ParseKeyValues(line, &first);
    if (first == "input-node") {
      if (kv["name"] == "ivector") { fm->ivector_dim = atoi(kv["dim"].c_str()); if (fm->ivector_dim <= 0) { *err = "input-node ivector with a bad dim"; return false; } continue; }
      if (kv["name"] != "input") { *err = "input-node '" + kv["name"] + "': only 'input' and 'ivector' are supported"; return false; }
      fm->input_dim = atoi(kv["dim"].c_str());
    } else if (first == "component-node" || first == "output-node") {
      CfgNode n; n.kind = first; n.name = kv["name"]; n.component = kv["component"];
      DescParser dp(kv["input"]);
      if (!dp.parse(&n.desc)) { *err = dp.err; return false; }
      std::vector<std::string> used; CollectNodes(n.desc, &used);
      for (auto &u : used) consumers[u]++;
      if (first == "output-node" && n.name != "output") continue;   // e.g. output-xent: unused at inference
      cfg.push_back(std::move(n));
    } else if (first == "dim-range-node") {
      *err = "dim-range-node is unsupported (LSTM-style models are out of scope)"; return false;
    } else { *err = "unexpected config line: " + line; return false; }
  }
  if (fm->input_dim <= 0) { *err = "model has no input-node name=input"; return false; }

  std::map<std::string, int> producer;            // node name -> fused node whose CURRENT output carries that name
  std::map<std::string, int> dims; dims["input"] = fm->input_dim;
  producer["input"] = -1;
  auto lookup = [&](const std::string &name, int *idx) {
    auto it = producer.find(name);
    if (it == producer.end()) { *err = "descriptor refers to '" + name + "' which is not (or no longer) an addressable node"; return false; }
    *idx = it->second; return true;
  };
  // input descriptor of a GEMM node -> (source, offsets)
  auto splice_of = [&](const Desc &d, std::string *src, std::vector<int> *offs, bool *with_ivector) -> bool {
    std::vector<const Desc *> parts; *with_ivector = false;
    if (d.kind == Desc::kAppend) for (const Desc &a : d.args) parts.push_back(&a); else parts.push_back(&d);
    if (parts.size() > 1 && parts.back()->kind == Desc::kIvector) {
      if (fm->ivector_dim <= 0) { *err = "ReplaceIndex(ivector, t, 0) in a model without input-node name=ivector"; return false; }
      *with_ivector = true; parts.pop_back();
    }
    for (const Desc *p : parts) {
      int o = 0; const Desc *q = p;
      while (q->kind == Desc::kOffset) { o += q->offset; q = &q->args[0]; }
      if (q->kind != Desc::kNode) {
        *err = "unsupported input descriptor for an affine component (need Append(Offset(x,t)..., [ReplaceIndex(ivector, t, 0)]) of one node, the i-vector last)";
        return false;
      }
      if (!src->empty() && *src != q->node) { *err = "Append() of different nodes (" + *src + ", " + q->node + ") is unsupported (no ivector/multi-stream models)"; return false; }
      *src = q->node; offs->push_back(o);
    }
    return true;
  };

  for (const CfgNode &n : cfg) {
    if (n.kind == "output-node") {
      if (n.desc.kind != Desc::kNode) { *err = "output-node with a non-trivial descriptor is unsupported"; return false; }
      int idx; if (!lookup(n.desc.node, &idx)) return false;
      if (idx < 0) { *err = "output-node fed directly by the input"; return false; }
      fm->output_node = idx; fm->output_dim = fm->nodes[idx].out_dim;
      continue;
    }
    auto ci = comps.find(n.component);
    if (ci == comps.end()) { *err = "component-node " + n.name + ": unknown component " + n.component; return false; }
    const RawComponent &c = *ci->second;
    if (IsAffineLike(c.type)) {
      std::string src; std::vector<int> desc_offs; bool with_iv = false;
      if (!splice_of(n.desc, &src, &desc_offs, &with_iv)) return false;
      FusedNode f; f.has_gemm = true; f.name = n.name;
      if (!lookup(src, &f.input)) return false;
      f.in_dim = dims[src];
      const Field *w = c.get(c.type == "LinearComponent" ? "<Params>" : "<LinearParams>");
      const Field *b = c.get("<BiasParams>");
      if (!w || !w->is_array || w->rows <= 0) { *err = c.type + " " + c.name + ": missing weight matrix"; return false; }
      std::vector<int> comp_offs{0};
      if (c.type == "TdnnComponent") {
        const Field *t = c.get("<TimeOffsets>");
        if (!t || t->ints.empty()) { *err = "TdnnComponent " + c.name + ": missing <TimeOffsets>"; return false; }
        comp_offs = t->ints;
      }
      // K blocks: component offset major, descriptor part minor (the component sees the appended vector)
      for (int co : comp_offs) for (int dofs : desc_offs) f.offsets.push_back(co + dofs);
      f.out_dim = w->rows;
      const int iv_cols = with_iv ? fm->ivector_dim : 0;
      if (with_iv && comp_offs.size() != 1) {
        *err = "TdnnComponent " + c.name + " with several time offsets over an Append() that holds the i-vector is unsupported";
        return false;
      }
      if ((int64_t)w->cols != (int64_t)f.offsets.size() * f.in_dim + iv_cols) {
        char m[256];
        snprintf(m, sizeof m, "%s %s: weight is %d x %d but input is %zu x %d + %d", c.type.c_str(), c.name.c_str(), w->rows, w->cols, f.offsets.size(),
            f.in_dim, iv_cols);
        *err = m; return false;
      }
      if (!with_iv) f.W = w->data;
      else {                                              // split the columns: [ spliced | i-vector ]
        const int ks = w->cols - iv_cols; f.W.resize((size_t)w->rows * ks); f.W_iv.resize((size_t)w->rows * iv_cols);
        for (int r = 0; r < w->rows; r++) {
          memcpy(&f.W[(size_t)r * ks], &w->data[(size_t)r * w->cols], sizeof(float) * ks);
          memcpy(&f.W_iv[(size_t)r * iv_cols], &w->data[(size_t)r * w->cols + ks], sizeof(float) * iv_cols);
        }
      }
      if (b && !b->data.empty()) { if ((int)b->data.size() != f.out_dim) { *err = c.name + ": bias dim mismatch"; return false; } f.bias = b->data; }
      fm->nodes.push_back(std::move(f));
      producer[n.name] = (int)fm->nodes.size() - 1; dims[n.name] = fm->nodes.back().out_dim;
      continue;
    }
    // ---- element-wise component: fold into the producing node when it is that node's only consumer ----
    const bool is_relu = c.type == "RectifiedLinearComponent", is_bn = c.type == "BatchNormComponent", is_id = IsIdentityAtTest(c.type);
    const bool is_sig = c.type == "SigmoidComponent", is_tanh = c.type == "TanhComponent";
    // nnet-simple-component.cc:3618-3625 / :3494-3504 (the output layer of non-chain nnet3 models); nnet-normalize-component.cc (relu-renorm layers)
    const int row_op = c.type == "LogSoftmaxComponent" ? 1 : c.type == "SoftmaxComponent" ? 2 : c.type == "NormalizeComponent" ? 3 : 0;
    float row_param = 0.0f;
    if (row_op == 3) {      // NormalizeComponent::Read: <TargetRms> and <AddLogStddev> (absent in old models = 1.0 / false)
      const Field *tr = c.get("<TargetRms>"), *als = c.get("<AddLogStddev>");
      row_param = tr && !tr->scalars.empty() ? tr->scalars[0].as_float() : 1.0f;
      if (c.get("<BlockDim>")) { *err = "NormalizeComponent " + c.name + " with block-dim != dim is not supported by the fused MI355X path"; return false; }
      if (als && !als->scalars.empty() && als->scalars[0].num != 0.0) {
        *err = "NormalizeComponent " + c.name + " with add-log-stddev=true is not supported by the fused MI355X path (its output has one more column)";
        return false;
      }
      if (!(row_param > 0.0f)) { *err = "NormalizeComponent " + c.name + ": bad <TargetRms>"; return false; }
    }
    if (!is_relu && !is_bn && !is_id && !row_op && !is_sig && !is_tanh) {
      *err = "component type " + c.type + " (" + c.name + ") is not supported by the MI355X TDNN/TDNN-F path"; return false;
    }
    // main input + optional residual term
    std::string main_name; std::string res_name; float res_scale = 1.0f; bool has_res = false;
    auto unscale = [&](const Desc &d, std::string *name, float *sc) -> bool {
      const Desc *q = &d; *sc = 1.0f;
      while (q->kind == Desc::kScale) { *sc *= q->scale; q = &q->args[0]; }
      if (q->kind != Desc::kNode) return false;
      *name = q->node; return true;
    };
    if (n.desc.kind == Desc::kNode) main_name = n.desc.node;
    else if (n.desc.kind == Desc::kSum && n.desc.args.size() == 2) {
      std::string a, b; float sa, sb;
      if (!unscale(n.desc.args[0], &a, &sa) || !unscale(n.desc.args[1], &b, &sb)) { *err = "unsupported Sum() descriptor at " + n.name; return false; }
      // prefer as 'main' an unscaled term whose producer can absorb this node
      auto absorbable = [&](const std::string &nm, float sc) {
        auto it = producer.find(nm);
        return sc == 1.0f && it != producer.end() && it->second >= 0 && consumers[nm] == 1 && fm->nodes[it->second].name == nm;
      };
      if (absorbable(b, sb)) { main_name = b; res_name = a; res_scale = sa; }
      else if (absorbable(a, sa)) { main_name = a; res_name = b; res_scale = sb; }
      else if (sb == 1.0f) { main_name = b; res_name = a; res_scale = sa; }
      else if (sa == 1.0f) { main_name = a; res_name = b; res_scale = sb; }
      else { *err = "Sum(Scale(..), Scale(..)) at " + n.name + " is unsupported"; return false; }
      has_res = true;
    } else { *err = "unsupported descriptor at element-wise node " + n.name; return false; }
    int main_idx; if (!lookup(main_name, &main_idx)) return false;
    int target;
    if (main_idx >= 0 && consumers[main_name] == 1 && fm->nodes[main_idx].name == main_name && fm->nodes[main_idx].row_op == 0) {
      target = main_idx;                                  // fold
      producer.erase(main_name);
    } else {                                              // stand-alone element-wise node
      FusedNode f; f.has_gemm = false; f.input = main_idx; f.in_dim = f.out_dim = dims[main_name]; f.offsets = {0};
      fm->nodes.push_back(std::move(f)); target = (int)fm->nodes.size() - 1;
    }
    FusedNode &t = fm->nodes[target];
    if (has_res) {
      for (const EpiOp &o : t.ops) if (o.kind == kEpiResidual) { *err = "two residual terms on one fused node (" + n.name + ")"; return false; }
      EpiOp op; op.kind = kEpiResidual; op.res_scale = res_scale;
      if (!lookup(res_name, &op.res_node)) return false;
      if (dims[res_name] != t.out_dim) { *err = "Sum() dimension mismatch at " + n.name; return false; }
      t.ops.push_back(std::move(op));
    }
    if (is_relu) { EpiOp op; op.kind = kEpiRelu; t.ops.push_back(std::move(op)); }
    if (is_sig || is_tanh) { EpiOp op; op.kind = is_sig ? kEpiSigmoid : kEpiTanh; t.ops.push_back(std::move(op)); }
    if (is_bn) { EpiOp op; op.kind = kEpiScaleOffset; if (!BatchNormScaleOffset(c, &op.scale, &op.offset, err)) return false;
                 if ((int)op.scale.size() != t.out_dim) { *err = "BatchNorm dim mismatch at " + n.name; return false; } t.ops.push_back(std::move(op)); }
    if (row_op) { t.row_op = row_op; t.row_param = row_param; }
    t.name = n.name; producer[n.name] = target; dims[n.name] = t.out_dim;
  }
  if (fm->output_node < 0) { *err = "model has no output-node name=output"; return false; }
  // contexts (ComputeSimpleNnetContext): propagate forward
  {
    std::vector<int> L(fm->nodes.size()), R(fm->nodes.size());
    for (size_t i = 0; i < fm->nodes.size(); i++) {
      const FusedNode &f = fm->nodes[i];
      int l = f.input >= 0 ? L[f.input] : 0, r = f.input >= 0 ? R[f.input] : 0;
      int mn = *std::min_element(f.offsets.begin(), f.offsets.end()), mx = *std::max_element(f.offsets.begin(), f.offsets.end());
      l -= mn; r += mx;
      for (const EpiOp &o : f.ops) if (o.kind == kEpiResidual && o.res_node >= 0) { l = std::max(l, L[o.res_node]); r = std::max(r, R[o.res_node]); }
      L[i] = l; R[i] = r;
    }
    fm->left_context = L[fm->output_node]; fm->right_context = R[fm->output_node];
  }
  fm->priors = raw.priors;
  return true;
}

}  // namespace k3
