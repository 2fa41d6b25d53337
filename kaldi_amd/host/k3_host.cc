// k3_host.cc -- see k3_host.h.  Formats are restated from the reference's readers/writers (cited per function); OpenFst's
// binary container is restated from its documented layout (FstHeader + vector/const bodies) and is exercised against
// kaldi_amd/fst.py's writer only, because OpenFst itself is not available here.
#include "k3_host.h"
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>

namespace k3host {

std::string g_program = "k3";
int g_verbose = 0;

LogLine::LogLine(const char *severity, const char *func, const char *file, int line, bool is_fatal) : sev(severity), fatal(is_fatal) {
  const char *base = strrchr(file, '/');
  ss << severity << " (" << g_program << "[k3hip]:" << func << "():" << (base ? base + 1 : file) << ":" << line << ") ";
}
LogLine::~LogLine() noexcept(false) {
  std::cerr << ss.str() << "\n";
  if (fatal) throw FatalError(ss.str());
}

// ------------------------------------------------------------------------------------------------ ParseOptions ----
ParseOptions::ParseOptions(const char *usage) : usage_(usage) {}
std::string ParseOptions::Normalize(const std::string &n) { std::string r = n; for (char &c : r) { if (c == '_') c = '-'; c = (char)tolower(c); } return r; }
void ParseOptions::RegisterImpl(const std::string &name, Kind k, void *p, const std::string &doc, const std::string &def) {
  opts_[Normalize(name)] = Opt{k, p, doc, def};
}
void ParseOptions::Register(const std::string &n, bool *p, const std::string &d) { RegisterImpl(n, kBool, p, d, *p ? "true" : "false"); }
void ParseOptions::Register(const std::string &n, int32_t *p, const std::string &d) { RegisterImpl(n, kInt, p, d, std::to_string(*p)); }
void ParseOptions::Register(const std::string &n, float *p, const std::string &d) { std::ostringstream o; o << *p; RegisterImpl(n, kFloat, p, d, o.str()); }
void ParseOptions::Register(const std::string &n, double *p, const std::string &d) { std::ostringstream o; o << *p; RegisterImpl(n, kDouble, p, d, o.str()); }
void ParseOptions::Register(const std::string &n, std::string *p, const std::string &d) { RegisterImpl(n, kString, p, d, *p); }

bool ParseOptions::SetOption(const std::string &key_in, const std::string &value, bool has_value) {
  const std::string key = Normalize(key_in);
  auto it = opts_.find(key);
  if (it == opts_.end()) return false;
  Opt &o = it->second; char *end = nullptr; errno = 0;
  switch (o.kind) {
    case kBool: {
      std::string v = value; for (char &c : v) c = (char)tolower(c);
      if (!has_value || v == "true" || v == "t" || v == "1" || v.empty()) *(bool *)o.ptr = true;
      else if (v == "false" || v == "f" || v == "0") *(bool *)o.ptr = false;
      else K3H_ERR << "Invalid format for boolean argument [expected true or false]: --" << key << "=" << value;
      break;
    }
    case kInt: {
      long v = strtol(value.c_str(), &end, 10);
      if (!has_value || value.empty() || *end || errno) K3H_ERR << "Invalid integer option \"" << value << "\" for --" << key;
      *(int32_t *)o.ptr = (int32_t)v;
      break;
    }
    case kFloat: {
      double v = strtod(value.c_str(), &end);
      if (!has_value || value.empty() || *end) K3H_ERR << "Invalid floating-point option \"" << value << "\" for --" << key;
      *(float *)o.ptr = (float)v;
      break;
    }
    case kDouble: {
      double v = strtod(value.c_str(), &end);
      if (!has_value || value.empty() || *end) K3H_ERR << "Invalid floating-point option \"" << value << "\" for --" << key;
      *(double *)o.ptr = v;
      break;
    }
    case kString: *(std::string *)o.ptr = value; break;
  }
  set_[key] = 1;
  return true;
}

void ParseOptions::ReadConfigFile(const std::string &path) {          // util/parse-options.cc:434-490: one --opt=value per line, # comments
  const std::string txt = ReadWholeInput(path);
  std::istringstream is(txt); std::string line; int ln = 0;
  while (std::getline(is, line)) {
    ln++;
    const size_t h = line.find('#'); if (h != std::string::npos) line.erase(h);
    const size_t b = line.find_first_not_of(" \t\r"); if (b == std::string::npos) continue;
    line = line.substr(b, line.find_last_not_of(" \t\r") - b + 1);
    if (line.compare(0, 2, "--") != 0) K3H_ERR << "Reading config file " << path << ": line " << ln <<
        " does not look like a line from a Kaldi command-line program's config file: should be of the form --x=y.";
    const size_t eq = line.find('=');
    const std::string key = line.substr(2, eq == std::string::npos ? std::string::npos : eq - 2), val = eq == std::string::npos ? "" : line.substr(eq + 1);
    if (!SetOption(key, val, eq != std::string::npos)) K3H_ERR << "Invalid option " << line << " in config file " << path;
  }
}

int ParseOptions::Read(int argc, const char *const *argv) {
  argv_.assign(argv, argv + argc);
  if (argc > 0) { const char *b = strrchr(argv[0], '/'); g_program = b ? b + 1 : argv[0]; }
  std::string config; Register("config", &config, "Configuration file to read (this option may be repeated)");
  bool help = false, print_args = true; Register("help", &help, "Print out usage message"); Register("print-args", &print_args, "Print the command line arguments (to stderr)");
  Register("verbose", &g_verbose, "Verbose level (higher->more logging)");
  // util/parse-options.cc:329-371: a first pass reads every --config file (and answers --help), the second pass applies the
  // command-line options, so the command line always wins whatever the order of the arguments
  auto split = [](const std::string &a, std::string *key, std::string *val) {
    const size_t eq = a.find('=');
    *key = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2); *val = eq == std::string::npos ? "" : a.substr(eq + 1);
    return eq != std::string::npos;
  };
  for (int k = 1; k < argc; k++) {
    const std::string a = argv[k];
    if (a == "--" || a.compare(0, 2, "--") != 0) break;
    std::string key, val; const bool has = split(a, &key, &val);
    if (Normalize(key) == "config") { if (!has || val.empty()) K3H_ERR << "Invalid option " << a; ReadConfigFile(val); }
    if (Normalize(key) == "help") { PrintUsage(); exit(0); }
  }
  int i = 1;
  for (; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--") { i++; break; }
    if (a.compare(0, 2, "--") != 0) break;
    std::string key, val; const bool has = split(a, &key, &val);
    if (!SetOption(key, val, has)) { PrintUsage(true); K3H_ERR << "Invalid option " << a; }
  }
  (void)help;
  for (; i < argc; i++) args_.push_back(argv[i]);
  if (print_args) { std::ostringstream o; for (int k = 0; k < argc; k++) o << argv[k] << " "; std::cerr << o.str() << "\n"; }
  return i;
}
const std::string &ParseOptions::GetArg(int i) const {
  if (i < 1 || i > (int)args_.size()) K3H_ERR << "ParseOptions::GetArg, invalid index " << i;
  return args_[i - 1];
}
void ParseOptions::PrintUsage(bool print_command_line) const {
  std::cerr << "\n" << usage_ << "\nOptions:\n";
  for (auto &kv : opts_) std::cerr << "  --" << kv.first << " : " << kv.second.doc << " (default = " << kv.second.default_str << ")\n";
  if (print_command_line) { std::cerr << "Command line was: "; for (auto &a : argv_) std::cerr << a << " "; std::cerr << "\n"; }
}

// ------------------------------------------------------------------------------------------------ streams ----
static std::string Trim(const std::string &s) {
  const size_t b = s.find_first_not_of(" \t\r\n");
  if (b == std::string::npos) return "";
  return s.substr(b, s.find_last_not_of(" \t\r\n") - b + 1);
}
std::shared_ptr<FILE> OpenInput(const std::string &rx_in) {
  const std::string rx = Trim(rx_in);
  if (rx.empty()) K3H_ERR << "empty rxfilename";
  if (rx == "-") return std::shared_ptr<FILE>(stdin, [](FILE *) {});
  if (rx.back() == '|') {
    FILE *f = popen(rx.substr(0, rx.size() - 1).c_str(), "r");
    if (!f) K3H_ERR << "Failed opening pipe for reading, command is: " << rx;
    return std::shared_ptr<FILE>(f, [](FILE *p) { pclose(p); });
  }
  FILE *f = fopen(rx.c_str(), "rb");
  if (!f) K3H_ERR << "Failed to open input file " << rx << ": " << strerror(errno);
  return std::shared_ptr<FILE>(f, [](FILE *p) { fclose(p); });
}
std::shared_ptr<FILE> OpenOutput(const std::string &wx_in) {
  const std::string wx = Trim(wx_in);
  if (wx.empty() || wx == "-") return std::shared_ptr<FILE>(stdout, [](FILE *p) { fflush(p); });
  if (wx[0] == '|') {
    FILE *f = popen(wx.substr(1).c_str(), "w");
    if (!f) K3H_ERR << "Failed opening pipe for writing, command is: " << wx;
    return std::shared_ptr<FILE>(f, [](FILE *p) { pclose(p); });
  }
  FILE *f = fopen(wx.c_str(), "wb");
  if (!f) K3H_ERR << "Failed to open output file " << wx << ": " << strerror(errno);
  return std::shared_ptr<FILE>(f, [](FILE *p) { fclose(p); });
}
std::string ReadWholeInput(const std::string &rx) {
  auto f = OpenInput(rx); std::string out; char buf[1 << 16]; size_t n;
  while ((n = fread(buf, 1, sizeof buf, f.get())) > 0) out.append(buf, n);
  return out;
}

std::vector<std::pair<std::string, std::string>> ReadScp(const std::string &rspecifier) {
  const size_t colon = rspecifier.find(':');
  if (colon == std::string::npos || rspecifier.compare(0, 3, "scp") != 0)
    K3H_ERR << "Invalid rspecifier " << rspecifier << " (this program reads script files: scp:<file>)";
  const std::string txt = ReadWholeInput(rspecifier.substr(colon + 1));
  std::vector<std::pair<std::string, std::string>> out; std::istringstream is(txt); std::string line;
  while (std::getline(is, line)) {
    line = Trim(line); if (line.empty()) continue;
    const size_t sp = line.find_first_of(" \t");
    if (sp == std::string::npos) K3H_ERR << "Invalid line in script file: " << line;
    out.push_back({line.substr(0, sp), Trim(line.substr(sp + 1))});
  }
  return out;
}

// ------------------------------------------------------------------------------------------------ wave ----
Wave ReadWave(const std::string &rxfilename, int channel, int *num_channels) {      // feat/wave-reader.cc:113-330 (PCM, 16 bit, incl. WAVE_FORMAT_EXTENSIBLE with a PCM sub-format)
  const std::string b = ReadWholeInput(rxfilename);
  auto u32 = [&](size_t p) { uint32_t v; memcpy(&v, b.data() + p, 4); return v; };
  auto u16 = [&](size_t p) { uint16_t v; memcpy(&v, b.data() + p, 2); return v; };
  if (b.size() < 44 || b.compare(0, 4, "RIFF") != 0 || b.compare(8, 4, "WAVE") != 0) K3H_ERR << "WaveData: expected RIFF/WAVE header in " << rxfilename;
  size_t pos = 12; int channels = 0, bits = 0; Wave w;
  while (pos + 8 <= b.size()) {
    const std::string id = b.substr(pos, 4); const uint32_t sz = u32(pos + 4);
    if (id == "fmt ") {
      const unsigned fmt = u16(pos + 8);      // 1 = PCM; 0xFFFE = extensible: the sub-format GUID's first two bytes carry the real format (wave-reader.cc:176-205)
      if (fmt != 1 && !(fmt == 0xFFFE && sz >= 26 && u16(pos + 8 + 24) == 1)) K3H_ERR << "WaveData: can read only PCM data, audio_format is not 1 in " << rxfilename;
      channels = u16(pos + 10); w.samp_freq = (float)u32(pos + 12); bits = u16(pos + 22);
      if (bits != 16 || channels < 1) K3H_ERR << "WaveData: unsupported bits_per_sample / channels = " << bits << " / " << channels;
    } else if (id == "data") {
      if (!channels) K3H_ERR << "WaveData: data chunk before fmt chunk in " << rxfilename;
      size_t n = std::min<size_t>(sz == 0xFFFFFFFFu || sz == 0 ? b.size() - pos - 8 : sz, b.size() - pos - 8) / (2 * channels);
      if (num_channels) *num_channels = channels;
      if (channel < 0 || channel >= channels) K3H_ERR << "WaveData: channel " << channel << " requested but the file has " << channels << " in " << rxfilename;
      w.samples.resize(n);
      const int16_t *s = reinterpret_cast<const int16_t *>(b.data() + pos + 8);
      for (size_t i = 0; i < n; i++) { int16_t v; memcpy(&v, s + i * channels + channel, 2); w.samples[i] = (float)v; }
      return w;
    }
    pos += 8 + sz + (sz & 1);
  }
  K3H_ERR << "WaveData: no data chunk in " << rxfilename;
  return w;
}

// ------------------------------------------------------------------------------------------------ Kaldi basic IO ----
namespace {
struct In {          // in-memory reader for Kaldi's text/binary object format (base/io-funcs-inl.h)
  const std::string &b; size_t p = 0; bool binary = false;
  explicit In(const std::string &buf) : b(buf) { if (b.size() >= 2 && b[0] == '\0' && b[1] == 'B') { binary = true; p = 2; } }
  void SkipWs() { while (p < b.size() && isspace((unsigned char)b[p])) p++; }
  std::string Token() {
    SkipWs(); const size_t s = p;
    while (p < b.size() && !isspace((unsigned char)b[p])) p++;
    if (p == s) K3H_ERR << "unexpected end of file while reading a token";
    std::string t = b.substr(s, p - s);
    if (binary && p < b.size()) p++;          // the single space after a token
    return t;
  }
  void Expect(const char *tok) { const std::string t = Token(); if (t != tok) K3H_ERR << "Expected token " << tok << ", got " << t; }
  template <class T> T Basic() {
    if (binary) {
      if (p + 1 + sizeof(T) > b.size()) K3H_ERR << "unexpected end of file";
      const int sz = (signed char)b[p];
      if (sz != (int)sizeof(T)) K3H_ERR << "ReadBasicType: did not get expected integer type, " << sz << " vs. " << sizeof(T);
      T v; memcpy(&v, b.data() + p + 1, sizeof(T)); p += 1 + sizeof(T); return v;
    }
    const std::string t = Token(); char *e = nullptr;
    const double d = strtod(t.c_str(), &e);
    if (*e) K3H_ERR << "ReadBasicType: could not parse \"" << t << "\"";
    return (T)d;
  }
  std::vector<int32_t> IntVector() {
    std::vector<int32_t> v;
    if (binary) {
      const int sz = (signed char)b[p]; if (sz != 4) K3H_ERR << "ReadIntegerVector: expected 4-byte integers, got " << sz;
      int32_t n; memcpy(&n, b.data() + p + 1, 4); p += 5;
      v.resize(n); if (n) memcpy(v.data(), b.data() + p, 4 * (size_t)n); p += 4 * (size_t)n;
    } else {
      Expect("["); for (std::string t = Token(); t != "]"; t = Token()) v.push_back(atoi(t.c_str()));
    }
    return v;
  }
  std::vector<float> FloatVector() {
    std::vector<float> v;
    if (binary) {
      const std::string t = Token();
      if (t != "FV" && t != "DV") K3H_ERR << "expected a vector (FV/DV), got " << t;
      const int32_t n = Basic<int32_t>(); v.resize(n);
      if (t == "FV") { if (n) memcpy(v.data(), b.data() + p, 4 * (size_t)n); p += 4 * (size_t)n; }
      else { for (int32_t i = 0; i < n; i++) { double d; memcpy(&d, b.data() + p + 8 * (size_t)i, 8); v[i] = (float)d; } p += 8 * (size_t)n; }
    } else {
      Expect("["); for (std::string t = Token(); t != "]"; t = Token()) v.push_back((float)atof(t.c_str()));
    }
    return v;
  }
  // binary Kaldi containers as doubles: "FV"/"DV" n, "FM"/"DM" rows cols, "FP"/"DP" n (packed lower triangle, n(n+1)/2 values)
  std::vector<double> BinaryDoubles(const char *what, const char *f_tok, const char *d_tok, int ndims, int32_t *dims) {
    if (!binary) K3H_ERR << "reading " << what << ": text-format models are not supported, copy the model with --binary=true";
    const std::string t = Token();
    if (t != f_tok && t != d_tok) K3H_ERR << "reading " << what << ": expected " << f_tok << " or " << d_tok << ", got " << t;
    size_t n = 1;
    for (int i = 0; i < ndims; i++) { dims[i] = Basic<int32_t>(); if (dims[i] < 0) K3H_ERR << "reading " << what << ": negative dimension"; n *= (size_t)dims[i]; }
    if (ndims == 1 && t[1] == 'P') n = (size_t)dims[0] * ((size_t)dims[0] + 1) / 2;
    std::vector<double> v(n); const size_t w = t[0] == 'F' ? 4 : 8;
    if (p + w * n > b.size()) K3H_ERR << "reading " << what << ": unexpected end of file";
    for (size_t i = 0; i < n; i++) { if (w == 4) { float f; memcpy(&f, b.data() + p + 4 * i, 4); v[i] = f; } else memcpy(&v[i], b.data() + p + 8 * i, 8); }
    p += w * n;
    return v;
  }
};
struct HmmState { int32_t fwd_pdf_class = -1, self_pdf_class = -1; std::vector<std::pair<int32_t, float>> trans; };
}  // namespace

TransitionInfo ReadTransitionModel(const std::string &mdl_rxfilename) {
  const std::string buf = ReadWholeInput(mdl_rxfilename);
  In in(buf);
  in.Expect("<TransitionModel>"); in.Expect("<Topology>");
  std::vector<int32_t> phone2idx; std::vector<std::vector<HmmState>> entries;
  if (!in.binary) {                                                       // hmm-topology.cc:33-118
    for (std::string t = in.Token(); t != "</Topology>"; t = in.Token()) {
      if (t != "<TopologyEntry>") K3H_ERR << "Reading HmmTopology object, expected </Topology> or <TopologyEntry>, got " << t;
      in.Expect("<ForPhones>"); std::vector<int32_t> phones;
      for (std::string s = in.Token(); s != "</ForPhones>"; s = in.Token()) phones.push_back(atoi(s.c_str()));
      std::vector<HmmState> entry;
      std::string tok = in.Token();
      while (tok != "</TopologyEntry>") {
        if (tok != "<State>") K3H_ERR << "Expected </TopologyEntry> or <State>, got instead " << tok;
        const int32_t st = in.Basic<int32_t>();
        if (st != (int32_t)entry.size()) K3H_ERR << "States are expected to be in order from zero";
        HmmState hs; tok = in.Token();
        if (tok == "<PdfClass>") { hs.fwd_pdf_class = hs.self_pdf_class = in.Basic<int32_t>(); tok = in.Token(); }
        else if (tok == "<ForwardPdfClass>") { hs.fwd_pdf_class = in.Basic<int32_t>(); in.Expect("<SelfLoopPdfClass>"); hs.self_pdf_class = in.Basic<int32_t>(); tok = in.Token(); }
        while (tok == "<Transition>") { const int32_t d = in.Basic<int32_t>(); const float pr = in.Basic<float>(); hs.trans.push_back({d, pr}); tok = in.Token(); }
        if (tok != "</State>") K3H_ERR << "Expected </State>, got instead " << tok;
        entry.push_back(hs); tok = in.Token();
      }
      for (int32_t ph : phones) { if ((int32_t)phone2idx.size() <= ph) phone2idx.resize(ph + 1, -1); phone2idx[ph] = (int32_t)entries.size(); }
      entries.push_back(entry);
    }
  } else {                                                                // hmm-topology.cc:119-152
    (void)in.IntVector(); phone2idx = in.IntVector();
    int32_t sz = in.Basic<int32_t>(); bool is_hmm = true;
    if (sz == -1) { is_hmm = false; sz = in.Basic<int32_t>(); }
    entries.resize(sz);
    for (auto &e : entries) {
      e.resize(in.Basic<int32_t>());
      for (auto &hs : e) {
        hs.fwd_pdf_class = in.Basic<int32_t>(); hs.self_pdf_class = is_hmm ? hs.fwd_pdf_class : in.Basic<int32_t>();
        hs.trans.resize(in.Basic<int32_t>());
        for (auto &tr : hs.trans) { tr.first = in.Basic<int32_t>(); tr.second = in.Basic<float>(); }
      }
    }
    in.Expect("</Topology>");
  }
  const std::string tok = in.Token();                                    // transition-model.cc:229-243
  if (tok != "<Triples>" && tok != "<Tuples>") K3H_ERR << "TransitionModel: expected <Triples> or <Tuples>, got " << tok;
  const int32_t n = in.Basic<int32_t>();
  TransitionInfo ti; ti.id2pdf.assign(1, 0); ti.id2phone.assign(1, 0); ti.self_loop.assign(1, 0); ti.phone_start.assign(1, 0); ti.is_final.assign(1, 0);
  for (int32_t i = 0; i < n; i++) {                                       // ComputeDerived (:90-124): transition-ids in tuple order
    const int32_t phone = in.Basic<int32_t>(), hs = in.Basic<int32_t>(), fpdf = in.Basic<int32_t>();
    const int32_t spdf = tok == "<Tuples>" ? in.Basic<int32_t>() : fpdf;
    if (phone <= 0 || phone >= (int32_t)phone2idx.size() || phone2idx[phone] < 0) K3H_ERR << "TransitionModel: phone " << phone << " has no topology entry";
    const auto &entry = entries[phone2idx[phone]];
    if (hs < 0 || hs >= (int32_t)entry.size()) K3H_ERR << "TransitionModel: bad hmm-state " << hs;
    for (const auto &tr : entry[hs].trans) {
      ti.id2pdf.push_back(tr.first == hs ? spdf : fpdf);     // IsSelfLoop
      // IsFinal (transition-model.cc:511-518)
      ti.id2phone.push_back(phone);
      ti.self_loop.push_back(tr.first == hs);
      ti.phone_start.push_back(hs == 0);
      ti.is_final.push_back(tr.first + 1 == (int32_t)entry.size());
    }
    ti.num_pdfs = std::max(ti.num_pdfs, 1 + std::max(fpdf, spdf));
  }
  const std::string endtok = in.Token();
  if (endtok != "</Triples>" && endtok != "</Tuples>") K3H_ERR << "TransitionModel: expected </Triples> or </Tuples>, got " << endtok;
  in.Expect("<LogProbs>");
  const std::vector<float> lp = in.FloatVector();
  if (lp.size() != ti.id2pdf.size()) K3H_ERR << "TransitionModel: " << lp.size() - 1 << " log-probs for " << ti.id2pdf.size() - 1 << " transition-ids";
  in.Expect("</LogProbs>"); in.Expect("</TransitionModel>");
  return ti;
}

// ------------------------------------------------------------------------------------------------ i-vector extractor models ----
DiagGmmModel ReadDiagGmm(const std::string &rxfilename) {               // DiagGmm::Read (gmm/diag-gmm.cc:728-756)
  const std::string buf = ReadWholeInput(rxfilename); In in(buf); DiagGmmModel g; int32_t d[2];
  std::string tok = in.Token();
  if (tok != "<DiagGMM>" && tok != "<DiagGMMBegin>") K3H_ERR << "Expected <DiagGMM>, got " << tok;
  tok = in.Token();
  if (tok == "<GCONSTS>") { g.gconsts = in.BinaryDoubles("DiagGmm gconsts", "FV", "DV", 1, d); in.Expect("<WEIGHTS>"); }
  else if (tok != "<WEIGHTS>") K3H_ERR << "DiagGmm::Read, expected <WEIGHTS> or <GCONSTS>, got " << tok;
  g.weights = in.BinaryDoubles("DiagGmm weights", "FV", "DV", 1, d); g.num_gauss = d[0];
  in.Expect("<MEANS_INVVARS>"); g.means_invvars = in.BinaryDoubles("DiagGmm means_invvars", "FM", "DM", 2, d);
  if (d[0] != g.num_gauss) K3H_ERR << "DiagGmm: " << d[0] << " mean rows for " << g.num_gauss << " weights";
  g.dim = d[1];
  in.Expect("<INV_VARS>"); g.inv_vars = in.BinaryDoubles("DiagGmm inv_vars", "FM", "DM", 2, d);
  if (d[0] != g.num_gauss || d[1] != g.dim) K3H_ERR << "DiagGmm: inv_vars is " << d[0] << " x " << d[1] << ", expected " << g.num_gauss << " x " << g.dim;
  tok = in.Token();
  if (tok != "</DiagGMM>" && tok != "<DiagGMMEnd>") K3H_ERR << "Expected </DiagGMM>, got " << tok;
  return g;
}
IvectorExtractorModel ReadIvectorExtractor(const std::string &rxfilename) {     // IvectorExtractor::Read (ivector/ivector-extractor.cc:828-849)
  const std::string buf = ReadWholeInput(rxfilename); In in(buf); IvectorExtractorModel m; int32_t d[2];
  in.Expect("<IvectorExtractor>"); in.Expect("<w>");
  m.w = in.BinaryDoubles("IvectorExtractor w", "FM", "DM", 2, d); m.w_rows = d[0]; m.w_cols = d[1];
  in.Expect("<w_vec>"); m.w_vec = in.BinaryDoubles("IvectorExtractor w_vec", "FV", "DV", 1, d);
  in.Expect("<M>"); m.num_gauss = in.Basic<int32_t>();
  if (m.num_gauss <= 0) K3H_ERR << "IvectorExtractor: bad number of Gaussians " << m.num_gauss;
  for (int32_t i = 0; i < m.num_gauss; i++) {
    std::vector<double> mi = in.BinaryDoubles("IvectorExtractor M", "FM", "DM", 2, d);
    if (i == 0) { m.feat_dim = d[0]; m.ivector_dim = d[1]; } else if (d[0] != m.feat_dim || d[1] != m.ivector_dim) K3H_ERR << "IvectorExtractor: M matrices of different sizes";
    m.M.insert(m.M.end(), mi.begin(), mi.end());
  }
  in.Expect("<SigmaInv>");
  for (int32_t i = 0; i < m.num_gauss; i++) {
    std::vector<double> si = in.BinaryDoubles("IvectorExtractor SigmaInv", "FP", "DP", 1, d);
    if (d[0] != m.feat_dim) K3H_ERR << "IvectorExtractor: SigmaInv of dimension " << d[0] << ", expected " << m.feat_dim;
    m.sigma_inv.insert(m.sigma_inv.end(), si.begin(), si.end());
  }
  in.Expect("<IvectorOffset>"); m.prior_offset = in.Basic<double>();
  in.Expect("</IvectorExtractor>");
  return m;
}

// ------------------------------------------------------------------------------------------------ OpenFst binary ----
namespace {
const int32_t kFstMagic = 2125659606;
struct Bin {
  const std::string &b; size_t p = 0;
  explicit Bin(const std::string &buf) : b(buf) {}
  template <class T> T Get() { if (p + sizeof(T) > b.size()) K3H_ERR << "unexpected end of FST file"; T v; memcpy(&v, b.data() + p, sizeof(T)); p += sizeof(T); return v; }
  std::string Str() { const int32_t n = Get<int32_t>(); if (n < 0 || p + n > b.size()) K3H_ERR << "corrupt FST header"; std::string s = b.substr(p, n); p += n; return s; }
  void Align(size_t a) { p = (p + a - 1) / a * a; }
};
void PutStr(std::string *o, const std::string &s) { const int32_t n = (int32_t)s.size(); o->append((const char *)&n, 4); o->append(s); }
template <class T> void Put(std::string *o, T v) { o->append((const char *)&v, sizeof(T)); }
}  // namespace

HostFst ReadFstKaldiGeneric(const std::string &rxfilename) {
  const std::string buf = ReadWholeInput(rxfilename);
  Bin in(buf);
  if (in.Get<int32_t>() != kFstMagic) K3H_ERR << "Reading FST: error reading FST header from " << rxfilename << " (text FSTs are not supported: compile it with fstcompile)";
  const std::string ftype = in.Str(), atype = in.Str();
  const int32_t version = in.Get<int32_t>(), flags = in.Get<int32_t>(); (void)in.Get<uint64_t>();
  const int64_t start = in.Get<int64_t>(), ns = in.Get<int64_t>(), na = in.Get<int64_t>();
  if (atype != "standard") K3H_ERR << "FST with arc type " << atype << " is not supported.";
  if (flags & 3) K3H_ERR << "FST files with embedded symbol tables are not supported (write it with --keep_isymbols=false --keep_osymbols=false)";
  HostFst f; f.start = (int32_t)start;
  if (ftype == "vector") {
    f.arc_offsets.push_back(0);
    for (int64_t s = 0; s < ns; s++) {
      f.final_cost.push_back(in.Get<float>());
      const int64_t n = in.Get<int64_t>();
      for (int64_t a = 0; a < n; a++) {
        f.ilabel.push_back(in.Get<int32_t>());
        f.olabel.push_back(in.Get<int32_t>());
        f.weight.push_back(in.Get<float>());
        f.nextstate.push_back(in.Get<int32_t>());
      }
      f.arc_offsets.push_back((int32_t)f.ilabel.size());
    }
  } else if (ftype == "const") {                     // ConstFst<StdArc, uint32>: states {final, pos, narcs, niepsilons, noepsilons} then arcs
    const bool aligned = (flags & 4) != 0 || version == 1;
    if (aligned) in.Align(16);
    std::vector<uint32_t> pos(ns), cnt(ns);
    for (int64_t s = 0; s < ns; s++) {
      f.final_cost.push_back(in.Get<float>());
      pos[s] = in.Get<uint32_t>();
      cnt[s] = in.Get<uint32_t>();
      (void)in.Get<uint32_t>();
      (void)in.Get<uint32_t>();
    }
    if (aligned) in.Align(16);
    const size_t arcs0 = in.p;
    f.arc_offsets.push_back(0);
    for (int64_t s = 0; s < ns; s++) {
      in.p = arcs0 + 16 * (size_t)pos[s];
      for (uint32_t a = 0; a < cnt[s]; a++) {
        f.ilabel.push_back(in.Get<int32_t>());
        f.olabel.push_back(in.Get<int32_t>());
        f.weight.push_back(in.Get<float>());
        f.nextstate.push_back(in.Get<int32_t>());
      }
      f.arc_offsets.push_back((int32_t)f.ilabel.size());
    }
  } else K3H_ERR << "Reading FST: unsupported FST type: " << ftype;
  if ((int64_t)f.ilabel.size() != na && ftype == "const") K3H_ERR << "FST arc count mismatch";
  if (f.start < 0 || f.start >= f.NumStates()) K3H_ERR << "FST has no start state";
  return f;
}

void WriteFstVector(const HostFst &f, const std::string &wxfilename) {
  std::string o; Put(&o, kFstMagic); PutStr(&o, "vector"); PutStr(&o, "standard");
  Put<int32_t>(&o, 2); Put<int32_t>(&o, 0); Put<uint64_t>(&o, 3); Put<int64_t>(&o, f.start); Put<int64_t>(&o, f.NumStates()); Put<int64_t>(&o, (int64_t)f.ilabel.size());
  for (int32_t s = 0; s < f.NumStates(); s++) {
    Put(&o, f.final_cost[s]); Put<int64_t>(&o, f.arc_offsets[s + 1] - f.arc_offsets[s]);
    for (int32_t a = f.arc_offsets[s]; a < f.arc_offsets[s + 1]; a++) { Put(&o, f.ilabel[a]); Put(&o, f.olabel[a]); Put(&o, f.weight[a]); Put(&o, f.nextstate[a]); }
  }
  auto out = OpenOutput(wxfilename); fwrite(o.data(), 1, o.size(), out.get());
}

// ------------------------------------------------------------------------------------------------ lattices ----
void Connect(Lattice *lat) {
  const int32_t n = lat->NumStates(); const size_t na = lat->arc_src.size();
  if (n == 0) return;
  std::vector<int32_t> foff(n + 1, 0), roff(n + 1, 0);
  for (size_t a = 0; a < na; a++) { foff[lat->arc_src[a] + 1]++; roff[lat->arc_dst[a] + 1]++; }
  for (int32_t s = 0; s < n; s++) { foff[s + 1] += foff[s]; roff[s + 1] += roff[s]; }
  std::vector<int32_t> fadj(na), radj(na), fp(foff.begin(), foff.end() - 1), rp(roff.begin(), roff.end() - 1);
  for (size_t a = 0; a < na; a++) { fadj[fp[lat->arc_src[a]]++] = lat->arc_dst[a]; radj[rp[lat->arc_dst[a]]++] = lat->arc_src[a]; }
  std::vector<char> acc(n, 0), co(n, 0); std::vector<int32_t> st;
  if (lat->start >= 0) { acc[lat->start] = 1; st.push_back(lat->start); }
  while (!st.empty()) {
    const int32_t s = st.back();
    st.pop_back();
    for (int32_t k = foff[s]; k < foff[s + 1]; k++) if (!acc[fadj[k]]) {
      acc[fadj[k]] = 1;
      st.push_back(fadj[k]);
    }
  }
  for (int32_t s = 0; s < n; s++) if (std::isfinite(lat->st_final[s])) { co[s] = 1; st.push_back(s); }
  while (!st.empty()) { const int32_t s = st.back(); st.pop_back(); for (int32_t k = roff[s]; k < roff[s + 1]; k++) if (!co[radj[k]]) { co[radj[k]] = 1; st.push_back(radj[k]); } }
  std::vector<int32_t> newid(n, -1); int32_t m = 0;
  if (lat->start >= 0 && acc[lat->start] && co[lat->start]) newid[lat->start] = m++;     // start state first (state 0), like GetRawLattice
  for (int32_t s = 0; s < n; s++) if (newid[s] < 0 && acc[s] && co[s]) newid[s] = m++;
  Lattice o; o.st_frame.resize(m); o.st_state.resize(m); o.st_final.resize(m); o.start = m ? 0 : -1;
  if (!lat->st_final_ac.empty()) o.st_final_ac.resize(m);
  for (int32_t s = 0; s < n; s++) if (newid[s] >= 0) {
    o.st_frame[newid[s]] = lat->st_frame[s]; o.st_state[newid[s]] = lat->st_state[s]; o.st_final[newid[s]] = lat->st_final[s];
    if (!lat->st_final_ac.empty()) o.st_final_ac[newid[s]] = lat->st_final_ac[s];
  }
  for (size_t a = 0; a < na; a++) {
    const int32_t s = newid[lat->arc_src[a]], d = newid[lat->arc_dst[a]];
    if (s < 0 || d < 0) continue;
    o.arc_src.push_back(s); o.arc_dst.push_back(d); o.arc_ilabel.push_back(lat->arc_ilabel[a]); o.arc_olabel.push_back(lat->arc_olabel[a]);
    o.arc_graph.push_back(lat->arc_graph[a]); o.arc_ac.push_back(lat->arc_ac[a]);
  }
  *lat = std::move(o);
}
void ScaleAcoustic(Lattice *lat, double scale) { for (float &a : lat->arc_ac) a = (float)(a * scale); for (float &a : lat->st_final_ac) a = (float)(a * scale); }

TableWriter::TableWriter(const std::string &wspecifier) {
  const size_t colon = wspecifier.find(':');
  if (colon == std::string::npos || wspecifier.compare(0, 3, "ark") != 0) K3H_ERR << "Invalid wspecifier " << wspecifier << " (supported: ark:<wxfilename>, ark,t:<wxfilename>)";
  const std::string opts = wspecifier.substr(0, colon);
  if (opts.find("scp") != std::string::npos) K3H_ERR << "wspecifier " << wspecifier << ": ark,scp output is not supported by this program";
  binary_ = opts.find(",t") == std::string::npos;
  f_ = OpenOutput(wspecifier.substr(colon + 1));
}
void TableWriter::Flush() { fflush(f_.get()); }

static void PrintWeight(std::string *o, float g, float a) {     // operator<< of LatticeWeightTpl: "graph,acoustic"
  char buf[64];
  auto num = [&](float v) {
    if (std::isinf(v)) return std::string(v > 0 ? "Infinity" : "-Infinity");
    snprintf(buf, sizeof buf, "%g", (double)v);
    return std::string(buf);
  };
  *o += num(g); *o += ","; *o += num(a);
}
void TableWriter::WriteLattice(const std::string &key, const Lattice &lat) {
  // arcs grouped by source state in state order (a VectorFst stores them per state)
  const int32_t n = lat.NumStates(); const size_t na = lat.arc_src.size();
  std::vector<int32_t> off(n + 1, 0), order(na);
  for (size_t a = 0; a < na; a++) off[lat.arc_src[a] + 1]++;
  for (int32_t s = 0; s < n; s++) off[s + 1] += off[s];
  { std::vector<int32_t> p(off.begin(), off.end() - 1); for (size_t a = 0; a < na; a++) order[p[lat.arc_src[a]]++] = (int32_t)a; }
  std::string o = key + " ";
  if (!binary_) {          // lat/kaldi-lattice.cc:401-417: newline, fst::FstPrinter (start state first, tab separated, weight omitted when One), newline
    o += "\n";
    auto print_state = [&](int32_t s) {
      for (int32_t k = off[s]; k < off[s + 1]; k++) {
        const int32_t a = order[k];
        o += std::to_string(s) + "\t" + std::to_string(lat.arc_dst[a]) + "\t" + std::to_string(lat.arc_ilabel[a]) + "\t" + std::to_string(lat.arc_olabel[a]);
        if (!(lat.arc_graph[a] == 0.0f && lat.arc_ac[a] == 0.0f)) { o += "\t"; PrintWeight(&o, lat.arc_graph[a], lat.arc_ac[a]); }
        o += "\n";
      }
      if (std::isfinite(lat.st_final[s])) {
        const float fa = lat.st_final_ac.empty() ? 0.0f : lat.st_final_ac[s];
        o += std::to_string(s);
        if (lat.st_final[s] != 0.0f || fa != 0.0f) {
          o += "\t";
          PrintWeight(&o, lat.st_final[s], fa);
        }
        o += "\n";
      }
    };
    if (lat.start >= 0) print_state(lat.start);
    for (int32_t s = 0; s < n; s++) if (s != lat.start) print_state(s);
    o += "\n";
  } else {                  // VectorFst<LatticeArc>::Write: FstHeader (arc type "lattice4") + per state {final (2 floats), narcs, arcs of 20 B}
    Put(&o, kFstMagic); PutStr(&o, "vector"); PutStr(&o, "lattice4");
    Put<int32_t>(&o, 2); Put<int32_t>(&o, 0); Put<uint64_t>(&o, 3); Put<int64_t>(&o, lat.start); Put<int64_t>(&o, n); Put<int64_t>(&o, (int64_t)na);
    const float inf = std::numeric_limits<float>::infinity();
    for (int32_t s = 0; s < n; s++) {
      const bool fin = std::isfinite(lat.st_final[s]);
      Put<float>(&o, fin ? lat.st_final[s] : inf); Put<float>(&o, fin ? (lat.st_final_ac.empty() ? 0.0f : lat.st_final_ac[s]) : inf); Put<int64_t>(&o, off[s + 1] - off[s]);
      for (int32_t k = off[s]; k < off[s + 1]; k++) {
        const int32_t a = order[k];
        Put(&o, lat.arc_ilabel[a]);
        Put(&o, lat.arc_olabel[a]);
        Put(&o, lat.arc_graph[a]);
        Put(&o, lat.arc_ac[a]);
        Put(&o, lat.arc_dst[a]);
      }
    }
  }
  if (fwrite(o.data(), 1, o.size(), f_.get()) != o.size()) K3H_ERR << "Write failure on lattice " << key;
}

void TableWriter::WriteMatrix(const std::string &key, const float *data, int32_t rows, int32_t cols, int64_t stride) {
  std::string o = key + " ";
  if (binary_) {          // "\0B" "FM " \4 rows \4 cols data   (matrix/kaldi-matrix.cc:1382-1400)
    o.append("\0B", 2); o += "FM "; o.push_back(4); Put(&o, rows); o.push_back(4); Put(&o, cols);
    for (int32_t r = 0; r < rows; r++) o.append((const char *)(data + (int64_t)r * stride), 4 * (size_t)cols);
  } else {
    o += " [";
    char buf[32];
    for (int32_t r = 0; r < rows; r++) { o += "\n  "; for (int32_t c = 0; c < cols; c++) { snprintf(buf, sizeof buf, "%g ", (double)data[(int64_t)r * stride + c]); o += buf; } }
    o += "]\n";
  }
  if (fwrite(o.data(), 1, o.size(), f_.get()) != o.size()) K3H_ERR << "Write failure on matrix " << key;
}
void TableWriter::WriteVector(const std::string &key, const float *data, int32_t dim) {
  std::string o = key + " ";
  if (binary_) { o.append("\0B", 2); o += "FV "; o.push_back(4); Put(&o, dim); o.append((const char *)data, 4 * (size_t)dim); }      // matrix/kaldi-vector.cc Vector::Write
  else { o += " [ "; char buf[32]; for (int32_t i = 0; i < dim; i++) { snprintf(buf, sizeof buf, "%g ", (double)data[i]); o += buf; } o += "]\n"; }
  if (fwrite(o.data(), 1, o.size(), f_.get()) != o.size()) K3H_ERR << "Write failure on vector " << key;
}


// ------------------------------------------------------------------------------------------------ matrices ----
namespace {
size_t ReadOneMatrix(const std::string &b, size_t p, Matrix *m, const std::string &what) {
  auto need = [&](size_t n) { if (p + n > b.size()) K3H_ERR << "unexpected end of data reading matrix " << what; };
  auto i32 = [&]() { need(5); if (b[p] != 4) K3H_ERR << "bad integer marker in matrix " << what; int32_t v; memcpy(&v, b.data() + p + 1, 4); p += 5; return v; };
  if (p + 2 <= b.size() && b[p] == '\0' && b[p + 1] == 'B') {
    p += 2; const size_t t0 = p; while (p < b.size() && b[p] != ' ') p++;
    const std::string tok = b.substr(t0, p - t0); p++;
    if (tok == "FM" || tok == "DM") {
      m->rows = i32(); m->cols = i32(); const size_t n = (size_t)m->rows * m->cols; m->data.resize(n);
      if (tok == "FM") { need(4 * n); memcpy(m->data.data(), b.data() + p, 4 * n); p += 4 * n; }
      else { need(8 * n); for (size_t i = 0; i < n; i++) { double d; memcpy(&d, b.data() + p + 8 * i, 8); m->data[i] = (float)d; } p += 8 * n; }
    } else if (tok == "FV" || tok == "DV") {      // a vector table read as one-row matrices (--ivectors)
      m->rows = 1; m->cols = i32(); const size_t n = (size_t)m->cols; m->data.resize(n);
      if (tok == "FV") { need(4 * n); memcpy(m->data.data(), b.data() + p, 4 * n); p += 4 * n; }
      else { need(8 * n); for (size_t i = 0; i < n; i++) { double d; memcpy(&d, b.data() + p + 8 * i, 8); m->data[i] = (float)d; } p += 8 * n; }
    } else if (tok == "CM" || tok == "CM2" || tok == "CM3") {
      need(16);
      float mn, range;
      int32_t nr, nc;
      memcpy(&mn, b.data() + p, 4);
      memcpy(&range, b.data() + p + 4, 4);
      memcpy(&nr, b.data() + p + 8, 4);
      memcpy(&nc, b.data() + p + 12, 4);
      p += 16;
      m->rows = nr; m->cols = nc; m->data.resize((size_t)nr * nc);
      if (tok == "CM") {                 // per-column percentile headers + column-major bytes (compressed-matrix.cc:626-648)
        need((size_t)nc * 8 + (size_t)nc * nr);
        const uint16_t *ch = reinterpret_cast<const uint16_t *>(b.data() + p); const uint8_t *bytes = reinterpret_cast<const uint8_t *>(b.data() + p + (size_t)nc * 8);
        for (int32_t c = 0; c < nc; c++) {
          uint16_t h[4]; memcpy(h, ch + 4 * c, 8);
          const float p0 = mn + range * 1.52590218966964e-05F * h[0], p25 = mn + range * 1.52590218966964e-05F * h[1],
              p75 = mn + range * 1.52590218966964e-05F * h[2], p100 = mn + range * 1.52590218966964e-05F * h[3];
          for (int32_t r = 0; r < nr; r++) {
            const uint8_t v = bytes[(size_t)c * nr + r]; float f;
            if (v <= 64) f = p0 + (p25 - p0) * v * (1 / 64.0);
            else if (v <= 192) f = p25 + (p75 - p25) * (v - 64) * (1 / 128.0);
            else f = p75 + (p100 - p75) * (v - 192) * (1 / 63.0);
            m->data[(size_t)r * nc + c] = f;
          }
        }
        p += (size_t)nc * 8 + (size_t)nc * nr;
      } else if (tok == "CM2") {
        need(2 * (size_t)nr * nc); const float inc = range * (1.0f / 65535.0f);
        for (size_t i = 0; i < (size_t)nr * nc; i++) { uint16_t v; memcpy(&v, b.data() + p + 2 * i, 2); m->data[i] = mn + v * inc; }
        p += 2 * (size_t)nr * nc;
      } else {
        need((size_t)nr * nc); const float inc = range * (1.0f / 255.0f);
        for (size_t i = 0; i < (size_t)nr * nc; i++) m->data[i] = mn + (uint8_t)b[p + i] * inc;
        p += (size_t)nr * nc;
      }
    } else K3H_ERR << "Expected a matrix (FM, DM, CM, CM2, CM3), got token " << tok << " reading " << what;
    return p;
  }
  // text: " [\n a b c\n d e f ]"
  while (p < b.size() && isspace((unsigned char)b[p])) p++;
  if (p >= b.size() || b[p] != '[') K3H_ERR << "Expected \"[\" reading text matrix " << what;
  p++; std::vector<float> row; m->rows = 0; m->cols = 0; m->data.clear();
  while (p < b.size()) {
    while (p < b.size() && (b[p] == ' ' || b[p] == '\t' || b[p] == '\r')) p++;
    if (p >= b.size()) break;
    if (b[p] == '\n' || b[p] == ']') {
      if (!row.empty()) {
        if (m->cols && (int32_t)row.size() != m->cols) K3H_ERR << "Inconsistent row lengths in text matrix " << what;
        m->cols = (int32_t)row.size();
        m->data.insert(m->data.end(), row.begin(), row.end());
        m->rows++;
        row.clear();
      }
      if (b[p++] == ']') { while (p < b.size() && b[p] != '\n') p++; if (p < b.size()) p++; return p; }
      continue;
    }
    char *e = nullptr; const float v = strtof(b.c_str() + p, &e);
    if (e == b.c_str() + p) K3H_ERR << "Bad number in text matrix " << what;
    row.push_back(v); p = e - b.c_str();
  }
  K3H_ERR << "Unterminated text matrix " << what;
  return p;
}
}  // namespace

MatrixD ReadDoubleMatrix(const std::string &rxfilename) {
  const std::string b = ReadWholeInput(rxfilename); MatrixD m; size_t p = 0;
  auto need = [&](size_t n) { if (p + n > b.size()) K3H_ERR << "unexpected end of data reading matrix from " << rxfilename; };
  if (b.size() >= 2 && b[0] == '\0' && b[1] == 'B') {
    p = 2; const size_t t0 = p; while (p < b.size() && b[p] != ' ') p++;
    const std::string tok = b.substr(t0, p - t0); p++;
    if (tok != "DM" && tok != "FM") K3H_ERR << "Expected a matrix (DM or FM) in " << rxfilename << ", got token " << tok;
    auto i32 = [&]() { need(5); if (b[p] != 4) K3H_ERR << "bad integer marker in " << rxfilename; int32_t v; memcpy(&v, b.data() + p + 1, 4); p += 5; return v; };
    m.rows = i32(); m.cols = i32(); const size_t n = (size_t)m.rows * m.cols; m.data.resize(n);
    if (tok == "DM") { need(8 * n); memcpy(m.data.data(), b.data() + p, 8 * n); }
    else { need(4 * n); for (size_t i = 0; i < n; i++) { float f; memcpy(&f, b.data() + p + 4 * i, 4); m.data[i] = f; } }
    return m;
  }
  while (p < b.size() && isspace((unsigned char)b[p])) p++;
  if (p >= b.size() || b[p] != '[') K3H_ERR << "Expected \"[\" reading text matrix from " << rxfilename;
  p++; std::vector<double> row;
  while (p < b.size()) {
    while (p < b.size() && (b[p] == ' ' || b[p] == '\t' || b[p] == '\r')) p++;
    if (p >= b.size()) break;
    if (b[p] == '\n' || b[p] == ']') {
      if (!row.empty()) {
        if (m.cols && (int32_t)row.size() != m.cols) K3H_ERR << "Inconsistent row lengths in text matrix " << rxfilename;
        m.cols = (int32_t)row.size();
        m.data.insert(m.data.end(), row.begin(), row.end());
        m.rows++;
        row.clear();
      }
      if (b[p++] == ']') return m;
      continue;
    }
    char *e = nullptr; const double v = strtod(b.c_str() + p, &e);
    if (e == b.c_str() + p) K3H_ERR << "Bad number in text matrix " << rxfilename;
    row.push_back(v); p = e - b.c_str();
  }
  K3H_ERR << "Unterminated text matrix " << rxfilename;
  return m;
}

std::vector<std::pair<std::string, std::vector<std::string>>> ReadTokenVectorTable(const std::string &rspecifier) {
  const size_t colon = rspecifier.find(':');
  if (colon == std::string::npos || rspecifier.compare(0, 3, "ark") != 0) K3H_ERR << "Expected an ark: rspecifier for a token-vector table, got " << rspecifier;
  std::istringstream in(ReadWholeInput(rspecifier.substr(colon + 1))); std::string line;
  std::vector<std::pair<std::string, std::vector<std::string>>> out;
  while (std::getline(in, line)) {
    std::istringstream ls(line); std::string key, tok; if (!(ls >> key)) continue;
    std::vector<std::string> v; while (ls >> tok) v.push_back(tok);
    out.push_back({key, std::move(v)});
  }
  return out;
}

std::vector<std::pair<std::string, Matrix>> ReadMatrixTable(const std::string &rspecifier) {
  const size_t colon = rspecifier.find(':');
  if (colon == std::string::npos) K3H_ERR << "Invalid rspecifier " << rspecifier;
  const std::string kind = rspecifier.substr(0, 3), rest = rspecifier.substr(colon + 1);
  std::vector<std::pair<std::string, Matrix>> out;
  if (kind == "ark") {
    const std::string b = ReadWholeInput(rest); size_t p = 0;
    while (true) {
      while (p < b.size() && isspace((unsigned char)b[p])) p++;
      if (p >= b.size()) break;
      const size_t k0 = p; while (p < b.size() && !isspace((unsigned char)b[p])) p++;
      const std::string key = b.substr(k0, p - k0); p++;
      Matrix m; p = ReadOneMatrix(b, p, &m, key); out.push_back({key, std::move(m)});
    }
  } else if (kind == "scp") {
    std::map<std::string, std::string> cache;
    for (auto &kv : ReadScp(rspecifier)) {
      std::string path = kv.second; size_t off = 0;
      const size_t c = path.rfind(':');
      if (c != std::string::npos && c + 1 < path.size() && path.find_first_not_of("0123456789", c + 1) == std::string::npos) {
        off = strtoull(path.c_str() + c + 1, nullptr, 10);
        path = path.substr(0, c);
      }
      if (!cache.count(path)) { if (cache.size() > 4) cache.clear(); cache[path] = ReadWholeInput(path); }
      Matrix m; ReadOneMatrix(cache[path], off, &m, kv.first); out.push_back({kv.first, std::move(m)});
    }
  } else K3H_ERR << "Invalid rspecifier " << rspecifier << " (supported: ark:, scp:)";
  return out;
}

bool BestPath(const Lattice &lat, std::vector<int32_t> *ali, std::vector<int32_t> *words, double *gcost, double *acost) {
  const int32_t n = lat.NumStates(); const size_t na = lat.arc_src.size();
  if (n == 0 || lat.start < 0) return false;
  std::vector<double> best(n, std::numeric_limits<double>::infinity()); std::vector<int64_t> back(n, -1);
  best[lat.start] = 0.0;
  std::vector<int32_t> order(na); for (size_t a = 0; a < na; a++) order[a] = (int32_t)a;
  std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return lat.st_frame[lat.arc_src[x]] < lat.st_frame[lat.arc_src[y]]; });
  for (bool changed = true; changed;) {       // arcs in frame order; epsilon chains inside a frame may need another sweep
    changed = false;
    for (int32_t a : order) {
      const double c = best[lat.arc_src[a]] + (double)lat.arc_graph[a] + (double)lat.arc_ac[a];
      if (c < best[lat.arc_dst[a]]) {
        best[lat.arc_dst[a]] = c;
        back[lat.arc_dst[a]] = a;
        changed = true;
      }
    }
  }
  int32_t end = -1; double bc = std::numeric_limits<double>::infinity();
  auto final_ac = [&](int32_t s) { return lat.st_final_ac.empty() ? 0.0 : (double)lat.st_final_ac[s]; };      // zero for the decoder's lattices
  for (int32_t s = 0; s < n; s++) if (std::isfinite(lat.st_final[s]) && best[s] + lat.st_final[s] + final_ac(s) < bc) { bc = best[s] + lat.st_final[s] + final_ac(s); end = s; }
  if (end < 0) return false;
  ali->clear(); words->clear(); *gcost = lat.st_final[end]; *acost = final_ac(end);
  for (int32_t s = end; s != lat.start;) {
    const int64_t a = back[s];
    if (lat.arc_ilabel[a]) ali->push_back(lat.arc_ilabel[a]);
    if (lat.arc_olabel[a]) words->push_back(lat.arc_olabel[a]);
    *gcost += lat.arc_graph[a];
    *acost += lat.arc_ac[a];
    s = lat.arc_src[a];
  }
  std::reverse(ali->begin(), ali->end()); std::reverse(words->begin(), words->end());
  return true;
}

void TableWriter::WriteInt32Vector(const std::string &key, const std::vector<int32_t> &v) {
  std::string o = key + " ";
  if (binary_) { o.append("\0B", 2); o.push_back(4); Put<int32_t>(&o, (int32_t)v.size()); for (int32_t x : v) { o.push_back(4); Put(&o, x); } }     // BasicVectorHolder::Write
  else { for (int32_t x : v) o += std::to_string(x) + " "; o += "\n"; }
  if (fwrite(o.data(), 1, o.size(), f_.get()) != o.size()) K3H_ERR << "Write failure on vector " << key;
}

// ---------------------------------------------------------------------------------------------------------------- i-vector extraction config
void IvectorExtractionInfo::Register(ParseOptions *po) {       // option names and meanings of OnlineIvectorExtractionConfig::Register (online2/online-ivector-feature.h:113-160)
  po->Register("lda-matrix", &lda_mat_rxfilename, "Filename of LDA matrix, e.g. final.mat; used for iVector extraction.");
  po->Register("global-cmvn-stats", &global_cmvn_stats_rxfilename,
      "(Extended) filename for global CMVN stats, used in iVector extraction, obtained for example from 'matrix-sum scp:data/train/cmvn.scp -'");
  po->Register("cmvn-config", &cmvn_config_rxfilename, "Configuration file for online CMVN features (e.g. conf/online_cmvn.conf), only used for iVector extraction.");
  po->Register("online-cmvn-iextractor", &online_cmvn_iextractor, "add online-cmvn to feature pipeline of ivector extractor (the statistics side).");
  po->Register("splice-config", &splice_config_rxfilename, "Configuration file for frame splicing (--left-context and --right-context options); used for iVector extraction.");
  po->Register("diag-ubm", &diag_ubm_rxfilename, "Filename of diagonal UBM used to obtain posteriors for iVector extraction, e.g. final.dubm");
  po->Register("ivector-extractor", &ivector_extractor_rxfilename, "Filename of iVector extractor, e.g. final.ie");
  po->Register("ivector-period", &ivector_period, "Frequency with which we extract iVectors for neural network adaptation");
  po->Register("num-gselect", &num_gselect, "Number of Gaussians to select for iVector extraction");
  po->Register("min-post", &min_post, "Threshold for posterior pruning in iVector extraction");
  po->Register("posterior-scale", &posterior_scale, "Scale for posteriors in iVector extraction (may be viewed as inverse of prior scale)");
  po->Register("max-count", &max_count, "Maximum data count we allow before we start scaling the prior term up (0 = off)");
  po->Register("num-cg-iters", &num_cg_iters, "Number of iterations of conjugate gradient descent to perform each time we re-estimate the iVector.");
  po->Register("use-most-recent-ivector", &use_most_recent_ivector, "If true, always use most recent available iVector, rather than the one for the designated frame.");
  po->Register("greedy-ivector-extractor", &greedy_ivector_extractor, "If true, 'read ahead' as many frames as we currently have available when extracting the iVector.");
  po->Register("max-remembered-frames", &max_remembered_frames, "The maximum number of frames of adaptation history that we carry through to later utterances of the same speaker");
}
void IvectorExtractionInfo::Init() {
  const char *note = "(note: this may be needed in the file supplied to --ivector-extractor-config)";
  if (lda_mat_rxfilename.empty()) K3H_ERR << "--lda-matrix option must be set " << note;
  { const MatrixD m = ReadDoubleMatrix(lda_mat_rxfilename); lda_rows = m.rows; lda_cols = m.cols; lda.assign(m.data.begin(), m.data.end()); }
  if (global_cmvn_stats_rxfilename.empty()) K3H_ERR << "--global-cmvn-stats option must be set " << note;
  global_cmvn_stats = ReadDoubleMatrix(global_cmvn_stats_rxfilename);
  if (cmvn_config_rxfilename.empty()) K3H_ERR << "--cmvn-config option must be set " << note;
  { ParseOptions po(""); po.Register("cmn-window", &cmn_window, ""); po.Register("global-frames", &global_frames, ""); po.Register("speaker-frames", &speaker_frames, "");
    po.Register("norm-vars", &normalize_variance, ""); po.Register("norm-means", &normalize_mean, ""); std::string skip; po.Register("skip-dims", &skip, "");
    po.ReadConfigFile(cmvn_config_rxfilename);
    if (!skip.empty()) K3H_ERR << "--skip-dims in the i-vector extractor's cmvn config is not supported"; }
  if (splice_config_rxfilename.empty()) K3H_ERR << "--splice-config option must be set " << note;
  { ParseOptions po(""); po.Register("left-context", &left_context, ""); po.Register("right-context", &right_context, ""); po.ReadConfigFile(splice_config_rxfilename); }
  if (diag_ubm_rxfilename.empty()) K3H_ERR << "--diag-ubm option must be set " << note;
  ubm = ReadDiagGmm(diag_ubm_rxfilename);
  if (ivector_extractor_rxfilename.empty()) K3H_ERR << "--ivector-extractor option must be set " << note;
  ie = ReadIvectorExtractor(ivector_extractor_rxfilename);
  // Check()
  if (global_cmvn_stats.rows != 2) K3H_ERR << "global CMVN stats must have two rows, got " << global_cmvn_stats.rows;
  const int32_t base = global_cmvn_stats.cols - 1, spliced = base * (left_context + 1 + right_context);
  if (lda_cols != spliced && lda_cols != spliced + 1) K3H_ERR << "LDA matrix has " << lda_cols << " columns, the spliced features have dimension " << spliced;
  if (lda_rows != ubm.dim) K3H_ERR << "LDA matrix has " << lda_rows << " rows, the diagonal UBM has dimension " << ubm.dim;
  if (ubm.dim != ie.feat_dim) K3H_ERR << "the diagonal UBM has dimension " << ubm.dim << ", the iVector extractor " << ie.feat_dim;
  if (ubm.num_gauss != ie.num_gauss) K3H_ERR << "the diagonal UBM has " << ubm.num_gauss << " Gaussians, the iVector extractor " << ie.num_gauss;
  if (ivector_period <= 0 || num_gselect <= 0 || !(min_post < 0.5f) || !(posterior_scale > 0.0f && posterior_scale <= 1.0f) || max_remembered_frames < 0.0f)
    K3H_ERR << "invalid value among --ivector-period, --num-gselect, --min-post, --posterior-scale, --max-remembered-frames";
}
IvectorExtractionInfo ReadIvectorExtractionConfig(const std::string &config_rxfilename) {
  IvectorExtractionInfo info; ParseOptions po(""); info.Register(&po); po.ReadConfigFile(config_rxfilename); info.Init(); return info;
}

}  // namespace k3host
