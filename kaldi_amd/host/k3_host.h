// k3_host.h -- host-side C++ glue above the C ABI of libk3hip.so: the pieces of Kaldi's CLI contract that the drop-in
// binaries need (SURVEY.md 8b "CLI" and "File formats" rows), restated, not linked -- OpenFst and Kaldi's table code are not
// available to link against.  Nothing here touches the GPU except through include/k3hip.h and hipMalloc/hipMemcpy.
//
//   ParseOptions            util/parse-options.{h,cc}: --name=value, --config=file, --help, typed Register(), exit codes
//   ReadScp / OpenInput     util/kaldi-table (scp: rspecifiers; rxfilenames "file", "-", "cmd |")
//   ReadWave                feat/wave-reader.{h,cc}: RIFF/WAVE PCM16
//   ReadTransitionModel     hmm/transition-model.cc:225-251 + hmm/hmm-topology.cc:31-158 (text and binary) -> tid -> pdf map
//   ReadFstKaldiGeneric     fstext/kaldi-fst-io.cc:51-107: OpenFst binary "vector"/"const" StdArc FST -> CSR
//   LatticeWriter           lat/kaldi-lattice.cc:392-420 + util/kaldi-holder: "ark:" / "ark,t:" Lattice (text = FstPrinter format)
//   MatrixWriter            matrix/kaldi-matrix.cc:1382-1400 ("FM" binary) / text
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <functional>
#include <string>
#include <vector>

namespace k3host {

extern std::string g_program;      // basename(argv[0]), used in the log prefix
extern int g_verbose;

struct LogLine {                   // KALDI_LOG / KALDI_WARN / KALDI_ERR look-alike: "LOG (prog:func():file:line) msg"
  std::ostringstream ss; const char *sev; bool fatal;
  LogLine(const char *severity, const char *func, const char *file, int line, bool is_fatal);
  ~LogLine() noexcept(false);
  template <class T> LogLine &operator<<(const T &v) { ss << v; return *this; }
};
struct FatalError : std::runtime_error { using std::runtime_error::runtime_error; };
#define K3H_LOG ::k3host::LogLine("LOG", __func__, __FILE__, __LINE__, false)
#define K3H_WARN ::k3host::LogLine("WARNING", __func__, __FILE__, __LINE__, false)
#define K3H_ERR ::k3host::LogLine("ERROR", __func__, __FILE__, __LINE__, true)
#define K3H_VLOG(n) if ((n) <= ::k3host::g_verbose) ::k3host::LogLine("VLOG", __func__, __FILE__, __LINE__, false)
#define K3H_CHECK_K3(expr) do { if ((expr) != 0) K3H_ERR << #expr << ": " << k3_last_error(); } while (0)
// k3_decoder_lattice_info for a batch in which single lanes may have failed: a lane that ran out of token / link capacity has its
// status in info[10 u + 2] (and 0 states / arcs); the other lanes of the batch are good and the job goes on (the reference decoder
// never fails a whole job for one utterance).  Only errors that are not per-lane (HIP failure, bad argument) are fatal.
#define K3H_LATTICE_INFO(dec, info_ptr) do { const int rc__ = k3_decoder_lattice_info((dec), (info_ptr)); \
    if (rc__ == K3_ERR_OVERFLOW) K3H_WARN << "some utterances of this batch exceeded the decoder capacities and will be reported as failed: " << k3_last_error(); \
    else if (rc__ != 0) K3H_ERR << "k3_decoder_lattice_info: " << k3_last_error(); } while (0)

class ParseOptions {
 public:
  explicit ParseOptions(const char *usage);
  void Register(const std::string &name, bool *p, const std::string &doc);
  void Register(const std::string &name, int32_t *p, const std::string &doc);
  void Register(const std::string &name, float *p, const std::string &doc);
  void Register(const std::string &name, double *p, const std::string &doc);
  void Register(const std::string &name, std::string *p, const std::string &doc);
  // returns the index of the first positional argument; exits(0) on --help; throws FatalError on an invalid option
  int Read(int argc, const char *const *argv);
  void ReadConfigFile(const std::string &path);
  int NumArgs() const { return (int)args_.size(); }
  const std::string &GetArg(int i) const;          // 1-based like Kaldi
  void PrintUsage(bool print_command_line = false) const;
  bool IsSet(const std::string &name) const { return set_.count(Normalize(name)) != 0; }
 private:
  enum Kind { kBool, kInt, kFloat, kDouble, kString };
  struct Opt { Kind kind; void *ptr; std::string doc, default_str; };
  static std::string Normalize(const std::string &n);
  void RegisterImpl(const std::string &name, Kind k, void *p, const std::string &doc, const std::string &def);
  bool SetOption(const std::string &key, const std::string &value, bool has_value);
  std::string usage_; std::map<std::string, Opt> opts_; std::map<std::string, int> set_; std::vector<std::string> args_, argv_;
};

// rxfilename: "path", "-" (stdin) or "command |"; the returned FILE is closed (fclose / pclose) by the deleter
std::shared_ptr<FILE> OpenInput(const std::string &rxfilename);
std::shared_ptr<FILE> OpenOutput(const std::string &wxfilename);    // "path", "-" or "| command"
std::string ReadWholeInput(const std::string &rxfilename);

// "scp:wav.scp" (options before the colon are ignored) -> (key, rxfilename) pairs in file order
std::vector<std::pair<std::string, std::string>> ReadScp(const std::string &rspecifier);

struct Wave { float samp_freq = 0; std::vector<float> samples; };   // channel 0, values in int16 range like WaveData
Wave ReadWave(const std::string &rxfilename, int channel = 0, int *num_channels = nullptr);      // one channel of the file (an out-of-range channel is an error)

// .mdl (text or binary): parses the TransitionModel in front of the nnet; id2pdf[0] is unused
// id2phone / self_loop / phone_start = TransitionIdToPhone, IsSelfLoop, TransitionIdIsStartOfPhone (hmm/transition-model.cc:790,925)
// is_final: TransitionModel::IsFinal (the transition enters the topology's last, non-emitting state)
struct TransitionInfo {
  int32_t num_pdfs = 0;
  std::vector<int32_t> id2pdf, id2phone;
  std::vector<char> self_loop, phone_start, is_final;
};
TransitionInfo ReadTransitionModel(const std::string &mdl_rxfilename);

// model files of the online i-vector extractor (binary files only), values widened to double, matrices row-major:
// DiagGmm (gmm/diag-gmm.h: the UBM that selects Gaussians and gives posteriors) and IvectorExtractor (ivector/ivector-extractor.h:
// per Gaussian i the projection M_i [feat_dim x ivector_dim] and the packed lower triangle of Sigma_i^-1, the weight projection w,
// the prior offset).  Readers only, for the GPU path to come.
struct DiagGmmModel { int32_t num_gauss = 0, dim = 0; std::vector<double> gconsts, weights, means_invvars, inv_vars; };
DiagGmmModel ReadDiagGmm(const std::string &rxfilename);
struct IvectorExtractorModel { int32_t num_gauss = 0, feat_dim = 0, ivector_dim = 0, w_rows = 0, w_cols = 0; double prior_offset = 0; std::vector<double> w, w_vec, M, sigma_inv; };
IvectorExtractorModel ReadIvectorExtractor(const std::string &rxfilename);

struct HostFst {          // generic CSR in FST arc order (input of k3_fst_create)
  int32_t start = -1; std::vector<int32_t> arc_offsets, ilabel, olabel, nextstate; std::vector<float> weight, final_cost;
  int32_t NumStates() const { return (int32_t)final_cost.size(); }
};
HostFst ReadFstKaldiGeneric(const std::string &rxfilename);
void WriteFstVector(const HostFst &fst, const std::string &wxfilename);

struct Lattice {          // one utterance of k3_decoder_get_raw_lattices, already trimmed by Connect()
  std::vector<int32_t> st_frame, st_state; std::vector<float> st_final;
  std::vector<float> st_final_ac;          // acoustic part of the final weights; empty = all zero (what the decoder produces)
  std::vector<int32_t> arc_src, arc_dst, arc_ilabel, arc_olabel; std::vector<float> arc_graph, arc_ac;
  int32_t start = -1;
  int32_t NumStates() const { return (int32_t)st_final.size(); }
};
// fst::Connect (decoder/decoder-wrappers.cc:353): drop states that are not accessible from `start` or not co-accessible to a
// final state, renumber with the start state first
void Connect(Lattice *lat);
// scales acoustic costs by 1/acoustic_scale (ScaleLattice(AcousticLatticeScale(1/acwt)), decoder-wrappers.cc:366-370)
void ScaleAcoustic(Lattice *lat, double scale);

// ---- word-level lattice determinization (k3_lattice.cc) ----------------------------------------------------------------------
// CompactLattice (lat/kaldi-lattice.h:46): acceptor over word labels whose weights carry (graph, acoustic) costs and the
// transition-id string of the best alignment for that stretch of words.
struct CompactLattice {
  int32_t start = -1;
  std::vector<char> is_final; std::vector<float> fin_graph, fin_ac; std::vector<std::vector<int32_t>> fin_str;
  std::vector<int32_t> arc_src, arc_dst, arc_label; std::vector<float> arc_graph, arc_ac; std::vector<std::vector<int32_t>> arc_str;
  int32_t NumStates() const { return (int32_t)is_final.size(); }
  int32_t AddState() { is_final.push_back(0); fin_graph.push_back(0); fin_ac.push_back(0); fin_str.emplace_back(); return NumStates() - 1; }
};
struct DeterminizeLatticePrunedOptions {     // lat/determinize-lattice-pruned.h:113-148, same defaults
  float delta = 1.0f / 1024.0f; int32_t max_mem = -1, max_loop = -1, max_states = -1, max_arcs = -1; float retry_cutoff = 0.5f;
};
// DeterminizeLatticePruned (lat/determinize-lattice-pruned.cc:1190-1236) behind the preparation its callers do
// (Invert, TopSort, ArcSort on the word label: lattice-determinize-pruned.cc:104-112, determinize-lattice-pruned.cc:1479-1496)
// and followed by fst::Connect.  Keeps, for every word sequence whose best path is within `beam` of the best path, that best
// path only (costs and transition-id string).  Returns false when a limit of `opts` stopped it early (output pruned tighter).
// Throws FatalError when the lattice has a cycle.
bool DeterminizeLatticePruned(const Lattice &lat, double beam, CompactLattice *clat, const DeterminizeLatticePrunedOptions &opts = DeterminizeLatticePrunedOptions());
// DeterminizeLatticePhonePrunedWrapper (lat/determinize-lattice-pruned.cc:1410-1499): what the decoders call.  With phone_determinize a
// first pass runs over the lattice with phone labels inserted at the phone boundaries (keeps the word pass's subsets small), then the
// word-level pass, then (minimize) push + minimize; with word_determinize=false the first pass's result re-packed by ConvertLattice.
struct DeterminizeLatticePhonePrunedOptions { float delta = 1.0f / 1024.0f; int32_t max_mem = 50000000; bool phone_determinize = true, word_determinize = true, minimize = false; };
bool DeterminizeLatticePhonePruned(const Lattice &lat, const TransitionInfo &trans, double beam, CompactLattice *clat,
                                   const DeterminizeLatticePhonePrunedOptions &opts = DeterminizeLatticePhonePrunedOptions());
// ConvertLattice(Lattice -> CompactLattice) (fstext/lattice-utils-inl.h:33-86): no determinization; Factor (fstext/factor-inl.h:70-150)
// folds linear chains of states (one arc in, one arc out, not final, no word on the arc out) into single arcs that carry the chain's
// transition-ids as their string, then the states are sorted topologically.  What the reference's CUDA pipeline writes with
// --determinize-lattice=false.
void ConvertLattice(const Lattice &lat, CompactLattice *clat);
// the other direction (fstext/lattice-utils-inl.h:88-152): every arc / final weight with a string of k transition-ids becomes a chain of
// k arcs through new states (word and weight on the first); what Kaldi's lattice readers do when a CompactLattice table is read as Lattice
void ConvertLattice(const CompactLattice &clat, Lattice *lat);
void Connect(CompactLattice *clat);                        // fst::Connect; keeps the relative order of the surviving states
void ScaleAcoustic(CompactLattice *clat, double scale);
bool TopSortIfNeeded(CompactLattice *clat);                // TopSortCompactLatticeIfNeeded (lat/lattice-functions.cc); false on a cycle
// --minimize of the determinization programs (determinize-lattice-pruned.cc:1455-1460, lattice-determinize-pruned.cc:122-126):
// PushCompactLatticeStrings / PushCompactLatticeWeights (lat/push-lattice.cc: transition-ids and weights moved as far towards the start
// state as all paths allow) then MinimizeCompactLattice (lat/minimize-lattice.cc: states with the same future merged; delta = tolerance on
// weights).  All three sort the lattice topologically first and return false when that fails.
bool PushCompactLatticeStrings(CompactLattice *clat);
bool PushCompactLatticeWeights(CompactLattice *clat);
bool MinimizeCompactLattice(CompactLattice *clat, float delta = 1.0f / 1024.0f);
// kaldi::PruneLattice (lat/lattice-functions.cc:233-318) on a raw lattice (any state order; acyclic): drops arcs and final
// weights that are on no path within `beam` of the best path, then trims.
bool PruneLattice(double beam, Lattice *lat);
// "ark:rxfilename" / "ark,t:rxfilename" table of state-level lattices (LatticeHolder::Read, lat/kaldi-lattice.cc:422-459: the
// first byte after the key tells text from OpenFst binary).  CompactLattice records (arc type compactlattice44, or text weights with a
// transition-id string) are accepted too and expanded with ConvertLattice, like the reference's SequentialLatticeReader.
std::vector<std::pair<std::string, Lattice>> ReadLatticeTable(const std::string &rspecifier);

// Kaldi float matrices from an archive or script file: "ark:rxfilename", "scp:rxfilename" (entries "key file" or
// "key file:offset"); binary FM / DM / CM / CM2 / CM3 (matrix/kaldi-matrix.cc:1402-1520, compressed-matrix.cc:560-660) and text
struct Matrix { int32_t rows = 0, cols = 0; std::vector<float> data; };
std::vector<std::pair<std::string, Matrix>> ReadMatrixTable(const std::string &rspecifier);
// one Matrix<double> object from an rxfilename (ReadKaldiObject: binary "DM"/"FM" after the \0B header, or text) -- CMVN stats files
struct MatrixD { int32_t rows = 0, cols = 0; std::vector<double> data; };
MatrixD ReadDoubleMatrix(const std::string &rxfilename);
// OnlineIvectorExtractionConfig (online2/online-ivector-feature.h:60-150) with the files it names read in, i.e. OnlineIvectorExtractionInfo::Init
// (online-ivector-feature.cc:29-68) + Check (:80-98).  Paths inside the config are used as written, like the reference.
struct IvectorExtractionInfo {
  std::string lda_mat_rxfilename, global_cmvn_stats_rxfilename, cmvn_config_rxfilename, splice_config_rxfilename, diag_ubm_rxfilename, ivector_extractor_rxfilename;
  bool online_cmvn_iextractor = false, use_most_recent_ivector = true, greedy_ivector_extractor = false;
  int32_t ivector_period = 10, num_gselect = 5, num_cg_iters = 15; float min_post = 0.025f, posterior_scale = 0.1f, max_count = 0.0f, max_remembered_frames = 1000.0f;
  int32_t left_context = 4, right_context = 4;                                                        // OnlineSpliceOptions (feat/online-feature.h:478-487)
  int32_t cmn_window = 600, speaker_frames = 600, global_frames = 200; bool normalize_mean = true, normalize_variance = false;   // OnlineCmvnOptions
  std::vector<float> lda; int32_t lda_rows = 0, lda_cols = 0; MatrixD global_cmvn_stats; DiagGmmModel ubm; IvectorExtractorModel ie;
  void Register(ParseOptions *po);
  void Init();                                                                                        // reads the files, checks the dimensions
};
IvectorExtractionInfo ReadIvectorExtractionConfig(const std::string &config_rxfilename);

// Kaldi token-vector table (util/kaldi-holder-inl.h TokenVectorHolder): "ark:file" with lines "key tok1 tok2 ..." (spk2utt)
std::vector<std::pair<std::string, std::vector<std::string>>> ReadTokenVectorTable(const std::string &rspecifier);

// best path through a raw lattice (LatticeFasterDecoder::GetBestPath = ShortestPath over the raw lattice, :103-111):
// transition-ids (alignment) and output labels (words) along it, total graph and acoustic cost.  false if no final state is reachable.
bool BestPath(const Lattice &lat, std::vector<int32_t> *alignment, std::vector<int32_t> *words, double *graph_cost, double *acoustic_cost);

class TableWriter {        // "ark:wxfilename" | "ark,t:wxfilename" (other options ignored); one stream
 public:
  explicit TableWriter(const std::string &wspecifier);
  bool Binary() const { return binary_; }
  void WriteLattice(const std::string &key, const Lattice &lat);
  void WriteCompactLattice(const std::string &key, const CompactLattice &clat);
  void WriteMatrix(const std::string &key, const float *data, int32_t rows, int32_t cols, int64_t stride);
  void WriteVector(const std::string &key, const float *data, int32_t dim);      // BaseFloatVectorWriter
  void WriteInt32Vector(const std::string &key, const std::vector<int32_t> &v);      // Int32VectorWriter (util/kaldi-holder-inl.h BasicVectorHolder)
  void Flush();
 private:
  std::shared_ptr<FILE> f_; bool binary_ = true;
};

// Determinization on worker threads with the output order of the submissions -- util/kaldi-thread.h:171-260 TaskSequencer as
// lattice-determinize-pruned-parallel.cc:30-100 uses it, and the job the reference's CUDA pipeline gives its --cuda-worker-threads
// pool (batched-threaded-nnet3-cuda-online-pipeline.cc:735-810): per lattice  scale acoustic costs by pre_scale -> DeterminizeLatticePruned
// -> (optional TopSortIfNeeded) -> scale by post_scale -> WriteCompactLattice.  Run() blocks while num_threads + 20 lattices are
// in flight (bounds memory).  An error inside a worker is re-thrown from the next Run() / Wait().
class DeterminizeSequencer {
 public:
  struct Config {
    int32_t num_threads = 1; double beam = 10.0, pre_scale = 1.0, post_scale = 1.0; bool topsort = false, minimize = false; DeterminizeLatticePrunedOptions det;
    const TransitionInfo *trans = nullptr; DeterminizeLatticePhonePrunedOptions phone_det;       // trans != nullptr: DeterminizeLatticePhonePruned with phone_det
    // the CUDA pipeline's lattice post-processor (SetLatticePostprocessor, batched-threaded-nnet3-cuda-pipeline2.h:204): applied to the determinized lattice;
    // with `ctm_out` the record written
    // is the utterance's CTM lines (LatticePostprocessor::GetCTM + MergeSegmentsToCTMOutput) instead of the lattice -- `writer` may then be null
    std::shared_ptr<class LatticePostprocessor> postprocessor; std::ostream *ctm_out = nullptr; const std::vector<std::string> *word_syms = nullptr; bool determinize = true;
    // called by a worker thread when `key`'s result exists (before it is written in its turn): the moment a streaming caller's latency clock stops
    std::function<void(const std::string &key)> on_done;
  };
  DeterminizeSequencer(const Config &config, TableWriter *writer);
  ~DeterminizeSequencer();
  void Run(std::string key, Lattice &&lat);
  void Wait();                              // returns when everything submitted so far has been written
  int32_t NumDone() const;
  int32_t NumWarn() const;                  // determinization stopped early, or empty output
 private:
  struct Impl; std::unique_ptr<Impl> impl_;
};


// ---- word-level Minimum Bayes Risk decoding, the CUDA pipeline's lattice post-processor, CTM output (k3_mbr.cc) ----------------------------------------
struct MinimumBayesRiskOptions { bool decode_mbr = true, print_silence = false; };      // lat/sausages.h:56-73
class MinimumBayesRisk {      // lat/sausages.h:80-266; the lattice is not required to be determinized
 public:
  explicit MinimumBayesRisk(const CompactLattice &clat, MinimumBayesRiskOptions opts = MinimumBayesRiskOptions());
  ~MinimumBayesRisk();
  const std::vector<int32_t> &GetOneBest() const;                                       // the MBR word sequence
  const std::vector<std::pair<float, float>> &GetOneBestTimes() const;                  // (begin, end) of every word of it, in frames
  const std::vector<float> &GetOneBestConfidences() const;
  const std::vector<std::vector<std::pair<int32_t, float>>> &GetSausageStats() const;   // per bin: (word or 0, posterior), most likely first
  const std::vector<std::pair<float, float>> &GetSausageTimes() const;
  double GetBayesRisk() const;                                                          // expected word errors
 private:
  struct Impl; std::unique_ptr<Impl> impl_;
};
struct CtmResult { std::vector<float> conf; std::vector<int32_t> words; std::vector<std::pair<float, float>> times_seconds; };      // cudadecoder/cuda-pipeline-common.h:53-57
struct LatticePostprocessorConfig {      // cudadecoder/lattice-postprocessor.h:35-76
  std::string word_boundary_rxfilename; MinimumBayesRiskOptions mbr_opts; int32_t silence_label = 0, partial_word_label = 0; bool reorder = true;
  float max_expand = 0.0f, acoustic_scale = 1.0f, lm_scale = 1.0f, acoustic2lm_scale = 0.0f, lm2acoustic_scale = 0.0f, word_ins_penalty = 0.0f;
  void Register(ParseOptions *po);
};
// ---- word alignment of a CompactLattice (lat/word-align-lattice.{h,cc}), k3_mbr.cc: every arc of the output is one word (or one silence, or a partial word at the end of a
// forced-out utterance) with exactly that word's transition-ids, so that state times are word boundaries.  The lexicon must use word-position-dependent phones; which phone is what
// comes from phones/word_boundary.int ("<phone-id> nonword|begin|end|internal|singleton").
struct WordBoundaryInfo {      // lat/word-align-lattice.h:119-170 (the WordBoundaryInfoNewOpts form)
  enum PhoneType { kNoPhone = 0, kWordBeginPhone, kWordEndPhone, kWordBeginAndEndPhone, kWordInternalPhone, kNonWordPhone };
  std::vector<PhoneType> phone_to_type; int32_t silence_label = 0, partial_word_label = 0; bool reorder = true;
  PhoneType TypeOfPhone(int32_t p) const;      // throws for a phone the file does not list
};
WordBoundaryInfo ReadWordBoundaryInfo(const std::string &word_boundary_rxfilename, bool reorder = true, int32_t silence_label = 0, int32_t partial_word_label = 0);
// WordAlignLattice (:724-731): false when the lattice could not be aligned cleanly (a broken / forced-out lattice, a mismatched model or --reorder option,
// max_states > 0 exceeded);
// lat_out then holds what could be made of it, as in the reference
bool WordAlignLattice(const CompactLattice &lat, const TransitionInfo &tmodel, const WordBoundaryInfo &info, int32_t max_states, CompactLattice *lat_out);

class LatticePostprocessor {      // cudadecoder/lattice-postprocessor.h:78-118
 public:
  explicit LatticePostprocessor(const LatticePostprocessorConfig &config);
  bool GetCTM(CompactLattice &clat, CtmResult *ctm_result) const;
  bool GetPostprocessedLattice(CompactLattice &clat, CompactLattice *out_clat) const;      // scales + word insertion penalty (clat is modified, like the reference's)
  void SetDecoderFrameShift(float seconds) { decoder_frame_shift_ = seconds; }
  void SetTransitionInformation(const TransitionInfo *tmodel) { tmodel_ = tmodel; }      // (:93-95; needed with --word-boundary-rxfilename)
 private:
  LatticePostprocessorConfig config_;
  bool use_lattice_scale_ = false;
  float decoder_frame_shift_ = 0.0f;
  const TransitionInfo *tmodel_ = nullptr;
  std::shared_ptr<WordBoundaryInfo> word_info_;
};
std::shared_ptr<LatticePostprocessor> LoadLatticePostprocessor(const std::string &config_rxfilename);      // LoadAndSetLatticePostprocessor's first half (:126-137)
// the CTM lines of one (un-segmented) utterance: "<key> 0  <begin> <duration> <word> <confidence>", two decimals (cuda-pipeline-common.cc:67-142)
void WriteCtm(const CtmResult &ctm, const std::string &key, std::ostream &os, const std::vector<std::string> *word_syms = nullptr);

}  // namespace k3host
