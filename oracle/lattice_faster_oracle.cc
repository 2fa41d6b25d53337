/*
 * oracle/lattice_faster_oracle.cc  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of Kaldi's LatticeFasterDecoder (the parity oracle north_star names:
 * "lattice-faster-decoder") over a plain CSR WFST.
 *
 * PINNED against the reference's own source: the reference holds no golden vectors or tests for its decoders
 * (decoder/Makefile:6 and cudadecoder/Makefile:16 have empty TESTFILES) and OpenFst 1.8.4 is neither installed nor
 * vendored (tools/Makefile:10), but decoder/lattice-faster-decoder.{h,cc} compile UNMODIFIED against a stand-in for the
 * small part of OpenFst they touch (third_party/minifst, recipe in oracle/build_ref.sh -> oracle/_ref/bin/
 * ref-lattice-decoder).  Mode 0 of this file reproduces that binary's GetRawLattice output exactly -- states per
 * frame, arcs, labels, float bits, sharing of states, up to state renaming -- on every case of tests/decoder_cases.py
 * (active-state limits, beams, prune intervals, hash sizes; tests/test_oracle_decoder.py, live where oracle/_ref is
 * present and through the digests recorded in tests/golden/decoder_ref_golden.json elsewhere).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the shared object built from
 * this file; kaldi_amd/ never does.
 *
 * What is restated (paths relative to /root/reference/src):
 *   decoder/lattice-faster-decoder.h:37-107   LatticeFasterDecoderConfig (beam, max_active, min_active, lattice_beam,
 *                                             prune_interval, beam_delta, hash_ratio, prune_scale)
 *   decoder/lattice-faster-decoder.cc:63-81   InitDecoding            -> Decoder::Init
 *                                   :227-233  PossiblyResizeHash      -> Decoder::MaybeGrowHash
 *                                   :259-302  FindOrAddToken          -> Decoder::FindOrAdd
 *                                   :308-379  PruneForwardLinks       -> Decoder::PruneLinks
 *                                   :385-467  PruneForwardLinksFinal  -> Decoder::PruneLinksFinal
 *                                   :488-507  PruneTokensForFrame     -> Decoder::PruneTokens
 *                                   :515-542  PruneActiveTokens       -> Decoder::PruneActive
 *                                   :545-586  ComputeFinalCosts       -> Decoder::FinalCosts
 *                                   :588-632  AdvanceDecoding         -> Decoder::Advance
 *                                   :634-649  FinalizeDecoding        -> Decoder::Finalize
 *                                   :653-720  GetCutoff               -> Decoder::Cutoff
 *                                   :723-814  ProcessEmitting         -> Decoder::Emitting
 *                                   :830-897  ProcessNonemitting      -> Decoder::Nonemitting
 *                                   :114-197  GetRawLattice           -> Decoder::RawLattice (states keyed by token,
 *                                             the per-frame numbering of TopSortTokens :927-1002 is not reproduced:
 *                                             parity is defined on the lattice canonicalised by (frame, fst state))
 *   util/hash-list-inl.h:37-165               HashList (bucket-chain list whose iteration order decides the token
 *                                             visit order of ProcessEmitting)            -> struct StateList
 *   decoder/decodable-matrix.h / nnet3/nnet-am-decodable-simple.h  LogLikelihood(frame, tid) = M(frame, tid2pdf[tid])
 *
 * Two evaluation modes:
 *   mode 0 "literal":   exactly the serial algorithm, next_cutoff tightened while the token list is traversed in
 *                       HashList order (so a few tokens/links exist only because the bound was still loose).
 *   mode 1 "two-pass":  the order-independent definition the GPU decoder implements: an arc is accepted iff
 *                       tot < min(prepass bound, min_tot + adaptive_beam), i.e. against the FINAL next_cutoff.
 *   mode 2 "literal, phase-parallel": the SAME results as mode 0, bit for bit, but computed the way the GPU kernel
 *                       (k3_decoder_config.literal_order) computes them: no serial walk over a hash list.  The token
 *                       visit order is derived from (first-occupation time of the bucket, insertion time) keys, arc
 *                       acceptance from an exclusive prefix-min over the arc sequence, token creation times from a
 *                       min over accepted arc sequence numbers, the eps closure from an order-free fixpoint followed
 *                       by a replay of the LIFO queue on the closure sub-graph that only recovers creation ORDER.
 *   mode 3              mode 2 with the work inside every parallel phase visited in a shuffled order (what a GPU does):
 *                       tests assert mode 0 == mode 2 == mode 3, i.e. the phase decomposition is order-free.
 *   mode 4              mode 3 with the replay split by connected components of the closure sub-graph (what the GPU kernel does).
 * SURVEY.md 9.1 argues that both give the same lattice after FinalizeDecoding except at exact float ties and in
 * the min_active corner; tests assert equality on the test sets and the GPU path is compared with both.
 * All arithmetic is float32 in the reference's evaluation order; compile WITHOUT -ffast-math / FMA contraction.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <limits>
#include <vector>

namespace {

const float kInf = std::numeric_limits<float>::infinity();

struct Fst {              // generic CSR acceptor/transducer: arcs of state s are [off[s], off[s+1]) in FST order
  int32_t num_states, start;
  const int32_t *off, *ilabel, *olabel, *next;
  const float *weight, *final_cost;   // final_cost[s] = +inf for non-final states
  std::vector<int32_t> num_ieps;      // NumInputEpsilons(s)
};

struct Config {           // LatticeFasterDecoderConfig, lattice-faster-decoder.h:37-107
  float beam; int32_t max_active, min_active; float lattice_beam; int32_t prune_interval;
  float beam_delta, hash_ratio, prune_scale;
};

struct Link { int32_t dst, ilabel, olabel; float graph, ac; int32_t next; };
struct Tok  { float tot, extra; int32_t links, next, state, frame; };
struct FrameList { int32_t head = -1; bool must_prune_links = true, must_prune_toks = true; };

// util/hash-list-inl.h: elements form ONE singly linked list; a bucket remembers its last element and the bucket
// that was occupied before it, so a bucket's elements are the run between the previous bucket's last element
// and its own.  New buckets go to the list tail; new elements of an occupied bucket go right after its last one.
struct StateList {
  struct Elem { int32_t key, val, tail; };
  struct Bucket { int64_t prev_bucket; int32_t last; };
  std::vector<Elem> elems; std::vector<int32_t> free_;
  std::vector<Bucket> buckets; size_t hash_size = 0;
  int32_t head = -1; int64_t tail_bucket = -1;
  size_t Size() const { return hash_size; }
  void SetSize(size_t n) { hash_size = n; if (n > buckets.size()) buckets.resize(n, Bucket{0, -1}); }
  int32_t NewElem() {
    if (!free_.empty()) { int32_t e = free_.back(); free_.pop_back(); return e; }
    elems.push_back(Elem{0, -1, -1}); return (int32_t)elems.size() - 1;
  }
  void Delete(int32_t e) { free_.push_back(e); }
  int32_t Clear() {   // hands the list to the caller, empties the index
    for (int64_t b = tail_bucket; b != -1; b = buckets[b].prev_bucket) buckets[b].last = -1;
    tail_bucket = -1; int32_t h = head; head = -1; return h;
  }
  int32_t BucketBegin(const Bucket &b) const { return b.prev_bucket == -1 ? head : elems[buckets[b.prev_bucket].last].tail; }
  int32_t Insert(int32_t key, int32_t val) {      // returns the element (existing or new)
    const size_t idx = (size_t)key % hash_size; Bucket &b = buckets[idx];
    if (b.last != -1) {
      const int32_t stop = elems[b.last].tail;
      for (int32_t e = BucketBegin(b); e != stop; e = elems[e].tail) if (elems[e].key == key) return e;
    }
    const int32_t e = NewElem(); elems[e].key = key; elems[e].val = val;
    Bucket &bb = buckets[idx];
    if (bb.last == -1) {
      if (tail_bucket == -1) head = e; else elems[buckets[tail_bucket].last].tail = e;
      elems[e].tail = -1; bb.last = e; bb.prev_bucket = tail_bucket; tail_bucket = (int64_t)idx;
    } else {
      elems[e].tail = elems[bb.last].tail; elems[bb.last].tail = e; bb.last = e;
    }
    return e;
  }
};

struct FrameStat { int32_t ntoks; float cur_cutoff, adaptive_beam, next_cutoff, cost_offset; };

struct Decoder {
  const Fst &fst; Config cfg; int mode;
  const float *loglikes; int64_t ld; const int32_t *tid2pdf; int32_t num_frames_ready;
  std::vector<Tok> toks; std::vector<int32_t> free_toks;
  std::vector<Link> links; std::vector<int32_t> free_links;
  std::vector<FrameList> active;      // active[f]: tokens after consuming f frames
  StateList cur;                      // state -> token of the newest frame
  std::vector<float> cost_offsets; std::vector<int32_t> queue; std::vector<float> tmp;
  int64_t num_toks = 0;
  bool finalized = false; float final_relative_cost = kInf, final_best_cost = kInf;
  std::vector<std::pair<int32_t, float>> final_costs;   // (token, final cost) for final-state tokens of the last frame
  std::vector<char> tok_is_final;                       // indexed by token when finalized
  // diagnostics
  std::vector<FrameStat> stats; int64_t n_extra_links = 0, n_extra_toks = 0, n_best_ties = 0, n_links_created = 0;

  Decoder(const Fst &f, const Config &c, int m) : fst(f), cfg(c), mode(m) { cur.SetSize(1000); }  // lattice-faster-decoder.cc:37-43

  float LogLike(int32_t frame, int32_t tid) const { return loglikes[(int64_t)frame * ld + tid2pdf[tid]]; }
  int32_t NumFramesDecoded() const { return (int32_t)active.size() - 1; }

  int32_t NewTok(float tot, float extra, int32_t next, int32_t state, int32_t frame) {
    int32_t t;
    if (!free_toks.empty()) { t = free_toks.back(); free_toks.pop_back(); } else { toks.push_back(Tok()); t = (int32_t)toks.size() - 1; }
    toks[t] = Tok{tot, extra, -1, next, state, frame}; return t;
  }
  int32_t NewLink(int32_t dst, int32_t il, int32_t ol, float g, float a, int32_t next) {
    int32_t l;
    if (!free_links.empty()) { l = free_links.back(); free_links.pop_back(); } else { links.push_back(Link()); l = (int32_t)links.size() - 1; }
    links[l] = Link{dst, il, ol, g, a, next}; n_links_created++; return l;
  }
  void DeleteLinks(int32_t t) { for (int32_t l = toks[t].links; l != -1;) { int32_t n = links[l].next; free_links.push_back(l); l = n; } toks[t].links = -1; }

  void MaybeGrowHash(size_t n) {   // :227-233
    const size_t want = (size_t)((float)n * cfg.hash_ratio);
    if (want > cur.Size()) cur.SetSize(want);
  }

  // :259-302.  Returns the list element; *changed = new token or cost lowered.
  int32_t FindOrAdd(int32_t state, int32_t frame_plus_one, float tot, bool *changed) {
    int32_t &head = active[frame_plus_one].head;
    const int32_t e = cur.Insert(state, -1);
    if (cur.elems[e].val == -1) {
      const int32_t t = NewTok(tot, 0.0f, head, state, frame_plus_one);
      head = t; num_toks++; cur.elems[e].val = t;
      if (changed) *changed = true;
    } else {
      Tok &t = toks[cur.elems[e].val];
      if (t.tot > tot) { t.tot = tot; if (changed) *changed = true; }
      else if (changed) *changed = false;
    }
    return e;
  }

  void Init() {   // :63-81
    active.assign(1, FrameList());
    const int32_t t = NewTok(0.0f, 0.0f, -1, fst.start, 0);
    active[0].head = t; const int32_t e = cur.Insert(fst.start, t); (void)e; num_toks++;
    Nonemitting(cfg.beam);
  }

  // :653-720.  list = element list handed over by cur.Clear()
  float Cutoff(int32_t list, size_t *count, float *adaptive_beam, int32_t *best_elem) {
    float best = kInf; size_t n = 0; *best_elem = -1;
    const bool plain = (cfg.max_active == std::numeric_limits<int32_t>::max() && cfg.min_active == 0);
    tmp.clear();
    for (int32_t e = list; e != -1; e = cur.elems[e].tail, n++) {
      const float w = toks[cur.elems[e].val].tot;
      if (!plain) tmp.push_back(w);
      if (w < best) { best = w; *best_elem = e; }
      else if (w == best && *best_elem != -1) {
        n_best_ties++;
        // mode 1 breaks ties by the smaller FST state (what the GPU decoder does); mode 0 keeps the first in list order
        if (mode == 1 && cur.elems[e].key < cur.elems[*best_elem].key) *best_elem = e;
      }
    }
    *count = n;
    if (plain) { *adaptive_beam = cfg.beam; return best + cfg.beam; }
    const float beam_cutoff = best + cfg.beam; float min_active_cutoff = kInf, max_active_cutoff = kInf;
    if (tmp.size() > (size_t)cfg.max_active) {
      std::nth_element(tmp.begin(), tmp.begin() + cfg.max_active, tmp.end());
      max_active_cutoff = tmp[cfg.max_active];
    }
    if (max_active_cutoff < beam_cutoff) { *adaptive_beam = max_active_cutoff - best + cfg.beam_delta; return max_active_cutoff; }
    if (tmp.size() > (size_t)cfg.min_active) {
      if (cfg.min_active == 0) min_active_cutoff = best;
      else {
        std::nth_element(tmp.begin(), tmp.begin() + cfg.min_active,
                         tmp.size() > (size_t)cfg.max_active ? tmp.begin() + cfg.max_active : tmp.end());
        min_active_cutoff = tmp[cfg.min_active];
      }
    }
    if (min_active_cutoff > beam_cutoff) { *adaptive_beam = min_active_cutoff - best + cfg.beam_delta; return min_active_cutoff; }
    *adaptive_beam = cfg.beam; return beam_cutoff;
  }

  // :723-814
  float Emitting() {
    const int32_t frame = (int32_t)active.size() - 1;
    active.resize(active.size() + 1);
    const int32_t list = cur.Clear();
    int32_t best_elem; float adaptive_beam; size_t cnt;
    const float cur_cutoff = Cutoff(list, &cnt, &adaptive_beam, &best_elem);
    MaybeGrowHash(cnt);
    float next_cutoff = kInf, cost_offset = 0.0f;
    if (best_elem != -1) {   // pre-pass over the best token's arcs (:753-768); note the different evaluation order
      const int32_t s = cur.elems[best_elem].key; const float tot = toks[cur.elems[best_elem].val].tot;
      cost_offset = -tot;
      for (int32_t a = fst.off[s]; a < fst.off[s + 1]; a++)
        if (fst.ilabel[a] != 0) {
          const float nw = fst.weight[a] + cost_offset - LogLike(frame, fst.ilabel[a]) + tot;
          if (nw + adaptive_beam < next_cutoff) next_cutoff = nw + adaptive_beam;
        }
    }
    cost_offsets.resize(frame + 1, 0.0f); cost_offsets[frame] = cost_offset;
    float accept_cutoff = kInf;     // mode 1: the final bound, computed by a first pass over all arcs
    if (mode == 1) {
      float min_tot = kInf;
      for (int32_t e = list; e != -1; e = cur.elems[e].tail) {
        const int32_t s = cur.elems[e].key; const float c = toks[cur.elems[e].val].tot;
        if (c <= cur_cutoff)
          for (int32_t a = fst.off[s]; a < fst.off[s + 1]; a++)
            if (fst.ilabel[a] != 0) {
              const float ac = cost_offset - LogLike(frame, fst.ilabel[a]);
              const float t = c + ac + fst.weight[a];
              if (t < min_tot) min_tot = t;
            }
      }
      accept_cutoff = next_cutoff;
      if (min_tot + adaptive_beam < accept_cutoff) accept_cutoff = min_tot + adaptive_beam;
    }
    std::vector<std::pair<int32_t, float>> made;   // (link, tot) for the extras diagnostic (mode 0)
    for (int32_t e = list, e_tail; e != -1; e = e_tail) {
      const int32_t s = cur.elems[e].key, t = cur.elems[e].val;
      if (toks[t].tot <= cur_cutoff) {
        for (int32_t a = fst.off[s]; a < fst.off[s + 1]; a++) {
          if (fst.ilabel[a] == 0) continue;
          const float ac_cost = cost_offset - LogLike(frame, fst.ilabel[a]), graph_cost = fst.weight[a], cur_cost = toks[t].tot;
          const float tot_cost = cur_cost + ac_cost + graph_cost;
          if (mode == 1) { if (tot_cost >= accept_cutoff) continue; }
          else {
            if (tot_cost >= next_cutoff) continue;
            else if (tot_cost + adaptive_beam < next_cutoff) next_cutoff = tot_cost + adaptive_beam;
          }
          const int32_t en = FindOrAdd(fst.next[a], frame + 1, tot_cost, nullptr);
          toks[t].links = NewLink(cur.elems[en].val, fst.ilabel[a], fst.olabel[a], graph_cost, ac_cost, toks[t].links);
          if (mode == 0) made.push_back({toks[t].links, tot_cost});
        }
      }
      e_tail = cur.elems[e].tail; cur.Delete(e);
    }
    if (mode == 1) next_cutoff = accept_cutoff;
    else {
      for (auto &m : made) if (m.second >= next_cutoff) n_extra_links++;
      for (int32_t t = active[frame + 1].head; t != -1; t = toks[t].next) if (toks[t].tot >= next_cutoff) n_extra_toks++;
    }
    stats.push_back(FrameStat{(int32_t)cnt, cur_cutoff, adaptive_beam, next_cutoff, cost_offset});
    return next_cutoff;
  }

  // :830-897
  void Nonemitting(float cutoff) {
    const int32_t frame = (int32_t)active.size() - 2;
    queue.clear();
    for (int32_t e = cur.head; e != -1; e = cur.elems[e].tail) if (fst.num_ieps[cur.elems[e].key] != 0) queue.push_back(e);
    while (!queue.empty()) {
      const int32_t e = queue.back(); queue.pop_back();
      const int32_t s = cur.elems[e].key, t = cur.elems[e].val;
      const float cur_cost = toks[t].tot;
      if (cur_cost >= cutoff) continue;
      DeleteLinks(t);
      for (int32_t a = fst.off[s]; a < fst.off[s + 1]; a++) {
        if (fst.ilabel[a] != 0) continue;
        const float graph_cost = fst.weight[a], tot_cost = cur_cost + graph_cost;
        if (tot_cost < cutoff) {
          bool changed;
          const int32_t en = FindOrAdd(fst.next[a], frame + 1, tot_cost, &changed);
          toks[t].links = NewLink(cur.elems[en].val, 0, fst.olabel[a], graph_cost, 0.0f, toks[t].links);
          if (changed && fst.num_ieps[fst.next[a]] != 0) queue.push_back(en);
        }
      }
    }
  }


  // =====================================================================================================================
  // modes 2 / 3: the literal algorithm in the phase structure of the GPU kernel.  Every loop marked PAR is a data-parallel
  // phase on the GPU (any visiting order must give the same result: mode 3 shuffles it); loops marked SER are the replay.
  struct Frame2 {                       // tokens of the newest frame, local index i
    std::vector<int32_t> state, tok, order;       // tok = pool index; order[r] = local index of the r-th element of the HashList
    std::vector<float> cost;
  };
  Frame2 cf; size_t hash_size2 = 1000;  // lattice-faster-decoder.cc:41 toks_.SetSize(1000)
  uint64_t rng_state = 0x9E3779B97F4A7C15ull;
  int64_t n_replay_pops = 0, n_replay_pushes = 0, n_replay_crit = 0, n_replay_comps = 0, n_replay_frames = 0;
  int64_t n_replay_pops_once = 0, n_replay_crit_once = 0, n_replay_big_pops = 0, n_replay_big_once = 0;      // K3O_COMP_STATS: pops in components where no existing token is improved (a forest walked once)
  uint32_t Rand() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33); }
  template <class T> void MaybeShuffle(std::vector<T> &v) { if (mode < 3) return; for (size_t i = v.size(); i > 1; i--) std::swap(v[i - 1], v[Rand() % i]); }
  std::vector<int32_t> Iota(size_t n) { std::vector<int32_t> v(n); for (size_t i = 0; i < n; i++) v[i] = (int32_t)i; MaybeShuffle(v); return v; }

  // HashList order (util/hash-list-inl.h:125-165) of tokens with unique creation labels: buckets in order of first occupation,
  // insertion order inside a bucket == sort by (label of the bucket's first occupant, own label).  Computed without a sort of
  // keys: dense creation rank d (a bitmap prefix count on the GPU), per-bucket first rank / count, prefix sum over the leaders.
  void HashOrder(const std::vector<int32_t> &state, const std::vector<uint32_t> &label, std::vector<int32_t> *order, std::vector<int32_t> *by_ins) {
    const size_t n = state.size();
    uint32_t M = 0; for (uint32_t l : label) M = std::max(M, l + 1);
    std::vector<uint32_t> bm((M + 31) / 32 + 1, 0), wpre((M + 31) / 32 + 1, 0);
    for (int32_t i : Iota(n)) { if (bm[label[i] >> 5] >> (label[i] & 31) & 1) abort(); bm[label[i] >> 5] |= 1u << (label[i] & 31); }       // PAR (atomicOr)
    for (size_t w = 1; w < bm.size(); w++) wpre[w] = wpre[w - 1] + (uint32_t)__builtin_popcount(bm[w - 1]);                                  // scan
    std::vector<int32_t> d(n); by_ins->assign(n, -1);
    std::vector<uint32_t> bfirst(hash_size2, 0xFFFFFFFFu), bcnt(hash_size2, 0), bfill(hash_size2, 0);
    for (int32_t i : Iota(n)) {                                                                                                                // PAR
      const uint32_t l = label[i]; d[i] = (int32_t)(wpre[l >> 5] + (uint32_t)__builtin_popcount(bm[l >> 5] & ((1u << (l & 31)) - 1u)));
      (*by_ins)[d[i]] = i;
      const size_t b = (size_t)state[i] % hash_size2; bfirst[b] = std::min(bfirst[b], (uint32_t)d[i]); bcnt[b]++;
    }
    std::vector<uint32_t> lead(n + 1, 0);
    for (int32_t dd : Iota(n)) { const int32_t i = (*by_ins)[dd]; const size_t b = (size_t)state[i] % hash_size2; lead[dd] = bfirst[b] == (uint32_t)dd ? bcnt[b] : 0; }   // PAR
    { uint32_t run = 0; for (size_t k = 0; k <= n; k++) { const uint32_t v = k < n ? lead[k] : 0; lead[k] = run; run += v; } }                 // exclusive scan
    std::vector<int32_t> grp(n, -1);
    for (int32_t i : Iota(n)) { const size_t b = (size_t)state[i] % hash_size2; if (bcnt[b] > 1) grp[lead[bfirst[b]] + bfill[b]++] = d[i]; }   // PAR (atomicAdd)
    order->assign(n, -1);
    for (int32_t i : Iota(n)) {                                                                                                                // PAR
      const size_t b = (size_t)state[i] % hash_size2; const uint32_t L = lead[bfirst[b]]; uint32_t rank = 0;
      if (bcnt[b] > 1) for (uint32_t k = 0; k < bcnt[b]; k++) rank += grp[L + k] < d[i];
      (*order)[L + rank] = i;
    }
  }

  struct PendingLink { int32_t src_tok, dst_local, ilabel, olabel; float graph, ac; };

  // ProcessNonemitting (:830-897) for the frame being built + publication of the frame.  state/cost/label: the tokens the
  // emitting pass created (label = creation time, unique); n_labels = number of labels handed out so far.
  void Nonemitting2(float cutoff, std::vector<int32_t> state, std::vector<float> cost, std::vector<uint32_t> label, uint32_t n_labels,
                    std::vector<PendingLink> links_in) {
    const int32_t frame_plus_one = (int32_t)active.size() - 1;
    const size_t n_e = state.size();
    std::vector<int32_t> order1, by_ins;
    HashOrder(state, label, &order1, &by_ins);          // the list ProcessNonemitting walks to fill its queue (:845-850)
    // ---- order-free fixpoint: final costs, the tokens the closure creates (no creation time yet)
    static thread_local std::vector<int32_t> index_of_state;      // state -> local index (a hash table on the GPU); all -1 between calls
    if ((int32_t)index_of_state.size() < fst.num_states) index_of_state.assign(fst.num_states, -1);
    for (size_t i = 0; i < n_e; i++) index_of_state[state[i]] = (int32_t)i;
    const std::vector<float> c0 = cost;                  // costs right after ProcessEmitting
    std::vector<int32_t> wl;
    for (int32_t i : Iota(n_e)) if (fst.num_ieps[state[i]] != 0) wl.push_back(i);
    while (!wl.empty()) {                                // rounds; PAR inside a round (atomicMin on the costs)
      std::vector<int32_t> nxt; MaybeShuffle(wl);
      for (int32_t i : wl) {
        const float c = cost[i]; if (c >= cutoff) continue;
        const int32_t s = state[i];
        for (int32_t a = fst.off[s]; a < fst.off[s + 1]; a++) {
          if (fst.ilabel[a] != 0) continue;
          const float tot = c + fst.weight[a];
          if (!(tot < cutoff)) continue;
          int32_t d = index_of_state[fst.next[a]];
          if (d < 0) { d = (int32_t)state.size(); index_of_state[fst.next[a]] = d; state.push_back(fst.next[a]); cost.push_back(kInf); label.push_back(0xFFFFFFFFu); }
          if (tot < cost[d]) { cost[d] = tot; if (fst.num_ieps[state[d]] != 0) nxt.push_back(d); }
        }
      }
      wl.swap(nxt);
    }
    const size_t n = state.size();
    // ---- closure sub-graph in token space: per source token its eps arcs, in arc order, that pass at the FINAL costs
    // (an arc that fails at the final cost of its source fails at every earlier, higher cost of it too)
    std::vector<int32_t> cbeg(n + 1, 0), cdst; std::vector<float> cw; std::vector<int32_t> carc;
    for (size_t i = 0; i < n; i++) {
      cbeg[i] = (int32_t)cdst.size();
      if (fst.num_ieps[state[i]] == 0 || !(cost[i] < cutoff)) continue;
      for (int32_t a = fst.off[state[i]]; a < fst.off[state[i] + 1]; a++)
        if (fst.ilabel[a] == 0) { const bool pass = cost[i] + fst.weight[a] < cutoff; cdst.push_back(pass ? index_of_state[fst.next[a]] : -1); cw.push_back(fst.weight[a]); carc.push_back(a); }
    }
    cbeg[n] = (int32_t)cdst.size();
    // ---- mode 4: the replay split by weakly connected COMPONENTS of the closure sub-graph.  The queue is consumed root by root (a root = an entry
    // of the initial queue; the stack is back at its initial level before the next root is taken), and a root's cascade only reads and writes the
    // costs of tokens it can reach, so cascades in different components commute: every component replays its own roots in queue order with a
    // stack of its own (PAR over components, SER inside one), and the creation labels are handed out afterwards root by root in queue order.
    if (mode == 4) {
      std::vector<int32_t> par(n); for (size_t i = 0; i < n; i++) par[i] = (int32_t)i;
      auto find = [&](int32_t x) { while (par[x] != x) x = par[x]; return x; };
      { std::vector<std::pair<int32_t, int32_t>> edges;
        for (size_t i = 0; i < n; i++) for (int32_t k = cbeg[i]; k < cbeg[i + 1]; k++) if (cdst[k] >= 0) edges.push_back({(int32_t)i, cdst[k]});
        MaybeShuffle(edges);
        for (auto &e : edges) { int32_t a = find(e.first), b = find(e.second); if (a == b) continue; if (a < b) std::swap(a, b); par[a] = b; } }      // PAR (lock-free union: the larger root hooks under the smaller)
      std::vector<int32_t> roots;                         // processing order = the initial queue from its back
      for (size_t p = n_e; p > 0; p--) { const int32_t i = order1[p - 1]; if (fst.num_ieps[state[i]] != 0) roots.push_back(i); }
      std::vector<std::vector<int32_t>> comp_roots(n); std::vector<int32_t> comps;
      for (size_t r = 0; r < roots.size(); r++) { const int32_t c = find(roots[r]); if (comp_roots[c].empty()) comps.push_back(c); comp_roots[c].push_back((int32_t)r); }
      MaybeShuffle(comps);
      std::vector<float> rc(n, kInf); std::vector<char> ex(n, 0);
      for (size_t i = 0; i < n_e; i++) { rc[i] = c0[i]; ex[i] = 1; }
      std::vector<std::vector<int32_t>> created(roots.size());
      int64_t crit = 0; bool crit_once = false;
      for (int32_t c : comps) {                           // PAR over components
        std::vector<int32_t> stack; const int64_t pops0 = n_replay_pops; int64_t improved = 0;
        for (int32_t r : comp_roots[c]) {                 // SER: this component's roots in queue order
          stack.push_back(roots[r]);
          while (!stack.empty()) {
            const int32_t e = stack.back(); stack.pop_back(); n_replay_pops++;
            const float cc = rc[e];
            if (cc >= cutoff) continue;
            for (int32_t k = cbeg[e]; k < cbeg[e + 1]; k++) {
              const int32_t d = cdst[k]; if (d < 0) continue;
              const float tot = cc + cw[k];
              if (!(tot < cutoff)) continue;
              bool changed = false;
              if (!ex[d]) { ex[d] = 1; rc[d] = tot; created[r].push_back(d); changed = true; }
              else if (rc[d] > tot) { rc[d] = tot; changed = true; improved++; }
              if (changed && fst.num_ieps[state[d]] != 0) { stack.push_back(d); n_replay_pushes++; }
            }
          }
        }
        { const int64_t cp = n_replay_pops - pops0; if (improved == 0) n_replay_pops_once += cp; if (cp >= 64) { n_replay_big_pops += cp; if (improved == 0) n_replay_big_once += cp; }
          if (cp > crit) crit_once = improved == 0; }
        crit = std::max(crit, n_replay_pops - pops0);
      }
      if (getenv("K3O_COMP_STATS") && atoi(getenv("K3O_COMP_STATS")) >= 2 && n_replay_frames < 12) fprintf(stderr, "  closure %lld: tokens %zu (emitting %zu) roots %zu components %zu critical-path pops %lld%s\n", (long long)n_replay_frames, n, (size_t)n_e, roots.size(), comps.size(), (long long)crit, crit_once ? " (no improvement)" : "");
      n_replay_crit += crit; if (crit_once) n_replay_crit_once += crit; n_replay_comps += (int64_t)comps.size(); n_replay_frames++;
      for (size_t r = 0; r < roots.size(); r++) for (int32_t d : created[r]) label[d] = n_labels++;      // exclusive scan over the roots' counts on the GPU
      for (size_t i = 0; i < n; i++) if (!ex[i] || rc[i] != cost[i]) abort();
    } else
    // ---- SER replay of the LIFO queue (:851-896) on the sub-graph: recovers only the ORDER in which the closure creates tokens
    {
      std::vector<float> rc(n, kInf); std::vector<char> ex(n, 0);
      for (size_t i = 0; i < n_e; i++) { rc[i] = c0[i]; ex[i] = 1; }
      std::vector<int32_t> stack; size_t p = n_e;         // the initial queue = order1 filtered by "has eps arcs", consumed from its back
      for (;;) {
        int32_t e = -1;
        if (!stack.empty()) { e = stack.back(); stack.pop_back(); }
        else { while (p > 0) { const int32_t i = order1[--p]; if (fst.num_ieps[state[i]] != 0) { e = i; break; } } if (e < 0) break; }
        n_replay_pops++;
        const float c = rc[e];
        if (c >= cutoff) continue;
        for (int32_t k = cbeg[e]; k < cbeg[e + 1]; k++) {
          const int32_t d = cdst[k]; if (d < 0) continue;
          const float tot = c + cw[k];
          if (!(tot < cutoff)) continue;
          bool changed = false;
          if (!ex[d]) { ex[d] = 1; rc[d] = tot; label[d] = n_labels++; changed = true; }
          else if (rc[d] > tot) { rc[d] = tot; changed = true; }
          if (changed && fst.num_ieps[state[d]] != 0) { stack.push_back(d); n_replay_pushes++; }
        }
      }
      for (size_t i = 0; i < n; i++) if (!ex[i] || rc[i] != cost[i]) abort();      // the replay must land on the fixpoint
    }
    for (size_t i = 0; i < n; i++) index_of_state[state[i]] = -1;
    // ---- publish: final HashList order, tokens materialised in creation order (active_toks_ list = reverse creation order)
    std::vector<int32_t> order;
    HashOrder(state, label, &order, &by_ins);
    std::vector<int32_t> tok(n, -1);
    for (size_t dd = 0; dd < n; dd++) {
      const int32_t i = by_ins[dd];
      int32_t &head = active[frame_plus_one].head;
      const int32_t t = NewTok(cost[i], 0.0f, head, state[i], frame_plus_one); head = t; num_toks++; tok[i] = t;
    }
    for (const PendingLink &k : links_in) toks[k.src_tok].links = NewLink(tok[k.dst_local], k.ilabel, k.olabel, k.graph, k.ac, toks[k.src_tok].links);
    for (size_t i = 0; i < n; i++)
      for (int32_t k = cbeg[i]; k < cbeg[i + 1]; k++)
        if (cdst[k] >= 0) toks[tok[i]].links = NewLink(tok[cdst[k]], 0, fst.olabel[carc[k]], cw[k], 0.0f, toks[tok[i]].links);
    cf.state = state; cf.cost = cost; cf.tok = tok; cf.order = order;
  }

  void Init2() {
    active.assign(1, FrameList());
    Nonemitting2(cfg.beam, {fst.start}, {0.0f}, {0u}, 1u, {});
  }

  float CutoffFromCosts(float best, size_t n, float *adaptive_beam) {     // the part of GetCutoff (:666-719) after the list walk; tmp = all costs
    const bool plain = (cfg.max_active == std::numeric_limits<int32_t>::max() && cfg.min_active == 0);
    (void)n;
    if (plain) { *adaptive_beam = cfg.beam; return best + cfg.beam; }
    const float beam_cutoff = best + cfg.beam; float min_active_cutoff = kInf, max_active_cutoff = kInf;
    if (tmp.size() > (size_t)cfg.max_active) { std::nth_element(tmp.begin(), tmp.begin() + cfg.max_active, tmp.end()); max_active_cutoff = tmp[cfg.max_active]; }
    if (max_active_cutoff < beam_cutoff) { *adaptive_beam = max_active_cutoff - best + cfg.beam_delta; return max_active_cutoff; }
    if (tmp.size() > (size_t)cfg.min_active) {
      if (cfg.min_active == 0) min_active_cutoff = best;
      else {
        std::nth_element(tmp.begin(), tmp.begin() + cfg.min_active, tmp.size() > (size_t)cfg.max_active ? tmp.begin() + cfg.max_active : tmp.end());
        min_active_cutoff = tmp[cfg.min_active];
      }
    }
    if (min_active_cutoff > beam_cutoff) { *adaptive_beam = min_active_cutoff - best + cfg.beam_delta; return min_active_cutoff; }
    *adaptive_beam = cfg.beam; return beam_cutoff;
  }

  float Emitting2() {
    const int32_t frame = (int32_t)active.size() - 1;
    active.resize(active.size() + 1);
    const size_t n = cf.state.size();
    // GetCutoff: best token = minimum cost, the FIRST such token in list order (:661-663 strict <)
    std::vector<int32_t> pos(n); for (size_t r = 0; r < n; r++) pos[cf.order[r]] = (int32_t)r;
    float best = kInf; int32_t best_i = -1; tmp.clear();
    for (int32_t i : Iota(n)) {                                                                       // PAR (min over (cost, position))
      tmp.push_back(cf.cost[i]);
      if (cf.cost[i] < best || (cf.cost[i] == best && best_i >= 0 && pos[i] < pos[best_i])) { if (cf.cost[i] == best) n_best_ties++; best = cf.cost[i]; best_i = i; }
    }
    float adaptive_beam; const float cur_cutoff = CutoffFromCosts(best, n, &adaptive_beam);
    { const size_t want = (size_t)((float)n * cfg.hash_ratio); if (want > hash_size2) hash_size2 = want; }     // PossiblyResizeHash :227-233
    float next0 = kInf, cost_offset = 0.0f;
    if (best_i != -1) {
      const int32_t s = cf.state[best_i]; const float tot = cf.cost[best_i]; cost_offset = -tot;
      for (int32_t a = fst.off[s]; a < fst.off[s + 1]; a++)
        if (fst.ilabel[a] != 0) { const float nw = fst.weight[a] + cost_offset - LogLike(frame, fst.ilabel[a]) + tot; if (nw + adaptive_beam < next0) next0 = nw + adaptive_beam; }
    }
    cost_offsets.resize(frame + 1, 0.0f); cost_offsets[frame] = cost_offset;
    // arc sequence: tokens in list order, arcs in FST order; seq_base[r] = exclusive prefix sum of the emitting out-degrees
    std::vector<uint32_t> seq_base(n + 1, 0);
    for (size_t r = 0; r < n; r++) {
      const int32_t i = cf.order[r]; uint32_t deg = 0;
      if (cf.cost[i] <= cur_cutoff) for (int32_t a = fst.off[cf.state[i]]; a < fst.off[cf.state[i] + 1]; a++) deg += fst.ilabel[a] != 0;
      seq_base[r + 1] = seq_base[r] + deg;
    }
    const uint32_t M = seq_base[n];
    // pass A (PAR): v[j] = tot_j + adaptive_beam for every arc; the cutoff in force when arc j is examined is
    // min(next0, min_{j' < j} v[j']) -- a rejected arc never lowers it (tot >= cutoff  =>  tot + beam >= cutoff)
    std::vector<float> v(M), before(M);
    auto for_arcs = [&](auto fn) {
      for (int32_t r : Iota(n)) {
        const int32_t i = cf.order[r]; if (!(cf.cost[i] <= cur_cutoff)) continue;
        uint32_t j = seq_base[r];
        for (int32_t a = fst.off[cf.state[i]]; a < fst.off[cf.state[i] + 1]; a++) if (fst.ilabel[a] != 0) fn(i, a, j++);
      }
    };
    for_arcs([&](int32_t i, int32_t a, uint32_t j) {
      const float ac = cost_offset - LogLike(frame, fst.ilabel[a]); const float tot = cf.cost[i] + ac + fst.weight[a];
      v[j] = tot + adaptive_beam;
    });
    float run = next0;
    for (uint32_t j = 0; j < M; j++) { before[j] = run; if (v[j] < run) run = v[j]; }                 // exclusive prefix-min (scan)
    const float next_cutoff = run;
    // pass B (PAR): accepted arcs claim their destination token: cost = min, creation label = min sequence number
    std::vector<int32_t> nstate; std::vector<float> ncost; std::vector<uint32_t> nlabel; std::vector<PendingLink> plinks;
    static thread_local std::vector<int32_t> index_of_state; if ((int32_t)index_of_state.size() < fst.num_states) index_of_state.assign(fst.num_states, -1);
    for_arcs([&](int32_t i, int32_t a, uint32_t j) {
      const float ac = cost_offset - LogLike(frame, fst.ilabel[a]), graph = fst.weight[a]; const float tot = cf.cost[i] + ac + graph;
      if (tot >= before[j]) return;
      if (tot >= next_cutoff) n_extra_links++;                                                        // order-sensitive event: exists only because the bound was still loose
      int32_t d = index_of_state[fst.next[a]];
      if (d < 0) { d = (int32_t)nstate.size(); index_of_state[fst.next[a]] = d; nstate.push_back(fst.next[a]); ncost.push_back(kInf); nlabel.push_back(0xFFFFFFFFu); }
      if (tot < ncost[d]) ncost[d] = tot;
      if (j < nlabel[d]) nlabel[d] = j;
      plinks.push_back(PendingLink{cf.tok[i], d, fst.ilabel[a], fst.olabel[a], graph, ac});
    });
    for (int32_t s : nstate) index_of_state[s] = -1;
    for (float c : ncost) if (c >= next_cutoff) n_extra_toks++;
    stats.push_back(FrameStat{(int32_t)n, cur_cutoff, adaptive_beam, next_cutoff, cost_offset});
    Nonemitting2(next_cutoff, nstate, ncost, nlabel, M, plinks);
    return next_cutoff;
  }

  // :308-379
  void PruneLinks(int32_t f, bool *extra_changed, bool *links_pruned, float delta) {
    *extra_changed = false; *links_pruned = false;
    bool changed = true;
    while (changed) {
      changed = false;
      for (int32_t t = active[f].head; t != -1; t = toks[t].next) {
        float tok_extra = kInf; int32_t prev = -1;
        for (int32_t l = toks[t].links; l != -1;) {
          const Tok &nt = toks[links[l].dst];
          float link_extra = nt.extra + ((toks[t].tot + links[l].ac + links[l].graph) - nt.tot);
          if (link_extra != link_extra) abort();
          if (link_extra > cfg.lattice_beam) {
            const int32_t nl = links[l].next;
            if (prev != -1) links[prev].next = nl; else toks[t].links = nl;
            free_links.push_back(l); l = nl; *links_pruned = true;
          } else {
            if (link_extra < 0.0f) link_extra = 0.0f;
            if (link_extra < tok_extra) tok_extra = link_extra;
            prev = l; l = links[l].next;
          }
        }
        if (std::fabs(tok_extra - toks[t].extra) > delta) changed = true;
        toks[t].extra = tok_extra;
      }
      if (changed) *extra_changed = true;
    }
  }

  // :545-586 (final_costs keyed by token)
  void FinalCosts(float *rel, float *best_out) {
    final_costs.clear();
    float best = kInf, best_final = kInf;
    auto one = [&](int32_t s, int32_t t) {
      const float fc = fst.final_cost[s], cost = toks[t].tot, with_final = cost + fc;
      best = std::min(cost, best); best_final = std::min(with_final, best_final);
      if (fc != kInf) final_costs.push_back({t, fc});
    };
    if (mode >= 2) for (size_t i = 0; i < cf.state.size(); i++) one(cf.state[i], cf.tok[i]);
    else for (int32_t e = cur.head; e != -1; e = cur.elems[e].tail) one(cur.elems[e].key, cur.elems[e].val);
    *rel = (best == kInf && best_final == kInf) ? kInf : best_final - best;
    *best_out = (best_final != kInf) ? best_final : best;
  }

  static bool ApproxEqual(float a, float b, float tol) {   // base/kaldi-math.h:265-275
    if (a == b) return true;
    const float diff = std::fabs(a - b);
    if (diff == kInf || diff != diff) return false;
    return diff <= tol * (std::fabs(a) + std::fabs(b));
  }

  // :385-467
  void PruneLinksFinal() {
    const int32_t f = (int32_t)active.size() - 1;
    FinalCosts(&final_relative_cost, &final_best_cost);
    finalized = true;
    std::vector<float> fc(toks.size(), final_costs.empty() ? 0.0f : kInf);
    for (auto &p : final_costs) fc[p.first] = p.second;
    for (int32_t e = cur.Clear(), n; e != -1; e = n) { n = cur.elems[e].tail; cur.Delete(e); }
    bool changed = true; const float delta = 1.0e-05f;
    while (changed) {
      changed = false;
      for (int32_t t = active[f].head; t != -1; t = toks[t].next) {
        float tok_extra = toks[t].tot + fc[t] - final_best_cost; int32_t prev = -1;
        for (int32_t l = toks[t].links; l != -1;) {
          const Tok &nt = toks[links[l].dst];
          float link_extra = nt.extra + ((toks[t].tot + links[l].ac + links[l].graph) - nt.tot);
          if (link_extra > cfg.lattice_beam) {
            const int32_t nl = links[l].next;
            if (prev != -1) links[prev].next = nl; else toks[t].links = nl;
            free_links.push_back(l); l = nl;
          } else {
            if (link_extra < 0.0f) link_extra = 0.0f;
            if (link_extra < tok_extra) tok_extra = link_extra;
            prev = l; l = links[l].next;
          }
        }
        if (tok_extra > cfg.lattice_beam) tok_extra = kInf;
        if (!ApproxEqual(toks[t].extra, tok_extra, delta)) changed = true;
        toks[t].extra = tok_extra;
      }
    }
  }

  // :488-507
  void PruneTokens(int32_t f) {
    int32_t prev = -1;
    for (int32_t t = active[f].head, n; t != -1; t = n) {
      n = toks[t].next;
      if (toks[t].extra == kInf) {
        if (prev != -1) toks[prev].next = n; else active[f].head = n;
        DeleteLinks(t); free_toks.push_back(t); toks[t].frame = -1; num_toks--;
      } else prev = t;
    }
  }

  // :515-542
  void PruneActive(float delta) {
    const int32_t cur_f = NumFramesDecoded();
    for (int32_t f = cur_f - 1; f >= 0; f--) {
      if (active[f].must_prune_links) {
        bool ec = false, lp = false;
        PruneLinks(f, &ec, &lp, delta);
        if (ec && f > 0) active[f - 1].must_prune_links = true;
        if (lp) active[f].must_prune_toks = true;
        active[f].must_prune_links = false;
      }
      if (f + 1 < cur_f && active[f + 1].must_prune_toks) { PruneTokens(f + 1); active[f + 1].must_prune_toks = false; }
    }
  }

  void Advance() {   // :588-632
    while (NumFramesDecoded() < num_frames_ready) {
      if (NumFramesDecoded() % cfg.prune_interval == 0) PruneActive(cfg.lattice_beam * cfg.prune_scale);
      if (mode >= 2) { Emitting2(); continue; }
      const float cutoff = Emitting();
      Nonemitting(cutoff);
    }
  }

  void Finalize() {  // :634-649
    const int32_t last = NumFramesDecoded();
    PruneLinksFinal();
    for (int32_t f = last - 1; f >= 0; f--) { bool b1, b2; PruneLinks(f, &b1, &b2, 0.0f); PruneTokens(f + 1); }
    PruneTokens(0);
  }
};

struct Lattice {   // GetRawLattice (:114-197) output with states identified by (frame, fst state)
  std::vector<int32_t> st_frame, st_state; std::vector<float> st_cost, st_final;   // st_final = +inf when not final
  std::vector<int32_t> arc_src, arc_dst, arc_ilabel, arc_olabel; std::vector<float> arc_graph, arc_ac;
  std::vector<FrameStat> stats; std::vector<float> cost_offsets;
  int64_t n_extra_links, n_extra_toks, n_best_ties, n_links_created, n_replay_pops, n_replay_pushes; int32_t reached_final; float final_relative_cost;
};

}  // namespace

extern "C" {

struct k3o_fst { int32_t num_states, start; const int32_t *arc_offsets, *ilabel, *olabel, *nextstate; const float *weight, *final_cost; };
struct k3o_lfd_config { float beam; int32_t max_active, min_active; float lattice_beam; int32_t prune_interval; float beam_delta, hash_ratio, prune_scale; };

// Decode one utterance: loglikes [num_frames x ld] (already scaled, as DecodableAmNnetSimple hands them over),
// tid2pdf maps a transition-id (arc ilabel) to a column.  mode 0 = literal, 1 = two-pass.  Returns a lattice handle
// (never NULL); an empty lattice (0 states) means "no tokens survived".
void *k3o_lfd_decode(const k3o_fst *f, const float *loglikes, int32_t num_frames, int64_t ld, const int32_t *tid2pdf,
                     const k3o_lfd_config *c, int32_t mode) {
  Fst fst{f->num_states, f->start, f->arc_offsets, f->ilabel, f->olabel, f->nextstate, f->weight, f->final_cost, {}};
  fst.num_ieps.assign(fst.num_states, 0);
  for (int32_t s = 0; s < fst.num_states; s++) for (int32_t a = fst.off[s]; a < fst.off[s + 1]; a++) if (fst.ilabel[a] == 0) fst.num_ieps[s]++;
  Config cfg{c->beam, c->max_active, c->min_active, c->lattice_beam, c->prune_interval, c->beam_delta, c->hash_ratio, c->prune_scale};
  Decoder d(fst, cfg, mode);
  struct StatPrinter { Decoder &d; ~StatPrinter() { if (getenv("K3O_COMP_STATS") && d.n_replay_frames) fprintf(stderr, "replay: frames %lld pops/frame %.1f comps/frame %.1f critical-path pops/frame %.1f\n", (long long)d.n_replay_frames, (double)d.n_replay_pops / d.n_replay_frames, (double)d.n_replay_comps / d.n_replay_frames, (double)d.n_replay_crit / d.n_replay_frames);
    if (getenv("K3O_COMP_STATS") && d.n_replay_pops) fprintf(stderr, "replay: pops in components without an improvement of an existing token %.3f of all; of the critical-path pops %.3f; components of >= 64 pops hold %.3f of the pops, %.3f of those without improvement\n",
        (double)d.n_replay_pops_once / d.n_replay_pops, d.n_replay_crit ? (double)d.n_replay_crit_once / d.n_replay_crit : 0.0, (double)d.n_replay_big_pops / d.n_replay_pops, d.n_replay_big_pops ? (double)d.n_replay_big_once / d.n_replay_big_pops : 0.0); } } stat_printer{d};
  d.loglikes = loglikes; d.ld = ld; d.tid2pdf = tid2pdf; d.num_frames_ready = num_frames;
  if (mode >= 2) d.Init2(); else d.Init();
  d.Advance(); d.Finalize();
  Lattice *L = new Lattice();
  L->stats = d.stats; L->cost_offsets = d.cost_offsets;
  L->n_extra_links = d.n_extra_links; L->n_extra_toks = d.n_extra_toks; L->n_best_ties = d.n_best_ties; L->n_links_created = d.n_links_created; L->n_replay_pops = d.n_replay_pops; L->n_replay_pushes = d.n_replay_pushes;
  L->reached_final = (d.final_relative_cost != kInf) ? 1 : 0;   // ReachedFinal(): FinalRelativeCost() != inf
  L->final_relative_cost = d.final_relative_cost;
  const int32_t T = d.NumFramesDecoded();
  std::vector<int32_t> id(d.toks.size(), -1);
  for (int32_t fr = 0; fr <= T; fr++) {
    if (d.active[fr].head == -1) { L->st_frame.clear(); L->st_state.clear(); L->st_cost.clear(); L->st_final.clear(); return L; }  // GetRawLattice returns false
    for (int32_t t = d.active[fr].head; t != -1; t = d.toks[t].next) {
      id[t] = (int32_t)L->st_frame.size();
      L->st_frame.push_back(fr); L->st_state.push_back(d.toks[t].state); L->st_cost.push_back(d.toks[t].tot); L->st_final.push_back(kInf);
    }
  }
  std::vector<float> fc(d.toks.size(), kInf);
  for (auto &p : d.final_costs) fc[p.first] = p.second;
  for (int32_t fr = 0; fr <= T; fr++)
    for (int32_t t = d.active[fr].head; t != -1; t = d.toks[t].next) {
      for (int32_t l = d.toks[t].links; l != -1; l = d.links[l].next) {
        const Link &k = d.links[l];
        float off = 0.0f;
        if (k.ilabel != 0) off = d.cost_offsets[fr];
        L->arc_src.push_back(id[t]); L->arc_dst.push_back(id[k.dst]); L->arc_ilabel.push_back(k.ilabel); L->arc_olabel.push_back(k.olabel);
        L->arc_graph.push_back(k.graph); L->arc_ac.push_back(k.ac - off);
      }
      if (fr == T) {
        if (!d.final_costs.empty()) { if (fc[t] != kInf) L->st_final[id[t]] = fc[t]; }
        else L->st_final[id[t]] = 0.0f;       // LatticeWeight::One()
      }
    }
  return L;
}

void k3o_lfd_sizes(const void *h, int64_t *out /* [10] */) {
  const Lattice *L = (const Lattice *)h;
  out[0] = (int64_t)L->st_frame.size(); out[1] = (int64_t)L->arc_src.size(); out[2] = (int64_t)L->stats.size();
  out[3] = L->n_extra_links; out[4] = L->n_extra_toks; out[5] = L->n_best_ties; out[6] = L->reached_final; out[7] = L->n_links_created; out[8] = L->n_replay_pops; out[9] = L->n_replay_pushes;
}
void k3o_lfd_states(const void *h, int32_t *frame, int32_t *state, float *cost, float *final_cost) {
  const Lattice *L = (const Lattice *)h; const size_t n = L->st_frame.size();
  memcpy(frame, L->st_frame.data(), 4 * n); memcpy(state, L->st_state.data(), 4 * n);
  memcpy(cost, L->st_cost.data(), 4 * n); memcpy(final_cost, L->st_final.data(), 4 * n);
}
void k3o_lfd_arcs(const void *h, int32_t *src, int32_t *dst, int32_t *ilabel, int32_t *olabel, float *graph, float *ac) {
  const Lattice *L = (const Lattice *)h; const size_t n = L->arc_src.size();
  memcpy(src, L->arc_src.data(), 4 * n); memcpy(dst, L->arc_dst.data(), 4 * n); memcpy(ilabel, L->arc_ilabel.data(), 4 * n);
  memcpy(olabel, L->arc_olabel.data(), 4 * n); memcpy(graph, L->arc_graph.data(), 4 * n); memcpy(ac, L->arc_ac.data(), 4 * n);
}
// per decoded frame: ntoks (size of the token list GetCutoff saw), cur_cutoff, adaptive_beam, next_cutoff, cost_offset
void k3o_lfd_frame_stats(const void *h, int32_t *ntoks, float *cur_cutoff, float *adaptive_beam, float *next_cutoff, float *cost_offset) {
  const Lattice *L = (const Lattice *)h;
  for (size_t i = 0; i < L->stats.size(); i++) {
    ntoks[i] = L->stats[i].ntoks; cur_cutoff[i] = L->stats[i].cur_cutoff; adaptive_beam[i] = L->stats[i].adaptive_beam;
    next_cutoff[i] = L->stats[i].next_cutoff; cost_offset[i] = L->stats[i].cost_offset;
  }
}
void k3o_lfd_free(void *h) { delete (Lattice *)h; }

}  // extern "C"
