/* k3host.h -- C ABI of the HOST tail of the path (no GPU work): what happens to a raw lattice after k3_decoder_get_raw_lattices.
 * Built into kaldi_amd/lib/libk3host.so from kaldi_amd/host/k3_lattice.cc (plain C++, no HIP).  Plain pointers and sizes; 0 = success,
 * negative = error with the message in k3h_last_error() (thread-local).
 *
 * Reference interfaces replaced (paths relative to the reference's src/):
 *   k3h_determinize_lattice   fst::DeterminizeLatticePhonePrunedWrapper   lat/determinize-lattice-pruned.h:284-289 (.cc:1479-1499)   [trans != NULL]
 *                             fst::DeterminizeLatticePruned               lat/determinize-lattice-pruned.h:209-214 (.cc:1190-1236)   [trans == NULL]
 *                             as called by DecodeUtteranceLatticeFaster (decoder/decoder-wrappers.cc:354-362) and by the CUDA pipeline
 *                             (cudadecoder/batched-threaded-nnet3-cuda-online-pipeline.cc:759-765)
 *   k3h_convert_lattice       fst::ConvertLattice(Lattice -> CompactLattice) fstext/lattice-utils.h:49-53 (the pipeline's --determinize-lattice=false)
 *   k3h_transitions_read      ReadKaldiObject(model, &trans_model): the transition-id -> phone / self-loop / phone-start map that
 *                             DeterminizeLatticeInsertPhones asks of TransitionInformation (lat/determinize-lattice-pruned.cc:1291-1343)
 *   k3h_clat_write            CompactLatticeWriter::Write (lat/kaldi-lattice.cc:65-94; "ark:" binary compactlattice44 / "ark,t:" text)
 *   k3h_ivector_config_read   OnlineIvectorExtractionInfo::Init                     online2/online-ivector-feature.cc:29-68
 * The lattice arrays are exactly what k3_decoder_get_raw_lattices returns for one utterance (include/k3hip.h), already trimmed or not.
 */
#ifndef K3HOST_H_
#define K3HOST_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct k3h_transitions k3h_transitions;
typedef struct k3h_clat k3h_clat;

/* fst::DeterminizeLatticePhonePrunedOptions (lat/determinize-lattice-pruned.h:145-180), same defaults */
typedef struct k3h_det_opts { float delta; int32_t max_mem; int32_t phone_determinize; int32_t word_determinize; int32_t minimize; } k3h_det_opts;
void k3h_det_opts_default(k3h_det_opts *opts);

const char *k3h_last_error(void);

int k3h_transitions_read(const char *model_rxfilename, k3h_transitions **out);
int32_t k3h_transitions_num_ids(const k3h_transitions *t);
void k3h_transitions_free(k3h_transitions *t);

/* *complete = 0 when a limit (max_mem) stopped the determinization before the beam was reached (output pruned tighter), else 1 */
int k3h_determinize_lattice(const k3h_transitions *trans, int32_t num_states, int32_t start, const float *st_final,
                            int64_t num_arcs, const int32_t *arc_src, const int32_t *arc_dst, const int32_t *arc_ilabel, const int32_t *arc_olabel,
                            const float *arc_graph, const float *arc_ac, double beam, const k3h_det_opts *opts, k3h_clat **out, int32_t *complete);
int k3h_convert_lattice(int32_t num_states, int32_t start, const float *st_final, int64_t num_arcs, const int32_t *arc_src, const int32_t *arc_dst,
                        const int32_t *arc_ilabel, const int32_t *arc_olabel, const float *arc_graph, const float *arc_ac, k3h_clat **out);

/* The host tail for a whole batch in the layout k3_decoder_get_raw_lattices returns (concatenated arrays; utterance u owns states
 * state_offsets[u]..[u+1] and arcs arc_offsets[u]..[u+1], arc end points local to the utterance): per utterance fst::Connect
 * (decoder/decoder-wrappers.cc:353) + the determinization above, on num_threads worker threads -- the CPU worker pool of the reference's
 * pipeline (cudadecoder/batched-threaded-nnet3-cuda-pipeline2.h:170-177).  graph_start = the decoding graph's start state (a lattice starts at
 * the state of frame 0 with that graph state).  h_out (nullable): a handle per utterance (NULL where no path survived), freed by the caller with
 * k3h_clat_free; the three count arrays (nullable) receive the size of each determinized lattice and whether the beam was reached. */
int k3h_postprocess_batch(const k3h_transitions *trans, int32_t num_utts, const int64_t *state_offsets, const int64_t *arc_offsets, int32_t graph_start,
                          const int32_t *st_frame, const int32_t *st_state, const float *st_final, const int32_t *arc_src, const int32_t *arc_dst, const int32_t *arc_ilabel,
                          const int32_t *arc_olabel, const float *arc_graph, const float *arc_ac, double beam, const k3h_det_opts *opts, int32_t num_threads,
                          k3h_clat **h_out, int32_t *h_clat_states, int64_t *h_clat_arcs, int32_t *h_complete);

/* sizes, then the arrays: transition-id strings are concatenated in `strings` (all final strings in state order, then all arc strings in
 * arc order); final_str_off has num_states + 1 entries, arc_str_off num_arcs + 1 entries, both index into `strings` */
int k3h_clat_sizes(const k3h_clat *c, int32_t *num_states, int64_t *num_arcs, int64_t *num_string_labels);
int k3h_clat_get(const k3h_clat *c, int32_t *start, uint8_t *is_final, float *final_graph, float *final_ac, int64_t *final_str_off,
                 int32_t *arc_src, int32_t *arc_dst, int32_t *arc_label, float *arc_graph, float *arc_ac, int64_t *arc_str_off, int32_t *strings);
int k3h_clat_scale_acoustic(k3h_clat *c, double scale);                 /* fst::ScaleLattice(fst::AcousticLatticeScale(scale), &clat) */
int k3h_clat_write(const k3h_clat *c, const char *key, const char *wspecifier);   /* one record; "ark:file" or "ark,t:file" */
void k3h_clat_free(k3h_clat *c);
/* cuda_decoder::LatticePostprocessor::GetCTM (cudadecoder/lattice-postprocessor.cc:88-110) on every lattice of a table: scales and word insertion penalty of the post-processor's
 * config file (lattice-postprocessor.h:35-76), word-level Minimum Bayes Risk decoding (lat/sausages.cc), times in seconds; CTM lines "<key> 0  <begin> <duration> <word> <conf>"
 * (cuda-pipeline-common.cc:67-142) into `out`.  Returns the bytes written (without the terminating 0), -1 on error. */
int64_t k3h_lattice_table_to_ctm(const char *lattice_rspecifier, const char *postprocessor_config_rxfilename, float decoder_frame_shift_seconds, char *out, int64_t out_cap);
/* the same with the acoustic model's file (final.mdl): needed when the config names --word-boundary-rxfilename (LatticePostprocessor::SetTransitionModel, lattice-postprocessor.h:93-95):
 * the lattice is word-aligned (lat/word-align-lattice.cc) in front of MBR, so that the CTM's times are word boundaries */
int64_t k3h_lattice_table_to_ctm_model(const char *lattice_rspecifier, const char *postprocessor_config_rxfilename, float decoder_frame_shift_seconds,
    const char *model_rxfilename, char *out, int64_t out_cap);

/* OnlineIvectorExtractionInfo(config) (online2/online-ivector-feature.cc:29-98): parse an --ivector-extraction-config file and read every file it
 * names (LDA matrix, global CMVN stats, cmvn / splice configs, diagonal UBM, i-vector extractor), with the reference's checks and messages.
 * k3h_ivector_config_get hands out what include/k3hip.h's k3_ivector_create takes; the pointers stay valid until k3h_ivector_config_free.
 *   ints[16]  = feat_dim, lda_rows, lda_cols, num_gauss, ivector_dim, left_context, right_context, ivector_period, num_gselect, num_cg_iters,
 *               cmn_window, speaker_frames, global_frames, normalize_mean, normalize_variance, online_cmvn_iextractor
 *   reals[5]  = min_post, posterior_scale, max_count, prior_offset, max_remembered_frames */
typedef struct k3h_ivector_config k3h_ivector_config;
int k3h_ivector_config_read(const char *config_rxfilename, k3h_ivector_config **out);
int k3h_ivector_config_get(const k3h_ivector_config *c, int32_t *ints, double *reals, const float **lda, const double **global_cmvn_stats, const double **gconsts,
                           const double **means_invvars, const double **inv_vars, const double **M, const double **sigma_inv);
void k3h_ivector_config_free(k3h_ivector_config *c);

#ifdef __cplusplus
}
#endif
#endif
