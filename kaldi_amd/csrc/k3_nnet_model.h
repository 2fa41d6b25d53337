// k3_nnet_model.h -- host-side nnet3 model reader and layer fuser (no device code).
//
// Reads Kaldi nnet3 models (text or binary; raw Nnet or .mdl = TransitionModel + AmNnetSimple) restating
// Nnet::Read (nnet3/nnet-nnet.cc:586-628), AmNnetSimple::Read (nnet3/am-nnet-simple.cc:47-57) and the Read()
// of the components a TDNN / TDNN-F chain model is made of, then pattern-matches the component-node graph
// into FUSED nodes:   out = epilogue( sum_i in[t + o_i] * W_i^T )
// with epilogue ops executed in graph order: +bias, ReLU, per-column scale/offset (test-mode BatchNorm,
// nnet-normalize-component.cc:209-247,460-462), + alpha * residual (the Sum(Scale(a, x), y) bypass descriptor).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace k3 {

struct Scalar {            // one basic-type value following a <Tag>
  bool from_binary4 = false, from_binary8 = false;
  uint32_t raw32 = 0;
  double num = 0.0;        // text value, bool (T=1/F=0) or binary double
  int as_int() const;
  float as_float() const;
};
struct Field {
  std::vector<Scalar> scalars;
  bool is_array = false;
  int rows = 0, cols = 0;           // vector: rows = 1 (or 0 when empty)
  std::vector<float> data;
  std::vector<int> ints;            // <TimeOffsets>
};
struct RawComponent {
  std::string name, type;
  std::map<std::string, Field> fields;
  const Field *get(const std::string &tag) const;
};
struct RawModel {
  std::vector<std::string> config_lines;
  std::vector<RawComponent> components;
  bool has_am = false;
  int left_context = 0, right_context = 0;
  std::vector<float> priors;
};

// (SigmoidComponent / TanhComponent: nnet-simple-component.cc Propagate = CuMatrixBase::Sigmoid / Tanh)
enum EpiKind { kEpiRelu = 0, kEpiScaleOffset = 1, kEpiResidual = 2, kEpiSigmoid = 3, kEpiTanh = 4 };
struct EpiOp {
  int kind;
  std::vector<float> scale, offset;  // kEpiScaleOffset
  int res_node = -1;                 // kEpiResidual: fused node index, -1 = network input
  float res_scale = 1.0f;
};
struct FusedNode {
  std::string name;            // name of the last component-node folded in
  int input = -1;              // fused node index, -1 = network input
  int in_dim = 0, out_dim = 0;
  bool has_gemm = false;
  std::vector<int> offsets;    // one per K block (time offsets in frames); {0} for elementwise nodes
  std::vector<float> W;        // [out_dim x (offsets.size() * in_dim)] row-major (Kaldi linear_params_ layout)
  std::vector<float> bias;     // empty = none
  std::vector<float> W_iv;     // [out_dim x ivector_dim]: the columns that multiply ReplaceIndex(ivector, t, 0), the last part of the Append(); empty = none
  std::vector<EpiOp> ops;
  // applied to the node's output rows after `ops`: 1 = LogSoftmaxComponent, 2 = SoftmaxComponent, 3 = NormalizeComponent (a reduction over the row: its own
  // kernel)
  int row_op = 0;
  float row_param = 0.0f;      // row_op 3: target_rms
};
struct FusedModel {
  int input_dim = 0, output_dim = 0, output_node = -1;
  int ivector_dim = 0;                          // input-node name=ivector (0 = the model has none)
  int left_context = 0, right_context = 0;      // of 'output' w.r.t. 'input'
  std::vector<FusedNode> nodes;
  std::vector<float> priors;                    // AmNnetSimple priors (may be empty)
  int64_t num_params = 0;
  int num_components = 0;
};

// All three return false and fill *err on failure.
bool ReadModelFile(const std::string &path, RawModel *out, std::string *err);
bool FuseModel(const RawModel &raw, FusedModel *out, std::string *err);

}  // namespace k3
