// TEST INFRASTRUCTURE: lattice-faster-decoder.cc instantiates the decoder for the two GrammarFst types as well; distinct
// placeholder types keep those explicit instantiations compilable (they are never used by the oracle driver).
#ifndef K3_MINIFST_GRAMMAR_FST_H_
#define K3_MINIFST_GRAMMAR_FST_H_
#include "fst/fstlib.h"
namespace fst {
class ConstGrammarFst : public ConstFst<StdArc> {};
class VectorGrammarFst : public VectorFst<StdArc> {};
}
#endif
