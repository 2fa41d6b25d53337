// TEST INFRASTRUCTURE: stands in for the reference's fstext/deterministic-fst.h (which needs OpenFst's matchers).  chain/chain-supervision.h only
// derives a class from the abstract interface below (reference: fstext/deterministic-fst.h, class DeterministicOnDemandFst).
#ifndef K3_MINIFST_DETERMINISTIC_FST_H_
#define K3_MINIFST_DETERMINISTIC_FST_H_
#include "fst/fstlib.h"
namespace fst {
template <class Arc> class DeterministicOnDemandFst {
 public:
  typedef typename Arc::StateId StateId; typedef typename Arc::Weight Weight; typedef typename Arc::Label Label;
  virtual StateId Start() = 0; virtual Weight Final(StateId s) = 0; virtual bool GetArc(StateId s, Label ilabel, Arc *oarc) = 0;
  virtual ~DeterministicOnDemandFst() {}
};
}
#endif
