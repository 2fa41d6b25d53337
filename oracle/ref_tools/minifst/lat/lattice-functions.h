// TEST INFRASTRUCTURE: lattice-faster-decoder.cc includes lat/lattice-functions.h but uses nothing from it.
#include "lat/kaldi-lattice.h"
