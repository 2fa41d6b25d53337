// moved: the pipeline classes are part of the drop-in boundary (include/); this forwarder keeps the host programs building
#include "../../include/k3_pipeline.h"
