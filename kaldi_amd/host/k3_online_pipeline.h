// moved: see include/k3_online_pipeline.h
#include "../../include/k3_online_pipeline.h"
