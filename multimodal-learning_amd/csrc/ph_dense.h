// Internal convenience header: the dense / CRD / optimiser entry points ARE the public C-ABI.
#pragma once
#include "../../include/pathomic_hip.h"
