// rollout_family.hip -- the rollout kernels of ONE (n, m, model) family (rollout_kernel.hpp), compiled once per family with
// -DISLS_FAM_NX= -DISLS_FAM_NU= -DISLS_FAM_MODEL= so that the families build in parallel.
#include "rollout_kernel.hpp"

namespace isls {
ISLS_ROLLOUT_FAMILY_DEFINE(ISLS_FAM_NX, ISLS_FAM_NU, ISLS_FAM_MODEL)
}  // namespace isls
