// libccsp_hip.so -- MI355X (gfx950 / CDNA4) implementation of the Diffusion-CCSP sampling path
// behind the C ABI of include/ccsp.h.  See DESIGN.md for the data layout and the kernel list.
//
// One network evaluation (reference networks/denoise_fn.py:453-537) is three launches:
//   k_ugemm   U[r,:]  = pose_emb[node(r),:] . Wp[type,slot]^T       (k_rowgemm<H,2H>) fp32 MFMA 32x32x2, LDS tiled
//   k_edge    O[k,s,:] = Dec( SiLU( U[u0(k)] + U[u1(k)] )[s-half] )      (U rows carry geometry + time terms)
//   k_node    eps[n] = ordered sum over the node's CSR / sqrt(cnt), mask fill; then the fused
//             Langevin / ancestral update of the pose rows and the pose encoder for the next
//             evaluation.
// No atomics anywhere: sums follow the reference's (type, edge, slot) order.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <algorithm>
#include <atomic>
#include <thread>
#include <dlfcn.h>
#include <type_traits>
#include <vector>

#include "../../include/ccsp.h"
#include "ccsp_philox.h"
#include "ccsp_plan.h"
#include "ccsp_trace.h"

namespace {

#include "ccsp_common.h"
#include "ccsp_kernels_setup.h"
#include "ccsp_kernels_f32.h"
#include "ccsp_kernels_node.h"
#include "ccsp_energy.h"
#include "ccsp_bf16x3.h"
#include "ccsp_f16x2.h"
#include "ccsp_edge_fb.h"    // energy mode, round 6: decoder forward + backward in one kernel
#ifdef CCSP_EXPERIMENTS
#include "ccsp_fused.h"      // the one-launch evaluation (k_eval_fused*, k_rowgemm_h2d): bitwise equal, slower at every batch size
#endif
#include "ccsp_struct.h"
#include "ccsp_hmc.h"
#include "ccsp_host_util.h"
}  // namespace

#include "ccsp_host_rccl.h"
#include "ccsp_host_objects.h"
namespace {
#include "ccsp_launch.h"
#include "ccsp_launch_struct.h"
#include "ccsp_launch_energy.h"
#include "ccsp_chain.h"
}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================

namespace {
#include "ccsp_graph_build.h"
#include "ccsp_compose.h"
}  // namespace

extern "C" {

#include "ccsp_abi_model.h"
#include "ccsp_abi_graph.h"
#include "ccsp_abi_eval.h"
}  // extern "C"

#include "ccsp_abi_compose.h"
