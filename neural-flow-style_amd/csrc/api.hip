// Error channel, version and device query of libnfs_hip.so.
#include "common.h"

namespace nfs {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace nfs

extern "C" {
int nfs_version(void) { return 151; }   // 1.5.1: + nfs_advect_bwd_adam_fwd_live_ever; 1.5: live masks (nfs_live_mask_words, nfs_advect_fwd_live, nfs_advect_bwd_adam_fwd_live, nfs_rotate_bwd_coef_live, nfs_rotate_live_workspace_ints); 1.4.1: the packed filters carry the 16 x 16 fragment order once more as bf16 limb planes (nfs_conv3x3_packed_floats grew by 54 (+ 73.5) floats per (ci, co)); 1.4: split-limb GEMMs by default (nfs_gemm_mode), round 2's limb planes gone from the packed filters (nfs_conv3x3_packed_floats shrank), nfs_gemm_timer_read_kind, nfs_slab_pack, nfs_lap_up_rms, the colour-chain operators; 1.3.3: packed filters carry two more layouts (nfs_conv3x3_packed_floats grew); 1.3.2: nfs_rotate_bwd overwrite; 1.3.1: addend_unmasked; 1.3: ReLU bit cache argument of the conv entry points; 1.2.1: + content loss; 1.2: fused pool conv, g2p, advect+Adam, max|g| hand-off
const char* nfs_last_error(void) { return nfs::g_err; }
int nfs_device_cus(void) {
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    nfs::set_error("nfs_device_cus: no HIP device");
    return NFS_ELAUNCH;
  }
  return p.multiProcessorCount;
}
}
