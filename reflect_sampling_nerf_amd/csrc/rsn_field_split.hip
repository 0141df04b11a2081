// rsn_field_split.hip -- the split-bf16 instantiations of the field kernel (MODE 1 = RSN_MMA_BF16X6: eval and training,
// MODE 2 = RSN_MMA_BF16X3: eval), compiled WITHOUT -amdgpu-mfma-vgpr-form (rsn_field_kernel.h says why).
#include "rsn_field_kernel.h"

int rsn_launch_field_split(int width, bool train, int mode, long long grid, hipStream_t st, const FieldJobs& J) {
  RSN_REQUIRE(mode == 1 || (mode == 2 && !train), RSN_ERR_INVALID_ARGUMENT, "split launch: mode=%d train=%d", mode, (int)train);
#define RSN_LAUNCH(NBV)                                                                                \
  do {                                                                                                 \
    if (train)                                                                                         \
      hipLaunchKernelGGL((rsn_field_kernel<NBV, true, 1>), dim3((unsigned)grid), dim3(256), 0, st, J);  \
    else if (mode == 1)                                                                                \
      hipLaunchKernelGGL((rsn_field_kernel<NBV, false, 1>), dim3((unsigned)grid), dim3(256), 0, st, J); \
    else                                                                                               \
      hipLaunchKernelGGL((rsn_field_kernel<NBV, false, 2>), dim3((unsigned)grid), dim3(256), 0, st, J); \
  } while (0)
  switch (width) {
    case 256: RSN_LAUNCH(8); break;
    case 128: RSN_LAUNCH(4); break;
    case 64: RSN_LAUNCH(2); break;
    default: RSN_REQUIRE(false, RSN_ERR_UNSUPPORTED, "width=%d unsupported", width);
  }
#undef RSN_LAUNCH
  return RSN_OK;
}
