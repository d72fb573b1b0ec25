// Host-side declarations shared between the kernel files of libouniverse (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include "ou_kernels.h"

namespace ou {
// per-family set-up (dynamic LDS limits); called once from init_conv_kernels()
hipError_t init_chain_kernels();
hipError_t init_direct3_kernels();
hipError_t init_block3_kernels();
hipError_t init_direct4_kernels();
hipError_t init_split_kernels();
// family launchers behind launch_conv(): hipErrorInvalidConfiguration = "not a layer for this family",
// hipErrorNotSupported = "not with this fused epilogue" (the caller runs conv + FIR pass)
hipError_t launch_conv_direct(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out);
hipError_t launch_conv_direct3(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out);
// (probe: only answer whether the family takes the layer, launch nothing)
hipError_t launch_conv_direct4(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out, bool probe);
hipError_t launch_conv_direct4w(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out);
// the bf16-split form of the stride-1 k3 / k5 convs (conv_split_kernel)
hipError_t launch_conv_split(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out);
}  // namespace ou
