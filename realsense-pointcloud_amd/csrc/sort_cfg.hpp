// sort_cfg.hpp — the rocPRIM radix sort configuration this engine uses for (uint32 key, uint32 value) pairs.
#pragma once

#include <rocprim/device/device_radix_sort.hpp>

namespace rsreg {

// 32-bit keys with 32-bit values (the cell keys of the target, the Morton keys of the source): 1024 threads x 4 items per
// workgroup and the match-based rank take 108 us for 10^6 pairs where rocPRIM's tuned default takes 146, 84 against 135 us
// at 3 x 10^5 (tools/microbench/sort_configs.hip, profiles/r03_sort_configs.txt)
using RadixCfg32 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                              rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<1024, 4>, 8,
                                                                                  rocprim::block_radix_rank_algorithm::match>,
                                              65536>;

}  // namespace rsreg
