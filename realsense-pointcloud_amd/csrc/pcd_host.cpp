// pcd_host.cpp — host-only helpers of the PCD reader/writer exported through the C ABI:
// the LZF coder of "DATA binary_compressed" bodies (include/rsreg/lzf.hpp), so that the Python
// host layer (cloud.py) and the header-only C++ layer (pcl_compat.hpp) share one implementation.
// Reference: pcl::io::loadPCDFile / savePCDFile* as called from src/main.cpp:53,81,87.
#include "../../include/rsreg.h"
#include "../../include/rsreg/lzf.hpp"

extern "C" {

size_t rsreg_lzf_max_encoded_size(size_t n) { return rsreg::lzf::max_encoded_size(n); }

size_t rsreg_lzf_encode(const void *in, size_t n, void *out, size_t capacity)
{
    if ((n && !in) || !out) return 0;
    return rsreg::lzf::encode(static_cast<const uint8_t *>(in), n, static_cast<uint8_t *>(out), capacity);
}

size_t rsreg_lzf_decode(const void *in, size_t n, void *out, size_t capacity)
{
    if ((n && !in) || !out) return 0;
    return rsreg::lzf::decode(static_cast<const uint8_t *>(in), n, static_cast<uint8_t *>(out), capacity);
}

}  // extern "C"
