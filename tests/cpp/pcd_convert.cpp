// CPU test helper: reads a .pcd with the header-only C++ layer (include/rsreg/pcl_compat.hpp, the
// PCL-named io functions the reference calls at src/main.cpp:53,81,87) and writes it back in
// the requested DATA mode.   pcd_convert <in.pcd> <out.pcd> <binary|binary_compressed>
#include <cstdio>
#include <cstring>

#include "rsreg/pcl_compat.hpp"

int main(int argc, char **argv)
{
    if (argc != 4) {
        std::fprintf(stderr, "usage: %s <in.pcd> <out.pcd> <binary|binary_compressed>\n", argv[0]);
        return 2;
    }
    rsreg::PointCloud<rsreg::PointXYZRGB> cloud;
    const int rc = rsreg::io::loadPCDFile(argv[1], cloud);
    if (rc != 0) {
        std::fprintf(stderr, "loadPCDFile(%s) = %d\n", argv[1], rc);
        return 1;
    }
    const int wc = std::strcmp(argv[3], "binary_compressed") == 0 ? rsreg::io::savePCDFileBinaryCompressed(argv[2], cloud)
                                                                   : rsreg::io::savePCDFileBinary(argv[2], cloud);
    std::printf("%zu %u %u %d\n", cloud.size(), cloud.width, cloud.height, (int)cloud.is_dense);
    return wc == 0 ? 0 : 1;
}
