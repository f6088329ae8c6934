// VolumeLoader.h — .dat/.raw volume files (the reference's loadVolumetricData,
// reference src/main_vulkan.cpp:26-76; file format also reference
// volume/walnut.dat and src/Volumetric.cpp).
#ifndef PATH_TRACER_VOLUME_LOADER_H
#define PATH_TRACER_VOLUME_LOADER_H

#include "glm_compat.h"

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

// Reads the "Resolution:" line of the .dat header and then resolution.x*y*z bytes of the .raw file, x
// fastest.  A header line counts when it begins with "Resolution:" and three integers follow, separated by
// spaces or tabs (what the reference's sscanf patterns accept, src/main_vulkan.cpp:42-46: the key at the very
// start of the line, the colon right behind it); other lines are ignored and the default is 1 1 1.  A short
// .raw file only warns: the missing voxels are zero.  Returns false if either file cannot be opened.
inline bool loadVolumetricData(const std::string &datFile, const std::string &rawFile,
                               std::vector<uint8_t> &volumeData, glm::ivec3 &resolution) {
  std::ifstream dat(datFile);
  if (!dat.is_open()) {
    std::cerr << "Failed to open .dat file: " << datFile << std::endl;
    return false;
  }
  resolution = glm::ivec3(1, 1, 1);
  static const char kKey[] = "Resolution:";
  std::string line;
  while (std::getline(dat, line)) {
    if (line.compare(0, sizeof kKey - 1, kKey) != 0) continue;
    int x = 0, y = 0, z = 0;
    if (std::sscanf(line.c_str() + (sizeof kKey - 1), " %d %d %d", &x, &y, &z) == 3) resolution = glm::ivec3(x, y, z);
  }
  std::ifstream raw(rawFile, std::ios::binary);
  if (!raw.is_open()) {
    std::cerr << "Failed to open .raw file: " << rawFile << std::endl;
    return false;
  }
  const size_t expected = static_cast<size_t>(resolution.x) * resolution.y * resolution.z;
  volumeData.assign(expected, 0);
  raw.read(reinterpret_cast<char *>(volumeData.data()), static_cast<std::streamsize>(expected));
  const size_t got = static_cast<size_t>(raw.gcount());
  if (got != expected)
    std::cerr << "Warning: Expected " << expected << " bytes, got " << got << " bytes" << std::endl;
  return true;
}

#endif
