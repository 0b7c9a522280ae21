// CacheUtils.cpp -- reader/writer for the format of 453-skeleton/CacheUtils.cpp:5-111.
#include "CacheUtils.h"

#include <cstdint>
#include <fstream>
#include <iostream>

namespace {
struct Header { int32_t dim[3]; float min[3]; float voxelSize; };

bool readHeader(std::ifstream& in, VoxelGrid& g, uint64_t& count) {
    Header h;
    in.read(reinterpret_cast<char*>(h.dim), sizeof h.dim);
    in.read(reinterpret_cast<char*>(h.min), sizeof h.min);
    in.read(reinterpret_cast<char*>(&h.voxelSize), sizeof h.voxelSize);
    in.read(reinterpret_cast<char*>(&count), sizeof count);
    if (!in) return false;
    g.dimX = h.dim[0]; g.dimY = h.dim[1]; g.dimZ = h.dim[2];
    g.minX = h.min[0]; g.minY = h.min[1]; g.minZ = h.min[2];
    g.voxelSize = h.voxelSize;
    return true;
}
}  // namespace

bool saveVoxelGrid(const std::string& filename, const VoxelGrid& grid) {
    std::ofstream out(filename, std::ios::binary);
    if (!out) {
        std::cerr << "Cannot open file for writing: " << filename << "\n";
        return false;
    }
    const int32_t dim[3] = { grid.dimX, grid.dimY, grid.dimZ };
    const float mn[3] = { grid.minX, grid.minY, grid.minZ };
    const uint64_t count = grid.data.size();
    out.write(reinterpret_cast<const char*>(dim), sizeof dim);
    out.write(reinterpret_cast<const char*>(mn), sizeof mn);
    out.write(reinterpret_cast<const char*>(&grid.voxelSize), sizeof grid.voxelSize);
    out.write(reinterpret_cast<const char*>(&count), sizeof count);
    out.write(reinterpret_cast<const char*>(grid.data.data()), (std::streamsize)count);
    return static_cast<bool>(out);
}

bool loadVoxelGrid(const std::string& filename, VoxelGrid& grid) {
    std::ifstream in(filename, std::ios::binary);
    if (!in) {
        std::cerr << "Cannot open file for reading: " << filename << "\n";
        return false;
    }
    uint64_t count = 0;
    if (!readHeader(in, grid, count)) return false;
    grid.data.resize(count);
    in.read(reinterpret_cast<char*>(grid.data.data()), (std::streamsize)count);
    return static_cast<bool>(in);
}

bool loadVoxelGridPartial(const std::string& filename, VoxelGrid& grid, int startLayer, int numLayers) {
    std::ifstream in(filename, std::ios::binary);
    if (!in) {
        std::cerr << "Cannot open file for reading: " << filename << "\n";
        return false;
    }
    uint64_t count = 0;
    if (!readHeader(in, grid, count)) return false;
    if (startLayer < 0 || startLayer >= grid.dimZ || startLayer + numLayers > grid.dimZ) {
        std::cerr << "Requested subvolume layers are out of bounds." << std::endl;
        return false;
    }
    const size_t layer = (size_t)grid.dimX * grid.dimY;
    std::vector<VoxelState> slab(layer * (size_t)numLayers);
    in.seekg((std::streamoff)(layer * (size_t)startLayer), std::ios::cur);
    in.read(reinterpret_cast<char*>(slab.data()), (std::streamsize)slab.size());
    if (!in) return false;
    grid.dimZ = numLayers;
    grid.minZ += startLayer * grid.voxelSize;
    grid.data = std::move(slab);
    return true;
}
