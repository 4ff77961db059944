// vlc_loader.hpp -- streaming .vlc reader (vlc_loader.cpp); reference: src/persistence.rs:149-176.
#pragma once

#include <cstdint>

namespace vl {

class GpuFlatIndex;
class HnswIndex;
struct VlcDoc;

int vlc_open(const char* path, VlcDoc** out);  // map + structural pass + header validation (host only)
void vlc_close(VlcDoc* d);
const char* vlc_name(const VlcDoc* d);
void vlc_info(const VlcDoc* d, int* index_type, int* metric, uint64_t* dim, uint64_t* rows, uint64_t* vector_count,
              uint64_t* dimension);
int vlc_side_table(const VlcDoc* d, uint64_t* ids, uint64_t* text_off, uint64_t* text_len, uint64_t* meta_off,
                   uint64_t* meta_len);
int vlc_read_values(const VlcDoc* d, uint64_t first, uint64_t n, double* out);  // host threads
int vlc_build_index(const VlcDoc* d, int device, GpuFlatIndex** out_flat, HnswIndex** out_hnsw);

}  // namespace vl
