// Segment sums by node id (segsum.hip): backward of the table-projection mode.
#pragma once
#include "common.h"

namespace pmgt {

int64_t seg_sort_temp_bytes(int M);
// keys/vals/skeys/perm: [M] uint32 scratch; seg_off: [n_rows + 1]; stable order by (ids[m], m)
int seg_sort(const int64_t* ids, int M, int n_rows, uint32_t* keys, uint32_t* vals, uint32_t* skeys, uint32_t* perm, int* seg_off,
             void* temp, int64_t temp_bytes, hipStream_t st);
int64_t seg_part_elems(int M, int cols);
// out[n, :cols] = sum of src[perm[p], :cols] over the sorted positions p of segment n (zeros for empty segments)
template <typename T, typename TO>
int seg_sum(const T* src, int64_t ld, const uint32_t* skeys, const uint32_t* perm, const int* seg_off, int M, int n_rows, int cols,
            TO* out, float* part, hipStream_t st);

}  // namespace pmgt
