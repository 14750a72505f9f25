// qf_sort.hip — sorted k-mer index for k > 8 (the reference accepts -kmatch 5..32, src/qmodel.cpp:773-779): every
// x-sequence's k-mers (64-bit) are radix-sorted with their positions, segment by segment (rocPRIM via hipCUB), and the
// seeding kernels binary-search them instead of reading direct-addressed buckets.  Built once per reference set.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "qf_kernels.hpp"

namespace qf {

// key = k-mer starting at each position (big-endian base-4, makeKmer src/fastseq.cpp:27-35); positions past len-k get
// the all-ones key and sort (stably) behind every real k-mer.
__global__ void k_kmer_keys(const uint8_t* __restrict__ tok, const uint64_t* __restrict__ off, uint32_t k,
                            unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t x = blockIdx.y;
  const uint64_t b = off[x], len = off[x + 1] - b;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  unsigned long long km = ~0ull;
  if (len >= k && i <= len - k) {
    km = 0;
    for (uint32_t a = 0; a < k; ++a) km = km * 4 + tok[b + i + a];
  }
  keys[b + i] = km;
  vals[b + i] = (uint32_t)i;
}

// Returns 0 on success, a hipError_t value otherwise.  keys_out / pos_out: sorted per segment.
int sort_kmer_index(const uint8_t* tok, const uint64_t* d_off, const int* d_off32, uint32_t n_seqs, uint64_t total,
                    uint64_t max_len, uint32_t k, unsigned long long* keys_tmp, uint32_t* vals_tmp,
                    unsigned long long* keys_out, uint32_t* pos_out, void** temp, size_t* temp_cap, hipStream_t s) {
  const dim3 grid((uint32_t)((max_len + 255) / 256), n_seqs);
  hipLaunchKernelGGL(k_kmer_keys, grid, dim3(256), 0, s, tok, d_off, k, keys_tmp, vals_tmp);
  size_t need = 0;
  hipError_t e = hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, need, keys_tmp, keys_out, vals_tmp, pos_out, (int)total,
                                                             (int)n_seqs, d_off32, d_off32 + 1, 0, 64, s);
  if (e != hipSuccess) return (int)e;
  if (need > *temp_cap) {
    if (*temp) (void)hipFree(*temp);
    *temp = nullptr;
    *temp_cap = 0;
    if ((e = hipMalloc(temp, need + 256)) != hipSuccess) return (int)e;
    *temp_cap = need + 256;
  }
  e = hipcub::DeviceSegmentedRadixSort::SortPairs(*temp, need, keys_tmp, keys_out, vals_tmp, pos_out, (int)total, (int)n_seqs,
                                                  d_off32, d_off32 + 1, 0, 64, s);
  return (int)e;
}

int sort_class_list(uint32_t* keys, uint32_t* list, uint32_t n, uint32_t* keys_tmp, uint32_t* list_tmp, void** temp,
                    size_t* temp_cap, hipStream_t s) {
  size_t need = 0;
  hipError_t e = hipcub::DeviceRadixSort::SortPairsDescending(nullptr, need, keys, keys_tmp, list, list_tmp, (int)n, 0, 24, s);
  if (e != hipSuccess) return (int)e;
  if (need > *temp_cap) {
    if (*temp) (void)hipFree(*temp);
    *temp = nullptr;
    *temp_cap = 0;
    if ((e = hipMalloc(temp, need + 256)) != hipSuccess) return (int)e;
    *temp_cap = need + 256;
  }
  e = hipcub::DeviceRadixSort::SortPairsDescending(*temp, need, keys, keys_tmp, list, list_tmp, (int)n, 0, 24, s);
  if (e != hipSuccess) return (int)e;
  return (int)hipMemcpyAsync(list, list_tmp, (size_t)n * 4, hipMemcpyDeviceToDevice, s);
}

}  // namespace qf
