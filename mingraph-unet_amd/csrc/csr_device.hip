// Device-side COO -> CSR-by-target for ARBITRARY graphs (the patch grid's maps are built on the host, patch_graph.cpp):
// the reference takes any (2, E) int64 edge_index (model/gat/graph_attention.py:40-58) and its scatter_add_ sums a target's
// messages in edge order, so the CSR must keep each target's sources in COO order -> a STABLE sort by target.
//   keys (targets, validated + narrowed to int32)  ->  rocprim::radix_sort_pairs (LSD radix sort: stable)  ->  col = sorted sources,
//   rowptr[j] = lower_bound(sorted targets, j).
// An id outside [0, N) sets *status (device int): the graph is rejected by the caller, as torch's indexing would raise IndexError.
#include <string.h>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "ctx.h"

namespace mgu {

__global__ __launch_bounds__(256) void csr_keys_kernel(const int64_t* __restrict__ coo, int64_t E, int N, int* __restrict__ keys,
                                                       int* __restrict__ vals, int* __restrict__ status) {
  bool bad = false;
  for (int64_t k = blockIdx.x * (int64_t)256 + threadIdx.x; k < E; k += (int64_t)gridDim.x * 256) {
    const int64_t s = coo[k], t = coo[E + k];
    if (s < 0 || s >= N || t < 0 || t >= N) bad = true;
    keys[k] = (int)min(max(t, (int64_t)0), (int64_t)N - 1);
    vals[k] = (int)min(max(s, (int64_t)0), (int64_t)N - 1);
  }
  if (bad) atomicOr(status, 1);
}
__global__ __launch_bounds__(256) void csr_rowptr_kernel(const int* __restrict__ sorted_keys, int64_t E, int N, int* __restrict__ rowptr) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j > N) return;
  int64_t lo = 0, hi = E;   // first position with key >= j
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted_keys[mid] < j) lo = mid + 1; else hi = mid;
  }
  rowptr[j] = (int)lo;
}

}  // namespace mgu

using namespace mgu;
using namespace mgud;

extern "C" int mgu_coo_to_csr_device(mgu_ctx* c, const int64_t* coo_dev, int64_t E, int num_nodes, int32_t* rowptr_dev, int32_t* col_dev,
                                     int* status_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (E < 0 || num_nodes < 0 || E >= (1ll << 31) || !rowptr_dev || !status_dev || (E > 0 && (!coo_dev || !col_dev)))
    return fail(c, MGU_ERR_INVALID, "bad coo_to_csr_device args (E < 2^31)");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  HIPCHK(c, hipMemsetAsync(status_dev, 0, sizeof(int), s));
  if (E == 0) {
    HIPCHK(c, hipMemsetAsync(rowptr_dev, 0, ((size_t)num_nodes + 1) * sizeof(int32_t), s));
    return MGU_OK;
  }
  int bits = 1;
  while ((1ll << bits) < num_nodes) ++bits;
  size_t temp_bytes = 0;
  HIPCHK(c, rocprim::radix_sort_pairs(nullptr, temp_bytes, (int*)nullptr, (int*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)E, 0, bits, s));
  const size_t o_keys = 0, o_vals = ((size_t)E * 4 + 255) / 256 * 256, o_skeys = 2 * o_vals, o_tmp = 3 * o_vals;
  int rc = ensure(c, &c->gws, &c->gws_bytes, o_tmp + temp_bytes + 256);
  if (rc) return rc;
  char* g = (char*)c->gws;
  int *keys = (int*)(g + o_keys), *vals = (int*)(g + o_vals), *skeys = (int*)(g + o_skeys);
  const int nblk = (int)std::min<int64_t>(2048, (E + 255) / 256);
  hipLaunchKernelGGL(csr_keys_kernel, dim3(nblk), dim3(256), 0, s, coo_dev, E, num_nodes, keys, vals, status_dev);
  HIPCHK(c, rocprim::radix_sort_pairs(g + o_tmp, temp_bytes, keys, skeys, vals, (int*)col_dev, (size_t)E, 0, bits, s));
  hipLaunchKernelGGL(csr_rowptr_kernel, dim3((num_nodes + 1 + 255) / 256), dim3(256), 0, s, skeys, E, num_nodes, (int*)rowptr_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}
