// Host-side index maps of the patch graph (bit-exact with the reference's emission order).
// Replaces PatchGraphConstructor.construct_patch_graph,
// preprocessing/graph_construction/patch_graph_construction.py:49-102.
#include <stdint.h>

#include <vector>

#include "../../include/mgunet.h"

extern "C" int mgu_coo_to_csr(const int64_t* coo, int64_t E, int num_nodes, int32_t* rowptr, int32_t* col) {
  if (E < 0 || num_nodes < 0 || (E > 0 && !coo) || !rowptr) return MGU_ERR_INVALID;
  const int64_t* src = coo;
  const int64_t* tgt = coo + E;
  for (int i = 0; i <= num_nodes; ++i) rowptr[i] = 0;
  for (int64_t k = 0; k < E; ++k) {
    if (tgt[k] < 0 || tgt[k] >= num_nodes || src[k] < 0 || src[k] >= num_nodes) return MGU_ERR_INVALID;
    rowptr[tgt[k] + 1]++;
  }
  for (int i = 0; i < num_nodes; ++i) rowptr[i + 1] += rowptr[i];
  if (col) {
    std::vector<int32_t> fill(rowptr, rowptr + num_nodes);
    for (int64_t k = 0; k < E; ++k) col[fill[tgt[k]]++] = (int32_t)src[k];  // stable: COO order kept per target
  }
  return MGU_OK;
}

extern "C" int mgu_patch_graph_build(int H, int W, int patch, int64_t* coo, int32_t* rowptr, int32_t* col,
                                     int64_t* E_out, int* nph_out, int* npw_out) {
  if (H <= 0 || W <= 0 || patch <= 0) return MGU_ERR_INVALID;
  const int nph = (H + patch - 1) / patch;  // ceil-div grid, :67-68
  const int npw = (W + patch - 1) / patch;
  const int64_t E = 2 * ((int64_t)nph * (npw - 1) + (int64_t)(nph - 1) * npw);
  if (E_out) *E_out = E;
  if (nph_out) *nph_out = nph;
  if (npw_out) *npw_out = npw;
  if (!coo && !rowptr && !col) return MGU_OK;
  std::vector<int64_t> local;
  int64_t* c = coo;
  if (!c) {
    local.resize((size_t)(2 * E > 0 ? 2 * E : 1));
    c = local.data();
  }
  int64_t k = 0;
  for (int r = 0; r < nph; ++r) {
    for (int q = 0; q < npw; ++q) {
      const int64_t n = (int64_t)r * npw + q;
      if (q + 1 < npw) {  // (n -> right), (right -> n): :81-84
        c[k] = n, c[E + k] = n + 1, ++k;
        c[k] = n + 1, c[E + k] = n, ++k;
      }
      if (r + 1 < nph) {  // (n -> down), (down -> n): :86-89
        c[k] = n, c[E + k] = n + npw, ++k;
        c[k] = n + npw, c[E + k] = n, ++k;
      }
    }
  }
  if (rowptr) return mgu_coo_to_csr(c, E, nph * npw, rowptr, col);
  return MGU_OK;
}
