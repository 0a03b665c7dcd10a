// AddressSanitizer / UBSan driver for the HOST routines of the C-ABI (include/mgunet.h): the patch-graph index maps and
// the NULL / bad-argument paths that return before any HIP call.  CPU only (GPU sanitizers are not available on the
// pool).  Built by `make -C mingraph-unet_amd/csrc asan-host`; run by tests/test_abi_host.py::test_host_routines_under_asan.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../include/mgunet.h"

#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      fprintf(stderr, "host_abi_check: %s failed at line %d\n", #cond, __LINE__); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

static int check_grid(int H, int W, int p) {
  int64_t E = -1;
  int nph = 0, npw = 0;
  CHECK(mgu_patch_graph_build(H, W, p, nullptr, nullptr, nullptr, &E, &nph, &npw) == MGU_OK);
  CHECK(nph == (H + p - 1) / p && npw == (W + p - 1) / p);
  CHECK(E == 2 * ((int64_t)nph * (npw - 1) + (int64_t)(nph - 1) * npw));
  const int N = nph * npw;
  // exact-size buffers: any write past the end is an ASan report
  std::vector<int64_t> coo((size_t)2 * E);
  std::vector<int32_t> rowptr((size_t)N + 1), col((size_t)E);
  CHECK(mgu_patch_graph_build(H, W, p, E ? coo.data() : nullptr, rowptr.data(), E ? col.data() : nullptr, nullptr, nullptr, nullptr) ==
        MGU_OK);
  CHECK(rowptr[0] == 0 && rowptr[N] == E);
  for (int64_t k = 0; k < E; ++k) {
    CHECK(coo[k] >= 0 && coo[k] < N && coo[E + k] >= 0 && coo[E + k] < N);
    const int64_t d = coo[k] > coo[E + k] ? coo[k] - coo[E + k] : coo[E + k] - coo[k];
    CHECK(d == 1 || d == npw);   // 4-connectivity
  }
  for (int n = 0; n < N; ++n) {
    CHECK(rowptr[n + 1] - rowptr[n] <= 4);
    for (int q = rowptr[n]; q < rowptr[n + 1]; ++q) CHECK(col[q] >= 0 && col[q] < N);
  }
  // CSR only (no COO buffer from the caller), rowptr only
  std::vector<int32_t> rp2((size_t)N + 1), col2((size_t)E);
  CHECK(mgu_patch_graph_build(H, W, p, nullptr, rp2.data(), E ? col2.data() : nullptr, nullptr, nullptr, nullptr) == MGU_OK);
  CHECK(rp2 == rowptr && col2 == col);
  CHECK(mgu_patch_graph_build(H, W, p, nullptr, rp2.data(), nullptr, nullptr, nullptr, nullptr) == MGU_OK);
  return 0;
}

int main() {
  // grids: the reference's own smoke sizes, ragged, single patch, single row / column, the BASELINE sizes
  const int cases[][3] = {{128, 128, 32}, {130, 140, 32}, {512, 512, 16}, {1024, 1024, 16}, {16, 16, 16}, {1, 1, 1},
                          {7, 300, 16},   {300, 7, 16},   {37, 45, 16},   {5, 5, 1}};
  for (auto& c : cases)
    if (check_grid(c[0], c[1], c[2])) return 1;
  CHECK(mgu_patch_graph_build(0, 8, 4, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == MGU_ERR_INVALID);
  CHECK(mgu_patch_graph_build(8, 8, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == MGU_ERR_INVALID);

  // arbitrary COO -> CSR: duplicates, self loops, isolated nodes, empty graph, out-of-range ids
  {
    const int N = 7;
    const int64_t coo[] = {3, 3, 0, 6, 6, 2, 2,   /* sources */
                           1, 1, 0, 5, 5, 5, 0};  /* targets */
    std::vector<int32_t> rowptr(N + 1), col(7);
    CHECK(mgu_coo_to_csr(coo, 7, N, rowptr.data(), col.data()) == MGU_OK);
    const int32_t want_rp[] = {0, 2, 4, 4, 4, 4, 7, 7}, want_col[] = {0, 2, 3, 3, 6, 6, 2};
    for (int i = 0; i <= N; ++i) CHECK(rowptr[i] == want_rp[i]);
    for (int i = 0; i < 7; ++i) CHECK(col[i] == want_col[i]);   // stable: COO order kept inside a target's row
    CHECK(mgu_coo_to_csr(nullptr, 0, N, rowptr.data(), nullptr) == MGU_OK);
    for (int i = 0; i <= N; ++i) CHECK(rowptr[i] == 0);
    CHECK(mgu_coo_to_csr(nullptr, 0, 0, rowptr.data(), nullptr) == MGU_OK);
    const int64_t bad[] = {0, 7, 0, 0};
    CHECK(mgu_coo_to_csr(bad, 2, N, rowptr.data(), col.data()) == MGU_ERR_INVALID);
    const int64_t neg[] = {0, 0, -1, 0};
    CHECK(mgu_coo_to_csr(neg, 2, N, rowptr.data(), col.data()) == MGU_ERR_INVALID);
    CHECK(mgu_coo_to_csr(coo, 7, N, nullptr, col.data()) == MGU_ERR_INVALID);
    CHECK(mgu_coo_to_csr(nullptr, 3, N, rowptr.data(), col.data()) == MGU_ERR_INVALID);
    CHECK(mgu_coo_to_csr(coo, -1, N, rowptr.data(), col.data()) == MGU_ERR_INVALID);
  }
  printf("host_abi_check ok\n");
  return 0;
}
