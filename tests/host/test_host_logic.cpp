// Host-only logic of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer.
// Build + run: tests/test_host_logic.py::test_host_logic_under_sanitizers
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all tests/host/test_host_logic.cpp
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../pir_amd/csrc/host_logic.h"

using namespace pir_host;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); std::exit(1); } } while (0)

static void test_trace_workspace() {
  const long sizes[] = {1, 2, 31, 32, 33, 255, 256, 257, 4096, 65535, 65536, 200000, 640000, 4194304, 0x7fffffffL / 4};
  for (long n : sizes)
    for (long chunk : {0L, 1L, 7L, 200000L, n})
      for (int mb : {1, 18, 64}) {
        if (chunk == 1 && n > 100000) continue;                 // (nchunks would be huge but still legal: skip for time)
        const int nchunks = trace_chunks(n, chunk);
        CHECK(nchunks >= 1 && (long)nchunks * (chunk <= 0 ? n : chunk) >= n);
        size_t off[kTrArrays];
        const size_t total = trace_ws_offsets(n, nchunks, mb, off);
        // arrays are 256-byte aligned, in order, non-overlapping, and the header holds counters + flags
        const size_t want[kTrArrays] = {sizeof(int) * kTraceCounters + sizeof(int) * (size_t)(mb + 2) * (size_t)nchunks,
                                        4u * (size_t)n, 4u * (size_t)n, 4u * (size_t)n, 4u * (size_t)n, 4u * (size_t)n,
                                        4u * (size_t)n, 4u * (size_t)n, 4u * (size_t)n, (size_t)n, 4u * (size_t)n};
        for (int a = 0; a < kTrArrays; ++a) {
          CHECK(off[a] % 256 == 0);
          const size_t end = off[a] + want[a];
          CHECK(end <= (a + 1 < kTrArrays ? off[a + 1] : total));
        }
        CHECK(total % 256 == 0);
        if (n <= 65536 && nchunks <= 64) {
          // really touch a buffer of that size through the offsets: ASan sees any overrun
          std::vector<unsigned char> buf(total);
          for (int a = 0; a < kTrArrays; ++a) std::memset(buf.data() + off[a], a + 1, want[a]);
          for (int a = 0; a < kTrArrays; ++a) CHECK(want[a] == 0 || buf[off[a] + want[a] - 1] == a + 1);
        }
      }
  // the allocation-free entry point's array (pir_trace_static): header for 64 chunks / 64 rounds + 9 int arrays + 1 byte array
  const long N = PIR_TRACE_STATIC_MAX_RAYS;
  size_t off[kTrArrays];
  const size_t header = align256(sizeof(int) * (kTraceCounters + (64 + 2) * 64));
  CHECK(trace_ws_offsets(N, 64, 64, off) <= header + 9 * sizeof(int) * (size_t)N + (size_t)N);
}

static void test_trace_args() {
  int x = 0;
  const void* p = &x;
  CHECK(trace_check_args(p, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 16, 128, 18, nullptr, nullptr, nullptr, nullptr) == PIR_OK);
  CHECK(trace_check_args(p, -1, p, p, p, p, p, 16, 128, 18, p, p, p, p) == PIR_ERR_ARG);
  CHECK(trace_check_args(p, 0x7fffffffL / 4 + 1, p, p, p, p, p, 16, 128, 18, p, p, p, p) == PIR_ERR_ARG);
  CHECK(trace_check_args(nullptr, 5, p, p, p, p, p, 16, 128, 18, p, p, p, p) == PIR_ERR_ARG);
  CHECK(trace_check_args(p, 5, p, p, p, p, p, 16, 128, 18, p, p, nullptr, p) == PIR_ERR_ARG);
  CHECK(trace_check_args(p, 5, p, p, p, p, p, 31, 128, 18, p, p, p, p) == PIR_ERR_ARG);
  CHECK(trace_check_args(p, 5, p, p, p, p, p, 16, 1, 18, p, p, p, p) == PIR_ERR_ARG);
  CHECK(trace_check_args(p, 5, p, p, p, p, p, 16, 128, 0, p, p, p, p) == PIR_ERR_ARG);
  CHECK(trace_check_args(p, 5, p, p, p, p, p, 16, 128, 65, p, p, p, p) == PIR_ERR_ARG);
  CHECK(trace_check_args(p, 5, p, p, p, p, p, 0, 2, 1, p, p, p, p) == PIR_OK);
}

static void test_wgrad_split() {
  const long slab_cap = 64L * 40 * 128 * 128 / (256L * 256);     // kKSplitMax * kMaxJobs * 128^2 / 256^2
  for (int n_jobs = 1; n_jobs <= 10; ++n_jobs)
    for (int steps : {1, 7, 8, 9, 64, 127, 8192})
      for (long cus : {1L, 8L, 64L, 256L, 304L, 100000L}) {
        int cost[10];
        for (int j = 0; j < n_jobs; ++j) cost[j] = (j == 0 ? 5 : 10) * (j == n_jobs - 1 && n_jobs > 2 ? 1 : 2);
        int first[12];
        const int n_wg = wgrad256_split(n_jobs, cost, steps, cus, slab_cap, first);
        CHECK(first[0] == 0 && n_wg >= n_jobs && first[n_jobs] == n_wg);
        CHECK(n_wg <= (cus < slab_cap ? cus : slab_cap) + n_jobs);   // at most one forced slice per job beyond the round
        CHECK(n_wg <= slab_cap + n_jobs);
        for (int j = 0; j < n_jobs; ++j) {
          const int k = first[j + 1] - first[j];
          CHECK(k >= 1 && (k == 1 || steps / k >= 8));               // >= 8 K steps per slice unless a single slice
          // every slice of the kernel's partition [ks * steps / k, (ks + 1) * steps / k) is inside [0, steps]
          for (int ks = 0; ks < k; ++ks) {
            const int it0 = (int)((long)ks * steps / k), it1 = (int)((long)(ks + 1) * steps / k);
            CHECK(0 <= it0 && it0 <= it1 && it1 <= steps);
          }
        }
        for (int j = n_jobs; j < 12; ++j) CHECK(first[j] == n_wg);
      }
  for (int n_jobs : {1, 2, 9, 30, 40})
    for (int n_rows : {32, 224, 256, 288, 8192, 262144})
      for (long slots : {2L, 16L, 512L, 608L}) {
        const int k = wgrad128_ksplit(n_jobs, n_rows, 32, slots, 64);
        CHECK(k >= 1 && k <= 64 && (k == 1 || n_rows / 32 / k >= 8));
      }
}

static void test_bwd_cap() {
  const size_t fixed = 171315200, row = 32 * 1024 + 512;
  CHECK(bwd_chunk_cap(0, fixed, row, 262144) == 0);
  CHECK(bwd_chunk_cap(fixed + 32 * row + 255, fixed, row, 262144) == 0);
  CHECK(bwd_chunk_cap(fixed + 32 * row + 256, fixed, row, 262144) == 32);
  CHECK(bwd_chunk_cap(fixed + 33 * row + 256, fixed, row, 262144) == 32);
  CHECK(bwd_chunk_cap(fixed + 64 * row + 256, fixed, row, 262144) == 64);
  CHECK(bwd_chunk_cap((size_t)1 << 40, fixed, row, 262144) == 262144);
  for (long pts : {4096L, 8192L, 131072L, 262144L}) {
    const size_t need = fixed + (size_t)((pts + 31) / 32 * 32) * row + 256;     // pir_sdf_backward_scratch_bytes
    CHECK(bwd_chunk_cap(need, fixed, row, 262144) == pts);
  }
}

int main() {
  test_trace_workspace();
  test_trace_args();
  test_wgrad_split();
  test_bwd_cap();
  std::puts("host logic ok");
  return 0;
}
