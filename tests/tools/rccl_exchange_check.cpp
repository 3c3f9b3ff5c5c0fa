// rccl_exchange_check.cpp — TEST DRIVER for include/blosc_gpu_rccl.h (run by tests/test_gpu_rccl_exchange.py; the product does not contain it).
// A C caller's view of SURVEY 8e: a many-chunk buffer is compressed, every rank takes its blosc_gpu_partition() range, the cbytes table
// is all-gathered and the compressed chunks are consolidated on one rank / on every rank / scattered back - on RCCL, without Python.
//   rccl_exchange_check threads [ndev]   one process, one thread per GPU (blosc_gpu_comm_create_all)
//   rccl_exchange_check procs [nranks]   one process per GPU (fork before HIP is touched; the unique id travels through pipes)
// ndev / nranks default to the GPUs of the node (1 on the test box: the same calls, a communicator of size 1).
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>
#include "../../include/blosc.h"
#include "../../include/blosc_gpu.h"
#include "../../include/blosc_gpu_rccl.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #x, __LINE__); exit(1); } } while (0)

static const int kChunks = 13;
static std::vector<std::vector<uint8_t>> g_plain, g_comp;      // the chunks, compressed once (host buffers) by the process that runs the ranks
static std::vector<int> g_cbytes;

static void make_chunks() {
  g_plain.resize(kChunks); g_comp.resize(kChunks); g_cbytes.resize(kChunks);
  std::vector<const void*> src(kChunks); std::vector<void*> dst(kChunks); std::vector<size_t> nb(kChunks), ds(kChunks);
  for (int c = 0; c < kChunks; c++) {
    const size_t n = (1u << 20) + 4096u * (size_t)c + (c == 5 ? 3 : 0);
    g_plain[c].resize(n);
    if (c == 7) { uint32_t s = 12345; for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; g_plain[c][i] = (uint8_t)(s >> 24); } }      // incompressible: a MEMCPYED chunk
    else { int32_t* v = (int32_t*)g_plain[c].data(); for (size_t i = 0; i < n / 4; i++) { const uint32_t k = (uint32_t)i + 1000u * (uint32_t)c; v[i] = (int32_t)(((k << 26) ^ (k << 18) ^ (k << 11) ^ (k << 3) ^ k) & ((1u << 19) - 1u)); } }
    g_comp[c].resize(n + 16);
    src[c] = g_plain[c].data(); dst[c] = g_comp[c].data(); nb[c] = n; ds[c] = n + 16;
  }
  CHECK(blosc_gpu_compress_batch_host(5, 1, 8, "lz4", 0, kChunks, src.data(), nb.data(), dst.data(), ds.data(), g_cbytes.data()) == 0);
  for (int c = 0; c < kChunks; c++) CHECK(g_cbytes[c] > 0 && (size_t)g_cbytes[c] <= g_plain[c].size() + 16);
}

// everything one rank does; returns 0 when all comparisons hold
static int run_rank(blosc_gpu_comm* comm) {
  int world, rank, dev;
  CHECK(blosc_gpu_comm_rank(comm, &world, &rank, &dev) == 0);
  CHECK(hipSetDevice(dev) == hipSuccess);
  size_t lo, hi;
  CHECK(blosc_gpu_partition(kChunks, world, rank, &lo, &hi) == 0);
  // this rank's compressed chunks in ITS device memory (what its own blosc_gpu_compress_batch would have left there)
  std::vector<void*> d_chunks(hi - lo, nullptr); std::vector<int> local(hi - lo);
  for (size_t c = lo; c < hi; c++) {
    local[c - lo] = g_cbytes[c];
    CHECK(hipMalloc(&d_chunks[c - lo], (size_t)g_cbytes[c]) == hipSuccess);
    CHECK(hipMemcpy(d_chunks[c - lo], g_comp[c].data(), (size_t)g_cbytes[c], hipMemcpyHostToDevice) == hipSuccess);
  }
  // (1) the table
  std::vector<int> table(kChunks, -7);
  CHECK(blosc_gpu_allgather_cbytes(comm, kChunks, local.data(), table.data()) == 0);
  for (int c = 0; c < kChunks; c++) CHECK(table[c] == g_cbytes[c]);
  size_t total = 0; std::vector<uint8_t> want;
  for (int c = 0; c < kChunks; c++) { want.insert(want.end(), g_comp[c].begin(), g_comp[c].begin() + g_cbytes[c]); total += (size_t)g_cbytes[c]; }
  // (2) onto rank 0, then onto every rank
  for (int root : {0, -1, world - 1}) {
    const bool recv = root < 0 || root == rank;
    void* d_cont = nullptr;
    if (recv) { CHECK(hipMalloc(&d_cont, total) == hipSuccess); CHECK(hipMemset(d_cont, 0xEE, total) == hipSuccess); }
    std::vector<size_t> off(kChunks);
    CHECK(blosc_gpu_gather_chunks(comm, kChunks, table.data(), (const void* const*)d_chunks.data(), d_cont, root, off.data()) == 0);
    size_t acc = 0; for (int c = 0; c < kChunks; c++) { CHECK(off[c] == acc); acc += (size_t)g_cbytes[c]; }
    if (recv) {
      std::vector<uint8_t> got(total);
      CHECK(hipMemcpy(got.data(), d_cont, total, hipMemcpyDeviceToHost) == hipSuccess);
      CHECK(memcmp(got.data(), want.data(), total) == 0);
      // the container decodes chunk by chunk (host-buffer batch call on the rank that holds it; rank 0 only: one check is enough)
      if (rank == 0 && root == 0) {
        std::vector<std::vector<uint8_t>> back(kChunks); std::vector<const void*> s(kChunks); std::vector<void*> d(kChunks); std::vector<size_t> ss(kChunks), dd(kChunks); std::vector<int> res(kChunks);
        for (int c = 0; c < kChunks; c++) { back[c].resize(g_plain[c].size()); s[c] = got.data() + off[c]; ss[c] = (size_t)g_cbytes[c]; d[c] = back[c].data(); dd[c] = back[c].size(); }
        CHECK(blosc_gpu_decompress_batch_host(kChunks, s.data(), ss.data(), d.data(), dd.data(), res.data()) == 0);
        for (int c = 0; c < kChunks; c++) CHECK(res[c] == (int)g_plain[c].size() && memcmp(back[c].data(), g_plain[c].data(), back[c].size()) == 0);
      }
    }
    // the inverse: from `r0` (the rank that certainly holds it) back to the ranges
    const int r0 = root < 0 ? 0 : root;
    size_t mine = 0; for (size_t c = lo; c < hi; c++) mine += (size_t)g_cbytes[c];
    void* d_mine = nullptr; CHECK(hipMalloc(&d_mine, mine ? mine : 1) == hipSuccess);
    std::vector<size_t> loff(hi - lo);
    CHECK(blosc_gpu_scatter_chunks(comm, kChunks, table.data(), d_cont, r0, d_mine, loff.data()) == 0);
    std::vector<uint8_t> got(mine);
    CHECK(hipMemcpy(got.data(), d_mine, mine, hipMemcpyDeviceToHost) == hipSuccess);
    for (size_t c = lo; c < hi; c++) CHECK(memcmp(got.data() + loff[c - lo], g_comp[c].data(), (size_t)g_cbytes[c]) == 0);
    CHECK(hipFree(d_mine) == hipSuccess);
    if (d_cont) CHECK(hipFree(d_cont) == hipSuccess);
  }
  for (void* p : d_chunks) CHECK(hipFree(p) == hipSuccess);
  // ---- the two one-call forms: compress + table + consolidation on rank 0, then scatter + decompress back into every rank's own memory ----
  {
    std::vector<const void*> src(kChunks, nullptr); std::vector<void*> dst(kChunks, nullptr), back(kChunks, nullptr); std::vector<size_t> nb(kChunks), ds(kChunks);
    size_t worst = 0, mine_worst = 0;
    for (int ch = 0; ch < kChunks; ch++) { nb[ch] = g_plain[ch].size(); ds[ch] = nb[ch] + 16; worst += ds[ch]; }
    for (size_t ch = lo; ch < hi; ch++) {      // only the own range lives on this device; the other entries stay NULL and must not be touched
      void* p; CHECK(hipMalloc(&p, nb[ch]) == hipSuccess); CHECK(hipMemcpy(p, g_plain[ch].data(), nb[ch], hipMemcpyHostToDevice) == hipSuccess); src[ch] = p;
      CHECK(hipMalloc(&dst[ch], ds[ch]) == hipSuccess); CHECK(hipMalloc(&back[ch], nb[ch]) == hipSuccess); CHECK(hipMemset(back[ch], 0xEE, nb[ch]) == hipSuccess);
      mine_worst += ds[ch];
    }
    void* d_cont = nullptr; if (rank == 0) CHECK(hipMalloc(&d_cont, worst) == hipSuccess);
    std::vector<int> tab(kChunks, -7); std::vector<size_t> off(kChunks); size_t used = 0;
    CHECK(blosc_gpu_compress_sharded(comm, 5, 1, 8, "lz4", 0, kChunks, src.data(), nb.data(), dst.data(), ds.data(), tab.data(), d_cont, worst, 0, off.data(), &used) == 0);
    for (int ch = 0; ch < kChunks; ch++) CHECK(tab[ch] == g_cbytes[ch]);      // (the LZ4 encoder writes the same bytes call after call: include/blosc.h)
    CHECK(used == total);
    if (rank == 0) { std::vector<uint8_t> got(total); CHECK(hipMemcpy(got.data(), d_cont, total, hipMemcpyDeviceToHost) == hipSuccess); CHECK(memcmp(got.data(), want.data(), total) == 0); }
    void* d_packed = nullptr; CHECK(hipMalloc(&d_packed, mine_worst ? mine_worst : 1) == hipSuccess);
    std::vector<int> res(kChunks, -7);
    CHECK(blosc_gpu_decompress_sharded(comm, kChunks, tab.data(), d_cont, 0, d_packed, mine_worst, back.data(), nb.data(), res.data()) == 0);
    for (int ch = 0; ch < kChunks; ch++) {
      if ((size_t)ch < lo || (size_t)ch >= hi) { CHECK(res[ch] == -7); continue; }
      CHECK(res[ch] == (int)nb[ch]);
      std::vector<uint8_t> got(nb[ch]); CHECK(hipMemcpy(got.data(), back[ch], nb[ch], hipMemcpyDeviceToHost) == hipSuccess);
      CHECK(memcmp(got.data(), g_plain[ch].data(), nb[ch]) == 0);
    }
    for (size_t ch = lo; ch < hi; ch++) { CHECK(hipFree((void*)src[ch]) == hipSuccess); CHECK(hipFree(dst[ch]) == hipSuccess); CHECK(hipFree(back[ch]) == hipSuccess); }
    CHECK(hipFree(d_packed) == hipSuccess); if (d_cont) CHECK(hipFree(d_cont) == hipSuccess);
  }
  return 0;
}

static void* rank_thread(void* p) { return (void*)(intptr_t)run_rank((blosc_gpu_comm*)p); }

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "threads";
  int want = argc > 2 ? atoi(argv[2]) : 0;
  if (!strcmp(mode, "threads")) {
    int ndev = blosc_gpu_device_count();
    CHECK(ndev >= 1);
    if (want > 0 && want < ndev) ndev = want;
    make_chunks();
    std::vector<blosc_gpu_comm*> comms((size_t)ndev, nullptr);
    CHECK(blosc_gpu_comm_create_all(comms.data(), ndev, nullptr) == 0);
    std::vector<pthread_t> th((size_t)ndev);
    for (int r = 0; r < ndev; r++) CHECK(pthread_create(&th[(size_t)r], nullptr, rank_thread, comms[(size_t)r]) == 0);
    for (int r = 0; r < ndev; r++) { void* rc; pthread_join(th[(size_t)r], &rc); CHECK(rc == nullptr); }
    for (auto* c : comms) blosc_gpu_comm_destroy(c);
    int dup[2] = {0, 0};
    CHECK(blosc_gpu_comm_create_all(comms.data(), 2, dup) == -1 || ndev < 1);      // two ranks on one device: refused
    printf("rccl exchange ok: threads, world %d\n", ndev);
    return 0;
  }
  // one process per rank.  HIP must not be touched before fork(): the parent only relays the id.
  int n = want > 0 ? want : 0;
  if (n == 0) {      // ask a child how many GPUs there are
    int pfd[2]; CHECK(pipe(pfd) == 0);
    pid_t pid = fork(); CHECK(pid >= 0);
    if (pid == 0) { int c = 0; if (hipGetDeviceCount(&c) != hipSuccess) c = 0; CHECK(write(pfd[1], &c, sizeof c) == sizeof c); _exit(0); }
    CHECK(read(pfd[0], &n, sizeof n) == sizeof n); waitpid(pid, nullptr, 0);
    CHECK(n >= 1);
  }
  std::vector<int> to_child((size_t)n), from0(2);
  int up[2]; CHECK(pipe(up) == 0);
  std::vector<pid_t> pids((size_t)n);
  std::vector<int> down_r((size_t)n), down_w((size_t)n);
  for (int r = 0; r < n; r++) { int p[2]; CHECK(pipe(p) == 0); down_r[(size_t)r] = p[0]; down_w[(size_t)r] = p[1]; }
  for (int r = 0; r < n; r++) {
    pids[(size_t)r] = fork(); CHECK(pids[(size_t)r] >= 0);
    if (pids[(size_t)r] == 0) {
      unsigned char id[BLOSC_GPU_COMM_ID_BYTES];
      if (r == 0) { CHECK(blosc_gpu_comm_unique_id(id) == 0); CHECK(write(up[1], id, sizeof id) == (ssize_t)sizeof id); }
      CHECK(read(down_r[(size_t)r], id, sizeof id) == (ssize_t)sizeof id);
      CHECK(blosc_gpu_set_device(r) == 0);
      make_chunks();
      blosc_gpu_comm* comm = nullptr;
      CHECK(blosc_gpu_comm_create(&comm, n, r, id, r) == 0);
      const int rc = run_rank(comm);
      blosc_gpu_comm_destroy(comm);
      _exit(rc);
    }
  }
  unsigned char id[BLOSC_GPU_COMM_ID_BYTES];
  CHECK(read(up[0], id, sizeof id) == (ssize_t)sizeof id);
  for (int r = 0; r < n; r++) CHECK(write(down_w[(size_t)r], id, sizeof id) == (ssize_t)sizeof id);
  int bad = 0;
  for (int r = 0; r < n; r++) { int st = 0; waitpid(pids[(size_t)r], &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad++; }
  CHECK(bad == 0);
  printf("rccl exchange ok: procs, world %d\n", n);
  return 0;
}
