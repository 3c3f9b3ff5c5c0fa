// rccl_exchange.hip — include/blosc_gpu_rccl.h: the cbytes table and the payload consolidation of a sharded many-chunk buffer, in C on
// RCCL (c-blosc_amd/multigpu.py does the same through torch.distributed for bench.py).  Built into
// libblosc_amd_rccl.so, NOT into the drop-in library.  Host code only: RCCL's own kernels move the bytes.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../include/blosc_gpu_rccl.h"
#include "../../include/blosc_gpu.h"

static_assert(BLOSC_GPU_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id the caller carries between ranks is RCCL's");

struct blosc_gpu_comm {
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0, device = 0;
  hipStream_t stream = nullptr;
  uint8_t* scratch = nullptr;      // device: the padded cbytes exchange, then this rank's chunks back to back
  size_t scratch_bytes = 0;
  int32_t* flag = nullptr;         // device, two words, made with the communicator: agree() never allocates
};

#define X_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "blosc_amd (rccl exchange): %s: %s\n", #call, hipGetErrorString(e_)); return -2; } } while (0)
#define X_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { fprintf(stderr, "blosc_amd (rccl exchange): %s: %s\n", #call, ncclGetErrorString(r_)); return -2; } } while (0)

// chunk range of `rank`: the rule of blosc_gpu_partition (chunk c -> rank floor(c * world / nchunks), contiguous ranges)
static void range_of(size_t nchunks, int world, int rank, size_t* lo, size_t* hi) {
  const size_t w = (size_t)world, r = (size_t)rank;
  size_t l = (r * nchunks + w - 1) / w, h = ((r + 1) * nchunks + w - 1) / w;
  if (h > nchunks) h = nchunks;
  if (l > h) l = h;
  *lo = l; *hi = h;
}
static size_t bytes_of(const int* table, size_t lo, size_t hi) {
  size_t n = 0;
  for (size_t c = lo; c < hi; c++) if (table[c] > 0) n += (size_t)table[c];
  return n;
}
static int need_scratch(blosc_gpu_comm* c, size_t bytes) {
  if (bytes <= c->scratch_bytes) return 0;
  if (c->scratch) X_HIP(hipFree(c->scratch));
  c->scratch = nullptr; c->scratch_bytes = 0;
  bytes = (bytes + 4095) & ~(size_t)4095;
  X_HIP(hipMalloc((void**)&c->scratch, bytes));
  c->scratch_bytes = bytes;
  return 0;
}
static int finish_create(blosc_gpu_comm* c) {
  X_HIP(hipSetDevice(c->device));
  X_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  X_HIP(hipMalloc((void**)&c->flag, 2 * sizeof(int32_t)));
  return 0;
}
// One verdict for all ranks: the minimum of everybody's status (0 = fine, negative = this rank cannot go on).  Every exchange below first
// agrees, THEN enters its group of sends and receives - a rank that returned on a condition only it can see (no container on the receiver, a
// failed batch call, too small a buffer) would leave its peers inside a group nobody completes.  A rank whose own status was 0 gets -3 when a
// peer's was not.  What cannot be agreed on is a call without a communicator, and arguments that must be equal on all ranks by contract
// (nchunks, root, the table): see include/blosc_gpu_rccl.h.
static int agree(blosc_gpu_comm* c, int status) {
  if (c->world == 1) return status;
  int32_t h = status, worst = 0;
  X_HIP(hipMemcpyAsync(c->flag, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
  X_NCCL(ncclAllReduce(c->flag, c->flag + 1, 1, ncclInt32, ncclMin, c->comm, c->stream));
  X_HIP(hipMemcpyAsync(&worst, c->flag + 1, sizeof worst, hipMemcpyDeviceToHost, c->stream));
  X_HIP(hipStreamSynchronize(c->stream));
  return status ? status : (worst ? -3 : 0);
}
// inside ncclGroupStart / ncclGroupEnd: remember the first failure, keep going, ALWAYS close the group (an early return would leave the thread's
// group open and queue the next RCCL call of this thread behind it)
#define G_HIP(call) do { if (!gerr) { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "blosc_amd (rccl exchange): %s: %s\n", #call, hipGetErrorString(e_)); gerr = -2; } } } while (0)
#define G_NCCL(call) do { if (!gerr) { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { fprintf(stderr, "blosc_amd (rccl exchange): %s: %s\n", #call, ncclGetErrorString(r_)); gerr = -2; } } } while (0)

extern "C" {

void blosc_gpu_comm_destroy(blosc_gpu_comm* c);

int blosc_gpu_comm_unique_id(void* id) {
  if (!id) return -1;
  ncclUniqueId u;
  X_NCCL(ncclGetUniqueId(&u));
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

int blosc_gpu_comm_create(blosc_gpu_comm** out, int world, int rank, const void* id, int device) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world || device < 0) return -1;
  int ndev = 0;
  X_HIP(hipGetDeviceCount(&ndev));
  if (device >= ndev) return -1;
  blosc_gpu_comm* c = new blosc_gpu_comm;
  c->world = world; c->rank = rank; c->device = device;
  ncclUniqueId u;
  memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  if (hipSetDevice(device) != hipSuccess || ncclCommInitRank(&c->comm, world, u, rank) != ncclSuccess || finish_create(c) != 0) {
    fprintf(stderr, "blosc_amd (rccl exchange): communicator rank %d of %d on device %d could not be created\n", rank, world, device);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->flag) (void)hipFree(c->flag);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
    return -2;
  }
  *out = c;
  return 0;
}

int blosc_gpu_comm_create_all(blosc_gpu_comm** comms, int ndev, const int* devices) {
  if (!comms || ndev < 1) return -1;
  int have = 0;
  X_HIP(hipGetDeviceCount(&have));
  std::vector<int> devs((size_t)ndev);
  for (int r = 0; r < ndev; r++) {
    devs[(size_t)r] = devices ? devices[r] : r;
    if (devs[(size_t)r] < 0 || devs[(size_t)r] >= have) return -1;
    for (int q = 0; q < r; q++) if (devs[(size_t)q] == devs[(size_t)r]) return -1;      // RCCL wants distinct devices inside one process
  }
  std::vector<ncclComm_t> raw((size_t)ndev, nullptr);
  X_NCCL(ncclCommInitAll(raw.data(), ndev, devs.data()));
  int rc = 0;
  for (int r = 0; r < ndev; r++) {
    blosc_gpu_comm* c = new blosc_gpu_comm;
    c->comm = raw[(size_t)r]; c->world = ndev; c->rank = r; c->device = devs[(size_t)r];
    comms[r] = c;
    if (!rc && finish_create(c) != 0) rc = -2;
  }
  if (rc) {      // nothing half-made is handed out: every communicator of the set goes, the caller's slots are cleared
    for (int r = 0; r < ndev; r++) { blosc_gpu_comm_destroy(comms[r]); comms[r] = nullptr; }
  }
  return rc;
}

void blosc_gpu_comm_destroy(blosc_gpu_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->flag) (void)hipFree(c->flag);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
}

int blosc_gpu_comm_rank(const blosc_gpu_comm* c, int* world, int* rank, int* device) {
  if (!c) return -1;
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  if (device) *device = c->device;
  return 0;
}

// (1) ranks own one chunk more or less: every rank contributes `per` = ceil(nchunks / world) ints, the tail padded
int blosc_gpu_allgather_cbytes(blosc_gpu_comm* c, size_t nchunks, const int* local_cbytes, int* table) {
  if (!c || !table) return -1;
  if (nchunks == 0) return 0;
  X_HIP(hipSetDevice(c->device));
  const size_t per = (nchunks + (size_t)c->world - 1) / (size_t)c->world;
  size_t lo, hi;
  range_of(nchunks, c->world, c->rank, &lo, &hi);
  if (hi - lo > per || (hi > lo && !local_cbytes)) return -1;
  if (need_scratch(c, (per + per * (size_t)c->world) * sizeof(int32_t))) return -2;
  int32_t* d_send = (int32_t*)c->scratch;
  int32_t* d_recv = d_send + per;
  std::vector<int32_t> h(per, -1);
  for (size_t i = 0; i < hi - lo; i++) h[i] = local_cbytes[i];
  X_HIP(hipMemcpyAsync(d_send, h.data(), per * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  X_NCCL(ncclAllGather(d_send, d_recv, per, ncclInt32, c->comm, c->stream));
  std::vector<int32_t> all(per * (size_t)c->world);
  X_HIP(hipMemcpyAsync(all.data(), d_recv, all.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  X_HIP(hipStreamSynchronize(c->stream));
  for (int r = 0; r < c->world; r++) {
    size_t l, u;
    range_of(nchunks, c->world, r, &l, &u);
    for (size_t i = 0; i < u - l; i++) table[l + i] = all[(size_t)r * per + i];
  }
  return 0;
}

// this rank's chunks back to back in c->scratch; *n = their bytes
static int pack_own(blosc_gpu_comm* c, size_t nchunks, const int* table, const void* const* local_chunks, size_t* n) {
  size_t lo, hi;
  range_of(nchunks, c->world, c->rank, &lo, &hi);
  const size_t mine = bytes_of(table, lo, hi);
  if (mine && !local_chunks) return -1;
  if (need_scratch(c, mine ? mine : 1)) return -2;
  size_t acc = 0;
  for (size_t ch = lo; ch < hi; ch++) {
    if (table[ch] <= 0) continue;
    if (!local_chunks[ch - lo]) return -1;
    X_HIP(hipMemcpyAsync(c->scratch + acc, local_chunks[ch - lo], (size_t)table[ch], hipMemcpyDeviceToDevice, c->stream));
    acc += (size_t)table[ch];
  }
  *n = mine;
  return 0;
}

static int gather_impl(blosc_gpu_comm* c, size_t nchunks, const int* table, const void* const* local_chunks, void* container, int root, size_t* offsets, int status) {
  if (offsets && table) { size_t acc = 0; for (size_t ch = 0; ch < nchunks; ch++) { offsets[ch] = acc; if (table[ch] > 0) acc += (size_t)table[ch]; } }
  if (nchunks == 0) return status;            // (nchunks is the same on every rank: nobody enters anything)
  X_HIP(hipSetDevice(c->device));
  const bool receiver = root < 0 || root == c->rank;
  size_t mine = 0;
  if (!status && receiver && !container && bytes_of(table, 0, nchunks)) status = -1;
  if (!status) status = pack_own(c, nchunks, table, local_chunks, &mine);
  status = agree(c, status);
  if (status) return status;
  // one group: what every receiver takes from every other rank, what this rank sends to every receiver
  int gerr = 0;
  G_NCCL(ncclGroupStart());
  if (gerr) return gerr;
  size_t base = 0;
  for (int r = 0; r < c->world; r++) {
    size_t l, u;
    range_of(nchunks, c->world, r, &l, &u);
    const size_t n = bytes_of(table, l, u);
    if (receiver && n) {
      if (r == c->rank) G_HIP(hipMemcpyAsync((uint8_t*)container + base, c->scratch, n, hipMemcpyDeviceToDevice, c->stream));
      else G_NCCL(ncclRecv((uint8_t*)container + base, n, ncclUint8, r, c->comm, c->stream));
    }
    base += n;
  }
  if (mine)
    for (int r = 0; r < c->world; r++)
      if (r != c->rank && (root < 0 || root == r)) G_NCCL(ncclSend(c->scratch, mine, ncclUint8, r, c->comm, c->stream));
  { const ncclResult_t r_ = ncclGroupEnd(); if (r_ != ncclSuccess && !gerr) { fprintf(stderr, "blosc_amd (rccl exchange): ncclGroupEnd: %s\n", ncclGetErrorString(r_)); gerr = -2; } }
  if (gerr) return gerr;
  X_HIP(hipStreamSynchronize(c->stream));
  return 0;
}
int blosc_gpu_gather_chunks(blosc_gpu_comm* c, size_t nchunks, const int* table, const void* const* local_chunks, void* container, int root, size_t* offsets) {
  if (!c) return -1;
  // (arguments that are equal on all ranks by contract decide for all ranks alike; what only this rank can see goes through agree())
  if (root < -1 || root >= c->world) return -1;
  return gather_impl(c, nchunks, table, local_chunks, container, root, offsets, (nchunks && !table) ? -1 : 0);
}

static int scatter_impl(blosc_gpu_comm* c, size_t nchunks, const int* table, const void* container, int root, void* local_packed, size_t* local_offsets, int status) {
  size_t lo, hi;
  range_of(nchunks, c->world, c->rank, &lo, &hi);
  if (local_offsets && table) { size_t acc = 0; for (size_t ch = lo; ch < hi; ch++) { local_offsets[ch - lo] = acc; if (table[ch] > 0) acc += (size_t)table[ch]; } }
  if (nchunks == 0) return status;
  X_HIP(hipSetDevice(c->device));
  const size_t mine = status ? 0 : bytes_of(table, lo, hi);
  if (!status && mine && !local_packed) status = -1;
  if (!status && c->rank == root && !container && bytes_of(table, 0, nchunks)) status = -1;
  status = agree(c, status);
  if (status) return status;
  int gerr = 0;
  G_NCCL(ncclGroupStart());
  if (gerr) return gerr;
  if (c->rank == root) {
    size_t base = 0;
    for (int r = 0; r < c->world; r++) {
      size_t l, u;
      range_of(nchunks, c->world, r, &l, &u);
      const size_t n = bytes_of(table, l, u);
      if (n) {
        if (r == root) G_HIP(hipMemcpyAsync(local_packed, (const uint8_t*)container + base, n, hipMemcpyDeviceToDevice, c->stream));
        else G_NCCL(ncclSend((const uint8_t*)container + base, n, ncclUint8, r, c->comm, c->stream));
      }
      base += n;
    }
  } else if (mine) {
    G_NCCL(ncclRecv(local_packed, mine, ncclUint8, root, c->comm, c->stream));
  }
  { const ncclResult_t r_ = ncclGroupEnd(); if (r_ != ncclSuccess && !gerr) { fprintf(stderr, "blosc_amd (rccl exchange): ncclGroupEnd: %s\n", ncclGetErrorString(r_)); gerr = -2; } }
  if (gerr) return gerr;
  X_HIP(hipStreamSynchronize(c->stream));
  return 0;
}
int blosc_gpu_scatter_chunks(blosc_gpu_comm* c, size_t nchunks, const int* table, const void* container, int root, void* local_packed, size_t* local_offsets) {
  if (!c) return -1;
  if (root < 0 || root >= c->world) return -1;
  return scatter_impl(c, nchunks, table, container, root, local_packed, local_offsets, (nchunks && !table) ? -1 : 0);
}

// ---- the sharded calls: partition + the drop-in's batched call on the own range + the exchanges above ----
int blosc_gpu_compress_sharded(blosc_gpu_comm* c, int clevel, int doshuffle, size_t typesize, const char* compressor, size_t blocksize,
                               size_t nchunks, const void* const* src, const size_t* nbytes, void* const* dest, const size_t* destsize,
                               int* table, void* container, size_t container_capacity, int root, size_t* offsets, size_t* container_bytes) {
  if (!c) return -1;
  if (root < -1 || root >= c->world || nchunks > 0x7fffffffu) return -1;      // (equal on all ranks by contract)
  int status = (!table || (nchunks && (!src || !nbytes || !dest || !destsize))) ? -1 : 0;
  size_t lo, hi;
  range_of(nchunks, c->world, c->rank, &lo, &hi);
  X_HIP(hipSetDevice(c->device));
  std::vector<int> local(hi - lo + 1, 0);
  if (!status && hi > lo) {
    // blosc_gpu_set_device is process-wide: in the thread-per-GPU layout the call must not move other threads, so the device is bound the way
    // the _multi calls bind it - through a one-device _multi call on this rank's device
    const int dev = c->device;
    if (blosc_gpu_compress_batch_multi(1, &dev, clevel, doshuffle, typesize, compressor, blocksize, (int)(hi - lo), src + lo, nbytes + lo, dest + lo, destsize + lo, local.data()) != 0) status = -2;
  }
  // a rank whose batch call failed (or whose arguments are no good) tells the others BEFORE anybody enters the table's all-gather
  status = agree(c, status);
  if (status) return status;
  int rc = blosc_gpu_allgather_cbytes(c, nchunks, local.data(), table);
  if (rc) return rc;
  const size_t total = bytes_of(table, 0, nchunks);
  if (container_bytes) *container_bytes = total;
  // the receivers' capacity is theirs alone to know: the verdict is agreed on inside the gather, before its group
  return gather_impl(c, nchunks, table, (const void* const*)(dest + lo), container, root, offsets, ((root < 0 || root == c->rank) && total > container_capacity) ? -1 : 0);
}

int blosc_gpu_decompress_sharded(blosc_gpu_comm* c, size_t nchunks, const int* table, const void* container, int root,
                                 void* packed, size_t packed_capacity, void* const* dest, const size_t* destsize, int* nbytes_out) {
  if (!c) return -1;
  if (root < 0 || root >= c->world || nchunks > 0x7fffffffu) return -1;       // (equal on all ranks by contract)
  int status = (nchunks && (!table || !dest || !destsize || !nbytes_out)) ? -1 : 0;
  size_t lo, hi;
  range_of(nchunks, c->world, c->rank, &lo, &hi);
  if (!status && bytes_of(table, lo, hi) > packed_capacity) status = -1;
  std::vector<size_t> loff(hi - lo + 1, 0);
  int rc = scatter_impl(c, nchunks, table, container, root, packed, loff.data(), status);      // (agrees on `status` before its group)
  if (rc || hi == lo) return rc;
  std::vector<const void*> s(hi - lo); std::vector<size_t> ss(hi - lo);
  for (size_t ch = lo; ch < hi; ch++) { s[ch - lo] = (const uint8_t*)packed + loff[ch - lo]; ss[ch - lo] = table[ch] > 0 ? (size_t)table[ch] : 0; }
  const int dev = c->device;
  return blosc_gpu_decompress_batch_multi(1, &dev, (int)(hi - lo), s.data(), ss.data(), dest + lo, destsize + lo, nbytes_out + lo) == 0 ? 0 : -2;
}

}  // extern "C"
