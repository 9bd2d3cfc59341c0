// Native multi-GPU plumbing of the data-parallel path (BASELINE north_star: "RCCL broadcast of weights and gather
// of depth maps over xGMI" reachable from the host language through the FFI layer): one process per GPU, RCCL over xGMI,
// NO collective inside DepthPro::infer -- images of a batch never interact (encoder.rs:216-225,249-255).
//   * one-time broadcast of the fp32 parameter arena from the root (1-GiB buckets: a ring broadcast is bound by one xGMI
//     link, ~153 GB/s; the bucket keeps the counts inside 32 bits), then every rank packs its own MFMA operand copies;
//   * per batch: the root scatters [B,3,H,W] shards and gathers the depth maps as ONE group of point-to-point
//     ncclSend / ncclRecv on the engine's stream (xGMI is point to point: the root's 7 links carry the 7 shards in parallel).
// The reference has no collectives at all; these entry points are what a reference-side host would call
// next to md_depth_pro_infer.
#include <rccl/rccl.h>

#include <cstring>

#include "md_engine.h"

using namespace md;

struct md_comm_s {
  md_device_t dev = nullptr;
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
};

#define MD_NCCL(expr)                                                                                   \
  do {                                                                                                  \
    ncclResult_t _r = (expr);                                                                           \
    if (_r != ncclSuccess) {                                                                            \
      ::md::set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, __LINE__);      \
      return MD_ERR_HIP;                                                                                \
    }                                                                                                   \
  } while (0)

// Inside an ncclGroupStart / ncclGroupEnd pair: a failing call must not leave the group open (every later RCCL call of this
// thread would be queued into it and never run) -- close it, then report the FIRST error.
#define MD_NCCL_G(expr)                                                                                 \
  do {                                                                                                  \
    ncclResult_t _r = (expr);                                                                           \
    if (_r != ncclSuccess) {                                                                            \
      (void)ncclGroupEnd();                                                                             \
      ::md::set_error("%s failed inside a group: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, __LINE__); \
      return MD_ERR_HIP;                                                                                \
    }                                                                                                   \
  } while (0)

extern "C" {

int md_comm_unique_id(uint8_t id[MD_COMM_ID_BYTES]) {
  if (!id) MD_FAIL(MD_ERR_INVALID_ARG, "id is null");
  static_assert(MD_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "md_comm id size must match ncclUniqueId");
  ncclUniqueId u;
  MD_NCCL(ncclGetUniqueId(&u));
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return MD_OK;
}

int md_comm_init_rank(md_device_t dev, const uint8_t id[MD_COMM_ID_BYTES], int world_size, int rank, md_comm_t* out) {
  if (!dev || !id || !out) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (world_size < 1 || rank < 0 || rank >= world_size) MD_FAIL(MD_ERR_INVALID_ARG, "rank %d of %d", rank, world_size);
  MD_HIP(hipSetDevice(dev->ordinal));
  ncclUniqueId u;
  memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  md_comm_s* c = new md_comm_s();
  c->dev = dev;
  c->world = world_size;
  c->rank = rank;
  ncclResult_t r = ncclCommInitRank(&c->comm, world_size, u, rank);
  if (r != ncclSuccess) {
    delete c;
    MD_FAIL(MD_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world_size, ncclGetErrorString(r));
  }
  *out = c;
  return MD_OK;
}

int md_comm_destroy(md_comm_t c) {
  if (!c) return MD_OK;
  if (c->dev) (void)hipSetDevice(c->dev->ordinal);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
  return MD_OK;
}

// ranks RCCL itself reports for the communicator (ncclCommCount) -- what `bench.py` prints as `ranks_seen` for N > 1
int md_comm_count(md_comm_t c, int* ranks_seen) {
  if (!c || !ranks_seen) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  int n = 0;
  MD_NCCL(ncclCommCount(c->comm, &n));
  *ranks_seen = n;
  return MD_OK;
}

int md_comm_rank(md_comm_t c, int* rank, int* world_size) {
  if (!c) MD_FAIL(MD_ERR_INVALID_ARG, "communicator is null");
  if (rank) *rank = c->rank;
  if (world_size) *world_size = c->world;
  return MD_OK;
}

int md_comm_broadcast_weights(md_comm_t c, md_model_t m, int root) {
  if (!c || !m) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (root < 0 || root >= c->world) MD_FAIL(MD_ERR_INVALID_ARG, "root %d of %d", root, c->world);
  if (m->parent) MD_FAIL(MD_ERR_INVALID_ARG, "a fork shares its root's weights: broadcast into the root model");
  if (m->dev != c->dev && m->dev->ordinal != c->dev->ordinal) MD_FAIL(MD_ERR_INVALID_ARG, "model and communicator live on different devices");
  MD_HIP(hipSetDevice(c->dev->ordinal));
  hipStream_t st = c->dev->stream;
  const size_t bucket = (size_t)1 << 30;
  for (size_t off = 0; off < m->w32_bytes; off += bucket) {
    const size_t n = std::min(bucket, m->w32_bytes - off);
    MD_NCCL(ncclBroadcast(m->w32_base + off, m->w32_base + off, n, ncclUint8, root, c->comm, st));
  }
  MD_HIP(hipStreamSynchronize(st));
  m->committed = false;
  return model_commit(m);
}

// root: `all` = [world][elems] (rank-major); every rank (the root included) receives its shard in `shard`
int md_comm_scatter_images(md_comm_t c, const float* all_dev, float* shard_dev, size_t elems_per_rank, int root, void* stream) {
  if (!c || !shard_dev) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (root < 0 || root >= c->world) MD_FAIL(MD_ERR_INVALID_ARG, "root %d of %d", root, c->world);
  if (c->rank == root && !all_dev) MD_FAIL(MD_ERR_INVALID_ARG, "the root rank needs the full batch");
  MD_HIP(hipSetDevice(c->dev->ordinal));
  hipStream_t st = stream ? (hipStream_t)stream : c->dev->stream;
  MD_NCCL(ncclGroupStart());
  if (c->rank == root) {
    for (int r = 0; r < c->world; ++r)
      if (r != root) MD_NCCL_G(ncclSend(all_dev + (size_t)r * elems_per_rank, elems_per_rank, ncclFloat, r, c->comm, st));
  } else {
    MD_NCCL_G(ncclRecv(shard_dev, elems_per_rank, ncclFloat, root, c->comm, st));
  }
  MD_NCCL(ncclGroupEnd());
  if (c->rank == root && shard_dev != all_dev + (size_t)root * elems_per_rank)
    MD_HIP(hipMemcpyAsync(shard_dev, all_dev + (size_t)root * elems_per_rank, elems_per_rank * 4, hipMemcpyDeviceToDevice, st));
  return MD_OK;
}

// every rank sends `shard`; root: `all` = [world][elems] (rank-major)
int md_comm_gather_depth(md_comm_t c, const float* shard_dev, float* all_dev, size_t elems_per_rank, int root, void* stream) {
  if (!c || !shard_dev) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (root < 0 || root >= c->world) MD_FAIL(MD_ERR_INVALID_ARG, "root %d of %d", root, c->world);
  if (c->rank == root && !all_dev) MD_FAIL(MD_ERR_INVALID_ARG, "the root rank needs the gather buffer");
  MD_HIP(hipSetDevice(c->dev->ordinal));
  hipStream_t st = stream ? (hipStream_t)stream : c->dev->stream;
  MD_NCCL(ncclGroupStart());
  if (c->rank == root) {
    for (int r = 0; r < c->world; ++r)
      if (r != root) MD_NCCL_G(ncclRecv(all_dev + (size_t)r * elems_per_rank, elems_per_rank, ncclFloat, r, c->comm, st));
  } else {
    MD_NCCL_G(ncclSend(shard_dev, elems_per_rank, ncclFloat, root, c->comm, st));
  }
  MD_NCCL(ncclGroupEnd());
  if (c->rank == root && shard_dev != all_dev + (size_t)root * elems_per_rank)
    MD_HIP(hipMemcpyAsync(all_dev + (size_t)root * elems_per_rank, shard_dev, elems_per_rank * 4, hipMemcpyDeviceToDevice, st));
  return MD_OK;
}

namespace {
struct TileExchange {
  md_comm_s* c;
  int root;
};
// Behind a rank's ViT window: every part but the root's sends its three row ranges to the root; the root posts the matching
// receives. Both sides walk the same table in the same order, so the k-th send of a peer meets the root's k-th receive for it.
int tile_exchange(void* ctx, int parts, const ShardSegment (*seg)[3], hipStream_t st) {
  TileExchange* x = (TileExchange*)ctx;
  md_comm_s* c = x->c;
  MD_NCCL(ncclGroupStart());
  if (c->rank == x->root) {
    for (int p = 0; p < parts; ++p)
      if (p != x->root)
        for (int k = 0; k < 3; ++k)
          if (seg[p][k].bytes) MD_NCCL_G(ncclRecv(seg[p][k].ptr, seg[p][k].bytes, ncclInt8, p, c->comm, st));
  } else {
    for (int k = 0; k < 3; ++k)
      if (seg[c->rank][k].bytes) MD_NCCL_G(ncclSend(seg[c->rank][k].ptr, seg[c->rank][k].bytes, ncclInt8, x->root, c->comm, st));
  }
  MD_NCCL(ncclGroupEnd());
  return MD_OK;
}
}  // namespace

namespace {
// Loopback transport of the tile-parallel mode: the "ranks" are models on ONE device; a non-root part records where its three row ranges
// live, the root copies them into its own workspace where the RCCL form would have received them.
struct LoopbackTable {
  std::vector<ShardSegment> sent;  // [parts][3]: the segments every non-root part "sent" (pointers into ITS workspace)
  int parts, root, me;
};
int loopback_exchange(void* ctx, int parts, const ShardSegment (*seg)[3], hipStream_t st) {
  LoopbackTable* t = (LoopbackTable*)ctx;
  if (parts != t->parts) MD_FAIL(MD_ERR_INVALID_ARG, "loopback exchange: %d parts, expected %d", parts, t->parts);
  if (t->me != t->root) {  // what tile_exchange's ncclSend calls would have moved
    for (int k = 0; k < 3; ++k) t->sent[(size_t)t->me * 3 + k] = seg[t->me][k];
    return MD_OK;
  }
  for (int p = 0; p < parts; ++p) {
    if (p == t->root) continue;
    for (int k = 0; k < 3; ++k) {
      const ShardSegment& from = t->sent[(size_t)p * 3 + k];
      // sender and receiver walk the same table: a disagreement about a segment's size is the bug this entry exists to catch
      if (from.bytes != seg[p][k].bytes)
        MD_FAIL(MD_ERR_INVALID_ARG, "loopback exchange: part %d segment %d is %zu bytes on the sender and %zu on the root", p, k, from.bytes, seg[p][k].bytes);
      if (from.bytes) MD_HIP(hipMemcpyAsync(seg[p][k].ptr, from.ptr, from.bytes, hipMemcpyDeviceToDevice, st));
    }
  }
  return MD_OK;
}
}  // namespace

int md_depth_pro_infer_tiles_loopback(const md_model_t* models, int parts, int root, const float* nchw, int B, int H, int W, int in_kind,
                                      float* depth, float* focallength_px, float* fovx_deg, float* fovy_rad, int out_kind, void* stream) {
  if (!models || !nchw) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (parts < 1 || parts > 64 || root < 0 || root >= parts) MD_FAIL(MD_ERR_INVALID_ARG, "%d parts, root %d", parts, root);
  for (int p = 0; p < parts; ++p) {
    if (!models[p] || models[p]->kind != 0) MD_FAIL(MD_ERR_INVALID_ARG, "models[%d] is not a Depth Pro model", p);
    if (models[p]->dev != models[0]->dev) MD_FAIL(MD_ERR_INVALID_ARG, "the loopback ranks live on one device");
    for (int q = 0; q < p; ++q)
      if (models[q] == models[p]) MD_FAIL(MD_ERR_INVALID_ARG, "models[%d] and models[%d] are the same context (a rank owns its workspace)", q, p);
  }
  MD_HIP(hipSetDevice(models[0]->dev->ordinal));
  hipStream_t st = stream ? (hipStream_t)stream : models[0]->dev->stream;  // ONE stream orders the parts and the copies
  LoopbackTable t;
  t.parts = parts; t.root = root;
  t.sent.assign((size_t)parts * 3, ShardSegment{nullptr, 0});
  const size_t elems = (size_t)B * 3 * H * W;
  auto run_part = [&](int p) -> int {
    float* x_dev = nullptr;
    MD_TRY(model_stage_input(models[p], nchw, elems, in_kind, st, &x_dev));  // the broadcast of the RCCL form
    t.me = p;
    ShardPlan sp;
    sp.parts = parts; sp.part = p; sp.root = root;
    sp.exchange = loopback_exchange; sp.ctx = &t;
    const bool is_root = p == root;
    return model_infer_sharded(models[p], x_dev, B, H, W, MD_MEM_DEVICE, is_root ? depth : nullptr, is_root ? focallength_px : nullptr,
                               is_root ? fovx_deg : nullptr, is_root ? fovy_rad : nullptr, out_kind, st, sp);
  };
  for (int p = 0; p < parts; ++p)
    if (p != root) MD_TRY(run_part(p));
  return run_part(root);
}

int md_comm_depth_pro_infer_tiles(md_comm_t c, md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth,
                                  float* focallength_px, float* fovx_deg, float* fovy_rad, int out_kind, int root, void* stream) {
  if (!c || !m) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (m->kind != 0) MD_FAIL(MD_ERR_INVALID_ARG, "not a Depth Pro model");
  if (root < 0 || root >= c->world) MD_FAIL(MD_ERR_INVALID_ARG, "root %d of %d", root, c->world);
  if (c->rank == root && !nchw) MD_FAIL(MD_ERR_INVALID_ARG, "the root rank needs the input");
  if (B <= 0 || H <= 0 || W <= 0) MD_FAIL(MD_ERR_SHAPE, "invalid input shape [%d,3,%d,%d]", B, H, W);
  if (m->dev != c->dev) MD_FAIL(MD_ERR_INVALID_ARG, "model and communicator live on different devices");
  // everything model_infer would refuse is refused HERE, before the first collective: these arguments are the same on every rank,
  // so every rank returns the same error instead of some ranks waiting in a broadcast the root never joins
  if (!model_root(m)->committed) MD_FAIL(MD_ERR_INVALID_ARG, "weights were modified; call md_model_commit_weights first");
  if (B > m->cfg.max_batch) MD_FAIL(MD_ERR_SHAPE, "batch %d exceeds max_batch %d", B, m->cfg.max_batch);
  if (!m->cfg.use_fov_head) MD_FAIL(MD_ERR_NO_FOV, "FOV head required for focal length");
  if (in_kind != MD_MEM_HOST && in_kind != MD_MEM_DEVICE) MD_FAIL(MD_ERR_INVALID_ARG, "unknown memory kind %d", in_kind);
  MD_HIP(hipSetDevice(c->dev->ordinal));
  hipStream_t st = stream ? (hipStream_t)stream : (m->own_stream ? m->own_stream : m->dev->stream);
  // 1. the root's image reaches every rank's staging buffer (28 MB per 1536^2 frame)
  const size_t elems = (size_t)B * 3 * H * W;
  float* x_dev = nullptr;
  MD_TRY(model_stage_input(m, c->rank == root ? nchw : nullptr, elems, in_kind, st, &x_dev));
  if (c->world > 1) MD_NCCL(ncclBroadcast(x_dev, x_dev, elems, ncclFloat, root, c->comm, st));
  // 2. - 4. this rank's window, the exchange, and on the root the rest of the network
  TileExchange x{c, root};
  ShardPlan sp;
  sp.parts = c->world;
  sp.part = c->rank;
  sp.root = root;
  sp.exchange = tile_exchange;
  sp.ctx = &x;
  return model_infer_sharded(m, x_dev, B, H, W, MD_MEM_DEVICE, depth, focallength_px, fovx_deg, fovy_rad, out_kind, st, sp);
}

}  // extern "C"
