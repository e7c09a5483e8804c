// sharded.cpp -- one process driving several contexts (one per GPU of a node): the multi-device entries of the C ABI.
//
// an MSM / commitment shards by INDEX RANGE -- every device runs the whole single-GPU pipeline on its
// contiguous slice and the per-device partial sums (one affine point each) are added.  With 72 B per device no
// collective library is needed when one host process owns all the contexts: a host thread per context issues the
// slice's kg_msm, the partials meet in host memory and are added with the host curve code (kg_points_sum_affine).
// (One process per GPU over torch.distributed / RCCL is the other deployment shape: kogarashi_amd/dist.py.)
//
// kg_sharded_key is the resident form of nova/src/pedersen.rs:6-13 PedersenCommitment { g }: slice i of the generators
// lives on device i in the MSM's internal form (kg_bases_register), the reference re-reads g on every commit.
#include "common.h"
#include <future>
#include <thread>
#include <vector>

using namespace kg;

namespace {

// dist.shard_range: contiguous slices that differ by at most one element
void shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi) {
  const size_t base = n / (size_t)world, extra = n % (size_t)world;
  const size_t r = (size_t)rank;
  *lo = r * base + (r < extra ? r : extra);
  *hi = *lo + base + (r < extra ? 1 : 0);
}

int words_of(int curve) { return curve == KG_G2 ? 16 : 8; }

// runs fn(i) for every context at once -- context 0 on the calling thread, the others on worker threads of context 0's pool; returns the
// first failing status
template <class Fn>
int for_each_ctx(kg_ctx* const* ctxs, int n_ctx, Fn fn) {
  std::vector<int> rc((size_t)n_ctx, KG_OK);
  std::vector<std::future<int>> fut;
  fut.reserve((size_t)n_ctx);
  int started = 1;
  try {
    for (int i = 1; i < n_ctx; ++i, ++started) fut.push_back(kg::pool(ctxs[0]).submit([&fn, i]() -> int { return fn(i); }));
  } catch (...) {                                        // a thread could not be started: its context and the later ones run here, in turn
  }
  rc[0] = fn(0);
  for (int i = started; i < n_ctx; ++i) rc[(size_t)i] = fn(i);
  for (size_t t = 0; t < fut.size(); ++t) rc[t + 1] = fut[t].get();
  for (int i = 0; i < n_ctx; ++i)
    if (rc[(size_t)i] != KG_OK) return rc[(size_t)i];
  return KG_OK;
}

// partial (x, y, z in {0, 1}) triples -> one affine sum
int combine(kg_ctx* ctx0, int curve, const std::vector<uint64_t>& xyz, int n_ctx, uint64_t* out_xy, uint8_t* out_inf) {
  const int E2 = words_of(curve), E = E2 / 2;
  std::vector<uint64_t> pts((size_t)n_ctx * E2);
  std::vector<uint8_t> inf((size_t)n_ctx);
  for (int i = 0; i < n_ctx; ++i) {
    const uint64_t* p = xyz.data() + (size_t)i * 3 * E;
    bool z0 = true;
    for (int k = 0; k < E; ++k) z0 = z0 && p[2 * E + k] == 0;
    inf[(size_t)i] = z0 ? 1 : 0;
    for (int k = 0; k < E2; ++k) pts[(size_t)i * E2 + k] = p[k];
  }
  return kg_points_sum_affine(ctx0, curve, pts.data(), inf.data(), (size_t)n_ctx, out_xy, out_inf);
}

}  // namespace

struct kg_sharded_key {
  std::vector<kg_ctx*> ctxs;
  int curve = 0;
  size_t n = 0;
  std::vector<size_t> lo, hi;
  std::vector<uint64_t*> d_bases;
  std::vector<uint8_t*> d_inf;
};

extern "C" {

int kg_shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi) {
  if (!lo || !hi || world < 1 || rank < 0 || rank >= world) return KG_ERR_BAD_ARG;
  shard_range(n, rank, world, lo, hi);
  return KG_OK;
}

int kg_commit_sharded(kg_ctx* const* ctxs, int n_ctx, int curve, const uint64_t* const* d_bases, const uint8_t* const* d_inf,
                      const uint64_t* const* d_scalars, const size_t* n_local, uint64_t* out_xy, uint8_t* out_inf) {
  return kg::kg_guarded((ctxs && n_ctx > 0 ? ctxs[0] : nullptr), [&]() -> int {
  if (!ctxs || n_ctx < 1 || n_ctx > 64 || curve < 0 || curve > KG_G2 || !d_bases || !d_scalars || !n_local || !out_xy || !out_inf) return KG_ERR_BAD_ARG;
  for (int i = 0; i < n_ctx; ++i)
    if (!ctxs[i] || (n_local[i] && (!d_bases[i] || !d_scalars[i]))) return KG_ERR_BAD_ARG;
  const int E = words_of(curve) / 2;
  std::vector<uint64_t> xyz((size_t)n_ctx * 3 * E);
  const int rc = for_each_ctx(ctxs, n_ctx, [&](int i) {
    return kg_msm(ctxs[i], curve, d_bases[i], d_inf ? d_inf[i] : nullptr, d_scalars[i], n_local[i], xyz.data() + (size_t)i * 3 * E);
  });
  if (rc != KG_OK) return rc;
  return combine(ctxs[0], curve, xyz, n_ctx, out_xy, out_inf);
  });
}

int kg_msm_sharded(kg_ctx* const* ctxs, int n_ctx, int curve, const uint64_t* const* d_bases, const uint8_t* const* d_inf,
                   const uint64_t* const* d_scalars, const size_t* n_local, uint64_t* out_xyz) {
  return kg::kg_guarded((ctxs && n_ctx > 0 ? ctxs[0] : nullptr), [&]() -> int {
  if (!out_xyz || curve < 0 || curve > KG_G2) return KG_ERR_BAD_ARG;
  uint64_t xy[16];
  uint8_t inf = 0;
  KG_TRY(kg_commit_sharded(ctxs, n_ctx, curve, d_bases, d_inf, d_scalars, n_local, xy, &inf));
  if (inf) { msm_identity(curve, out_xyz); return KG_OK; }
  const int E = words_of(curve) / 2;
  uint64_t one[24];
  msm_identity(curve, one);                        // (0, 1, 0): its y is the field's one in the ABI form
  for (int k = 0; k < 2 * E; ++k) out_xyz[k] = xy[k];
  for (int k = 0; k < E; ++k) out_xyz[2 * E + k] = one[E + k];
  return KG_OK;
  });
}

int kg_sharded_key_create(kg_ctx* const* ctxs, int n_ctx, int curve, const uint64_t* h_bases, const uint8_t* h_inf, size_t n,
                          kg_sharded_key** out) {
  return kg::kg_guarded((ctxs && n_ctx > 0 ? ctxs[0] : nullptr), [&]() -> int {
  if (!out) return KG_ERR_BAD_ARG;
  *out = nullptr;
  if (!ctxs || n_ctx < 1 || n_ctx > 64 || curve < 0 || curve > KG_G2 || (n && !h_bases)) return KG_ERR_BAD_ARG;
  for (int i = 0; i < n_ctx; ++i)
    if (!ctxs[i]) return KG_ERR_BAD_ARG;
  kg_sharded_key* K = new kg_sharded_key();
  K->ctxs.assign(ctxs, ctxs + n_ctx);
  K->curve = curve; K->n = n;
  K->lo.resize((size_t)n_ctx); K->hi.resize((size_t)n_ctx);
  K->d_bases.assign((size_t)n_ctx, nullptr); K->d_inf.assign((size_t)n_ctx, nullptr);
  const size_t wb = (size_t)words_of(curve) * 8;
  const int rc = for_each_ctx(ctxs, n_ctx, [&](int i) {
    size_t lo, hi;
    shard_range(n, i, n_ctx, &lo, &hi);
    K->lo[(size_t)i] = lo; K->hi[(size_t)i] = hi;
    const size_t cnt = hi - lo;
    if (!cnt) return (int)KG_OK;
    kg_ctx* c = K->ctxs[(size_t)i];
    KG_TRY(kg_malloc(c, cnt * wb, (void**)&K->d_bases[(size_t)i]));
    KG_TRY(kg_memcpy_h2d(c, K->d_bases[(size_t)i], (const char*)h_bases + lo * wb, cnt * wb));
    if (h_inf) {
      KG_TRY(kg_malloc(c, cnt, (void**)&K->d_inf[(size_t)i]));
      KG_TRY(kg_memcpy_h2d(c, K->d_inf[(size_t)i], h_inf + lo, cnt));
    }
    return kg_bases_register(c, curve, K->d_bases[(size_t)i], K->d_inf[(size_t)i], cnt);
  });
  if (rc != KG_OK) { kg_sharded_key_destroy(K); return rc; }
  *out = K;
  return KG_OK;
  });
}

void kg_sharded_key_destroy(kg_sharded_key* K) {
  if (!K) return;
  for (size_t i = 0; i < K->ctxs.size(); ++i) {
    kg_ctx* c = K->ctxs[i];
    if (K->d_bases[i]) kg_free(c, K->d_bases[i]);        // kg_free drops the registration with the array
    if (K->d_inf[i]) kg_free(c, K->d_inf[i]);
  }
  delete K;
}

size_t kg_sharded_key_len(const kg_sharded_key* K) { return K ? K->n : 0; }

int kg_sharded_key_commit(kg_sharded_key* K, const uint64_t* h_scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf) {
  return kg::kg_guarded((kg_ctx*)nullptr, [&]() -> int {
  if (!K || !out_xy || !out_inf || (n && !h_scalars)) return KG_ERR_BAD_ARG;
  if (n > K->n) n = K->n;                              // zip semantics of commit (pedersen.rs:16-17)
  const int n_ctx = (int)K->ctxs.size();
  const int E = words_of(K->curve) / 2;
  std::vector<uint64_t> xyz((size_t)n_ctx * 3 * E);
  const int rc = for_each_ctx(K->ctxs.data(), n_ctx, [&](int i) {
    const size_t lo = K->lo[(size_t)i] < n ? K->lo[(size_t)i] : n, hi = K->hi[(size_t)i] < n ? K->hi[(size_t)i] : n;
    const size_t cnt = hi - lo;
    kg_ctx* c = K->ctxs[(size_t)i];
    uint64_t* part = xyz.data() + (size_t)i * 3 * E;
    if (!cnt) { msm_identity(K->curve, part); return (int)KG_OK; }
    // the slice's scalars travel in index sub-slices, each sorted and accumulated while the next one is on the bus (msm_host.cpp):
    // one synchronous copy in front of a blocking MSM would add the whole upload (10 ms per 2^24 scalars) to every commit
    return kg_msm_host_scalars(c, K->curve, K->d_bases[(size_t)i], K->d_inf[(size_t)i], h_scalars + 4 * lo, cnt, part);
  });
  if (rc != KG_OK) return rc;
  return combine(K->ctxs[0], K->curve, xyz, n_ctx, out_xy, out_inf);
  });
}

}  // extern "C"
