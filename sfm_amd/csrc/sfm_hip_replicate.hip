// sfm_hip_replicate_images of include/sfm_hip.h: one process, several GPUs, every GPU gets every
// image -- each image crosses PCIe once (to the GPU that owns its share) and reaches the others in a
// single RCCL all-gather over xGMI.
//
// The reference has nothing like it: its images live in host memory and every (pair, direction)
// task gets deep copies (src/feature_matcher/brute_force_matcher.cc:47-52).
//
// RCCL is bound at run time (dlopen of librccl.so.1): the matcher itself has no use for it, so the
// library carries no link-time dependency, and inside a process that already holds an RCCL (a
// PyTorch process) the same copy is used.
#include "ctx.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <exception>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

using namespace sfm_host;

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

std::mutex g_mu;             // one replication at a time per process: the communicators are shared
Rccl g_rccl;
std::vector<int> g_comm_devices;       // devices of the cached communicators, in rank order
std::vector<ncclComm_t> g_comms;

bool load_rccl() {
  if (g_rccl.handle) return true;
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) {
    g_rccl.error = std::string("RCCL is not loadable: ") + dlerror();
    return false;
  }
  auto sym = [&](const char* n) { return dlsym(g_rccl.handle, n); };
  g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
  if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.AllGather ||
      !g_rccl.GetErrorString) {
    g_rccl.error = "the RCCL library found lacks a needed entry point";
    dlclose(g_rccl.handle);
    g_rccl.handle = nullptr;
    return false;
  }
  return true;
}

double seconds_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Runs f(r) for every r < n on a thread of its own (one per GPU) and returns the first failure.
template <class F>
int on_every_gpu(int n, F&& f) {
  std::vector<int> rc((size_t)n, SFM_HIP_OK);
  std::vector<std::thread> threads;
  for (int r = 1; r < n; r++) threads.emplace_back([&, r] { rc[(size_t)r] = f(r); });
  rc[0] = f(0);
  for (auto& t : threads) t.join();
  for (int v : rc)
    if (v != SFM_HIP_OK) return v;
  return SFM_HIP_OK;
}

}  // namespace

extern "C" int sfm_hip_replicate_images(sfm_hip_ctx* const* ctxs, int32_t n_ctx, const int64_t* image_ids,
                                        const float* const* descriptors, const int32_t* rows, int32_t n_images,
                                        sfm_hip_replicate_times* out_times) {
  if (!ctxs || n_ctx <= 0) return fail(nullptr, SFM_HIP_ERR_INVALID, "no contexts");
  for (int r = 0; r < n_ctx; r++)
    if (!ctxs[r]) return fail(nullptr, SFM_HIP_ERR_INVALID, "a context is null");
  sfm_hip_ctx* c0 = ctxs[0];
  if (n_images < 0 || (n_images > 0 && (!image_ids || !descriptors || !rows)))
    return fail(c0, SFM_HIP_ERR_INVALID, "null image list");
  for (int r = 0; r < n_ctx; r++)
    for (int q = 0; q < r; q++)
      if (ctxs[r]->device == ctxs[q]->device)
        return fail(c0, SFM_HIP_ERR_INVALID, "two contexts on one device: a collective needs one GPU per context");
  for (int i = 0; i < n_images; i++) {
    if (rows[i] < 0) return fail(c0, SFM_HIP_ERR_INVALID, "negative descriptor count");
    if (rows[i] > 0 && !descriptors[i]) return fail(c0, SFM_HIP_ERR_INVALID, "descriptors is null");
  }
  if (out_times) *out_times = sfm_hip_replicate_times{};
  if (n_images == 0) return SFM_HIP_OK;

  std::lock_guard<std::mutex> lock(g_mu);
  if (!load_rccl()) return fail(c0, SFM_HIP_ERR_HIP, g_rccl.error);
  auto t0 = std::chrono::steady_clock::now();
  std::vector<int> devices((size_t)n_ctx);
  for (int r = 0; r < n_ctx; r++) devices[(size_t)r] = ctxs[r]->device;
  if (devices != g_comm_devices) {
    for (ncclComm_t c : g_comms) (void)g_rccl.CommDestroy(c);
    g_comms.assign((size_t)n_ctx, nullptr);
    g_comm_devices.clear();
    const ncclResult_t e = g_rccl.CommInitAll(g_comms.data(), n_ctx, devices.data());
    if (e != ncclSuccess) {
      g_comms.clear();
      return fail(c0, SFM_HIP_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(e));
    }
    g_comm_devices = devices;
  }
  if (out_times) out_times->comm_init_s = seconds_since(t0);

  // Shares: images in ascending id, cut into n_ctx runs of about equal row totals (the all-gather
  // moves equal-sized shares, so the largest share sets the size of all).
  std::vector<int32_t> by_id((size_t)n_images);
  std::iota(by_id.begin(), by_id.end(), 0);
  std::sort(by_id.begin(), by_id.end(), [&](int32_t a, int32_t b) { return image_ids[a] < image_ids[b]; });
  for (int k = 1; k < n_images; k++)
    if (image_ids[by_id[(size_t)k]] == image_ids[by_id[(size_t)k - 1]])
      return fail(c0, SFM_HIP_ERR_INVALID, "image id " + std::to_string(image_ids[by_id[(size_t)k]]) + " appears twice");
  size_t total_rows = 0;
  for (int i = 0; i < n_images; i++) total_rows += (size_t)rows[i];
  std::vector<int> owner((size_t)n_images);          // by position in by_id
  std::vector<size_t> at((size_t)n_images);          // first row inside the owner's share
  std::vector<size_t> share_rows((size_t)n_ctx, 0);
  {
    size_t before = 0;
    for (int k = 0; k < n_images; k++) {
      const size_t n = (size_t)rows[by_id[(size_t)k]];
      int r = total_rows ? (int)((before + n / 2) * (size_t)n_ctx / total_rows) : 0;
      r = std::min(r, n_ctx - 1);
      owner[(size_t)k] = r;
      at[(size_t)k] = share_rows[(size_t)r];
      share_rows[(size_t)r] += n;
      before += n;
    }
  }
  const size_t slot_rows = std::max<size_t>(1, *std::max_element(share_rows.begin(), share_rows.end()));
  const size_t slot_floats = slot_rows * SFM_HIP_DIM;

  // 1. every GPU: room for all shares; its own share arrives from the host
  t0 = std::chrono::steady_clock::now();
  std::vector<float*> everything((size_t)n_ctx, nullptr);
  auto release = [&] {
    for (int r = 0; r < n_ctx; r++)
      if (everything[(size_t)r]) {
        (void)hipSetDevice(ctxs[r]->device);
        (void)hipFree(everything[(size_t)r]);
      }
  };
  int rc = on_every_gpu(n_ctx, [&](int r) -> int {
    sfm_hip_ctx* ctx = ctxs[r];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc((void**)&everything[(size_t)r], slot_floats * sizeof(float) * (size_t)n_ctx));
    float* mine = everything[(size_t)r] + (size_t)r * slot_floats;
    for (int k = 0; k < n_images; k++) {
      if (owner[(size_t)k] != r) continue;
      const int32_t i = by_id[(size_t)k];
      if (rows[i] > 0)
        HIP_TRY(ctx, hipMemcpyAsync(mine + at[(size_t)k] * SFM_HIP_DIM, descriptors[i],
                                    (size_t)rows[i] * SFM_HIP_DIM * sizeof(float), hipMemcpyHostToDevice, ctx->upload_stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->upload_stream));
    return SFM_HIP_OK;
  });
  if (rc != SFM_HIP_OK) {
    release();
    for (int r = 1; r < n_ctx; r++)
      if (!ctxs[r]->last_error.empty() && c0->last_error.empty()) c0->last_error = ctxs[r]->last_error;
    return rc;
  }
  if (out_times) out_times->upload_s = seconds_since(t0);

  // 2. one all-gather, in place (each GPU's share already sits in its slot of its own buffer)
  t0 = std::chrono::steady_clock::now();
  ncclResult_t ne = g_rccl.GroupStart();
  for (int r = 0; r < n_ctx && ne == ncclSuccess; r++) {
    float* buf = everything[(size_t)r];
    ne = g_rccl.AllGather(buf + (size_t)r * slot_floats, buf, slot_floats, ncclFloat, g_comms[(size_t)r], ctxs[r]->upload_stream);
  }
  const ncclResult_t ge = g_rccl.GroupEnd();
  if (ne == ncclSuccess) ne = ge;
  hipError_t he = hipSuccess;
  for (int r = 0; r < n_ctx; r++) {
    (void)hipSetDevice(ctxs[r]->device);
    const hipError_t e = hipStreamSynchronize(ctxs[r]->upload_stream);
    if (he == hipSuccess) he = e;
  }
  if (ne != ncclSuccess || he != hipSuccess) {
    release();
    return fail(c0, SFM_HIP_ERR_HIP,
                ne != ncclSuccess ? std::string("ncclAllGather: ") + g_rccl.GetErrorString(ne)
                                  : std::string("all-gather: ") + hipGetErrorString(he));
  }
  if (out_times) out_times->allgather_s = seconds_since(t0);

  // 3. every GPU registers every image from where it now lies (padding, norms and split planes are
  //    made by the same code path as for an upload from the host)
  t0 = std::chrono::steady_clock::now();
  rc = on_every_gpu(n_ctx, [&](int r) -> int {
    for (int k = 0; k < n_images; k++) {
      const int32_t i = by_id[(size_t)k];
      const float* src = everything[(size_t)r] + (size_t)owner[(size_t)k] * slot_floats + at[(size_t)k] * SFM_HIP_DIM;
      const int v = sfm_hip_upload_image_device(ctxs[r], image_ids[i], rows[i] > 0 ? src : nullptr, rows[i]);
      if (v != SFM_HIP_OK) return v;
    }
    return SFM_HIP_OK;
  });
  release();
  if (rc != SFM_HIP_OK) {
    for (int r = 1; r < n_ctx; r++)
      if (!ctxs[r]->last_error.empty() && c0->last_error.empty()) c0->last_error = ctxs[r]->last_error;
    // A GPU that ran out of memory part-way leaves every context with a different part of the set:
    // none of the call's images stays anywhere, so that all contexts agree again.
    const std::string why = c0->last_error;
    for (int r = 0; r < n_ctx; r++)
      for (int k = 0; k < n_images; k++)
        if (ctxs[r]->images.count(image_ids[k])) (void)sfm_hip_drop_image(ctxs[r], image_ids[k]);
    c0->last_error = why;
    return rc;
  }
  if (out_times) out_times->register_s = seconds_since(t0);
  return SFM_HIP_OK;
}
