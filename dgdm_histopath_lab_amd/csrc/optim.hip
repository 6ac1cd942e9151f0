// AdamW over ALL live parameters of the model in one launch (two beyond 96 tensors): the optimizer step of
// DGDMTrainer.configure_optimizers (reference training/trainer.py:217-226: torch.optim.AdamW(lr, weight_decay)).
// torch's fused AdamW walks its tensor lists through multi_tensor_apply, whose per-launch metadata (4 lists x 320-byte
// chunk tables) cuts the ~180 live tensors of DGDM-Base into 6 launches of 15-50 us (0.19 ms per step, 5x what the 150 MB
// of parameter / moment traffic cost at HBM rate).  Here a launch carries up to 96 tensor descriptors BY VALUE in its kernel
// arguments (so a HIP graph records them; nothing is read from a table in memory that a capture could not upload), a
// workgroup finds its tensor by a search over the block offsets, and streams 4096 elements with 16-byte accesses.
//
// Arithmetic = torch/optim/adamw.py (decoupled weight decay, no amsgrad, no maximize), the order of torch's fused kernel:
//   p -= lr * wd * p;  m += (1 - b1) (g - m);  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with t = step + 1 read from a device scalar that the launch itself advances: the workgroup that finishes last (a ticket)
// writes t back, after every workgroup has read the old value.  Bias corrections in float64 (1 - 0.999^t cancels).
#include "common.hpp"

namespace {

constexpr int ADAM_MAX = 96;          // descriptors per launch: 96 * 40 B = 3840 B of kernel arguments
constexpr int ADAM_BLOCK_ELEMS = 4096;

struct AdamArgs {
  float* p[ADAM_MAX];
  const float* g[ADAM_MAX];
  float* m[ADAM_MAX];
  float* v[ADAM_MAX];
  int32_t blk_end[ADAM_MAX];          // exclusive end of the tensor's workgroup range
  int32_t last_elems[ADAM_MAX];       // elements of the tensor's last workgroup (1 .. 4096)
  int32_t count;
};

struct AdamHyper { float lr, b1, b2, eps, wd; };

__global__ __launch_bounds__(256) void k_adamw_many(const AdamArgs a, const float* __restrict__ lr_dev, const AdamHyper h,
                                                    float* __restrict__ step_dev, unsigned* __restrict__ ticket, int advance) {
  // tensor of this workgroup: first t with blockIdx.x < blk_end[t] (wave-uniform binary search over kernel arguments)
  int lo = 0, hi = a.count - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x < a.blk_end[mid]) hi = mid; else lo = mid + 1;
  }
  const int t = lo;
  const int blk0 = t == 0 ? 0 : a.blk_end[t - 1];
  const int lb = blockIdx.x - blk0;
  const bool last_blk = (int)blockIdx.x == a.blk_end[t] - 1;
  const int n = last_blk ? a.last_elems[t] : ADAM_BLOCK_ELEMS;
  const int64_t off = (int64_t)lb * ADAM_BLOCK_ELEMS;
  float* __restrict__ p = a.p[t] + off;
  const float* __restrict__ g = a.g[t] + off;
  float* __restrict__ m = a.m[t] + off;
  float* __restrict__ v = a.v[t] + off;

  __shared__ float corr[2];
  if (threadIdx.x == 0) {
    const double ts = (double)step_dev[0] + 1.0;
    corr[0] = (float)(1.0 - pow((double)h.b1, ts));
    corr[1] = (float)sqrt(1.0 - pow((double)h.b2, ts));
  }
  __syncthreads();
  const float lr = lr_dev ? lr_dev[0] : h.lr;
  const float step_size = lr / corr[0], bc2s = corr[1];
  const float decay = lr * h.wd, omb1 = 1.f - h.b1, omb2 = 1.f - h.b2;

  auto upd = [&](float& pv, float gv, float& mv, float& vv) {
    pv -= decay * pv;
    mv = fmaf(omb1, gv - mv, mv);
    vv = h.b2 * vv + omb2 * gv * gv;
    const float denom = sqrtf(vv) / bc2s + h.eps;
    pv -= step_size * mv / denom;
  };
  // parameters and moments are allocations of their own (aligned); a gradient may be a slice of the data-parallel reducer's flat
  // buffer at any 4-byte offset: it is then read dword by dword, everything else still moves 16 bytes per access
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15u) == 0;
  const bool gvec = (reinterpret_cast<uintptr_t>(g) & 15u) == 0;
  if (vec) {
    const int n4 = n >> 2;
#pragma unroll
    for (int i = threadIdx.x; i < ADAM_BLOCK_ELEMS / 4; i += 256) {
      if (i < n4) {
        float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gv = gvec ? reinterpret_cast<const float4*>(g)[i] : make_float4(g[4 * i], g[4 * i + 1], g[4 * i + 2], g[4 * i + 3]);
        upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
        reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
      }
    }
    for (int i = 4 * n4 + threadIdx.x; i < n; i += 256) upd(p[i], g[i], m[i], v[i]);
  } else {
    for (int i = threadIdx.x; i < n; i += 256) upd(p[i], g[i], m[i], v[i]);
  }

  // the last workgroup to arrive has seen every other one read step_dev: it advances the count (and re-arms the ticket)
  __syncthreads();
  if (threadIdx.x == 0) {
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
      if (advance) step_dev[0] = step_dev[0] + 1.f;
      ticket[0] = 0u;
    }
  }
}

}  // namespace

extern "C" int dgdm_adamw_step(const DgdmAdamTensor* tensors, int32_t count, const float* lr_dev, float lr, float beta1, float beta2,
                               float eps, float weight_decay, float* step_dev, uint32_t* ticket_dev, void* stream_) {
  DGDM_REQUIRE(count >= 0 && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f);
  DGDM_REQUIRE(step_dev && ticket_dev && (count == 0 || tensors));
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const AdamHyper h{lr, beta1, beta2, eps, weight_decay};
  // validate everything before the first launch
  int32_t live = 0;
  for (int32_t i = 0; i < count; ++i) {
    const DgdmAdamTensor& d = tensors[i];
    DGDM_REQUIRE(d.numel >= 0);
    if (d.numel == 0) continue;
    DGDM_REQUIRE(d.param && d.grad && d.exp_avg && d.exp_avg_sq);
    if (d.numel > (int64_t)ADAM_BLOCK_ELEMS * 0x3fffffff) return DGDM_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(d.param) | reinterpret_cast<uintptr_t>(d.grad) | reinterpret_cast<uintptr_t>(d.exp_avg) |
         reinterpret_cast<uintptr_t>(d.exp_avg_sq)) & 3u) return DGDM_ERR_UNSUPPORTED;
    ++live;
  }
  if (live == 0) {    // nothing carries a gradient: torch's step leaves the counts alone as well
    return DGDM_OK;
  }
  AdamArgs a;
  int32_t k = 0, seen = 0;
  int64_t blocks = 0;
  auto flush = [&](bool last) {
    a.count = k;
    hipLaunchKernelGGL(k_adamw_many, dim3((unsigned)blocks), dim3(256), 0, st, a, lr_dev, h, step_dev, ticket_dev, last ? 1 : 0);
    k = 0; blocks = 0;
  };
  for (int32_t i = 0; i < count; ++i) {
    const DgdmAdamTensor& d = tensors[i];
    if (d.numel == 0) continue;
    const int64_t nb = (d.numel + ADAM_BLOCK_ELEMS - 1) / ADAM_BLOCK_ELEMS;
    if (k == ADAM_MAX || blocks + nb > 0x3fffffff) flush(false);
    a.p[k] = d.param; a.g[k] = d.grad; a.m[k] = d.exp_avg; a.v[k] = d.exp_avg_sq;
    blocks += nb;
    a.blk_end[k] = (int32_t)blocks;
    a.last_elems[k] = (int32_t)(d.numel - (nb - 1) * ADAM_BLOCK_ELEMS);
    ++k; ++seen;
    if (seen == live) flush(true);
  }
  return dgdm_launch_status();
}
