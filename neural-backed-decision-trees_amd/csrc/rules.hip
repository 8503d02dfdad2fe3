// NBDT decision-rule layer + SoftTreeSupLoss on gfx950.
//
// Replaces the reference's per-node Python loops over stock aten ops
//   get_node_logits ........ nbdt/model.py:83-99     (K gathers + means per inner node)
//   get_all_node_outputs ... nbdt/model.py:101-120   (argmax / softmax / entropy per node)
//   Soft traverse_tree ..... nbdt/model.py:207-242   (class_probs[:, old] *= probs[:, new])
//   Hard traverse_tree ..... nbdt/model.py:145-192   (per-sample python walk, D2H per node)
//   SoftTreeSupLoss ........ nbdt/loss.py:191-203, 260-266  (+ autograd backward)
//   HardTreeSupLoss ........ nbdt/loss.py:212-257, model.py:127-143 (+ autograd backward)
// by ONE launch per call: a group of TPS lanes owns one sample; the sample's logits, the R child
// logits and the R child probabilities live in LDS; the hierarchy is a pair of CSR maps
// (slot -> classes, class -> slots) read through L2.  HBM traffic is the algorithmic minimum
// (read z once, write P / gz once); the path is launch-latency bound, so everything
// for one call is fused into a single kernel and nothing round-trips through HBM.
//
// Arithmetic contract (matches oracle/nbdt_oracle.py bit-for-bit on the integer outputs):
// child logit = sequential fp32 sum over ascending class index, one IEEE division by the count;
// argmax = first maximum; path product = 1.0f * p(node_1) * p(node_2) ... in inode order.
// Compiled with -ffp-contract=off and correctly rounded division.
#include "common.h"

namespace nbdt {
thread_local char g_err[512] = "";
}

using namespace nbdt;

struct nbdt_tree {
  int device;
  int C, N, R, L, root, max_depth;
  int32_t* d_all;  // one allocation
  const int32_t *node_off, *slot_off, *slot_cls, *cls_off, *cls_slot, *slot_next;
};

struct TreeView {
  int C, N, R, root, max_depth;
  const int32_t *node_off, *slot_off, *slot_cls, *cls_off, *cls_slot, *slot_next;
};

static TreeView view_of(const nbdt_tree* t) {
  TreeView v;
  v.C = t->C; v.N = t->N; v.R = t->R; v.root = t->root; v.max_depth = t->max_depth;
  v.node_off = t->node_off; v.slot_off = t->slot_off; v.slot_cls = t->slot_cls;
  v.cls_off = t->cls_off; v.cls_slot = t->cls_slot; v.slot_next = t->slot_next;
  return v;
}

// ------------------------------------------------------------------------------------------
// device helpers

struct LoadF32 { static __device__ __forceinline__ float at(const void* p, int64_t i) { return ((const float*)p)[i]; } };
struct LoadBF16 { static __device__ __forceinline__ float at(const void* p, int64_t i) { return bf16_to_f32(((const bf16_t*)p)[i]); } };
struct LoadF16 { static __device__ __forceinline__ float at(const void* p, int64_t i) { return __half2float(((const __half*)p)[i]); } };

constexpr int kBlock = 256;

template <int TPS>
__device__ __forceinline__ float group_max(float v, float* red, int g_tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if (TPS > 64) {
    if ((g_tid & 63) == 0) red[g_tid >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < TPS / 64; ++i) r = fmaxf(r, red[i]);
    __syncthreads();
    v = r;
  }
  return v;
}

template <int TPS>
__device__ __forceinline__ float group_sum(float v, float* red, int g_tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (TPS > 64) {
    if ((g_tid & 63) == 0) red[g_tid >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < TPS / 64; ++i) r += red[i];
    __syncthreads();
    v = r;
  }
  return v;
}

// phase 0+1: logits row -> LDS, then child logits (nbdt/model.py:94-99)
template <int TPS, typename LD>
__device__ __forceinline__ void load_and_node_logits(const TreeView& t, const void* z, int64_t row_off,
                                                     bool active, int g_tid, float* zs, float* ss) {
  if (active)
    for (int c = g_tid; c < t.C; c += TPS) zs[c] = LD::at(z, row_off + c);
  __syncthreads();
  if (active)
    for (int s = g_tid; s < t.R; s += TPS) {
      const int b = t.slot_off[s], e = t.slot_off[s + 1];
      float acc = 0.f;
      for (int j = b; j < e; ++j) acc = acc + zs[t.slot_cls[j]];
      ss[s] = acc / (float)(e - b);
    }
  __syncthreads();
}

// phase 2: per-node softmax (nbdt/model.py:114)
template <int TPS>
__device__ __forceinline__ void node_softmax(const TreeView& t, bool active, int g_tid, const float* ss,
                                             float* ps) {
  if (active)
    for (int n = g_tid; n < t.N; n += TPS) {
      const int b = t.node_off[n], e = t.node_off[n + 1];
      float m = ss[b];
      for (int s = b + 1; s < e; ++s) m = fmaxf(m, ss[s]);
      float sum = 0.f;
      for (int s = b; s < e; ++s) {
        const float ex = expf(ss[s] - m);
        ps[s] = ex;
        sum = sum + ex;
      }
      for (int s = b; s < e; ++s) ps[s] = ps[s] / sum;
    }
  __syncthreads();
}

__device__ __forceinline__ float path_product(const TreeView& t, int c, const float* ps) {
  float p = 1.0f;
  const int b = t.cls_off[c], e = t.cls_off[c + 1];
  for (int j = b; j < e; ++j) p = p * ps[t.cls_slot[j]];
  return p;
}

// Categorical(probs=p).entropy() for one node (probs renormalised, log clamped to [eps, 1-eps])
__device__ __forceinline__ float node_entropy(const float* ps, int b, int e) {
  const float eps = 1.1920928955078125e-07f;
  float tot = 0.f;
  for (int s = b; s < e; ++s) tot = tot + ps[s];
  float h = 0.f;
  for (int s = b; s < e; ++s) {
    const float p = ps[s] / tot;
    const float cl = fminf(fmaxf(p, eps), 1.0f - eps);
    h = h + p * logf(cl);
  }
  return -h;
}

// ------------------------------------------------------------------------------------------
// kernels: grid = ceil(B / SPB) blocks of 256 threads, SPB = 256 / TPS samples per block

template <int TPS, typename LD>
__global__ __launch_bounds__(kBlock) void soft_fwd_kernel(TreeView t, const void* z, int64_t B, int64_t ldz,
                                                          float* __restrict__ P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SPB = kBlock / TPS;
  const int g = threadIdx.x / TPS, g_tid = threadIdx.x % TPS;
  const int64_t sample = (int64_t)blockIdx.x * SPB + g;
  const bool active = sample < B;
  const int stride = t.C + 2 * t.R;
  float* zs = lds + (size_t)g * stride;
  float* ss = zs + t.C;
  float* ps = ss + t.R;
  load_and_node_logits<TPS, LD>(t, z, sample * ldz, active, g_tid, zs, ss);
  node_softmax<TPS>(t, active, g_tid, ss, ps);
  if (active)
    for (int c = g_tid; c < t.C; c += TPS) P[sample * t.C + c] = path_product(t, c, ps);
}

// shared tail of the two backward flavours: Pg (= P*g per class) is in `pq`; ss is overwritten by
// G then ds; result accumulated on top of `base` per class.
template <int TPS>
__device__ __forceinline__ void tree_backward(const TreeView& t, bool active, int g_tid, float* ss,
                                              const float* ps, const float* pq) {
  if (active)
    for (int s = g_tid; s < t.R; s += TPS) {
      const int b = t.slot_off[s], e = t.slot_off[s + 1];
      float acc = 0.f;
      for (int j = b; j < e; ++j) acc = acc + pq[t.slot_cls[j]];
      ss[s] = acc;  // G[slot]
    }
  __syncthreads();
  if (active)
    for (int n = g_tid; n < t.N; n += TPS) {
      const int b = t.node_off[n], e = t.node_off[n + 1];
      float tot = 0.f;
      for (int s = b; s < e; ++s) tot = tot + ss[s];
      for (int s = b; s < e; ++s) ss[s] = ss[s] - ps[s] * tot;  // dL/ds
    }
  __syncthreads();
}

__device__ __forceinline__ float class_grad(const TreeView& t, int c, const float* ds) {
  float acc = 0.f;
  const int b = t.cls_off[c], e = t.cls_off[c + 1];
  for (int j = b; j < e; ++j) {
    const int s = t.cls_slot[j];
    acc = acc + ds[s] / (float)(t.slot_off[s + 1] - t.slot_off[s]);
  }
  return acc;
}

template <int TPS, typename LD>
__global__ __launch_bounds__(kBlock) void soft_bwd_kernel(TreeView t, const void* z, int64_t B, int64_t ldz,
                                                          const float* __restrict__ gP,
                                                          float* __restrict__ gz) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SPB = kBlock / TPS;
  const int g = threadIdx.x / TPS, g_tid = threadIdx.x % TPS;
  const int64_t sample = (int64_t)blockIdx.x * SPB + g;
  const bool active = sample < B;
  const int stride = 2 * t.C + 2 * t.R;
  float* zs = lds + (size_t)g * stride;
  float* ss = zs + t.C;
  float* ps = ss + t.R;
  float* pq = ps + t.R;
  load_and_node_logits<TPS, LD>(t, z, sample * ldz, active, g_tid, zs, ss);
  node_softmax<TPS>(t, active, g_tid, ss, ps);
  if (active)
    for (int c = g_tid; c < t.C; c += TPS) pq[c] = path_product(t, c, ps) * gP[sample * t.C + c];
  __syncthreads();
  tree_backward<TPS>(t, active, g_tid, ss, ps, pq);
  if (active)
    for (int c = g_tid; c < t.C; c += TPS) gz[sample * t.C + c] = class_grad(t, c, ss);
}

// SoftTreeSupLoss forward+backward for criterion = nn.CrossEntropyLoss() (mean reduction):
//   row = w_x*(lse(z) - z[y]) + w_t*(lse(P) - P[y])     (P fed to CE as if logits, loss.py:266)
//   gz  = scale*( w_x*(softmax(z) - 1[y]) + J^T * w_t*(softmax(P) - 1[y]) )
template <int TPS, typename LD>
__global__ __launch_bounds__(kBlock) void soft_loss_kernel(TreeView t, const void* z, int64_t B, int64_t ldz,
                                                           const int64_t* __restrict__ y, float w_x,
                                                           float w_t, float scale,
                                                           float* __restrict__ row_loss,
                                                           float* __restrict__ gz) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SPB = kBlock / TPS;
  const int g = threadIdx.x / TPS, g_tid = threadIdx.x % TPS;
  const int64_t sample = (int64_t)blockIdx.x * SPB + g;
  const bool active = sample < B;
  const int stride = 3 * t.C + 2 * t.R + 8;
  float* zs = lds + (size_t)g * stride;
  float* ss = zs + t.C;
  float* ps = ss + t.R;
  float* pq = ps + t.R;
  float* gx = pq + t.C;
  float* red = gx + t.C;
  load_and_node_logits<TPS, LD>(t, z, sample * ldz, active, g_tid, zs, ss);
  node_softmax<TPS>(t, active, g_tid, ss, ps);

  float mz = -INFINITY, mp = -INFINITY;
  if (active)
    for (int c = g_tid; c < t.C; c += TPS) {
      const float p = path_product(t, c, ps);
      pq[c] = p;
      mp = fmaxf(mp, p);
      mz = fmaxf(mz, zs[c]);
    }
  mz = group_max<TPS>(mz, red, g_tid);
  mp = group_max<TPS>(mp, red, g_tid);
  float sz = 0.f, sp = 0.f;
  if (active)
    for (int c = g_tid; c < t.C; c += TPS) {
      sz += expf(zs[c] - mz);
      sp += expf(pq[c] - mp);
    }
  sz = group_sum<TPS>(sz, red, g_tid);
  sp = group_sum<TPS>(sp, red, g_tid);

  if (active) {
    const int64_t yy = y[sample];
    const bool valid = yy >= 0 && yy < t.C;
    for (int c = g_tid; c < t.C; c += TPS) {
      const float hot = (c == yy) ? 1.f : 0.f;
      const float p = pq[c];
      if (c == yy) row_loss[sample] = w_x * ((logf(sz) + mz) - zs[c]) + w_t * ((logf(sp) + mp) - p);
      gx[c] = (expf(zs[c] - mz) / sz - hot) * (w_x * scale);
      const float gp = (expf(p - mp) / sp - hot) * (w_t * scale);
      pq[c] = p * gp;
    }
    if (!valid && g_tid == 0) row_loss[sample] = __uint_as_float(0x7fc00000u);  // loud: NaN loss
  }
  __syncthreads();
  tree_backward<TPS>(t, active, g_tid, ss, ps, pq);
  if (active)
    for (int c = g_tid; c < t.C; c += TPS) gz[sample * t.C + c] = gx[c] + class_grad(t, c, ss);
}

// HardTreeSupLoss forward+backward for criterion = nn.CrossEntropyLoss() (nbdt/loss.py:212-257):
//   row = w_x*(lse(z) - z[y]) + w_h * sum over inner nodes n with y under n of
//                                     (lse(s[n,:]) - s[n, child_of(n,y)])
//   (w_h folds the reference's pooled-by-child-count means: every (sample,node) term ends up with
//    the same weight tsw/(B*N/2), loss.py:228,250-256, times the scheduled tree weight, :195-203)
// The nodes on the label's path are exactly the class->slot CSR row of y; a class listed under two
// children of one node counts once, for the first child (model.py:135 `cls[0]`).
template <int TPS, typename LD>
__global__ __launch_bounds__(kBlock) void hard_loss_kernel(TreeView t, const void* z, int64_t B, int64_t ldz,
                                                           const int64_t* __restrict__ y, float w_x,
                                                           float w_h, float scale,
                                                           float* __restrict__ row_loss,
                                                           float* __restrict__ gz) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SPB = kBlock / TPS;
  const int g = threadIdx.x / TPS, g_tid = threadIdx.x % TPS;
  const int64_t sample = (int64_t)blockIdx.x * SPB + g;
  const bool active = sample < B;
  const int stride = 2 * t.C + 2 * t.R + 8;
  float* zs = lds + (size_t)g * stride;
  float* ss = zs + t.C;
  float* ds = ss + t.R;
  float* gx = ds + t.R;
  float* red = gx + t.C;
  load_and_node_logits<TPS, LD>(t, z, sample * ldz, active, g_tid, zs, ss);
  if (active)
    for (int s = g_tid; s < t.R; s += TPS) ds[s] = 0.f;
  __syncthreads();

  int64_t yy = 0;
  bool valid = false;
  if (active) {
    yy = y[sample];
    valid = yy >= 0 && yy < t.C;
  }
  float tree_rows = 0.f;
  if (valid) {
    const int pb = t.cls_off[yy], pe = t.cls_off[yy + 1];
    for (int j = pb + g_tid; j < pe; j += TPS) {
      const int s = t.cls_slot[j];
      // node owning slot s: last n with node_off[n] <= s
      int lo = 0, hi = t.N - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.node_off[mid] <= s) lo = mid; else hi = mid - 1;
      }
      const int b = t.node_off[lo], e = t.node_off[lo + 1];
      if (j > pb && t.cls_slot[j - 1] >= b) continue;  // same node, later child: not the first
      float m = ss[b];
      for (int q = b + 1; q < e; ++q) m = fmaxf(m, ss[q]);
      float sum = 0.f;
      for (int q = b; q < e; ++q) sum = sum + expf(ss[q] - m);
      tree_rows += (logf(sum) + m) - ss[s];
      for (int q = b; q < e; ++q)
        ds[q] = (expf(ss[q] - m) / sum - (q == s ? 1.f : 0.f)) * (w_h * scale);
    }
  }
  tree_rows = group_sum<TPS>(tree_rows, red, g_tid);

  float mz = -INFINITY;
  if (active)
    for (int c = g_tid; c < t.C; c += TPS) mz = fmaxf(mz, zs[c]);
  mz = group_max<TPS>(mz, red, g_tid);
  float sz = 0.f;
  if (active)
    for (int c = g_tid; c < t.C; c += TPS) sz += expf(zs[c] - mz);
  sz = group_sum<TPS>(sz, red, g_tid);
  if (active) {
    for (int c = g_tid; c < t.C; c += TPS) {
      const float hot = (c == yy) ? 1.f : 0.f;
      if (c == yy) row_loss[sample] = w_x * ((logf(sz) + mz) - zs[c]) + w_h * tree_rows;
      gx[c] = (expf(zs[c] - mz) / sz - hot) * (w_x * scale);
    }
    if (!valid && g_tid == 0) row_loss[sample] = __uint_as_float(0x7fc00000u);  // loud: NaN loss
  }
  __syncthreads();
  if (active)
    for (int c = g_tid; c < t.C; c += TPS) gz[sample * t.C + c] = gx[c] + class_grad(t, c, ds);
}

// VJP of the node-logit map (nbdt/model.py:94-99): gz[b,c] = sum over slots s holding c of
// gs[b,s] / |leaves(s)|.  Lets any torch criterion be composed on top of nbdt_node_outputs logits.
__global__ __launch_bounds__(kBlock) void node_logits_bwd_kernel(TreeView t, const float* __restrict__ gs,
                                                                 int64_t B, float* __restrict__ gz) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= B * t.C) return;
  const int64_t b = i / t.C;
  const int c = (int)(i - b * t.C);
  gz[i] = class_grad(t, c, gs + b * t.R);
}

__global__ __launch_bounds__(kBlock) void mean_kernel(const float* __restrict__ rows, int64_t n,
                                                      float* __restrict__ out) {
  __shared__ float red[kBlock / 64];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kBlock) acc += rows[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

template <int TPS, typename LD>
__global__ __launch_bounds__(kBlock) void hard_fwd_kernel(TreeView t, const void* z, int64_t B, int64_t ldz,
                                                          int64_t* __restrict__ pred,
                                                          float* __restrict__ onehot,
                                                          int32_t* __restrict__ path_node,
                                                          int32_t* __restrict__ path_child,
                                                          float* __restrict__ path_prob,
                                                          float* __restrict__ path_entropy) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SPB = kBlock / TPS;
  const int g = threadIdx.x / TPS, g_tid = threadIdx.x % TPS;
  const int64_t sample = (int64_t)blockIdx.x * SPB + g;
  const bool active = sample < B;
  const int stride = t.C + t.R + 8;
  float* zs = lds + (size_t)g * stride;
  float* ss = zs + t.C;
  int* res = (int*)(ss + t.R);
  load_and_node_logits<TPS, LD>(t, z, sample * ldz, active, g_tid, zs, ss);
  if (active && g_tid == 0) {
    int n = t.root, cls = -1;
    const bool want = path_node != nullptr;
    int d = 0;
    for (; d < t.max_depth; ++d) {
      const int b = t.node_off[n], e = t.node_off[n + 1];
      int best = b;
      for (int s = b + 1; s < e; ++s)
        if (ss[s] > ss[best]) best = s;  // first maximum wins (torch.max, model.py:113)
      if (want) {
        const float m = ss[best];
        float sum = 0.f;
        for (int s = b; s < e; ++s) sum = sum + expf(ss[s] - m);
        // entropy of the node's distribution (Categorical semantics)
        const float eps = 1.1920928955078125e-07f;
        float tot = 0.f;
        for (int s = b; s < e; ++s) tot = tot + expf(ss[s] - m) / sum;
        float h = 0.f;
        for (int s = b; s < e; ++s) {
          const float p = (expf(ss[s] - m) / sum) / tot;
          h = h + p * logf(fminf(fmaxf(p, eps), 1.0f - eps));
        }
        const int64_t o = sample * t.max_depth + d;
        path_node[o] = n;
        path_child[o] = best - b;
        path_prob[o] = expf(ss[best] - m) / sum;
        path_entropy[o] = -h;
      }
      const int nx = t.slot_next[best];
      if (nx >= 0) {
        n = nx;
      } else {
        cls = -nx - 1;
        ++d;
        break;
      }
    }
    if (want)
      for (; d < t.max_depth; ++d) {
        const int64_t o = sample * t.max_depth + d;
        path_node[o] = -1; path_child[o] = -1; path_prob[o] = 0.f; path_entropy[o] = 0.f;
      }
    pred[sample] = cls;
    res[0] = cls;
  }
  __syncthreads();
  if (active && onehot != nullptr) {
    const int cls = res[0];
    for (int c = g_tid; c < t.C; c += TPS) onehot[sample * t.C + c] = (c == cls) ? 1.f : 0.f;
  }
}

template <int TPS, typename LD>
__global__ __launch_bounds__(kBlock) void node_outputs_kernel(TreeView t, const void* z, int64_t B, int64_t ldz,
                                                              float* __restrict__ logits,
                                                              float* __restrict__ probs,
                                                              int64_t* __restrict__ preds,
                                                              float* __restrict__ entropy) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SPB = kBlock / TPS;
  const int g = threadIdx.x / TPS, g_tid = threadIdx.x % TPS;
  const int64_t sample = (int64_t)blockIdx.x * SPB + g;
  const bool active = sample < B;
  const int stride = t.C + 2 * t.R;
  float* zs = lds + (size_t)g * stride;
  float* ss = zs + t.C;
  float* ps = ss + t.R;
  load_and_node_logits<TPS, LD>(t, z, sample * ldz, active, g_tid, zs, ss);
  node_softmax<TPS>(t, active, g_tid, ss, ps);
  if (!active) return;
  for (int s = g_tid; s < t.R; s += TPS) {
    if (logits) logits[sample * t.R + s] = ss[s];
    if (probs) probs[sample * t.R + s] = ps[s];
  }
  for (int n = g_tid; n < t.N; n += TPS) {
    const int b = t.node_off[n], e = t.node_off[n + 1];
    if (preds) {
      int best = b;
      for (int s = b + 1; s < e; ++s)
        if (ss[s] > ss[best]) best = s;
      preds[sample * t.N + n] = best - b;
    }
    if (entropy) entropy[sample * t.N + n] = node_entropy(ps, b, e);
  }
}

// ------------------------------------------------------------------------------------------
// host side

extern "C" const char* nbdt_last_error(void) { return nbdt::g_err; }
extern "C" int nbdt_version(void) { return 100; }
extern "C" int nbdt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int nbdt_tree_create(int device, int C, int N, int root, const int32_t* node_off,
                                const int32_t* slot_off, const int32_t* slot_cls, const int32_t* cls_off,
                                const int32_t* cls_slot, const int32_t* slot_next, nbdt_tree** out) {
  NBDT_REQUIRE(out && node_off && slot_off && slot_cls && cls_off && cls_slot && slot_next, "null argument");
  NBDT_REQUIRE(C > 0 && N > 0 && root >= 0 && root < N, "bad tree sizes");
  const int R = node_off[N];
  const int L = slot_off[R];
  NBDT_REQUIRE(R > 0 && L > 0 && cls_off[C] == L, "inconsistent CSR maps");
  for (int n = 0; n < N; ++n) NBDT_REQUIRE(node_off[n + 1] > node_off[n], "inner node without children");
  for (int s = 0; s < R; ++s) {
    NBDT_REQUIRE(slot_off[s + 1] > slot_off[s], "child without leaves");
    NBDT_REQUIRE(slot_next[s] < N && slot_next[s] >= -C, "bad slot_next");
  }
  for (int j = 0; j < L; ++j) {
    NBDT_REQUIRE(slot_cls[j] >= 0 && slot_cls[j] < C, "class index out of range");
    NBDT_REQUIRE(cls_slot[j] >= 0 && cls_slot[j] < R, "slot index out of range");
  }
  // longest decision path (number of inner nodes visited); also rejects cycles
  int max_depth = 0;
  {
    // iterative relaxation: depth[n] = 1 + max(depth[child inner])
    int* depth = new int[N];
    for (int n = 0; n < N; ++n) depth[n] = 1;
    bool changed = true;
    int iters = 0;
    while (changed && iters <= N + 1) {
      changed = false;
      ++iters;
      for (int n = 0; n < N; ++n)
        for (int s = node_off[n]; s < node_off[n + 1]; ++s)
          if (slot_next[s] >= 0 && depth[n] < depth[slot_next[s]] + 1) {
            depth[n] = depth[slot_next[s]] + 1;
            changed = true;
          }
    }
    max_depth = depth[root];
    delete[] depth;
    NBDT_REQUIRE(iters <= N + 1, "hierarchy has a cycle");
  }

  int prev = 0;
  NBDT_HIP_CHECK(hipGetDevice(&prev));
  NBDT_HIP_CHECK(hipSetDevice(device));
  nbdt_tree* t = new nbdt_tree();
  t->device = device; t->C = C; t->N = N; t->R = R; t->L = L; t->root = root; t->max_depth = max_depth;
  const size_t n_ints = (size_t)(N + 1) + (R + 1) + L + (C + 1) + L + R;
  hipError_t e = hipMalloc((void**)&t->d_all, n_ints * sizeof(int32_t));
  if (e != hipSuccess) {
    delete t;
    (void)hipSetDevice(prev);
    return nbdt::fail(NBDT_ENOMEM, "hipMalloc(tree): %s", hipGetErrorString(e));
  }
  int32_t* host = new int32_t[n_ints];
  size_t o = 0;
  auto put = [&](const int32_t* src, size_t n, const int32_t** dst) {
    memcpy(host + o, src, n * sizeof(int32_t));
    *dst = t->d_all + o;
    o += n;
  };
  put(node_off, N + 1, &t->node_off);
  put(slot_off, R + 1, &t->slot_off);
  put(slot_cls, L, &t->slot_cls);
  put(cls_off, C + 1, &t->cls_off);
  put(cls_slot, L, &t->cls_slot);
  put(slot_next, R, &t->slot_next);
  e = hipMemcpy(t->d_all, host, n_ints * sizeof(int32_t), hipMemcpyHostToDevice);
  delete[] host;
  (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    (void)hipFree(t->d_all);
    delete t;
    return nbdt::fail(NBDT_EHIP, "hipMemcpy(tree): %s", hipGetErrorString(e));
  }
  *out = t;
  return NBDT_OK;
}

extern "C" int nbdt_tree_destroy(nbdt_tree* t) {
  if (!t) return NBDT_OK;
  (void)hipFree(t->d_all);
  delete t;
  return NBDT_OK;
}

extern "C" int nbdt_tree_max_depth(const nbdt_tree* t) { return t ? t->max_depth : 0; }

// threads per sample: one wave for small hierarchies, the whole block for large ones
static int pick_tps(const nbdt_tree* t) { return (t->C > 256 || t->R > 512) ? 256 : 64; }

#define NBDT_DISPATCH_RULES(KERNEL, FLOATS_PER_SAMPLE, ...)                                          \
  do {                                                                                               \
    const int tps = pick_tps(t);                                                                     \
    const int spb = kBlock / tps;                                                                    \
    const size_t shmem = (size_t)spb * (FLOATS_PER_SAMPLE) * sizeof(float);                          \
    NBDT_REQUIRE(shmem <= 64 * 1024, "hierarchy too large for LDS");                                \
    const unsigned grid = (unsigned)((B + spb - 1) / spb);                                           \
    hipStream_t st = (hipStream_t)stream;                                                            \
    if (tps == 64) {                                                                                 \
      if (ztype == NBDT_F32) hipLaunchKernelGGL((KERNEL<64, LoadF32>), dim3(grid), dim3(kBlock), shmem, st, __VA_ARGS__);   \
      else if (ztype == NBDT_BF16) hipLaunchKernelGGL((KERNEL<64, LoadBF16>), dim3(grid), dim3(kBlock), shmem, st, __VA_ARGS__); \
      else hipLaunchKernelGGL((KERNEL<64, LoadF16>), dim3(grid), dim3(kBlock), shmem, st, __VA_ARGS__);                     \
    } else {                                                                                         \
      if (ztype == NBDT_F32) hipLaunchKernelGGL((KERNEL<256, LoadF32>), dim3(grid), dim3(kBlock), shmem, st, __VA_ARGS__);  \
      else if (ztype == NBDT_BF16) hipLaunchKernelGGL((KERNEL<256, LoadBF16>), dim3(grid), dim3(kBlock), shmem, st, __VA_ARGS__); \
      else hipLaunchKernelGGL((KERNEL<256, LoadF16>), dim3(grid), dim3(kBlock), shmem, st, __VA_ARGS__);                    \
    }                                                                                                \
    NBDT_LAUNCH_CHECK();                                                                             \
  } while (0)

static int check_common(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz) {
  NBDT_REQUIRE(t != nullptr, "null tree handle");
  NBDT_REQUIRE(z != nullptr || B == 0, "null logits");
  NBDT_REQUIRE(ztype == NBDT_F32 || ztype == NBDT_BF16 || ztype == NBDT_F16, "unsupported logits dtype");
  NBDT_REQUIRE(B >= 0 && ldz >= t->C, "bad batch / row stride");
  return NBDT_OK;
}

extern "C" int nbdt_soft_forward(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz,
                                 float* P, void* stream) {
  int rc = check_common(t, z, ztype, B, ldz);
  if (rc) return rc;
  if (B == 0) return NBDT_OK;
  NBDT_REQUIRE(P != nullptr, "null output");
  TreeView v = view_of(t);
  NBDT_DISPATCH_RULES(soft_fwd_kernel, t->C + 2 * t->R, v, z, B, ldz, P);
  return NBDT_OK;
}

extern "C" int nbdt_soft_backward(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz,
                                  const float* gP, float* gz, void* stream) {
  int rc = check_common(t, z, ztype, B, ldz);
  if (rc) return rc;
  if (B == 0) return NBDT_OK;
  NBDT_REQUIRE(gP != nullptr && gz != nullptr, "null gradient buffer");
  TreeView v = view_of(t);
  NBDT_DISPATCH_RULES(soft_bwd_kernel, 2 * t->C + 2 * t->R, v, z, B, ldz, gP, gz);
  return NBDT_OK;
}

extern "C" int nbdt_soft_tree_loss(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz,
                                   const int64_t* y, float w_xent, float w_tree, float grad_scale,
                                   float* row_loss, float* loss, float* gz, void* stream) {
  int rc = check_common(t, z, ztype, B, ldz);
  if (rc) return rc;
  NBDT_REQUIRE(y && row_loss && loss && gz, "null buffer");
  NBDT_REQUIRE(B > 0, "empty batch has no mean loss");
  TreeView v = view_of(t);
  const float scale = grad_scale / (float)B;
  NBDT_DISPATCH_RULES(soft_loss_kernel, 3 * t->C + 2 * t->R + 8, v, z, B, ldz, y, w_xent, w_tree, scale,
                      row_loss, gz);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, row_loss, B, loss);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_hard_tree_loss(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz,
                                   const int64_t* y, float w_xent, float w_node, float grad_scale,
                                   float* row_loss, float* loss, float* gz, void* stream) {
  int rc = check_common(t, z, ztype, B, ldz);
  if (rc) return rc;
  NBDT_REQUIRE(y && row_loss && loss && gz, "null buffer");
  NBDT_REQUIRE(B > 0, "empty batch has no mean loss");
  TreeView v = view_of(t);
  const float scale = grad_scale / (float)B;
  NBDT_DISPATCH_RULES(hard_loss_kernel, 2 * t->C + 2 * t->R + 8, v, z, B, ldz, y, w_xent, w_node, scale,
                      row_loss, gz);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, row_loss, B, loss);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_node_logits_backward(const nbdt_tree* t, const float* gs, int64_t B, float* gz,
                                         void* stream) {
  NBDT_REQUIRE(t != nullptr, "null tree handle");
  NBDT_REQUIRE(B >= 0, "bad batch");
  if (B == 0) return NBDT_OK;
  NBDT_REQUIRE(gs && gz, "null gradient buffer");
  TreeView v = view_of(t);
  const int64_t n = B * t->C;
  hipLaunchKernelGGL(node_logits_bwd_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, v, gs, B, gz);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_hard_forward(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz,
                                 int64_t* pred, float* onehot, int32_t* path_node, int32_t* path_child,
                                 float* path_prob, float* path_entropy, void* stream) {
  int rc = check_common(t, z, ztype, B, ldz);
  if (rc) return rc;
  if (B == 0) return NBDT_OK;
  NBDT_REQUIRE(pred != nullptr, "null pred");
  const bool any = path_node || path_child || path_prob || path_entropy;
  NBDT_REQUIRE(!any || (path_node && path_child && path_prob && path_entropy),
               "decision buffers must be all set or all NULL");
  if (B == 0) return NBDT_OK;
  TreeView v = view_of(t);
  NBDT_DISPATCH_RULES(hard_fwd_kernel, t->C + t->R + 8, v, z, B, ldz, pred, onehot, path_node, path_child,
                      path_prob, path_entropy);
  return NBDT_OK;
}

extern "C" int nbdt_node_outputs(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz,
                                 float* logits, float* probs, int64_t* preds, float* entropy, void* stream) {
  int rc = check_common(t, z, ztype, B, ldz);
  if (rc) return rc;
  if (B == 0) return NBDT_OK;
  TreeView v = view_of(t);
  NBDT_DISPATCH_RULES(node_outputs_kernel, t->C + 2 * t->R, v, z, B, ldz, logits, probs, preds, entropy);
  return NBDT_OK;
}
