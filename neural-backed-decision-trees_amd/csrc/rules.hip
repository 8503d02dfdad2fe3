// NBDT decision-rule layer + SoftTreeSupLoss on gfx950.
//
// Replaces the reference's per-node Python loops over stock aten ops
//   get_node_logits ........ nbdt/model.py:83-99     (K gathers + means per inner node)
//   get_all_node_outputs ... nbdt/model.py:101-120   (argmax / softmax / entropy per node)
//   Soft traverse_tree ..... nbdt/model.py:207-242   (class_probs[:, old] *= probs[:, new])
//   Hard traverse_tree ..... nbdt/model.py:145-192   (per-sample python walk, D2H per node)
//   SoftTreeSupLoss ........ nbdt/loss.py:191-203, 260-266  (+ autograd backward)
//   HardTreeSupLoss ........ nbdt/loss.py:212-257, model.py:127-143 (+ autograd backward)
// by ONE launch per call: a group of TPS lanes (1, 4 or 16 waves, pick_tps) owns one sample; the sample's
// logits, the R child logits and the R child probabilities live in LDS, next to the hierarchy's offset arrays
// (stage_tree).  HBM traffic is the algorithmic minimum (read z once, write P / gz once); the path is latency
// bound, so everything for one call is fused into a single kernel, nothing round-trips through HBM,
// and the work inside a sample is laid out for short dependent chains:
//   - the two CSR maps (slot -> classes, class -> slots) are walked by ordered fp32 chains whose order is the
//     arithmetic contract below.  A chain cannot be split, but its indirection can: Gather copies
//     src[idx[j]] for all j into a staging row with every lane working (index reads issued a phase early),
//     and the chains then fold contiguous LDS (chains / long_chain);
//   - one lane folds one slot, so a wave is busy for its longest slot: the host sorts the slots by length and
//     deals chunks of 64 to the group's waves longest-processing-time first (build_slot_schedule);
//   - hierarchies too deep for the staging row (sum of leaf depths > ~35k) take direct-indexed chains.
// profiles/r02_rules*.{jsonl,txt}: (256, 1000) soft forward 67.6 -> 11.3 us, fused loss 141 -> 23 us.
//
// Arithmetic contract (matches oracle/nbdt_oracle.py bit-for-bit on the integer outputs):
// child logit = sequential fp32 sum over ascending class index, one IEEE division by the count;
// argmax = first maximum; path product = 1.0f * p(node_1) * p(node_2) ... in inode order.
// Compiled with -ffp-contract=off and correctly rounded division.
#include "common.h"

#include <algorithm>
#include <vector>

namespace nbdt {
thread_local char g_err[512] = "";
}

using namespace nbdt;

struct nbdt_tree {
  int device;
  int C, N, R, L, root, max_depth;
  int sched_len;   // rows * lanes-per-sample of the slot schedule (see build_slot_schedule)
  int32_t* d_all;  // one allocation
  const int32_t *node_off, *slot_off, *slot_cls, *cls_off, *cls_slot, *slot_next;
  const int32_t* sched;  // [3][sched_len]: slot id (or -1), first and one-past-last element in slot_cls
};

struct TreeView {
  int C, N, R, L, root, max_depth, sched_len;
  int tl_ints;  // LDS ints in front of the sample rows: the offset arrays (stage_tree), set per launch
  int staged;  // the per-sample LDS row ends in an L-float staging area (set per launch, see rules_lds_floats)
  const int32_t *node_off, *slot_off, *slot_cls, *cls_off, *cls_slot, *slot_next, *sched;
};

static TreeView view_of(const nbdt_tree* t) {
  TreeView v;
  v.C = t->C; v.N = t->N; v.R = t->R; v.L = t->L; v.root = t->root; v.max_depth = t->max_depth;
  v.staged = 0; v.tl_ints = 0;
  v.node_off = t->node_off; v.slot_off = t->slot_off; v.slot_cls = t->slot_cls;
  v.cls_off = t->cls_off; v.cls_slot = t->cls_slot; v.slot_next = t->slot_next;
  v.sched = t->sched; v.sched_len = t->sched_len;
  return v;
}

// ------------------------------------------------------------------------------------------
// device helpers

struct LoadF32 { static __device__ __forceinline__ float at(const void* p, int64_t i) { return ((const float*)p)[i]; } };
struct LoadBF16 { static __device__ __forceinline__ float at(const void* p, int64_t i) { return bf16_to_f32(((const bf16_t*)p)[i]); } };
struct LoadF16 { static __device__ __forceinline__ float at(const void* p, int64_t i) { return __half2float(((const __half*)p)[i]); } };

constexpr int kBlock = 256;
// threads per block of the per-sample kernels: a sample's group of TPS lanes is 1, 4 or 16 waves
constexpr int block_of(int tps) { return tps > kBlock ? tps : kBlock; }

#ifndef NBDT_RULES_TIMING
#define NBDT_RULES_TIMING 0   // 1: s_memtime phase stamps in soft_fwd_kernel (scratch/rules_timing.py); outputs are overwritten
#endif

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

// The three offset arrays (and, for the hard walk, slot -> next node) are read by every phase; each read from
// global memory is an L2 round trip in front of a dependent chain, so the block copies them to the front of its
// LDS once, under the latency of the logits load.  The caller's first __syncthreads() publishes them.
typedef const __attribute__((address_space(3))) int32_t* lds_iptr;
struct TreeLds { lds_iptr node_off, slot_off, cls_off, sched, slot_next; };

static int tree_lds_ints(const nbdt_tree* t, bool next) {
  return ((t->N + 1) + (t->R + 1) + (t->C + 1) + 3 * t->sched_len + (next ? t->R : 0) + 3) & ~3;
}

template <bool NEXT>
__device__ __forceinline__ TreeLds stage_tree(const TreeView& t, float* lds) {
  int32_t* d0 = (int32_t*)lds;
  int32_t* d1 = d0 + (t.N + 1);
  int32_t* d2 = d1 + (t.R + 1);
  int32_t* d3 = d2 + (t.C + 1);
  int32_t* d4 = d3 + 3 * t.sched_len;
  for (int i = threadIdx.x; i <= t.N; i += (int)blockDim.x) d0[i] = t.node_off[i];
  for (int i = threadIdx.x; i <= t.R; i += (int)blockDim.x) d1[i] = t.slot_off[i];
  for (int i = threadIdx.x; i <= t.C; i += (int)blockDim.x) d2[i] = t.cls_off[i];
  for (int i = threadIdx.x; i < 3 * t.sched_len; i += (int)blockDim.x) d3[i] = t.sched[i];
  if (NEXT)
    for (int i = threadIdx.x; i < t.R; i += (int)blockDim.x) d4[i] = t.slot_next[i];
  TreeLds o;
  o.node_off = (lds_iptr)d0; o.slot_off = (lds_iptr)d1; o.cls_off = (lds_iptr)d2; o.sched = (lds_iptr)d3;
  o.slot_next = (lds_iptr)d4;
  return o;
}

// The two CSR maps are walked in both directions by ordered fp32 chains (a child's leaves in ascending class
// order, a class's path in root-to-leaf order).  The order is the arithmetic contract, so a chain cannot be
// split; what can be taken off it is the indirection.  Gather spreads `stage[j] = src[idx[j]]` over all
// lanes of the group (coalesced index reads, independent LDS gathers); the chain that follows then walks
// contiguous LDS with its loads running one block of 8 ahead of the adds, which costs ~8 cycles per element
// instead of a dependent global-index + LDS round trip (ImageNet-1000: root children are 525 leaves long).
template <int TPS>
struct Gather {
  // Lane g_tid owns elements g_tid + k*TPS of `stage[j] = src[idx[j]]`.  The index reads are L2 round trips, so
  // issue() puts the first 3*U of them in flight (the ragged last round with clamped addresses, rounds 0 and 1)
  // and is called a phase early; finish() does the LDS gather/scatter, loading round r+2 while round r runs.
  static constexpr int U = TPS > 256 ? 4 : 8, S = U * TPS;  // a lane holds ~10 elements (pick_tps): all in flight
  int ir[U], ia[U], ib[U];
  int nr, left;
  const int32_t* ip;

  __device__ __forceinline__ void issue(const int32_t* __restrict__ idx, int L, bool active, int g_tid) {
    nr = 0;
    left = 0;
    ip = idx + g_tid;
    if (!active) return;
    const int cnt = (L > g_tid) ? (L - g_tid + TPS - 1) / TPS : 0;
    nr = cnt / U;
    left = cnt - nr * U;
    if (left > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) ir[u] = ip[nr * S + min(u, left - 1) * TPS];
    }
    if (nr > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) ia[u] = ip[u * TPS];
    }
    if (nr > 1) {
#pragma unroll
      for (int u = 0; u < U; ++u) ib[u] = ip[S + u * TPS];
    }
  }

  __device__ __forceinline__ void finish(const float* __restrict__ src, float* __restrict__ stage, int g_tid) {
    float* sp = stage + g_tid;
    float x[U];
    int r = 0;
    for (; r + 2 <= nr; r += 2) {
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = src[ia[u]];
#pragma unroll
      for (int u = 0; u < U; ++u) sp[r * S + u * TPS] = x[u];
      const int ra = min(r + 2, nr - 1);  // past the end: reload the last round, never used
#pragma unroll
      for (int u = 0; u < U; ++u) ia[u] = ip[ra * S + u * TPS];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = src[ib[u]];
#pragma unroll
      for (int u = 0; u < U; ++u) sp[(r + 1) * S + u * TPS] = x[u];
      const int rb = min(r + 3, nr - 1);
#pragma unroll
      for (int u = 0; u < U; ++u) ib[u] = ip[rb * S + u * TPS];
    }
    if (r < nr) {
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = src[ia[u]];
#pragma unroll
      for (int u = 0; u < U; ++u) sp[r * S + u * TPS] = x[u];
    }
    if (left > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = src[ir[u]];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (u < left) sp[nr * S + u * TPS] = x[u];
    }
    __syncthreads();
  }
};

struct ChainAdd { static __device__ __forceinline__ float ap(float a, float x) { return a + x; } };
struct ChainMul { static __device__ __forceinline__ float ap(float a, float x) { return a * x; } };

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) f32x4* lds_v4ptr;

template <typename OP>
__device__ __forceinline__ float fold4(float acc, f32x4 x) {
  acc = OP::ap(acc, x.x);
  acc = OP::ap(acc, x.y);
  acc = OP::ap(acc, x.z);
  return OP::ap(acc, x.w);
}

// The leading whole blocks of 16 of one chain of n >= 16 elements, acc = OP(...OP(OP(acc, v[0]), v[1])...): up to
// 3 elements bring the address to 16 bytes, then 32 elements per trip as eight ds_read_b128 issued together
// and folded as they arrive -- one exposed LDS round trip per 32 dependent operations (element-wise reads cost
// one per 8: 17 cycles per element on the 525-leaf root children).  Returns the number of elements folded.
// (A register-rotating version with hand-counted lgkmcnt was faster still but not safe: the compiler copies
// asm outputs at loop edges, before the data has arrived.)
template <typename OP>
__device__ __forceinline__ int long_chain(const float* v, int n, float& acc_io) {
  float acc = acc_io;
  const unsigned a = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)v;
  const int peel = (4 - ((a >> 2) & 3)) & 3;
  {
    const float x0 = v[0], x1 = v[1], x2 = v[2];
    acc = (0 < peel) ? OP::ap(acc, x0) : acc;
    acc = (1 < peel) ? OP::ap(acc, x1) : acc;
    acc = (2 < peel) ? OP::ap(acc, x2) : acc;
  }
  lds_v4ptr p = (lds_v4ptr)(size_t)(a + 4 * peel);
  int rem = n - peel;  // >= 13
  for (; rem >= 32; rem -= 32, p += 8) {
    f32x4 x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = p[u];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = fold4<OP>(acc, x[u]);
  }
  if (rem >= 16) {
    f32x4 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = p[u];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = fold4<OP>(acc, x[u]);
    rem -= 16;
  }
  acc_io = acc;
  return n - rem;
}

// acc[m] = OP(...OP(OP(acc[m], v[b[m]]), v[b[m]+1])..., v[e[m]-1]) in exactly that order, for the M independent
// chains one lane holds (an empty chain, b == e, leaves acc alone).  Chains of 16 or more first go through
// long_chain one after the other (the schedule puts the few there are in the first pass of one wave).  The
// last 0..15 elements of every chain are folded jointly: whole blocks of 4 are read (up to 3 floats past e) and
// folded under a per-element select; a block no lane of the wave needs is skipped, and the M chains share each
// LDS round trip.
template <typename OP, int M>
__device__ __forceinline__ void chains(const float* v, const int (&b)[M], const int (&e)[M], float (&acc)[M]) {
  int j[M], n[M];
  int most = 0;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    j[m] = b[m];
    n[m] = e[m] - b[m];
    if (n[m] >= 16) {
      const int done = long_chain<OP>(v + b[m], n[m], acc[m]);
      j[m] += done;
      n[m] -= done;
    }
    most = max(most, n[m]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (__ballot(most > 4 * q) == 0) break;
    float x[M][4];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int u = 0; u < 4; ++u) x[m][u] = v[j[m] + 4 * q + u];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (M > 1 && __ballot(n[m] > 4 * q) == 0) continue;  // e.g. the rows past a wave's last chunk
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[m] = (4 * q + u < n[m]) ? OP::ap(acc[m], x[m][u]) : acc[m];
    }
  }
}

// chains one lane folds per pass.  A 4-wave group leaves one wave per SIMD, so a lane folds its two slots (or
// classes) together; a 16-wave group has other waves to hide the round trips, and 128 VGPRs per lane.
template <int TPS> constexpr int joint_of() { return TPS == 256 ? 2 : 1; }

// out[s] = ordered sum over the leaves of slot s of src[class] (divided by the leaf count when MEAN)
template <int TPS, bool MEAN>
__device__ __forceinline__ void slot_sums(const TreeView& t, const TreeLds& o, const float* src, float* stage, float* out,
                                          bool active, int g_tid, Gather<TPS>& g, Gather<TPS>& gn,
                                          const int32_t* next_idx) {
  // g's indices were issued a phase ago; the next gather's are issued here, ahead of the chains
  if (t.staged) {
    g.finish(src, stage, g_tid);
    if (next_idx) gn.issue(next_idx, t.L, active, g_tid);
  }
  if (active)
    for (int k = g_tid; k < t.sched_len; k += joint_of<TPS>() * TPS) {
      int s[joint_of<TPS>()], b[joint_of<TPS>()], e[joint_of<TPS>()];
      float acc[joint_of<TPS>()];
#pragma unroll
      for (int m = 0; m < joint_of<TPS>(); ++m) {
        const int km = k + m * TPS;
        const bool in = km < t.sched_len;
        s[m] = in ? o.sched[km] : -1;  // -1 also pads the rows of a wave that has run out of slots
        b[m] = in ? o.sched[t.sched_len + km] : 0;
        e[m] = in ? o.sched[2 * t.sched_len + km] : 0;
        if (s[m] < 0) b[m] = e[m] = 0;
        acc[m] = 0.f;
      }
      if (t.staged) {
        chains<ChainAdd, joint_of<TPS>()>(stage, b, e, acc);
      } else {
#pragma unroll
        for (int m = 0; m < joint_of<TPS>(); ++m)
          for (int j = b[m]; j < e[m]; ++j) acc[m] = acc[m] + src[t.slot_cls[j]];
      }
#pragma unroll
      for (int m = 0; m < joint_of<TPS>(); ++m)
        if (s[m] >= 0) out[s[m]] = MEAN ? acc[m] / (float)(e[m] - b[m]) : acc[m];
    }
  __syncthreads();
}

// phase 0+1: logits row -> LDS, then child logits (nbdt/model.py:94-99)
template <int TPS, typename LD>
__device__ __forceinline__ void load_and_node_logits(const TreeView& t, const TreeLds& o, const void* z, int64_t row_off,
                                                     bool active, int g_tid, float* zs, float* ss,
                                                     float* stage, Gather<TPS>& g, Gather<TPS>& gn,
                                                     const int32_t* next_idx) {
  if (active)
    for (int c = g_tid; c < t.C; c += TPS) zs[c] = LD::at(z, row_off + c);
  __syncthreads();
  slot_sums<TPS, true>(t, o, zs, stage, ss, active, g_tid, g, gn, next_idx);
}

// phase 2: per-node softmax (nbdt/model.py:114)
template <int TPS>
__device__ __forceinline__ void node_softmax(const TreeView& t, const TreeLds& o, bool active, int g_tid, const float* ss,
                                             float* ps) {
  if (active)
    for (int n = g_tid; n < t.N; n += TPS) {
      const int b = o.node_off[n], e = o.node_off[n + 1];
      if (e - b == 2) {  // binary node (every induced hierarchy): same operations, one LDS round trip
        const float s0 = ss[b], s1 = ss[b + 1];
        const float m = fmaxf(s0, s1);
        const float e0 = expf(s0 - m), e1 = expf(s1 - m);
        float sum = 0.f;
        sum = sum + e0;
        sum = sum + e1;
        ps[b] = e0 / sum;
        ps[b + 1] = e1 / sum;
        continue;
      }
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

// per-slot values along every class's path, staged for class_chain (no-op when the row has no stage)
template <int TPS>
__device__ __forceinline__ void stage_paths(const TreeView& t, const float* per_slot, float* stage, int g_tid,
                                            Gather<TPS>& g) {
  if (t.staged) g.finish(per_slot, stage, g_tid);
}

// ordered chains over the slots on the paths of classes c0, c0+TPS, ... (joint_of<TPS>() of them, `init` where the class
// index runs past C): path products (OP = mul, init 1) of the child probabilities (nbdt/model.py:230-240), or
// sums (OP = add, init 0) of per-slot gradient terms
template <int TPS, typename OP>
__device__ __forceinline__ void class_chains(const TreeView& t, const TreeLds& o, int c0, const float* per_slot,
                                             const float* stage, float init, float (&acc)[joint_of<TPS>()]) {
  int b[joint_of<TPS>()], e[joint_of<TPS>()];
#pragma unroll
  for (int m = 0; m < joint_of<TPS>(); ++m) {
    const int c = c0 + m * TPS;
    const bool in = c < t.C;
    b[m] = in ? o.cls_off[c] : 0;
    e[m] = in ? o.cls_off[c + 1] : 0;
    acc[m] = init;
  }
  if (t.staged) {
    chains<OP, joint_of<TPS>()>(stage, b, e, acc);
  } else {
#pragma unroll
    for (int m = 0; m < joint_of<TPS>(); ++m)
      for (int j = b[m]; j < e[m]; ++j) acc[m] = OP::ap(acc[m], per_slot[t.cls_slot[j]]);
  }
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
__global__ __launch_bounds__(block_of(TPS)) void soft_fwd_kernel(TreeView t, const void* z, int64_t B, int64_t ldz,
                                                          float* __restrict__ P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SPB = block_of(TPS) / TPS;
  const int g = threadIdx.x / TPS, g_tid = threadIdx.x % TPS;
  const int64_t sample = (int64_t)blockIdx.x * SPB + g;
  const bool active = sample < B;
  const int stride = t.C + 2 * t.R + (t.staged ? t.L : 0);
  Gather<TPS> ga, gb;  // ga: slot -> classes (issued here, under the logits load), gb: class -> slots
  if (t.staged) ga.issue(t.slot_cls, t.L, active, g_tid);
  const TreeLds o = stage_tree<false>(t, lds);
  float* zs = lds + t.tl_ints + (size_t)g * stride;
  float* ss = zs + t.C;
  float* ps = ss + t.R;
  float* stage = ps + t.R;
#if NBDT_RULES_TIMING
  // phase stamps of block 0, one row of 8 per wave, written over P[0, :32] (scratch/rules_timing.py)
  unsigned tsv[8];
#define NBDT_RSTAMP(i) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); tsv[i] = (unsigned)t_; }
  NBDT_RSTAMP(0)
  if (active)
    for (int c = g_tid; c < t.C; c += TPS) zs[c] = LD::at(z, sample * ldz + c);
  __syncthreads();
  NBDT_RSTAMP(1)
  if (t.staged) {
    ga.finish(zs, stage, g_tid);
    gb.issue(t.cls_slot, t.L, active, g_tid);
  }
  NBDT_RSTAMP(2)
  {
    if (active)
      for (int k = g_tid; k < t.sched_len; k += joint_of<TPS>() * TPS) {
        int s[joint_of<TPS>()], b[joint_of<TPS>()], e[joint_of<TPS>()];
        float acc[joint_of<TPS>()];
#pragma unroll
        for (int m = 0; m < joint_of<TPS>(); ++m) {
          const int km = k + m * TPS;
          const bool in = km < t.sched_len;
          s[m] = in ? o.sched[km] : -1;
          b[m] = in ? o.sched[t.sched_len + km] : 0;
          e[m] = in ? o.sched[2 * t.sched_len + km] : 0;
          if (s[m] < 0) b[m] = e[m] = 0;
          acc[m] = 0.f;
        }
        if (t.staged) chains<ChainAdd, joint_of<TPS>()>(stage, b, e, acc);
#pragma unroll
        for (int m = 0; m < joint_of<TPS>(); ++m)
          if (s[m] >= 0) ss[s[m]] = acc[m] / (float)(e[m] - b[m]);
      }
    __syncthreads();
  }
  NBDT_RSTAMP(3)
#pragma nounroll
  for (int rep = 0; rep < 2; ++rep) {  // second trip: the same code with a warm instruction cache (tsv[7])
    node_softmax<TPS>(t, o, active, g_tid, ss, ps);
    if (rep == 0) NBDT_RSTAMP(7) else NBDT_RSTAMP(4)
  }
  stage_paths<TPS>(t, ps, stage, g_tid, gb);
  NBDT_RSTAMP(5)
  if (active)
    for (int c0 = g_tid; c0 < t.C; c0 += joint_of<TPS>() * TPS) {
      float p[joint_of<TPS>()];
      class_chains<TPS, ChainMul>(t, o, c0, ps, stage, 1.0f, p);
#pragma unroll
      for (int m = 0; m < joint_of<TPS>(); ++m)
        if (c0 + m * TPS < t.C) P[sample * t.C + c0 + m * TPS] = p[m];
    }
  __syncthreads();
  NBDT_RSTAMP(6)
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)
    for (int i = 0; i < 8; ++i) P[(threadIdx.x >> 6) * 8 + i] = (float)(tsv[i] - tsv[0]);
#else
  load_and_node_logits<TPS, LD>(t, o, z, sample * ldz, active, g_tid, zs, ss, stage, ga, gb, t.cls_slot);
  node_softmax<TPS>(t, o, active, g_tid, ss, ps);
  stage_paths<TPS>(t, ps, stage, g_tid, gb);
  if (active)
    for (int c0 = g_tid; c0 < t.C; c0 += joint_of<TPS>() * TPS) {
      float p[joint_of<TPS>()];
      class_chains<TPS, ChainMul>(t, o, c0, ps, stage, 1.0f, p);
#pragma unroll
      for (int m = 0; m < joint_of<TPS>(); ++m)
        if (c0 + m * TPS < t.C) P[sample * t.C + c0 + m * TPS] = p[m];
    }
#endif
}

// shared tail of the two backward flavours: Pg (= P*g per class) is in `pq`; ss is overwritten by
// G, then by dL/ds already divided by the slot's leaf count (the term every leaf of the slot receives), and
// those terms are staged along the class paths for class_chains<ChainAdd>.
template <int TPS>
__device__ __forceinline__ void tree_backward(const TreeView& t, const TreeLds& o, bool active, int g_tid, float* ss,
                                              const float* ps, const float* pq, float* stage, Gather<TPS>& ga,
                                              Gather<TPS>& gb) {
  slot_sums<TPS, false>(t, o, pq, stage, ss, active, g_tid, ga, gb, t.cls_slot);  // G[slot]
  if (active)
    for (int n = g_tid; n < t.N; n += TPS) {
      const int b = o.node_off[n], e = o.node_off[n + 1];
      float tot = 0.f;
      for (int s = b; s < e; ++s) tot = tot + ss[s];
      for (int s = b; s < e; ++s) {
        const float ds = ss[s] - ps[s] * tot;  // dL/ds
        ss[s] = ds / (float)(o.slot_off[s + 1] - o.slot_off[s]);
      }
    }
  __syncthreads();
  stage_paths<TPS>(t, ss, stage, g_tid, gb);
}

// VJP of the mean over a slot's leaves, from per-slot gradients in global memory (nbdt_node_logits_backward)
__device__ __forceinline__ float class_grad_direct(const TreeView& t, int c, const float* ds) {
  float acc = 0.f;
  const int b = t.cls_off[c], e = t.cls_off[c + 1];
  for (int j = b; j < e; ++j) {
    const int s = t.cls_slot[j];
    acc = acc + ds[s] / (float)(t.slot_off[s + 1] - t.slot_off[s]);
  }
  return acc;
}

template <int TPS, typename LD>
__global__ __launch_bounds__(block_of(TPS)) void soft_bwd_kernel(TreeView t, const void* z, int64_t B, int64_t ldz,
                                                          const float* __restrict__ gP,
                                                          float* __restrict__ gz) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SPB = block_of(TPS) / TPS;
  const int g = threadIdx.x / TPS, g_tid = threadIdx.x % TPS;
  const int64_t sample = (int64_t)blockIdx.x * SPB + g;
  const bool active = sample < B;
  const int stride = 2 * t.C + 2 * t.R + (t.staged ? t.L : 0);
  Gather<TPS> ga, gb;  // ga: slot -> classes (issued here, under the logits load), gb: class -> slots
  if (t.staged) ga.issue(t.slot_cls, t.L, active, g_tid);
  const TreeLds o = stage_tree<false>(t, lds);
  float* zs = lds + t.tl_ints + (size_t)g * stride;
  float* ss = zs + t.C;
  float* ps = ss + t.R;
  float* pq = ps + t.R;
  float* stage = pq + t.C;
  load_and_node_logits<TPS, LD>(t, o, z, sample * ldz, active, g_tid, zs, ss, stage, ga, gb, t.cls_slot);
  node_softmax<TPS>(t, o, active, g_tid, ss, ps);
  stage_paths<TPS>(t, ps, stage, g_tid, gb);
  if (t.staged) ga.issue(t.slot_cls, t.L, active, g_tid);  // for the G sums of tree_backward
  if (active)
    for (int c0 = g_tid; c0 < t.C; c0 += joint_of<TPS>() * TPS) {
      float p[joint_of<TPS>()];
      class_chains<TPS, ChainMul>(t, o, c0, ps, stage, 1.0f, p);
#pragma unroll
      for (int m = 0; m < joint_of<TPS>(); ++m)
        if (c0 + m * TPS < t.C) pq[c0 + m * TPS] = p[m] * gP[sample * t.C + c0 + m * TPS];
    }
  __syncthreads();
  tree_backward<TPS>(t, o, active, g_tid, ss, ps, pq, stage, ga, gb);
  if (active)
    for (int c0 = g_tid; c0 < t.C; c0 += joint_of<TPS>() * TPS) {
      float d[joint_of<TPS>()];
      class_chains<TPS, ChainAdd>(t, o, c0, ss, stage, 0.f, d);
#pragma unroll
      for (int m = 0; m < joint_of<TPS>(); ++m)
        if (c0 + m * TPS < t.C) gz[sample * t.C + c0 + m * TPS] = d[m];
    }
}

// SoftTreeSupLoss forward+backward for criterion = nn.CrossEntropyLoss() (mean reduction):
//   row = w_x*(lse(z) - z[y]) + w_t*(lse(P) - P[y])     (P fed to CE as if logits, loss.py:266)
//   gz  = scale*( w_x*(softmax(z) - 1[y]) + J^T * w_t*(softmax(P) - 1[y]) )
// Everything after the sample's logits are in LDS (zs, published by a __syncthreads()) -- shared by soft_loss_kernel,
// which loads them from HBM, and head_soft_loss_kernel, which computes them from the pooled features.  On return
// gx[c] holds dL/dz[c] for every class (each written by the lane that owns c; no trailing barrier); gz != nullptr
// also stores it to gz[sample][c].
template <int TPS>
__device__ __forceinline__ void soft_loss_from_lds_logits(const TreeView& t, const TreeLds& o, bool active, int g_tid,
                                                          int64_t sample, const int64_t* __restrict__ y, float w_x,
                                                          float w_t, float scale, float* __restrict__ row_loss,
                                                          float* zs, float* ss, float* ps, float* pq, float* gx,
                                                          float* red, float* stage, Gather<TPS>& ga, Gather<TPS>& gb,
                                                          float* __restrict__ gz) {
  slot_sums<TPS, true>(t, o, zs, stage, ss, active, g_tid, ga, gb, t.cls_slot);
  node_softmax<TPS>(t, o, active, g_tid, ss, ps);
  stage_paths<TPS>(t, ps, stage, g_tid, gb);
  if (t.staged) ga.issue(t.slot_cls, t.L, active, g_tid);  // for the G sums of tree_backward

  float mz = -INFINITY, mp = -INFINITY;
  if (active)
    for (int c0 = g_tid; c0 < t.C; c0 += joint_of<TPS>() * TPS) {
      float p[joint_of<TPS>()];
      class_chains<TPS, ChainMul>(t, o, c0, ps, stage, 1.0f, p);
#pragma unroll
      for (int m = 0; m < joint_of<TPS>(); ++m) {
        const int c = c0 + m * TPS;
        if (c < t.C) {
          pq[c] = p[m];
          mp = fmaxf(mp, p[m]);
          mz = fmaxf(mz, zs[c]);
        }
      }
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
  tree_backward<TPS>(t, o, active, g_tid, ss, ps, pq, stage, ga, gb);
  if (active)
    for (int c0 = g_tid; c0 < t.C; c0 += joint_of<TPS>() * TPS) {
      float d[joint_of<TPS>()];
      class_chains<TPS, ChainAdd>(t, o, c0, ss, stage, 0.f, d);
#pragma unroll
      for (int m = 0; m < joint_of<TPS>(); ++m) {
        const int c = c0 + m * TPS;
        if (c < t.C) {
          const float v = gx[c] + d[m];
          gx[c] = v;
          if (gz) gz[sample * t.C + c] = v;
        }
      }
    }
}

template <int TPS, typename LD>
__global__ __launch_bounds__(block_of(TPS)) void soft_loss_kernel(TreeView t, const void* z, int64_t B, int64_t ldz,
                                                           const int64_t* __restrict__ y, float w_x,
                                                           float w_t, float scale,
                                                           float* __restrict__ row_loss,
                                                           float* __restrict__ gz) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SPB = block_of(TPS) / TPS;
  const int g = threadIdx.x / TPS, g_tid = threadIdx.x % TPS;
  const int64_t sample = (int64_t)blockIdx.x * SPB + g;
  const bool active = sample < B;
  const int stride = 3 * t.C + 2 * t.R + 16 + (t.staged ? t.L : 0);
  Gather<TPS> ga, gb;  // ga: slot -> classes (issued here, under the logits load), gb: class -> slots
  if (t.staged) ga.issue(t.slot_cls, t.L, active, g_tid);
  const TreeLds o = stage_tree<false>(t, lds);
  float* zs = lds + t.tl_ints + (size_t)g * stride;
  float* ss = zs + t.C;
  float* ps = ss + t.R;
  float* pq = ps + t.R;
  float* gx = pq + t.C;
  float* red = gx + t.C;
  float* stage = red + 16;
  if (active)
    for (int c = g_tid; c < t.C; c += TPS) zs[c] = LD::at(z, sample * ldz + c);
  __syncthreads();
  soft_loss_from_lds_logits<TPS>(t, o, active, g_tid, sample, y, w_x, w_t, scale, row_loss, zs, ss, ps, pq, gx, red,
                                 stage, ga, gb, gz);
}

// ---- N1: the classifier head and the tree loss in ONE launch (north star: "the node-embedding inner product and
// per-node softmax / path-product as a fused wavefront-reduction kernel with coalesced HBM reads of the (batch x
// feature) x (nodes x feature) matrices").  Replaces, for a classifier of at most 512 classes,
//     nbdt_linear_fwd -> nbdt_soft_tree_loss -> nbdt_linear_bwd      (nn.Linear, reference nbdt/models/resnet.py:126,148 /
//                                                                    pytorchcv `output`; nbdt/loss.py:191-203, 264-266)
// The logits never touch HBM (unless z_out asks for them).  Per sample (a group of TPS lanes, as in every kernel of
// this file):  pooled row -> LDS;  z[c] = sum_k x[k] W[c][k] + b[c] with ONE WAVE PER CLASS -- lanes stride the
// feature axis (coalesced 256-byte reads of a W row), fused multiply-adds, xor-butterfly -- i.e. the arithmetic of
// linear_fwd_kernel (misc.hip), so the logits are bit-identical to the unfused path's for heads below 64 classes;
// then the ordered-chain rules + loss above on those logits in LDS (node logits / decisions bit-exact against the
// oracle on the logits z_out reports);  then the head's backward from the same block:
//     dpooled[b][k] = sum_c gz[c] W[c][k]              (ascending c, fused multiply-adds: linear_bwd_x_kernel's order)
//     dW[c][k] += sum over the block's SPB samples of gz[c] x[k],  db[c] += sum of gz[c]   (one add per block and
//     address; deterministic mode: a zeroed row per block, summed in block order by det_fold).
#ifndef NBDT_HEAD_SKIP
#define NBDT_HEAD_SKIP 0     // timing-only builds: bit 0 drops the forward, 1 the dpooled, 2 the dW / db phase
#endif
template <int TPS, int SPB>
__global__ __launch_bounds__(TPS * SPB) void head_soft_loss_kernel(TreeView t, const float* __restrict__ pooled,
                                                                   const float* __restrict__ W,
                                                                   const float* __restrict__ bias, int K, int64_t B,
                                                                   const int64_t* __restrict__ y, float w_x, float w_t,
                                                                   float scale, float* __restrict__ row_loss,
                                                                   float* __restrict__ z_out,
                                                                   float* __restrict__ gpooled, float* gW, float* gb,
                                                                   long long row_stride_w, long long row_stride_b) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int g = threadIdx.x / TPS, g_tid = threadIdx.x % TPS;
  const int64_t sample = (int64_t)blockIdx.x * SPB + g;
  const bool active = sample < B;
  const int stride = K + 3 * t.C + 2 * t.R + 16 + (t.staged ? t.L : 0);
  Gather<TPS> ga, gbk;
  if (t.staged) ga.issue(t.slot_cls, t.L, active, g_tid);
  const TreeLds o = stage_tree<false>(t, lds);
  float* rows = lds + t.tl_ints;
  float* xs = rows + (size_t)g * stride;
  float* zs = xs + K;
  float* ss = zs + t.C;
  float* ps = ss + t.R;
  float* pq = ps + t.R;
  float* gx = pq + t.C;
  float* red = gx + t.C;
  float* stage = red + 16;
  if (active)
    for (int k = g_tid; k < K; k += TPS) xs[k] = pooled[sample * K + k];
  __syncthreads();
  const int n_live = (int)((B - (int64_t)blockIdx.x * SPB) < SPB ? (B - (int64_t)blockIdx.x * SPB) : SPB);
#if !(NBDT_HEAD_SKIP & 1)
  {  // classifier forward, the whole block together: wave wv takes classes wv, wv + NWB, ... FOUR at a time, and forms
     // their inner products for ALL the block's samples from one read of the W rows (4 rows x 8 x 64 features in
     // registers, so 32 loads per lane are in flight together; a row used to be re-read per sample, one dependent load
     // at a time).  Every (sample, class) sum keeps linear_fwd_kernel's order: lanes stride the features, fused
     // multiply-adds in ascending k, xor-butterfly.
    constexpr int NWB = TPS * SPB / 64, CU = 4;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int zs_off = (int)(zs - xs);
    for (int c0 = wv; c0 < t.C; c0 += NWB * CU) {
      float s[CU][SPB];
#pragma unroll
      for (int u = 0; u < CU; ++u)
#pragma unroll
        for (int j = 0; j < SPB; ++j) s[u][j] = 0.f;
      for (int k0 = lane; k0 < K; k0 += 512) {
        float w[CU][8];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
          const int c = c0 + u * NWB;
          const float* wr = W + (size_t)(c < t.C ? c : c0) * K;
#pragma unroll
          for (int i = 0; i < 8; ++i) w[u][i] = k0 + i * 64 < K ? wr[k0 + i * 64] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (k0 + i * 64 < K) {
            float x[SPB];
#pragma unroll
            for (int j = 0; j < SPB; ++j) x[j] = rows[(size_t)j * stride + k0 + i * 64];
#pragma unroll
            for (int u = 0; u < CU; ++u)
#pragma unroll
              for (int j = 0; j < SPB; ++j) s[u][j] = fmaf(x[j], w[u][i], s[u][j]);
          }
      }
#pragma unroll
      for (int u = 0; u < CU; ++u)
#pragma unroll
        for (int j = 0; j < SPB; ++j) {
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) s[u][j] += __shfl_xor(s[u][j], off, 64);
        }
      if (lane == 0) {
#pragma unroll
        for (int u = 0; u < CU; ++u) {
          const int c = c0 + u * NWB;
          if (c < t.C) {
            const float bc = bias ? bias[c] : 0.f;
#pragma unroll
            for (int j = 0; j < SPB; ++j)
              if (j < n_live) {
                const float zc = s[u][j] + bc;
                rows[(size_t)j * stride + zs_off + c] = zc;
                if (z_out) z_out[((int64_t)blockIdx.x * SPB + j) * t.C + c] = zc;
              }
          }
        }
      }
    }
  }
#endif
  __syncthreads();
  soft_loss_from_lds_logits<TPS>(t, o, active, g_tid, sample, y, w_x, w_t, scale, row_loss, zs, ss, ps, pq, gx, red,
                                 stage, ga, gbk, nullptr);
  __syncthreads();      // every sample's gx is complete: the block-wide weight-gradient sums read all of them
#if !(NBDT_HEAD_SKIP & 2)
  if (gpooled) {      // thread = feature k for ALL the block's samples: one read of W's column per block, 32 rows in flight;
    const int gx_off = (int)(gx - xs);      // each (sample, k) sum stays one ascending chain of fused multiply-adds
    for (int k = threadIdx.x; k < K; k += TPS * SPB) {
      float s[SPB];
#pragma unroll
      for (int j = 0; j < SPB; ++j) s[j] = 0.f;
      for (int c0 = 0; c0 < t.C; c0 += 32) {
        float w[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) w[i] = c0 + i < t.C ? W[(size_t)(c0 + i) * K + k] : 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i)
          if (c0 + i < t.C) {
#pragma unroll
            for (int j = 0; j < SPB; ++j) s[j] = fmaf(rows[(size_t)j * stride + gx_off + c0 + i], w[i], s[j]);
          }
      }
#pragma unroll
      for (int j = 0; j < SPB; ++j)
        if (j < n_live) gpooled[((int64_t)blockIdx.x * SPB + j) * K + k] = s[j];
    }
  }
#endif
#if !(NBDT_HEAD_SKIP & 4)
  if (gW) {
    const int gx_off = (int)(gx - xs);
    float* dw = gW + (size_t)blockIdx.x * row_stride_w;
    constexpr int NT = TPS * SPB;
    const int dc = NT / K, dk = NT - dc * K;        // idx += NT as (c, k) += (dc, dk) with a carry: no division per element
    int c = threadIdx.x / K, k = threadIdx.x - c * K;
    for (int idx = threadIdx.x; idx < t.C * K; idx += NT) {
      float s = 0.f;
      for (int j = 0; j < n_live; ++j) {
        const float* xr = rows + (size_t)j * stride;
        s = fmaf(xr[gx_off + c], xr[k], s);
      }
      atomicAdd(dw + idx, s);
      c += dc; k += dk;
      if (k >= K) { k -= K; ++c; }
    }
    if (gb) {
      float* db = gb + (size_t)blockIdx.x * row_stride_b;
      for (int c = threadIdx.x; c < t.C; c += TPS * SPB) {
        float s = 0.f;
        for (int j = 0; j < n_live; ++j) s += rows[(size_t)j * stride + gx_off + c];
        atomicAdd(db + c, s);
      }
    }
  }
#endif
}

// HardTreeSupLoss forward+backward for criterion = nn.CrossEntropyLoss() (nbdt/loss.py:212-257):
//   row = w_x*(lse(z) - z[y]) + w_h * sum over inner nodes n with y under n of
//                                     (lse(s[n,:]) - s[n, child_of(n,y)])
//   (w_h folds the reference's pooled-by-child-count means: every (sample,node) term ends up with
//    the same weight tsw/(B*N/2), loss.py:228,250-256, times the scheduled tree weight, :195-203)
// The nodes on the label's path are exactly the class->slot CSR row of y; a class listed under two
// children of one node counts once, for the first child (model.py:135 `cls[0]`).
template <int TPS, typename LD>
__global__ __launch_bounds__(block_of(TPS)) void hard_loss_kernel(TreeView t, const void* z, int64_t B, int64_t ldz,
                                                           const int64_t* __restrict__ y, float w_x,
                                                           float w_h, float scale,
                                                           float* __restrict__ row_loss,
                                                           float* __restrict__ gz) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SPB = block_of(TPS) / TPS;
  const int g = threadIdx.x / TPS, g_tid = threadIdx.x % TPS;
  const int64_t sample = (int64_t)blockIdx.x * SPB + g;
  const bool active = sample < B;
  const int stride = 2 * t.C + 2 * t.R + 16 + (t.staged ? t.L : 0);
  Gather<TPS> ga, gb;  // ga: slot -> classes (issued here, under the logits load), gb: class -> slots
  if (t.staged) ga.issue(t.slot_cls, t.L, active, g_tid);
  const TreeLds o = stage_tree<false>(t, lds);
  float* zs = lds + t.tl_ints + (size_t)g * stride;
  float* ss = zs + t.C;
  float* ds = ss + t.R;
  float* gx = ds + t.R;
  float* red = gx + t.C;
  float* stage = red + 16;
  load_and_node_logits<TPS, LD>(t, o, z, sample * ldz, active, g_tid, zs, ss, stage, ga, gb, t.cls_slot);
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
    const int pb = o.cls_off[yy], pe = o.cls_off[yy + 1];
    for (int j = pb + g_tid; j < pe; j += TPS) {
      const int s = t.cls_slot[j];
      // node owning slot s: last n with node_off[n] <= s
      int lo = 0, hi = t.N - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (o.node_off[mid] <= s) lo = mid; else hi = mid - 1;
      }
      const int b = o.node_off[lo], e = o.node_off[lo + 1];
      if (j > pb && t.cls_slot[j - 1] >= b) continue;  // same node, later child: not the first
      float m = ss[b];
      for (int q = b + 1; q < e; ++q) m = fmaxf(m, ss[q]);
      float sum = 0.f;
      for (int q = b; q < e; ++q) sum = sum + expf(ss[q] - m);
      tree_rows += (logf(sum) + m) - ss[s];
      for (int q = b; q < e; ++q)  // the term each leaf of slot q receives: dL/ds over the slot's leaf count
        ds[q] = (expf(ss[q] - m) / sum - (q == s ? 1.f : 0.f)) * (w_h * scale) /
                (float)(o.slot_off[q + 1] - o.slot_off[q]);
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
  stage_paths<TPS>(t, ds, stage, g_tid, gb);
  if (active)
    for (int c0 = g_tid; c0 < t.C; c0 += joint_of<TPS>() * TPS) {
      float d[joint_of<TPS>()];
      class_chains<TPS, ChainAdd>(t, o, c0, ds, stage, 0.f, d);
#pragma unroll
      for (int m = 0; m < joint_of<TPS>(); ++m)
        if (c0 + m * TPS < t.C) gz[sample * t.C + c0 + m * TPS] = gx[c0 + m * TPS] + d[m];
    }
}

// VJP of the node-logit map (nbdt/model.py:94-99): gz[b,c] = sum over slots s holding c of
// gs[b,s] / |leaves(s)|.  Lets any torch criterion be composed on top of nbdt_node_outputs logits.
__global__ __launch_bounds__(kBlock) void node_logits_bwd_kernel(TreeView t, const float* __restrict__ gs,
                                                                 int64_t B, float* __restrict__ gz) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= B * t.C) return;
  const int64_t b = i / t.C;
  const int c = (int)(i - b * t.C);
  gz[i] = class_grad_direct(t, c, gs + b * t.R);
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
__global__ __launch_bounds__(block_of(TPS)) void hard_fwd_kernel(TreeView t, const void* z, int64_t B, int64_t ldz,
                                                          int64_t* __restrict__ pred,
                                                          float* __restrict__ onehot,
                                                          int32_t* __restrict__ path_node,
                                                          int32_t* __restrict__ path_child,
                                                          float* __restrict__ path_prob,
                                                          float* __restrict__ path_entropy) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SPB = block_of(TPS) / TPS;
  const int g = threadIdx.x / TPS, g_tid = threadIdx.x % TPS;
  const int64_t sample = (int64_t)blockIdx.x * SPB + g;
  const bool active = sample < B;
  const int stride = t.C + t.R + 8 + 2 * t.N + (t.staged ? t.L : 0);
  Gather<TPS> ga, gb;
  if (t.staged) ga.issue(t.slot_cls, t.L, active, g_tid);
  const TreeLds o = stage_tree<true>(t, lds);
  float* zs = lds + t.tl_ints + (size_t)g * stride;
  float* ss = zs + t.C;
  int* res = (int*)(ss + t.R);
  int* best_of = res + 8;        // [N] winning slot of every inner node
  int* next_of = best_of + t.N;  // [N] where that slot leads (inner node, or -(class+1))
  float* stage = (float*)(next_of + t.N);
  load_and_node_logits<TPS, LD>(t, o, z, sample * ldz, active, g_tid, zs, ss, stage, ga, gb, nullptr);
  // every node decides in parallel; the walk below is then one LDS hop per level
  if (active)
    for (int n = g_tid; n < t.N; n += TPS) {
      const int b = o.node_off[n], e = o.node_off[n + 1];
      int best = b;
      for (int s = b + 1; s < e; ++s)
        if (ss[s] > ss[best]) best = s;  // first maximum wins (torch.max, model.py:113)
      best_of[n] = best;
      next_of[n] = o.slot_next[best];
    }
  __syncthreads();
  if (active && g_tid == 0) {
    int n = t.root, cls = -1;
    const bool want = path_node != nullptr;
    int d = 0;
    for (; d < t.max_depth; ++d) {
      const int best = best_of[n];
      if (want) {
        const int b = o.node_off[n], e = o.node_off[n + 1];
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
      const int nx = next_of[n];
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
__global__ __launch_bounds__(block_of(TPS)) void node_outputs_kernel(TreeView t, const void* z, int64_t B, int64_t ldz,
                                                              float* __restrict__ logits,
                                                              float* __restrict__ probs,
                                                              int64_t* __restrict__ preds,
                                                              float* __restrict__ entropy) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SPB = block_of(TPS) / TPS;
  const int g = threadIdx.x / TPS, g_tid = threadIdx.x % TPS;
  const int64_t sample = (int64_t)blockIdx.x * SPB + g;
  const bool active = sample < B;
  const int stride = t.C + 2 * t.R + (t.staged ? t.L : 0);
  Gather<TPS> ga, gb;  // ga: slot -> classes (issued here, under the logits load), gb: class -> slots
  if (t.staged) ga.issue(t.slot_cls, t.L, active, g_tid);
  const TreeLds o = stage_tree<false>(t, lds);
  float* zs = lds + t.tl_ints + (size_t)g * stride;
  float* ss = zs + t.C;
  float* ps = ss + t.R;
  load_and_node_logits<TPS, LD>(t, o, z, sample * ldz, active, g_tid, zs, ss, ps + t.R, ga, gb, nullptr);
  node_softmax<TPS>(t, o, active, g_tid, ss, ps);
  if (!active) return;
  for (int s = g_tid; s < t.R; s += TPS) {
    if (logits) logits[sample * t.R + s] = ss[s];
    if (probs) probs[sample * t.R + s] = ps[s];
  }
  for (int n = g_tid; n < t.N; n += TPS) {
    const int b = o.node_off[n], e = o.node_off[n + 1];
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
extern "C" int nbdt_version(void) { return 108; }     // 108: nbdt_se_param_grad, nbdt_se_gate_bwd without parameter gradients; 107: nbdt_bn_act_se_sums / _se_bwd_apply; 106: nbdt_conv_desc.ksplit is live (was reserved), nbdt_conv_seg_*
extern "C" int nbdt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// lanes per sample: one wave for the smallest hierarchies, 4 waves up to 512 child slots, 16 beyond.  At one
// sample per block every phase is a chain of LDS round trips and dependent ALU work that only more waves of
// the same sample can overlap.
static int pick_tps(int C, int R) { return (C <= 64 && R <= 64) ? 64 : (C <= 512 && R <= 512) ? 256 : 1024; }

// Order in which the lanes of a sample's group take the slots.  One lane folds one slot (the fp32 order is the
// contract), so a wave is busy for the longest slot of each 64 it holds: slots are sorted by length, cut into
// chunks of 64, and the chunks dealt to the group's waves longest-processing-time first, so the wave that gets
// the 500-leaf root children gets little else.  Row k of the result is what the group does in its k-th pass:
// [rows][lanes] slot ids, -1 where a wave has run out; then the same shape of first / one-past-last elements.
static void build_slot_schedule(int R, const int32_t* slot_off, int lanes, std::vector<int32_t>* out, int* len) {
  std::vector<int> order(R);
  for (int s = 0; s < R; ++s) order[s] = s;
  auto length = [&](int s) { return slot_off[s + 1] - slot_off[s]; };
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return length(a) > length(b); });
  const int waves = lanes / 64, chunks = (R + 63) / 64;
  std::vector<std::vector<int>> mine(waves);
  std::vector<long> load(waves, 0);
  for (int c = 0; c < chunks; ++c) {  // chunks come longest first
    const int longest = length(order[c * 64]);
    int w = 0;
    for (int i = 1; i < waves; ++i)
      if (load[i] < load[w]) w = i;
    mine[w].push_back(c);
    load[w] += 64 + 8L * longest;  // a pass costs a fixed part plus the longest chain in it
  }
  size_t rows = 0;
  for (auto& m : mine) rows = std::max(rows, m.size());
  const int n = (int)rows * lanes;
  out->assign((size_t)3 * n, -1);
  for (int w = 0; w < waves; ++w)
    for (size_t k = 0; k < mine[w].size(); ++k)
      for (int l = 0; l < 64; ++l) {
        const int i = mine[w][k] * 64 + l;
        if (i >= R) break;
        const int at = (int)k * lanes + w * 64 + l, s = order[i];
        (*out)[at] = s;
        (*out)[n + at] = slot_off[s];
        (*out)[2 * n + at] = slot_off[s + 1];
      }
  *len = n;
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
  std::vector<int32_t> sched;
  int sched_len = 0;
  build_slot_schedule(R, slot_off, pick_tps(C, R), &sched, &sched_len);
  nbdt_tree* t = new nbdt_tree();
  t->device = device; t->C = C; t->N = N; t->R = R; t->L = L; t->root = root; t->max_depth = max_depth;
  t->sched_len = sched_len;
  const size_t n_ints = (size_t)(N + 1) + (R + 1) + L + (C + 1) + L + R + sched.size();
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
  put(sched.data(), sched.size(), &t->sched);
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

static int pick_tps(const nbdt_tree* t) { return pick_tps(t->C, t->R); }

// One launch.  The per-sample LDS row is `base_floats` plus, when it fits, an L-float staging area for the
// ordered chains (TreeView::staged); hierarchies too deep for that (L grows with the sum of leaf depths) take
// the direct-indexed chains, slower but the same arithmetic.
constexpr size_t kLdsBytes = 160 * 1024;
constexpr size_t kStagePad = 4;  // floats of slack behind the last staging row (chain() reads whole blocks)

template <typename... KA, typename... A>
static int launch_rules(void (*kernel)(KA...), unsigned grid, int block, size_t shmem, hipStream_t st,
                        A... args) {
  if (shmem > 64 * 1024)
    NBDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, st, args...);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

#define NBDT_DISPATCH_RULES(KERNEL, NEXT, BASE_FLOATS, ...)                                                \
  do {                                                                                               \
    const int tps = pick_tps(t);                                                                     \
    const int spb = block_of(tps) / tps;                                                             \
    size_t floats = (size_t)(BASE_FLOATS);                                                           \
    v.tl_ints = tree_lds_ints(t, NEXT);                                                              \
    v.staged = ((size_t)v.tl_ints + (size_t)spb * (floats + t->L) + kStagePad) * sizeof(float) <= kLdsBytes; \
    if (v.staged) floats += t->L;                                                                    \
    const size_t shmem = ((size_t)v.tl_ints + (size_t)spb * floats + kStagePad) * sizeof(float);     \
    NBDT_REQUIRE(shmem <= kLdsBytes, "hierarchy too large for LDS");                                 \
    const unsigned grid = (unsigned)((B + spb - 1) / spb);                                           \
    hipStream_t st = (hipStream_t)stream;                                                            \
    int lrc;                                                                                         \
    const int blk = block_of(tps);                                                                   \
    if (tps == 64) {                                                                                 \
      if (ztype == NBDT_F32) lrc = launch_rules(KERNEL<64, LoadF32>, grid, blk, shmem, st, __VA_ARGS__);   \
      else if (ztype == NBDT_BF16) lrc = launch_rules(KERNEL<64, LoadBF16>, grid, blk, shmem, st, __VA_ARGS__); \
      else lrc = launch_rules(KERNEL<64, LoadF16>, grid, blk, shmem, st, __VA_ARGS__);               \
    } else if (tps == 256) {                                                                         \
      if (ztype == NBDT_F32) lrc = launch_rules(KERNEL<256, LoadF32>, grid, blk, shmem, st, __VA_ARGS__);  \
      else if (ztype == NBDT_BF16) lrc = launch_rules(KERNEL<256, LoadBF16>, grid, blk, shmem, st, __VA_ARGS__); \
      else lrc = launch_rules(KERNEL<256, LoadF16>, grid, blk, shmem, st, __VA_ARGS__);              \
    } else {                                                                                         \
      if (ztype == NBDT_F32) lrc = launch_rules(KERNEL<1024, LoadF32>, grid, blk, shmem, st, __VA_ARGS__); \
      else if (ztype == NBDT_BF16) lrc = launch_rules(KERNEL<1024, LoadBF16>, grid, blk, shmem, st, __VA_ARGS__); \
      else lrc = launch_rules(KERNEL<1024, LoadF16>, grid, blk, shmem, st, __VA_ARGS__);             \
    }                                                                                                \
    if (lrc) return lrc;                                                                             \
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
  NBDT_DISPATCH_RULES(soft_fwd_kernel, false, t->C + 2 * t->R, v, z, B, ldz, P);
  return NBDT_OK;
}

extern "C" int nbdt_soft_backward(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz,
                                  const float* gP, float* gz, void* stream) {
  int rc = check_common(t, z, ztype, B, ldz);
  if (rc) return rc;
  if (B == 0) return NBDT_OK;
  NBDT_REQUIRE(gP != nullptr && gz != nullptr, "null gradient buffer");
  TreeView v = view_of(t);
  NBDT_DISPATCH_RULES(soft_bwd_kernel, false, 2 * t->C + 2 * t->R, v, z, B, ldz, gP, gz);
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
  NBDT_DISPATCH_RULES(soft_loss_kernel, false, 3 * t->C + 2 * t->R + 16, v, z, B, ldz, y, w_xent, w_tree, scale,
                      row_loss, gz);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, row_loss, B, loss);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

template <int TPS, int SPB>
static int launch_head(const nbdt_tree* t, TreeView v, const float* pooled, const float* W, const float* bias, int K,
                       int64_t B, const int64_t* y, float w_x, float w_t, float scale, float* row_loss, float* z_out,
                       float* gpooled, float* gW, float* gb, hipStream_t st) {
  size_t floats = (size_t)K + 3 * (size_t)t->C + 2 * (size_t)t->R + 16;
  v.tl_ints = tree_lds_ints(t, false);
  v.staged = ((size_t)v.tl_ints + (size_t)SPB * (floats + t->L) + kStagePad) * sizeof(float) <= kLdsBytes;
  if (v.staged) floats += t->L;
  const size_t shmem = ((size_t)v.tl_ints + (size_t)SPB * floats + kStagePad) * sizeof(float);
  if (shmem > kLdsBytes) return 1;     // caller tries fewer samples per block
  const unsigned grid = (unsigned)((B + SPB - 1) / SPB);
  float *dw = gW, *db = gb, *rows = nullptr;
  long long sw = 0, sb = 0;
  const size_t nw = (size_t)t->C * K, nb = (size_t)t->C;
  if (gW && deterministic()) {       // a zeroed row per block for dW and db, folded in block order afterwards
    rows = det_rows(st, (size_t)grid * (nw + nb));
    if (!rows) return nbdt::fail(NBDT_ENOMEM, "deterministic mode: %s (%s)", "no workspace for the per-block rows", nbdt::det_rows_why());
    NBDT_HIP_CHECK(hipMemsetAsync(rows, 0, (size_t)grid * (nw + nb) * sizeof(float), st));
    dw = rows; db = gb ? rows + (size_t)grid * nw : nullptr;
    sw = (long long)nw; sb = (long long)nb;
  }
  int rc = launch_rules(head_soft_loss_kernel<TPS, SPB>, grid, TPS * SPB, shmem, st, v, pooled, W, bias, K, B, y, w_x,
                        w_t, scale, row_loss, z_out, gpooled, dw, db, sw, sb);
  if (rc) return rc;
  if (rows) {
    rc = det_fold(st, rows, (int)grid, nw, gW);
    if (rc) return rc;
    if (gb) rc = det_fold(st, rows + (size_t)grid * nw, (int)grid, nb, gb);
  }
  return rc;
}

extern "C" int nbdt_head_soft_tree_loss(const nbdt_tree* t, const float* pooled, const float* W, const float* bias,
                                        int64_t B, int32_t K, const int64_t* y, float w_xent, float w_tree,
                                        float grad_scale, float* row_loss, float* loss, float* z_out, float* gpooled,
                                        float* gW, float* gb, void* stream) {
  NBDT_REQUIRE(t != nullptr, "null tree handle");
  NBDT_REQUIRE(pooled && W && y && row_loss && loss, "null buffer");
  NBDT_REQUIRE(B > 0, "empty batch has no mean loss");
  NBDT_REQUIRE(K > 0 && K <= 4096, "feature width must be 1..4096");
  NBDT_REQUIRE(gW != nullptr || gb == nullptr, "db without dW is not supported");
  const int tps = pick_tps(t);
  NBDT_REQUIRE(tps <= 256, "classifier too wide for the fused head (more than 512 classes or child slots): use "
                           "nbdt_linear_fwd + nbdt_soft_tree_loss + nbdt_linear_bwd");
  TreeView v = view_of(t);
  const float scale = grad_scale / (float)B;
  hipStream_t st = (hipStream_t)stream;
  int rc = 1;
#define NBDT_HEAD(TPS_, SPB_)                                                                                      \
  if (rc == 1) rc = launch_head<TPS_, SPB_>(t, v, pooled, W, bias, K, B, y, w_xent, w_tree, scale, row_loss, z_out, \
                                            gpooled, gW, gb, st)
  // samples per block: 8 for one-wave groups (16 / 8 / 4 measure the same inside the training step, 8 is the fastest
  // alone: profiles/r03_head.txt), fewer if the hierarchy's LDS rows do not fit
  if (tps == 64) { NBDT_HEAD(64, 8); NBDT_HEAD(64, 4); NBDT_HEAD(64, 1); }
#if defined(NBDT_HEAD_SPB) && NBDT_HEAD_SPB == 2
  else { NBDT_HEAD(256, 2); NBDT_HEAD(256, 1); }
#elif defined(NBDT_HEAD_SPB) && NBDT_HEAD_SPB == 1
  else { NBDT_HEAD(256, 1); }
#else
  else { NBDT_HEAD(256, 4); NBDT_HEAD(256, 2); NBDT_HEAD(256, 1); }
#endif
#undef NBDT_HEAD
  NBDT_REQUIRE(rc != 1, "hierarchy + feature row too large for LDS");
  if (rc) return rc;
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(kBlock), 0, st, row_loss, B, loss);
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
  NBDT_DISPATCH_RULES(hard_loss_kernel, false, 2 * t->C + 2 * t->R + 16, v, z, B, ldz, y, w_xent, w_node, scale,
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
  NBDT_DISPATCH_RULES(hard_fwd_kernel, true, t->C + t->R + 8 + 2 * t->N, v, z, B, ldz, pred, onehot, path_node, path_child,
                      path_prob, path_entropy);
  return NBDT_OK;
}

extern "C" int nbdt_node_outputs(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz,
                                 float* logits, float* probs, int64_t* preds, float* entropy, void* stream) {
  int rc = check_common(t, z, ztype, B, ldz);
  if (rc) return rc;
  if (B == 0) return NBDT_OK;
  TreeView v = view_of(t);
  NBDT_DISPATCH_RULES(node_outputs_kernel, false, t->C + 2 * t->R, v, z, B, ldz, logits, probs, preds, entropy);
  return NBDT_OK;
}
