// voxe_refine.hip -- whole-grid integer passes of the refinement stage:
//   * graph construction + exact minimum cut on the 6-connected voxel graph
//     (replaces the PyMaxflow python loops of modules/refinement_functions.py:182-298),
//   * 26-connected component labelling + k-largest selection
//     (replaces cc3d.largest_k in edit_pretrained_relu_field.py:384-389,411-416).
//
// Both are HBM / atomic bound integer work on [X,Y,Z] arrays (one thread per voxel, Z fastest => coalesced
// plane accesses); nothing here is GEMM shaped.  Results are exact (integer capacities, integer labels), so
// the oracle parity is bit for bit.
//
// Minimum cut: lock-free push-relabel (one thread owns one voxel; only the owner lowers its excess and its
// outgoing residual capacities, everybody else only raises them through atomics, heights are written by the
// owner only) with exact global relabelling (Bellman-Ford sweeps from the sink seeds) between bursts of
// push/relabel sweeps; the iterative kernels run over a compacted list of the free nodes.  Only the first phase is needed: once no voxel that can still reach a sink seed holds
// excess the preflow is maximum, and "can reach a sink seed in the residual graph" is the sink side of the
// cut -- the same set Boykov-Kolmogorov's sink tree spans when PyMaxflow terminates.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "voxe.h"
#include "voxe_launch.hpp"

namespace voxe {
namespace {

constexpr int kInf = 0x3fffffff;
constexpr int kThreads = 256;

struct Dims {
  int X, Y, Z;
  int sx, sy;  // linear strides of x and y (z stride is 1)
  int N;
};

__host__ __device__ inline Dims make_dims(int X, int Y, int Z) {
  Dims g;
  g.X = X, g.Y = Y, g.Z = Z;
  g.sx = Y * Z, g.sy = Z;
  g.N = X * Y * Z;
  return g;
}

struct Vox {
  int x, y, z;
};

__device__ __forceinline__ Vox decode(const Dims& g, int v) {
  Vox p;
  p.x = v / g.sx;
  const int r = v - p.x * g.sx;
  p.y = r / g.sy;
  p.z = r - p.y * g.sy;
  return p;
}

// neighbour of v in direction d (VOXE_DIR_*), -1 outside the grid
__device__ __forceinline__ int neighbour(const Dims& g, int v, const Vox& p, int d) {
  switch (d) {
    case VOXE_DIR_XP: return p.x + 1 < g.X ? v + g.sx : -1;
    case VOXE_DIR_XM: return p.x > 0 ? v - g.sx : -1;
    case VOXE_DIR_YP: return p.y + 1 < g.Y ? v + g.sy : -1;
    case VOXE_DIR_YM: return p.y > 0 ? v - g.sy : -1;
    case VOXE_DIR_ZP: return p.z + 1 < g.Z ? v + 1 : -1;
    default: return p.z > 0 ? v - 1 : -1;
  }
}

template <typename T>
__device__ __forceinline__ T ld(const T* p) {
  return __atomic_load_n(p, __ATOMIC_RELAXED);
}
template <typename T>
__device__ __forceinline__ void st(T* p, T v) {
  __atomic_store_n(p, v, __ATOMIC_RELAXED);
}

// ------------------------------------------------------------------------------------------------
// graph construction
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_node(const Dims& g, const float* __restrict__ dens, int v, const Vox& p,
                                        int dilate_yz) {
  if (!dilate_yz) return dens[v] > 0.0f;
  bool any = false;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz) {
      const int y = p.y + dy, z = p.z + dz;
      if (y >= 0 && y < g.Y && z >= 0 && z < g.Z) any |= dens[v + dy * g.sy + dz] > 0.0f;
    }
  return any;
}

__global__ __launch_bounds__(kThreads) void graph_build_kernel(Dims g, int F, float sigma, int dilate_yz,
                                                               const float* __restrict__ dens,
                                                               const float* __restrict__ feat,
                                                               uint8_t* __restrict__ node_mask,
                                                               int32_t* __restrict__ cap) {
  const int v = blockIdx.x * kThreads + threadIdx.x;
  if (v >= g.N) return;
  const Vox p = decode(g, v);
  const bool node = is_node(g, dens, v, p, dilate_yz);
  node_mask[v] = node ? 1 : 0;
  // the reference's bounds test compares every coordinate of the neighbour with each of x, y and z (:264-266),
  // i.e. with min(X, Y, Z): on a non-cubic grid a voxel beyond that limit is never *visited as a neighbour*
  const int lim = min(g.X, min(g.Y, g.Z));
  const bool dense_u = dens[v] > 0.0f && max(p.x, max(p.y, p.z)) < lim;
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    int32_t q = 0;
    const int n = neighbour(g, v, p, d);
    if (node && n >= 0) {
      const Vox pn = decode(g, n);
      if (is_node(g, dens, n, pn, dilate_yz)) {
        const bool dense_n = dens[n] > 0.0f && max(pn.x, max(pn.y, pn.z)) < lim;
        const int mult = (dense_u ? 1 : 0) + (dense_n ? 1 : 0);
        if (mult) {
          float s = 0.0f;  // sqrt(((a - b) ** 2).sum()), refinement_functions.py:281
          for (int c = 0; c < F; ++c) {
            const float df = feat[(size_t)v * F + c] - feat[(size_t)n * F + c];
            s = s + df * df;
          }
          const float l2 = sqrtf(s);
          const float e = (float)exp(-(double)(l2 / sigma));  // :284 (the `l2_probs * 0.0` term dropped); evaluated
                                                             // in double so that device and oracle round identically
          q = (int32_t)llrint((double)e * (double)VOXE_GRAPH_CAP_ONE) * mult;
        }
      }
    }
    cap[(size_t)d * g.N + v] = q;
  }
}

// ------------------------------------------------------------------------------------------------
// minimum cut
// ------------------------------------------------------------------------------------------------
struct CutState {
  const uint8_t* node;
  const int8_t* term;
  int32_t* cap;     // [6, N] residual capacities
  long long* excess;  // [N]
  int32_t* height;  // [N]
  long long* flow;  // [1] units absorbed by the sink seeds
  int32_t* flags;   // [0] relabel changed, [1] active count, [2] number of free nodes
  int32_t* list;    // [n_free] voxels that are nodes without a t-link (the only ones that relax / push)
  int n_free;
};

// wave-aggregated append of the free nodes to s.list (order is irrelevant to the exact result)
__global__ __launch_bounds__(kThreads) void cut_compact_kernel(Dims g, CutState s) {
  const int v = blockIdx.x * kThreads + threadIdx.x;
  const bool keep = v < g.N && s.node[v] && s.term[v] == 0;
  const unsigned long long b = __ballot(keep);
  if (!b) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == __ffsll((long long)b) - 1) base = atomicAdd(&s.flags[2], __popcll(b));
  base = __shfl(base, __ffsll((long long)b) - 1, 64);
  if (keep) s.list[base + __popcll(b & ((1ull << lane) - 1ull))] = v;
}

__global__ __launch_bounds__(kThreads) void cut_init_kernel(Dims g, CutState s) {
  const int v = blockIdx.x * kThreads + threadIdx.x;
  if (v >= g.N) return;
  s.excess[v] = 0;
  s.height[v] = kInf;
  if (v == 0) {
    *s.flow = 0;
    s.flags[0] = 0;
    s.flags[1] = 0;
    s.flags[2] = 0;
  }
}

// source seeds saturate every outgoing n-link (their supply is infinite)
__global__ __launch_bounds__(kThreads) void cut_saturate_kernel(Dims g, CutState s) {
  const int v = blockIdx.x * kThreads + threadIdx.x;
  if (v >= g.N || !s.node[v] || s.term[v] <= 0) return;
  const Vox p = decode(g, v);
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    const int n = neighbour(g, v, p, d);
    if (n < 0) continue;
    const int32_t c = s.cap[(size_t)d * g.N + v];
    if (c <= 0 || !s.node[n] || s.term[n] > 0) continue;
    s.cap[(size_t)d * g.N + v] = 0;
    atomicAdd(&s.cap[(size_t)(d ^ 1) * g.N + n], c);
    if (s.term[n] < 0)
      atomicAdd((unsigned long long*)s.flow, (unsigned long long)c);
    else
      atomicAdd((unsigned long long*)&s.excess[n], (unsigned long long)c);
  }
}

__global__ __launch_bounds__(kThreads) void relabel_init_kernel(Dims g, CutState s) {
  const int v = blockIdx.x * kThreads + threadIdx.x;
  if (v >= g.N) return;
  s.height[v] = (s.node[v] && s.term[v] < 0) ? 0 : kInf;
}

// one Bellman-Ford relaxation sweep of "distance to a sink seed over residual edges" (in place, so a single
// launch propagates many levels along the thread order); the fixed point is the exact BFS distance
__global__ __launch_bounds__(kThreads) void relabel_sweep_kernel(Dims g, CutState s, int inner) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= s.n_free) return;
  const int v = s.list[i];
  const Vox p = decode(g, v);
  int nb[6];
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    const int n = neighbour(g, v, p, d);
    nb[d] = (n >= 0 && s.node[n] && s.cap[(size_t)d * g.N + v] > 0) ? n : -1;
  }
  int h = ld(&s.height[v]);
  bool changed = false;
  for (int it = 0; it < inner; ++it) {
    int best = kInf;
#pragma unroll
    for (int d = 0; d < 6; ++d)
      if (nb[d] >= 0) best = min(best, ld(&s.height[nb[d]]));
    if (best + 1 < h) {
      h = best + 1;
      st(&s.height[v], h);
      changed = true;
    }
  }
  if (changed) s.flags[0] = 1;
}

__global__ __launch_bounds__(kThreads) void count_active_kernel(Dims g, CutState s) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  bool active = false;
  if (i < s.n_free) {
    const int v = s.list[i];
    active = s.excess[v] > 0 && s.height[v] < g.N;
  }
  const unsigned long long b = __ballot(active);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&s.flags[1], __popcll(b));
}

__global__ __launch_bounds__(kThreads) void push_relabel_kernel(Dims g, CutState s, int inner) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= s.n_free) return;
  const int v = s.list[i];
  if (ld(&s.excess[v]) <= 0) return;
  const Vox p = decode(g, v);
  int nb[6];
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    const int n = neighbour(g, v, p, d);
    nb[d] = (n >= 0 && s.node[n]) ? n : -1;
  }
  int h = ld(&s.height[v]);
  for (int it = 0; it < inner; ++it) {
    const long long ex = ld(&s.excess[v]);
    if (ex <= 0 || h >= g.N) break;
    int best = kInf, bd = -1;
    int32_t bc = 0;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      if (nb[d] < 0) continue;
      const int32_t c = ld(&s.cap[(size_t)d * g.N + v]);
      if (c <= 0) continue;
      const int hn = ld(&s.height[nb[d]]);
      if (hn < best) best = hn, bd = d, bc = c;
    }
    if (bd < 0) {  // no residual edge leaves this voxel: its excess can never move
      h = kInf;
      st(&s.height[v], h);
      break;
    }
    if (h > best) {
      const long long delta = ex < (long long)bc ? ex : (long long)bc;
      const int n = nb[bd];
      atomicAdd(&s.cap[(size_t)bd * g.N + v], (int32_t)(-delta));
      atomicAdd(&s.cap[(size_t)(bd ^ 1) * g.N + n], (int32_t)delta);
      atomicAdd((unsigned long long*)&s.excess[v], (unsigned long long)(-delta));
      if (s.term[n] < 0)
        atomicAdd((unsigned long long*)s.flow, (unsigned long long)delta);
      else
        atomicAdd((unsigned long long*)&s.excess[n], (unsigned long long)delta);
    } else {
      h = min(best + 1, kInf);
      st(&s.height[v], h);
    }
  }
}

__global__ __launch_bounds__(kThreads) void cut_finalize_kernel(Dims g, CutState s, uint8_t* __restrict__ segment) {
  const int v = blockIdx.x * kThreads + threadIdx.x;
  if (v >= g.N) return;
  uint8_t out = 255;
  if (s.node[v]) {
    const int t = s.term[v];
    out = t > 0 ? 0 : (t < 0 ? 1 : (s.height[v] < kInf ? 1 : 0));
  }
  segment[v] = out;
}

// ------------------------------------------------------------------------------------------------
// connected components (26-connectivity): lock-free union-find, roots = smallest linear index
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int uf_find(int* parent, int i) {
  for (;;) {
    const int q = ld(&parent[i]);
    if (q == i) return i;
    const int qq = ld(&parent[q]);
    if (qq != q) atomicMin(&parent[i], qq);  // path halving: qq is an ancestor of i with a smaller index
    i = q;
  }
}

__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
  for (;;) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) {
      const int t = a;
      a = b;
      b = t;
    }
    const int old = atomicMin(&parent[a], b);  // hook the larger root under the smaller one
    if (old == a) return;
    a = old;  // somebody re-parented `a` first: continue from there
  }
}

__global__ __launch_bounds__(kThreads) void cc_init_kernel(Dims g, const uint8_t* __restrict__ mask,
                                                           int* __restrict__ parent, int* __restrict__ count,
                                                           int* __restrict__ ncomp, unsigned long long* sel, int k) {
  const int v = blockIdx.x * kThreads + threadIdx.x;
  if (v < g.N) {
    parent[v] = mask[v] ? v : -1;
    count[v] = 0;
  }
  if (v == 0) *ncomp = 0;
  if (v < k) sel[v] = 0ull;
}

__global__ __launch_bounds__(kThreads) void cc_union_kernel(Dims g, int* __restrict__ parent) {
  const int v = blockIdx.x * kThreads + threadIdx.x;
  if (v >= g.N || ld(&parent[v]) < 0) return;
  const Vox p = decode(g, v);
  // the 13 neighbours that follow v in memory order
  for (int dx = 0; dx <= 1; ++dx)
    for (int dy = (dx ? -1 : 0); dy <= 1; ++dy)
      for (int dz = ((dx || dy) ? -1 : 1); dz <= 1; ++dz) {
        const int x = p.x + dx, y = p.y + dy, z = p.z + dz;
        if (x >= g.X || y < 0 || y >= g.Y || z < 0 || z >= g.Z) continue;
        const int n = v + dx * g.sx + dy * g.sy + dz;
        if (ld(&parent[n]) >= 0) uf_union(parent, v, n);
      }
}

__global__ __launch_bounds__(kThreads) void cc_compress_kernel(Dims g, int* __restrict__ parent,
                                                               int* __restrict__ count, int* __restrict__ ncomp) {
  const int v = blockIdx.x * kThreads + threadIdx.x;
  const bool fg = v < g.N && ld(&parent[v]) >= 0;
  int r = -1;
  if (fg) {
    r = uf_find(parent, v);
    if (r != v) st(&parent[v], r);  // r is an ancestor of v: concurrent finds through v stay valid
    if (r == v) atomicAdd(ncomp, 1);
  }
  // size count, aggregated per wave: the lanes of a wave mostly share one root (a big component funnels
  // hundreds of thousands of adds into ONE address otherwise)
  unsigned long long todo = __ballot(fg);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int rl = __shfl(r, leader, 64);
    const unsigned long long same = __ballot(fg && r == rl) & todo;
    if ((threadIdx.x & 63) == leader) atomicAdd(&count[rl], __popcll(same));
    todo &= ~same;
  }
}

// round j: the not yet selected root with the largest (count, -index)
__global__ __launch_bounds__(kThreads) void cc_select_kernel(Dims g, const int* __restrict__ parent,
                                                             const int* __restrict__ count,
                                                             unsigned long long* __restrict__ sel, int j) {
  const int v = blockIdx.x * kThreads + threadIdx.x;
  unsigned long long key = 0ull;
  if (v < g.N && parent[v] == v && count[v] > 0)
    key = ((unsigned long long)(unsigned)count[v] << 32) | (unsigned long long)(0xffffffffu - (unsigned)v);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(key, off, 64);
    key = o > key ? o : key;
  }
  if ((threadIdx.x & 63) == 0 && key > ld(&sel[j])) atomicMax(&sel[j], key);
}

__global__ void cc_mark_kernel(int* __restrict__ count, const unsigned long long* __restrict__ sel, int j) {
  const unsigned long long key = sel[j];
  if (key) count[0xffffffffu - (unsigned)(key & 0xffffffffull)] = -(j + 1);  // rank j (0 = largest)
}

__global__ __launch_bounds__(kThreads) void cc_relabel_kernel(Dims g, const int* __restrict__ parent,
                                                              const int* __restrict__ count,
                                                              const int* __restrict__ ncomp, int k,
                                                              int32_t* __restrict__ labels) {
  const int v = blockIdx.x * kThreads + threadIdx.x;
  if (v >= g.N) return;
  int out = 0;
  const int r = parent[v];
  if (r >= 0) {
    const int c = count[r];
    if (c < 0) out = min(k, *ncomp) - (-c - 1);
  }
  labels[v] = out;
}

inline int blocks(int n) { return (n + kThreads - 1) / kThreads; }

struct CutScratch {
  long long* excess;
  int32_t* height;
  long long* flow;
  int32_t* flags;
};

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

void launch_graph_build(const float* dens, const float* feat, int X, int Y, int Z, int F, float sigma,
                        int dilate_yz, uint8_t* node_mask, int32_t* cap, hipStream_t stream) {
  const Dims g = make_dims(X, Y, Z);
  graph_build_kernel<<<blocks(g.N), kThreads, 0, stream>>>(g, F, sigma, dilate_yz, dens, feat, node_mask, cap);
}

size_t graphcut_scratch_bytes(int X, int Y, int Z) {
  const size_t n = (size_t)X * Y * Z;
  return align256(n * sizeof(long long)) + 2 * align256(n * sizeof(int32_t)) + 256;
}

// returns hipSuccess or the first failing runtime call
hipError_t run_graphcut(const uint8_t* node_mask, const int8_t* terminal, int32_t* cap, int X, int Y, int Z,
                        uint8_t* segment, int64_t* flow, void* scratch, hipStream_t stream) {
  const Dims g = make_dims(X, Y, Z);
  char* base = (char*)scratch;
  CutState s;
  s.node = node_mask;
  s.term = terminal;
  s.cap = cap;
  s.excess = (long long*)base;
  base += align256((size_t)g.N * sizeof(long long));
  s.height = (int32_t*)base;
  base += align256((size_t)g.N * sizeof(int32_t));
  s.list = (int32_t*)base;
  base += align256((size_t)g.N * sizeof(int32_t));
  s.flow = (long long*)base;
  s.flags = (int32_t*)(base + 64);
  s.n_free = 0;
  const int nb = blocks(g.N);

  int32_t host_flags[2];
  hipError_t err = hipSuccess;
  cut_init_kernel<<<nb, kThreads, 0, stream>>>(g, s);
  cut_compact_kernel<<<nb, kThreads, 0, stream>>>(g, s);
  cut_saturate_kernel<<<nb, kThreads, 0, stream>>>(g, s);
  if ((err = hipMemcpyAsync(&s.n_free, s.flags + 2, sizeof(int32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess)
    return err;
  if ((err = hipStreamSynchronize(stream)) != hipSuccess) return err;
  const int nf = blocks(s.n_free > 0 ? s.n_free : 1);  // the iterative kernels run over the free nodes only
  int kRelabelBatch = 2, kRelabelInner = 16, push_sweeps = 16, push_inner = 32, push_max = 64;  // swept on hardware
  if (const char* e = getenv("VOXE_CUT_PARAMS"))  // tuning hook: "relabel_batch,relabel_inner,push_sweeps,push_inner,push_max"
    sscanf(e, "%d,%d,%d,%d,%d", &kRelabelBatch, &kRelabelInner, &push_sweeps, &push_inner, &push_max);
  long rounds = 0, relabel_launches = 0, push_launches = 0;
  for (;;) {
    if (++rounds > 100000) return hipErrorLaunchFailure;  // never observed; push-relabel terminates, this only bounds a bug
    // exact distances to the sink seeds over the residual graph
    relabel_init_kernel<<<nb, kThreads, 0, stream>>>(g, s);
    for (;;) {
      if ((err = hipMemsetAsync(s.flags, 0, sizeof(int32_t), stream)) != hipSuccess) return err;
      for (int i = 0; i < kRelabelBatch; ++i)
        relabel_sweep_kernel<<<nf, kThreads, 0, stream>>>(g, s, kRelabelInner);
      relabel_launches += kRelabelBatch;
      if ((err = hipMemcpyAsync(host_flags, s.flags, sizeof(int32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess)
        return err;
      if ((err = hipStreamSynchronize(stream)) != hipSuccess) return err;
      if (!host_flags[0]) break;
    }
    if ((err = hipMemsetAsync(s.flags + 1, 0, sizeof(int32_t), stream)) != hipSuccess) return err;
    count_active_kernel<<<nf, kThreads, 0, stream>>>(g, s);
    if ((err = hipMemcpyAsync(host_flags + 1, s.flags + 1, sizeof(int32_t), hipMemcpyDeviceToHost, stream)) !=
        hipSuccess)
      return err;
    if ((err = hipStreamSynchronize(stream)) != hipSuccess) return err;
    if (host_flags[1] == 0) break;
    for (int i = 0; i < push_sweeps; ++i) push_relabel_kernel<<<nf, kThreads, 0, stream>>>(g, s, push_inner);
    push_launches += push_sweeps;
    if (getenv("VOXE_REFINE_VERBOSE"))
      fprintf(stderr, "[voxe_graphcut] round %ld: %d active voxels, %ld relabel / %ld push launches so far\n", rounds,
              host_flags[1], relabel_launches, push_launches);
    if (push_sweeps < push_max) push_sweeps *= 2;  // the tail moves little flow per relabel: lengthen the bursts
  }
  cut_finalize_kernel<<<nb, kThreads, 0, stream>>>(g, s, segment);
  if ((err = hipMemcpyAsync(flow, s.flow, sizeof(int64_t), hipMemcpyDeviceToDevice, stream)) != hipSuccess) return err;
  return hipStreamSynchronize(stream);
}

size_t cc_scratch_bytes(int X, int Y, int Z, int k) {
  const size_t n = (size_t)X * Y * Z;
  return 2 * align256(n * sizeof(int32_t)) + align256((size_t)(k > 0 ? k : 1) * sizeof(unsigned long long)) + 256;
}

void launch_cc_largest_k(const uint8_t* mask, int X, int Y, int Z, int k, int32_t* labels, int32_t* ncomp,
                         void* scratch, hipStream_t stream) {
  const Dims g = make_dims(X, Y, Z);
  char* base = (char*)scratch;
  int* parent = (int*)base;
  base += align256((size_t)g.N * sizeof(int32_t));
  int* count = (int*)base;
  base += align256((size_t)g.N * sizeof(int32_t));
  unsigned long long* sel = (unsigned long long*)base;
  const int nb = blocks(g.N > k ? g.N : k);
  cc_init_kernel<<<nb, kThreads, 0, stream>>>(g, mask, parent, count, ncomp, sel, k);
  cc_union_kernel<<<nb, kThreads, 0, stream>>>(g, parent);
  cc_compress_kernel<<<nb, kThreads, 0, stream>>>(g, parent, count, ncomp);
  for (int j = 0; j < k; ++j) {
    cc_select_kernel<<<nb, kThreads, 0, stream>>>(g, parent, count, sel, j);
    cc_mark_kernel<<<1, 1, 0, stream>>>(count, sel, j);
  }
  cc_relabel_kernel<<<nb, kThreads, 0, stream>>>(g, parent, count, ncomp, k, labels);
}

}  // namespace voxe
