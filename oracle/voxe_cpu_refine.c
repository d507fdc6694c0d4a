/*
 * voxe_cpu_refine.c -- CPU oracle of the refinement-stage grid passes (part of oracle/libvoxe_oracle.so).
 *
 * TEST INFRASTRUCTURE ONLY (see voxe_cpu.c): the product path never links or calls this file.
 *
 * Restates, with deliberately different algorithms from the HIP kernels so that agreement means something:
 *   voxe_cpu_graph_build   graph construction of build_graph  (modules/refinement_functions.py:182-287)
 *                          PINNED: tests/golden/refine_graph.npz holds the node / t-link / n-link lists the
 *                          reference's build_graph emits on small grids (recorded through a `maxflow` module
 *                          stub that only logs add_nodes / add_tedge / add_edge calls; tools/gen_golden.py).
 *   voxe_cpu_graphcut      g.maxflow() + get_segment (:289-294) -- Dinic's algorithm + reverse residual BFS.
 *                          PyMaxflow (third party, `PyMaxflow` in the reference's requirements, Boykov-Kolmogorov
 *                          v3.01) is absent from this image: parity with it is UNPINNED; the labels are pinned
 *                          instead against scipy.sparse.csgraph.maximum_flow and brute-force enumeration of all
 *                          cuts on small graphs (tests/test_oracle_refine.py), using the published semantics of
 *                          Graph::what_segment (SINK iff the node is in the sink tree, default SOURCE).
 *   voxe_cpu_cc_largest_k  cc3d.largest_k(connectivity=26) (edit_pretrained_relu_field.py:384-389) -- flood fill.
 *                          connected-components-3d is absent from this image: UNPINNED against it, pinned against
 *                          scipy.ndimage.label with a full 3x3x3 structuring element.
 */
#include "../include/voxe.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int X, Y, Z, sx, sy, N;
} Dims;

static Dims make_dims(int X, int Y, int Z) {
  Dims g = {X, Y, Z, Y * Z, Z, X * Y * Z};
  return g;
}

static int nbr(const Dims* g, int v, int d) {
  const int x = v / g->sx, r = v - x * g->sx, y = r / g->sy, z = r - y * g->sy;
  switch (d) {
    case VOXE_DIR_XP: return x + 1 < g->X ? v + g->sx : -1;
    case VOXE_DIR_XM: return x > 0 ? v - g->sx : -1;
    case VOXE_DIR_YP: return y + 1 < g->Y ? v + g->sy : -1;
    case VOXE_DIR_YM: return y > 0 ? v - g->sy : -1;
    case VOXE_DIR_ZP: return z + 1 < g->Z ? v + 1 : -1;
    default: return z > 0 ? v - 1 : -1;
  }
}

static int dims_ok(int X, int Y, int Z) { return X > 0 && Y > 0 && Z > 0 && (long long)X * Y * Z < (1ll << 30); }

/* ---------------------------------------------------------------------------------------------- */
int voxe_cpu_graph_build(const float* dens, const float* feat, int32_t X, int32_t Y, int32_t Z, int32_t F,
                         float sigma, int32_t dilate_yz, uint8_t* node_mask, int32_t* cap) {
  if (!dens || !feat || !node_mask || !cap) return VOXE_ERR_NULL_POINTER;
  if (!dims_ok(X, Y, Z) || F <= 0 || !(sigma > 0.0f)) return VOXE_ERR_BAD_SHAPE;
  const Dims g = make_dims(X, Y, Z);
  /* nodes: refinement_functions.py:186,200 (MaxPool3d over the Y-Z plane of a [X,Y,Z,1] tensor) or :194 */
  for (int x = 0; x < X; ++x)
    for (int y = 0; y < Y; ++y)
      for (int z = 0; z < Z; ++z) {
        const int v = x * g.sx + y * g.sy + z;
        int any = 0;
        if (dilate_yz) {
          for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
              const int yy = y + dy, zz = z + dz;
              if (yy >= 0 && yy < Y && zz >= 0 && zz < Z && dens[x * g.sx + yy * g.sy + zz] > 0.0f) any = 1;
            }
        } else {
          any = dens[v] > 0.0f;
        }
        node_mask[v] = (uint8_t)any;
      }
  memset(cap, 0, (size_t)6 * g.N * sizeof(int32_t));
  const int lim = X < Y ? (X < Z ? X : Z) : (Y < Z ? Y : Z);
  /* n-links: every node i visits its 6 neighbours n and adds (w, w) when density[n] > 0 (:261-287) */
  for (int i = 0; i < g.N; ++i) {
    if (!node_mask[i]) continue;
    for (int d = 0; d < 6; ++d) {
      const int n = nbr(&g, i, d);
      if (n < 0) continue;
      { /* :264-266 compares EVERY coordinate of the neighbour with x, then y, then z => with min(X, Y, Z) */
        const int nx = n / g.sx, nr = n - nx * g.sx, ny = nr / g.sy, nz = nr - ny * g.sy;
        if (nx >= lim || ny >= lim || nz >= lim) continue;
      }
      if (!(dens[n] > 0.0f)) continue; /* :272 */
      float s = 0.0f;
      for (int c = 0; c < F; ++c) {
        const float df = feat[(size_t)i * F + c] - feat[(size_t)n * F + c];
        s = s + df * df;
      }
      const float l2 = sqrtf(s);
      const float e = (float)exp(-(double)(l2 / sigma));  /* float32 result, evaluated in double so that both
                                                                 implementations round identically */
      const int32_t q = (int32_t)llrint((double)e * (double)VOXE_GRAPH_CAP_ONE);
      cap[(size_t)d * g.N + i] += q;       /* add_edge(i, n, w, w): capacity i -> n ... */
      cap[(size_t)(d ^ 1) * g.N + n] += q; /* ... and n -> i */
    }
  }
  return VOXE_OK;
}

/* ---------------------------------------------------------------------------------------------- */
/* Dinic's algorithm on the grid graph; sources = all +1 seeds (infinite supply), sinks = all -1 seeds. */
typedef struct {
  Dims g;
  const uint8_t* node;
  const int8_t* term;
  int32_t* cap;
  int* level;
  int* it;     /* current-arc pointer */
  int* queue;
} Dinic;

static int dinic_bfs(Dinic* D) {
  const int N = D->g.N;
  int head = 0, tail = 0, reached = 0;
  for (int v = 0; v < N; ++v) {
    D->level[v] = -1;
    if (D->node[v] && D->term[v] > 0) {
      D->level[v] = 0;
      D->queue[tail++] = v;
    }
  }
  while (head < tail) {
    const int u = D->queue[head++];
    if (D->term[u] < 0) {
      reached = 1;
      continue; /* sinks absorb: never route through them */
    }
    for (int d = 0; d < 6; ++d) {
      if (D->cap[(size_t)d * N + u] <= 0) continue;
      const int n = nbr(&D->g, u, d);
      if (n < 0 || !D->node[n] || D->level[n] >= 0) continue;
      D->level[n] = D->level[u] + 1;
      D->queue[tail++] = n;
    }
  }
  return reached;
}

/* one augmenting path from source seed `s` along the level graph; returns the pushed amount (0: none left) */
static long long dinic_augment(Dinic* D, int s, int* path_v, int* path_d) {
  const int N = D->g.N;
  int depth = 0;
  path_v[0] = s;
  for (;;) {
    const int u = path_v[depth];
    if (D->term[u] < 0) { /* reached a sink: bottleneck and push */
      long long b = (long long)1 << 60;
      for (int i = 0; i < depth; ++i) {
        const long long c = D->cap[(size_t)path_d[i] * N + path_v[i]];
        if (c < b) b = c;
      }
      for (int i = 0; i < depth; ++i) {
        D->cap[(size_t)path_d[i] * N + path_v[i]] -= (int32_t)b;
        D->cap[(size_t)(path_d[i] ^ 1) * N + path_v[i + 1]] += (int32_t)b;
      }
      return b;
    }
    int advanced = 0;
    while (D->it[u] < 6) {
      const int d = D->it[u];
      const int n = nbr(&D->g, u, d);
      if (n >= 0 && D->node[n] && D->cap[(size_t)d * N + u] > 0 && D->level[n] == D->level[u] + 1) {
        path_d[depth] = d;
        path_v[++depth] = n;
        advanced = 1;
        break;
      }
      ++D->it[u];
    }
    if (advanced) continue;
    if (depth == 0) return 0;
    --depth; /* dead end: retreat and drop the arc that led here */
    ++D->it[path_v[depth]];
  }
}

int voxe_cpu_graphcut(const uint8_t* node_mask, const int8_t* terminal, int32_t* cap, int32_t X, int32_t Y,
                      int32_t Z, uint8_t* segment, int64_t* flow) {
  if (!node_mask || !terminal || !cap || !segment || !flow) return VOXE_ERR_NULL_POINTER;
  if (!dims_ok(X, Y, Z)) return VOXE_ERR_BAD_SHAPE;
  Dinic D;
  D.g = make_dims(X, Y, Z);
  const int N = D.g.N;
  D.node = node_mask;
  D.term = terminal;
  D.cap = cap;
  D.level = (int*)malloc((size_t)N * sizeof(int));
  D.it = (int*)malloc((size_t)N * sizeof(int));
  D.queue = (int*)malloc((size_t)N * sizeof(int));
  int* path_v = (int*)malloc(((size_t)N + 1) * sizeof(int));
  int* path_d = (int*)malloc(((size_t)N + 1) * sizeof(int));
  long long total = 0;
  while (dinic_bfs(&D)) {
    memset(D.it, 0, (size_t)N * sizeof(int));
    for (int s = 0; s < N; ++s) {
      if (!node_mask[s] || terminal[s] <= 0) continue;
      long long f;
      while ((f = dinic_augment(&D, s, path_v, path_d)) > 0) total += f;
    }
  }
  *flow = total;
  /* sink side = nodes that reach a sink seed over residual edges (reverse BFS) */
  int head = 0, tail = 0;
  for (int v = 0; v < N; ++v) {
    D.level[v] = 0;
    if (node_mask[v] && terminal[v] < 0) {
      D.level[v] = 1;
      D.queue[tail++] = v;
    }
  }
  while (head < tail) {
    const int v = D.queue[head++];
    for (int d = 0; d < 6; ++d) {
      const int u = nbr(&D.g, v, d); /* u -> v uses u's plane of the opposite direction */
      if (u < 0 || !node_mask[u] || D.level[u] || terminal[u] != 0) continue;
      if (cap[(size_t)(d ^ 1) * N + u] > 0) {
        D.level[u] = 1;
        D.queue[tail++] = u;
      }
    }
  }
  for (int v = 0; v < N; ++v)
    segment[v] = !node_mask[v] ? 255 : (terminal[v] > 0 ? 0 : (terminal[v] < 0 ? 1 : (uint8_t)D.level[v]));
  free(D.level); free(D.it); free(D.queue); free(path_v); free(path_d);
  return VOXE_OK;
}

/* ---------------------------------------------------------------------------------------------- */
typedef struct {
  int count, root;
} Comp;

static int comp_cmp(const void* a, const void* b) {
  const Comp* p = (const Comp*)a;
  const Comp* q = (const Comp*)b;
  if (p->count != q->count) return p->count > q->count ? -1 : 1; /* larger first */
  return p->root < q->root ? -1 : (p->root > q->root ? 1 : 0);   /* then earlier first voxel */
}

int voxe_cpu_cc_largest_k(const uint8_t* mask, int32_t X, int32_t Y, int32_t Z, int32_t k, int32_t* labels,
                          int32_t* num_components) {
  if (!mask || !labels || !num_components) return VOXE_ERR_NULL_POINTER;
  if (!dims_ok(X, Y, Z) || k < 0) return VOXE_ERR_BAD_SHAPE;
  const Dims g = make_dims(X, Y, Z);
  const int N = g.N;
  int* comp = (int*)malloc((size_t)N * sizeof(int)); /* component id per voxel, -1 background / unvisited */
  int* queue = (int*)malloc((size_t)N * sizeof(int));
  Comp* comps = (Comp*)malloc((size_t)N * sizeof(Comp));
  int nc = 0;
  for (int v = 0; v < N; ++v) comp[v] = -1;
  for (int s = 0; s < N; ++s) {
    if (!mask[s] || comp[s] >= 0) continue;
    int head = 0, tail = 0;
    comp[s] = nc;
    queue[tail++] = s;
    while (head < tail) {
      const int v = queue[head++];
      const int x = v / g.sx, r = v - x * g.sx, y = r / g.sy, z = r - y * g.sy;
      for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dz = -1; dz <= 1; ++dz) {
            const int xx = x + dx, yy = y + dy, zz = z + dz;
            if (xx < 0 || xx >= X || yy < 0 || yy >= Y || zz < 0 || zz >= Z) continue;
            const int n = xx * g.sx + yy * g.sy + zz;
            if (mask[n] && comp[n] < 0) {
              comp[n] = nc;
              queue[tail++] = n;
            }
          }
    }
    comps[nc].count = tail;
    comps[nc].root = s;
    ++nc;
  }
  *num_components = nc;
  qsort(comps, (size_t)nc, sizeof(Comp), comp_cmp);
  const int M = k < nc ? k : nc;
  int* label_of = (int*)calloc((size_t)(nc > 0 ? nc : 1), sizeof(int));
  for (int j = 0; j < M; ++j) label_of[comp[comps[j].root]] = M - j; /* largest -> M, ascending sizes 1..M */
  for (int v = 0; v < N; ++v) labels[v] = comp[v] >= 0 ? label_of[comp[v]] : 0;
  free(comp); free(queue); free(comps); free(label_of);
  return VOXE_OK;
}
