// voxe_render_scatter.hip -- backward render kernel for rays in ARBITRARY order (random ray batches of the
// reconstruction trainer, ray lists without image structure): line-dense global atomics.
//
// Unordered rays share no voxels inside a wave, so there is nothing to combine on chip (the LDS-window kernel
// does not apply); the scatter is bound by the ~20 G atomic cache-line REQUESTS/s of the memory side
// (profiles/r01_microbench_atomics.md), and one request can carry up to 16 dwords.  render_bwd_kernel issues
// one request per (corner, channel, lane).  Here every wave stages its 64 samples' footprints in LDS and
// re-reads them transposed, so that the lanes of one atomic instruction are
//     (sample s, corner j, channel ch)         ->  2 samples x [8 corners x 4 channels]
// and the gradient buffer is 2x2x2-BRICKED (brick_slot(): the 8 voxels of a brick share one 128-byte line), so the
// 32 lanes of a sample fall into (1.5)^3 = 3.4 lines on average instead of the 4.5 of z-pairs in the linear layout (a
// request carries at most 16 dwords, so a full brick still costs two): measured -7 % on the kernel (32768 random
// rays: 1.24 -> 1.17 ms), the un-brick pass of the gradient costs 0.015 ms more.
// Per-ray math: identical to render_bwd_kernel / render_bwd_tile_kernel.
#include <limits.h>

#include "voxe_device.hpp"
#include "voxe_launch.hpp"
#include "voxe_render_common.hpp"

namespace voxe {

// View-dependent grids (NCU = 4 / 9 / 16 used coefficients per colour, texels of CM = 13 / 28 / 49 channels): the wave
// stages the 4 gradient SOURCES of a sample (d rad_0..2, d v) and the SH basis of its ray; the lanes of one atomic
// instruction are (corner j, 8 consecutive gradient channels) of ONE sample, i.e. 32 contiguous bytes per corner, and a
// sample takes ceil((3 NCU + 1) / 8) instructions -- 16 requests per sample at SH-1 instead of the 104 of
// render_bwd_kernel.
//
// Consecutive samples of a ray are about a voxel apart (S = 256 over 160 voxels), so the 2x2x2 footprints of samples
// k and k + 1 share a face: every lane keeps the footprint of its current cell PENDING in registers (8 corners x C
// channels), adds the next sample into it when the cell did not change, and otherwise hands over only the corners that
// left the footprint -- the others move to their place in the new cell's footprint.  The scatter is bound by atomic
// REQUESTS, and a request is now a voxel-channel that a ray is done with (~4 texels per sample instead of 8).
#ifndef VOXE_SCATTER_PEND
#define VOXE_SCATTER_PEND 1
#endif
// slot of voxel (x, y, z) in the gradient region: 2x2x2 bricks by default, the linear [X,Y,Z] order on request
__device__ __forceinline__ long long grad_slot(const DevCfg& c, int x, int y, int z, int Y, int Z) {
  return c.linear_grad ? ((long long)x * Y + y) * Z + z : brick_slot(x, y, z, Y, Z);
}

template <int COUT, int NCM, int NCU, bool WANT_D, bool WANT_F>
__global__ __launch_bounds__(64) void render_bwd_packed_scatter_kernel(
    DevGrid g, DevCfg c, const float* __restrict__ packed, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ jitter,
    const float* __restrict__ colour, const float* __restrict__ depth,
    const float* __restrict__ acc, const float* __restrict__ d_colour,
    const float* __restrict__ d_depth, const float* __restrict__ d_acc,
    const float* __restrict__ ray_state, float* __restrict__ gpacked) {
  constexpr int C = COUT + 1;                             // gradient sources per sample (= channels when NCU == 1)
  constexpr int CM = COUT * NCM + 1;                      // channels of a packed texel
  constexpr int NG = COUT * NCU + 1;                      // channels that receive a gradient
  constexpr int kLanesPerSample = 8 * C;                  // (corner, channel)
  constexpr int kSamplesPerInstr = 64 / kLanesPerSample;  // 2 (C = 4) or 4 (C = 2)
  __shared__ int s_cell[3][64];  // cell corner (x0, y0, z0); x0 = -1: nothing to deposit
  __shared__ float s_w[8][64];
  __shared__ float s_g[C][64];
  __shared__ float s_basis[NCU > 1 ? NCU : 1][64];
  constexpr bool kPend = VOXE_SCATTER_PEND != 0;   // (view-dependent grids: the 4 SOURCES per corner are pending)
  __shared__ float s_val[kPend ? 8 * C : 1][64];          // handed-over corner values (weights already applied)
  const int lane = threadIdx.x;

  // With the forward's depth-segment states (ray_state != nullptr) a block is (64 rays, 32-sample depth segment),
  // segment-major like the other render kernels: a 32768-ray batch is 512 waves if every wave marches whole rays --
  // half a wave per SIMD, latency bound far below the atomic-request ceiling -- and 4096 with segments.
  const int nt = (int)((c.R + 63) / 64);
  const int nseg = ray_state ? num_segments(c.S, c.seg_len) : 1;
  const int nrb = gridDim.x / nseg;
  const int seg = blockIdx.x / nrb;
  const int logical = logical_tile_of(c, blockIdx.x - seg * nrb, nrb, 1, nt);
  if (logical < 0) return;
  const long long r0 = (long long)logical * 64 + lane;
  const bool alive = r0 < c.R;
  const long long r = alive ? r0 : 0;

  RayCtx<COUT, NCM, NCU> rc;
  rc.init(g, c, r, rays_o, rays_d, jitter);
  if constexpr (NCU > 1) {   // one ray per lane for the whole kernel: its basis is staged once (visible after the first barrier)
#pragma unroll
    for (int t = 0; t < NCU; ++t) s_basis[t][lane] = rc.basis[t];
  }
  const int ks = ray_state ? seg * c.seg_len : 0, ke = ray_state ? min(c.S, ks + c.seg_len) - 1 : c.S - 1;
  const int k_lo = max(rc.k_lo, ks);
  int k_hi = alive ? min(rc.k_hi, ke) : k_lo - 1;
  bool has = k_lo <= k_hi;
  // state at the segment start (transmittance + partial sums of the forward), as in render_bwd_tile_kernel
  float T = 1.0f, suf_c[COUT], suf_a = 0.0f, suf_d = 0.0f;   // (suffix sums of the samples from the segment start on)
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) suf_c[ch] = 0.0f;
  if (has && seg > 0) {
    constexpr int NC = COUT + 3;
    T = ray_state[ray_state_index(seg, 0, NC, c.R, r)];
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) suf_c[ch] = ray_state[ray_state_index(seg, 1 + ch, NC, c.R, r)];
    suf_a = ray_state[ray_state_index(seg, 1 + COUT, NC, c.R, r)];
    suf_d = ray_state[ray_state_index(seg, 2 + COUT, NC, c.R, r)];
    if (c.term_eps > 0.0f && T < c.term_eps) { has = false; k_hi = k_lo - 1; }  // gradient truncation: nothing behind T < term_eps receives a gradient
  }

  float gc[COUT], gsum = 0.0f;
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) { gc[ch] = d_colour[r * COUT + ch]; gsum += gc[ch]; }
  const float gdep = d_depth ? d_depth[r] : 0.0f;
  const float gacc = d_acc ? d_acc[r] : 0.0f;
  const bool white = c.white && !c.attn;
  const float asum = acc[r];
  float total = gdep * depth[r] + gacc * asum;
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) {
    const float csum = white ? colour[r * COUT + ch] - (1.0f - asum) : colour[r * COUT + ch];
    total += gc[ch] * csum;
  }
  if (white) total -= gsum * asum;

  // every lane walks ITS OWN sample range: iteration i handles sample k_lo + i of the lane
  int trips = has ? (k_hi - k_lo + 1) : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) trips = max(trips, __shfl_xor(trips, off, 64));

  // suffix0 = sum_{j >= segment start} dL/dw_j w_j: the whole ray (from the forward outputs) for the first segment, the
  // saved suffix sums (back-to-front sums of the forward: no cancellation against the part in front) otherwise
  float suffix0 = total;
  if (seg > 0) {
    suffix0 = gdep * suf_d + gacc * suf_a;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) suffix0 += gc[ch] * suf_c[ch];
    if (white) suffix0 -= gsum * suf_a;
  }
  float run = 0.0f;   // sum of dL/dw_j w_j over the samples of this segment up to and including the current one
  float z_next = has ? rc.dg.z(k_lo) : 0.0f;
  // pending footprint of this lane's ray: cell (pc0 == INT_MIN: none) and its 8 x C accumulated values
  int pc0 = INT_MIN, pc1 = 0, pc2 = 0;
  float pend[8][C];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int ch = 0; ch < C; ++ch) pend[j][ch] = 0.0f;
  for (int i = 0; i < trips + (kPend ? 1 : 0); ++i) {   // (+1: the iteration after a lane's last sample hands over the rest)
    const int k = k_lo + i;
    const bool active = has && k <= k_hi;
    int base = -1, cy0 = 0, cz0 = 0;
    float wc[8], gch[C];
#pragma unroll
    for (int j = 0; j < 8; ++j) wc[j] = 0.0f;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) gch[ch] = 0.0f;
    if (active) {
      const float z = z_next;
      const bool last = (k == c.S - 1);
      if (!last) z_next = rc.dg.z(k + 1);
      float p[3];
      rc.point(z, p);
      Footprint fp;
      footprint(g, p, fp);
      if (fp.inside) {
        Cell cell;
        make_cell_fast(g, fp, cell);
        float v, rad[COUT];
        gather<COUT, NCM, NCU>(g, packed, cell, rc.basis, v, rad);
        float sigma, dpost;
        post_activate_vg(g.post_act, v, sigma, dpost);
        const float dl = last ? kInfinity : (z_next - z);
        const float delta = dl * rc.dnorm;
        const float e = fast_exp(-(sigma * delta));
        const float alpha = 1.0f - e;
        const float om = 1.0f - alpha;
        const float wk = alpha * T;
        float col[COUT], dldw = fmaf(gdep, z, gacc);
#pragma unroll
        for (int ch = 0; ch < COUT; ++ch) { col[ch] = sigmoidf(rad[ch]); dldw = fmaf(gc[ch], col[ch], dldw); }
        if (white) dldw -= gsum;
        run = fmaf(dldw, wk, run);
        const float suffix = last ? 0.0f : (suffix0 - run);
        const float tail = (om > 0.0f) ? suffix * fast_rcp(om) : 0.0f;
        const float dsig = (delta * e) * fmaf(T, dldw, -tail);
        bool any = false;
#pragma unroll
        for (int ch = 0; ch < COUT; ++ch) {
          // NCU == 1: the channel gradient itself (x C0); otherwise the source d rad_ch (x basis_j at the deposit)
          const float src = (wk * gc[ch]) * (col[ch] * (1.0f - col[ch]));
          gch[ch] = WANT_F ? (NCU == 1 ? src * kC0 : src) : 0.0f;
          any = any || (gch[ch] != 0.0f);
        }
        gch[COUT] = WANT_D ? dsig * dpost : 0.0f;
        any = any || (gch[COUT] != 0.0f);
        T = T * om;
        if (any) {
          base = cell.i[0];
          cy0 = cell.i[1];
          cz0 = cell.i[2];
#pragma unroll
          for (int j = 0; j < 8; ++j) wc[j] = (cell.w[0][j & 1] * cell.w[1][(j >> 1) & 1]) * cell.w[2][j >> 2];
        }
        if (c.term_eps > 0.0f && T < c.term_eps) k_hi = k;
      }
    }
    float fv[kPend ? 8 : 1][C];   // corners handed over this iteration (cell base / cy0 / cz0 after the block below)
    if constexpr (kPend) {
      const bool contrib = base >= 0;
      const int n0 = base, n1 = cy0, n2 = cz0;
      base = -1;
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int ch = 0; ch < C; ++ch) fv[j][ch] = 0.0f;
      if (pc0 != INT_MIN && (!active || (contrib && (n0 != pc0 || n1 != pc1 || n2 != pc2)))) {
        // the ray is done (or moved on): hand over the pending corners that are not part of the new footprint.
        // Corner j of the pending cell sits at pc + bits(j); it stays iff on every axis bit - d is 0 or 1 (d = new - old)
        const int d0 = active ? n0 - pc0 : 4, d1 = active ? n1 - pc1 : 4, d2 = active ? n2 - pc2 : 4;
        const bool k00 = d0 == 0 || d0 == -1, k01 = d0 == 0 || d0 == 1;   // axis 0: corner bit 0 / bit 1 stays
        const bool k10 = d1 == 0 || d1 == -1, k11 = d1 == 0 || d1 == 1;
        const bool k20 = d2 == 0 || d2 == -1, k21 = d2 == 0 || d2 == 1;
        base = pc0; cy0 = pc1; cz0 = pc2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool stay = ((j & 1) ? k01 : k00) && ((j & 2) ? k11 : k10) && ((j & 4) ? k21 : k20);
#pragma unroll
          for (int ch = 0; ch < C; ++ch) {
            fv[j][ch] = stay ? 0.0f : pend[j][ch];
            pend[j][ch] = stay ? pend[j][ch] : 0.0f;
          }
        }
        // move the survivors to their corner of the new cell, one axis at a time (what is shifted out is already 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const int d = a == 0 ? d0 : (a == 1 ? d1 : d2);
          const int bit = 1 << a;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j & bit) continue;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
              const float lo = pend[j][ch], hi = pend[j | bit][ch];
              pend[j][ch] = d == 1 ? hi : (d == -1 ? 0.0f : lo);
              pend[j | bit][ch] = d == -1 ? lo : (d == 1 ? 0.0f : hi);
            }
          }
        }
        pc0 = INT_MIN;
      }
      if (contrib) {
        pc0 = n0; pc1 = n1; pc2 = n2;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int ch = 0; ch < C; ++ch) pend[j][ch] = fmaf(gch[ch], wc[j], pend[j][ch]);
      }
    }
    if (__ballot(base >= 0) == 0ull) continue;  // wave-uniform: nothing to deposit this iteration
    // ---- stage the 64 footprints, then deposit them transposed ------------------------------------
    __syncthreads();
    s_cell[0][lane] = base;
    s_cell[1][lane] = cy0;
    s_cell[2][lane] = cz0;
    if constexpr (kPend) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int ch = 0; ch < C; ++ch) s_val[j * C + ch][lane] = fv[j][ch];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) s_w[j][lane] = wc[j];
#pragma unroll
      for (int ch = 0; ch < C; ++ch) s_g[ch][lane] = gch[ch];
    }
    __syncthreads();
    if constexpr (kPend && NCU == 1) {
      const int j = (lane / C) & 7, ch = lane % C;  // corner (x bit 0, y bit 1, z bit 2) and channel of this lane
      const int mem = (ch == COUT) ? CM - 1 : ch * NCM;   // (diffuse renders of wider texels: coefficient 0 of colour ch)
#pragma unroll 4
      for (int grp = 0; grp < 64 / kSamplesPerInstr; ++grp) {
        const int s = grp * kSamplesPerInstr + lane / kLanesPerSample;
        const int b = s_cell[0][s];
        if (b >= 0) {
          const float gv = s_val[j * C + ch][s];
          if (gv != 0.0f) {
            const int x = min(b + (j & 1), g.X - 1), y = min(s_cell[1][s] + ((j >> 1) & 1), g.Y - 1);
            const int z = min(s_cell[2][s] + (j >> 2), g.Z - 1);
            atomicAdd(gpacked + grad_slot(c, x, y, z, g.Y, g.Z) * CM + mem, gv);
          }
        }
      }
    } else if constexpr (NCU == 1) {
      const int j = (lane / C) & 7, ch = lane % C;  // corner (x bit 0, y bit 1, z bit 2) and channel of this lane
      const int mem = (ch == COUT) ? CM - 1 : ch * NCM;   // (diffuse renders of wider texels: coefficient 0 of colour ch)
#pragma unroll 4
      for (int grp = 0; grp < 64 / kSamplesPerInstr; ++grp) {
        const int s = grp * kSamplesPerInstr + lane / kLanesPerSample;
        const int b = s_cell[0][s];
        if (b >= 0) {
          const float gv = s_g[ch][s];
          const float w = s_w[j][s];
          if (gv != 0.0f && w != 0.0f) {
            // (a size-1 axis has both corners on the same voxel; make_cell gives the second one weight 0)
            const int x = min(b + (j & 1), g.X - 1), y = min(s_cell[1][s] + ((j >> 1) & 1), g.Y - 1);
            const int z = min(s_cell[2][s] + (j >> 2), g.Z - 1);
            atomicAdd(gpacked + grad_slot(c, x, y, z, g.Y, g.Z) * CM + mem, gv * w);
          }
        }
      }
    } else {
      const int j = lane >> 3, cc = lane & 7;       // corner and channel-in-chunk of this lane
      for (int s = 0; s < 64; ++s) {
        const int b = s_cell[0][s];
        if (b < 0) continue;                        // wave-uniform
        const float w = kPend ? 1.0f : s_w[j][s];   // (pending footprints carry their weights already)
        const int x = min(b + (j & 1), g.X - 1), y = min(s_cell[1][s] + ((j >> 1) & 1), g.Y - 1);
        const int z = min(s_cell[2][s] + (j >> 2), g.Z - 1);
        float* __restrict__ texel = gpacked + grad_slot(c, x, y, z, g.Y, g.Z) * CM;
#pragma unroll
        for (int chunk = 0; chunk < (NG + 7) / 8; ++chunk) {
          const int q = chunk * 8 + cc;             // gradient channel: (colour ch, coefficient jj) or, last, the density
          if (q < NG) {
            const int ch = q / NCU, jj = q - ch * NCU;
            const bool dens = (q == NG - 1);
            const float src = kPend ? s_val[j * C + (dens ? COUT : ch)][s] : s_g[dens ? COUT : ch][s];
            const float gv = dens ? src : src * s_basis[jj][s];
            if (gv != 0.0f && w != 0.0f) atomicAdd(texel + (dens ? CM - 1 : ch * NCM + jj), gv * w);
          }
        }
      }
    }
  }
}

bool packed_scatter_supported(int deg) { (void)deg; return true; }

template <int COUT, int NCM, int NCU>
static void launch_bwd_packed_scatter_t(const DevGrid& g, const DevCfg& c, const BwdArgs& a, hipStream_t st) {
  const int nb = blocks_for_tiles(c.map_mode, 1, (c.R + 63) / 64) * (a.ray_state ? num_segments(c.S, c.seg_len) : 1);
#define VOXE_PBWD(WD, WF)                                                                            \
  render_bwd_packed_scatter_kernel<COUT, NCM, NCU, WD, WF><<<nb, 64, 0, st>>>(                       \
      g, c, a.packed, a.rays_o, a.rays_d, a.jitter, a.colour, a.depth, a.acc, a.d_colour, a.d_depth, \
      a.d_acc, a.ray_state, a.gpacked)
  if (a.want_d && a.want_f) VOXE_PBWD(true, true);
  else if (a.want_d) VOXE_PBWD(true, false);
  else VOXE_PBWD(false, true);
#undef VOXE_PBWD
}

void launch_bwd_packed_scatter(const DevGrid& g, const DevCfg& c, int deg, int diffuse, const BwdArgs& a, hipStream_t st) {
  if (c.attn) launch_bwd_packed_scatter_t<1, 1, 1>(g, c, a, st);
  else if (deg == 0) launch_bwd_packed_scatter_t<3, 1, 1>(g, c, a, st);
  else if (deg == 1) { if (diffuse) launch_bwd_packed_scatter_t<3, 4, 1>(g, c, a, st); else launch_bwd_packed_scatter_t<3, 4, 4>(g, c, a, st); }
  else if (deg == 2) { if (diffuse) launch_bwd_packed_scatter_t<3, 9, 1>(g, c, a, st); else launch_bwd_packed_scatter_t<3, 9, 9>(g, c, a, st); }
  else { if (diffuse) launch_bwd_packed_scatter_t<3, 16, 1>(g, c, a, st); else launch_bwd_packed_scatter_t<3, 16, 16>(g, c, a, st); }
}

}  // namespace voxe
