// gather_gemm.hip -- the one contraction kernel of the codec: an im2col-free, NHWC gather GEMM on
// the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 fma chain, 64 FLOP/clk/SIMD).
//
// Every Conv2D / Conv2DTranspose / SignalConv2D / GDN norm-pool on the hot path is
//     out[m, col] = sum_t sum_c x[src(m, t), c] * Wp[col][t*Cin + c]          (sntc_internal.h)
// with pixels on the MFMA row (A) side and output columns on the column (B) side.
//
// Block: 256 threads = 4 waves arranged WM x WN; each wave owns TM x TN tiles of 32x32; BK = 32.
//   HBM/L2 -> registers: buffer_load_dwordx4 through two wave-uniform descriptors (input, packed
//   weights) with a 32-bit per-lane byte offset that changes only when the tap changes and a scalar
//   offset that walks the channel slabs / K steps; a tile row is one full 128-B line; rows whose source
//   pixel falls outside the image carry an out-of-range offset and read zeros from the bounds check
//   (no branches, no 64-bit address arithmetic in the loop).
//   registers -> LDS: 128-B rows, 16-B chunks XOR-swizzled by (row>>1)&7 so both the staging
//   ds_write_b128 and the fragment ds_read_b128 are bank-conflict free without padding.
//   Two LDS buffers, one barrier per K step; the next step's loads are issued before the MFMAs of
//   the current one.  Per lane a ds_read_b128 fetches k = 8g+4h..8g+4h+3 (h = lane>>5): MFMA j of
//   k-group g sums k in {8g+j, 8g+4+j}; A and B use the same permutation, so the products pair up.
//   C/D layout: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
#include <algorithm>
#include <cstdlib>
#include "sntc_internal.h"

namespace sntc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kOutOfRange = 0x80000000u;   // > any in-range offset: buffers are < 2 GiB (host check)

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case SNTC_ACT_RELU: return fmaxf(v, 0.0f);
    case SNTC_ACT_LEAKY_RELU: return v >= 0.0f ? v : 0.2f * v;
    case SNTC_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0);
  return __builtin_bit_cast(f32x4, v);
}

template <int TM, int TN, int WM, int WN, bool VEC, bool PRO>
__global__ void __launch_bounds__(256, 2) gg_kernel(const GGArgs a) {
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int A_CH = BM / 32;   // 16-B chunks per thread per K step (A)
  constexpr int B_CH = BN / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);          // [2][BM][32]
  float* Bs = As + 2 * BM * 32;                        // [2][BN][32]
  int4* rinfo = reinterpret_cast<int4*>(Bs + 2 * BN * 32);   // [BM] (n, qy, qx, valid)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;

  int gi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroups; ++i)
    if (i < a.ngroups && (int)blockIdx.x >= a.g[i].blk0) gi = i;
  const GGGroup G = a.g[gi];
  const int lb0 = blockIdx.x - G.blk0;
  const int split = lb0 % a.ksplit;          // K range of this block (deterministic split-K)
  const int lb = lb0 / a.ksplit;
  // XCD-aware tile order: blocks b and b + 8 share an XCD (and its 4 MB L2).  Inside one XCD's sequence the column
  // tile runs fastest, so the ~64 blocks resident on an XCD cover a few row strips x all column tiles: an
  // activation strip is fetched into ONE L2 instead of eight, and the weight tiles are shared by the co-resident strips.
  int mt, nt;
  const int full = a.ntm & ~7;
  if (lb < full * G.ntn) {
    const int l = lb >> 3;
    mt = (l / G.ntn) * 8 + (lb & 7);
    nt = l % G.ntn;
  } else {                                   // ragged tail: fewer than 8 row strips left
    const int r = lb - full * G.ntn, rem = a.ntm - full;
    mt = full + r % rem;
    nt = r / rem;
  }
  const int m0 = mt * BM;
  const int n0 = nt * BN;

  for (int r = tid; r < BM; r += 256) {
    const int m = m0 + r;
    int4 ri = make_int4(0, 0, 0, 0);
    if (m < a.M) {
      const int per = a.Qh * a.Qw;
      const int n = m / per;
      const int rem = m - n * per;
      const int qy = rem / a.Qw;
      ri = make_int4(n, qy + G.q0y, rem - qy * a.Qw + G.q0x, 1);   // per-group origin of the macro grid
    }
    rinfo[r] = ri;
  }
  __syncthreads();

  // ---------------- loader state ----------------
  const int c = tid & 7;     // 16-B chunk inside the 128-B K slab
  const int r0 = tid >> 3;   // row 0..31 (+32 i)
  int a_iy0[A_CH], a_ix0[A_CH];
  unsigned a_img[A_CH];      // byte offset of the row's image (+ chunk), or kOutOfRange for padding rows
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int4 ri = rinfo[r0 + 32 * i];
    a_iy0[i] = ri.y * a.sA + a.offy;
    a_ix0[i] = ri.z * a.sA + a.offx;
    a_img[i] = ri.w ? (unsigned)ri.x * (unsigned)(a.H * a.W) * (unsigned)a.Cin * 4u + (unsigned)c * 16u : kOutOfRange;
  }
  unsigned b_off[B_CH];
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {   // rows past Ncol re-read the last column (finite, discarded)
    const int brow = min(n0 + r0 + 32 * i, G.Ncol - 1);
    b_off[i] = (unsigned)brow * (unsigned)G.K * 4u + (unsigned)c * 16u;
  }
  const __amdgpu_buffer_rsrc_t xs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ws =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G.wp), 0, G.Ncol * G.K * 4, 0x00020000);

  const int nsteps_all = G.K >> 5;
  const int ks0 = (int)(((long long)split * nsteps_all) / a.ksplit);
  const int ks1 = (int)(((long long)(split + 1) * nsteps_all) / a.ksplit);
  const int nsteps = ks1 - ks0;
  const int ncc = VEC ? (a.Cin >> 5) : 1;
  const int ktrue = G.T * a.Cin;
  int ld_step = ks0;
  int ld_t = VEC ? ks0 / ncc : 0, ld_cc = VEC ? ks0 % ncc : 0;
  unsigned a_off[A_CH];

  auto set_tap = [&](int t) {
    const int tap = G.taps[t];
    const int ty = tap >> 16, tx = tap & 0xffff;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int iy = a_iy0[i] + ty * a.tstep;
      const int ix = a_ix0[i] + tx * a.tstep;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && !(a_img[i] & kOutOfRange);
      const unsigned pix = (unsigned)(iy * a.W + ix) * (unsigned)a.Cin * 4u;
      a_off[i] = ok ? a_img[i] + pix : kOutOfRange;
    }
  };
  if (VEC && ld_t < G.T) set_tap(ld_t);

  f32x4 ra[A_CH], rb[B_CH];
  auto load_regs = [&]() {
    if (VEC) {
      const unsigned soff = (unsigned)ld_cc * 128u;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) ra[i] = buf_load(xs, a_off[i], soff);
    } else {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = ld_step * 32 + c * 4 + e;
          if (k < ktrue && !(a_img[i] & kOutOfRange)) {
            const int t = k / a.Cin;
            const int ch = k - t * a.Cin;
            const int tap = G.taps[t];
            const int iy = a_iy0[i] + (tap >> 16) * a.tstep;
            const int ix = a_ix0[i] + (tap & 0xffff) * a.tstep;
            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
              v[e] = a.x[(size_t)(a_img[i] >> 2) - c * 4 + ((size_t)iy * a.W + ix) * a.Cin + ch];
          }
        }
        ra[i] = v;
      }
    }
    const unsigned wsoff = (unsigned)ld_step * 128u;
#pragma unroll
    for (int i = 0; i < B_CH; ++i) rb[i] = buf_load(ws, b_off[i], wsoff);
    ++ld_step;
    if (VEC) {
      if (++ld_cc == ncc) {
        ld_cc = 0;
        if (++ld_t < G.T) set_tap(ld_t);
      }
    }
  };
  auto write_lds = [&](int buf) {
    float* Ab = As + buf * BM * 32;
    float* Bb = Bs + buf * BN * 32;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int r = r0 + 32 * i;
      f32x4 v = ra[i];
      if (PRO) {
        if (a.pro == SNTC_PRO_ABS) {
          v[0] = fabsf(v[0]); v[1] = fabsf(v[1]); v[2] = fabsf(v[2]); v[3] = fabsf(v[3]);
        } else if (a.pro == SNTC_PRO_SQUARE) {
          v = v * v;
        }
      }
      *reinterpret_cast<f32x4*>(Ab + r * 32 + ((c ^ ((r >> 1) & 7)) << 2)) = v;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int r = r0 + 32 * i;
      *reinterpret_cast<f32x4*>(Bb + r * 32 + ((c ^ ((r >> 1) & 7)) << 2)) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int swz = (l31 >> 1) & 7;

  if (nsteps > 0) {
    load_regs();
    write_lds(0);
  }
  __syncthreads();

  for (int ks = 0; ks < nsteps; ++ks) {
    const bool more = ks + 1 < nsteps;
    if (more && !SNTC_DBG(a, 1)) load_regs();
    const float* Ab = As + (ks & 1) * BM * 32 + (wm * TM * 32 + l31) * 32;
    const float* Bb = Bs + (ks & 1) * BN * 32 + (wn * TN * 32 + l31) * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int off = ((2 * g + h) ^ swz) << 2;
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * 32 + off);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * 32 + off);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
    if (more && !SNTC_DBG(a, 2)) write_lds((ks + 1) & 1);
    if (!SNTC_DBG(a, 4)) __syncthreads();
  }
  if (SNTC_DBG(a, 4)) __syncthreads();

  // ---------------- epilogue ----------------
  // Wide path (Cout % 4 == 0): each wave transposes its accumulators through a private 8-KB LDS
  // slice (the staging buffers are dead after the last barrier) so that every lane owns 4
  // consecutive channels of one pixel: bias / residual / gate operands are read and the output is
  // written with 16-B accesses, 256 contiguous bytes per 16 lanes.
  if (a.ksplit > 1) {
    // split-K: raw partial sums go to slab[group][split][m][col]; gg_reduce_kernel adds the splits in a
    // fixed order and applies bias / activation / epilogue, so the result does not depend on scheduling.
    float* slab = a.slab + G.slab_off + (size_t)split * a.M * G.Ncol;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + l31;
      if (col >= G.Ncol) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < a.M) slab[(size_t)m * G.Ncol + col] = acc[i][j][r];
        }
    }
    return;
  }
  if ((a.Cout & 3) == 0) {
    static_assert(TM == 1, "wide epilogue assumes one M tile per wave");
    float* stage = reinterpret_cast<float*>(smem) + wave * 2048;   // 32 rows x 64 floats
#pragma unroll
    for (int j0 = 0; j0 < TN; j0 += 2) {
      const int ct = (TN - j0) >= 2 ? 2 : 1;          // tiles in this chunk
      const int wfl = ct * 32;                         // chunk width in floats
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        if (jj < ct) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            stage[((r & 3) + 8 * (r >> 2) + 4 * h) * wfl + jj * 32 + l31] = acc[0][j0 + jj][r];
        }
      }
      const int lanes_per_row = wfl >> 2;              // 16 or 8
      const int rows_per_pass = 64 / lanes_per_row;    // 4 or 8
      const int c4 = (lane % lanes_per_row) << 2;
      const int rsub = lane / lanes_per_row;
      const int col = n0 + (wn * TN + j0) * 32 + c4;
      const bool col_ok = col < G.Ncol;
      unsigned ce = 0;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (col_ok) {
        ce = G.cols[col];
        if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + (ce & 0xffff));
      }
      const int ch = ce & 0xffff;
      const int oyo = (int)((ce >> 24) & 0xff) - 128;
      const int oxo = (int)((ce >> 16) & 0xff) - 128;
      for (int rp = 0; rp < 32; rp += rows_per_pass) {
        const int rloc = rp + rsub;
        const int4 ri = rinfo[wm * 32 + rloc];
        const int oy = ri.y * a.sO + oyo;
        const int ox = ri.z * a.sO + oxo;
        if (!col_ok || !ri.w || (unsigned)oy >= (unsigned)a.Ho || (unsigned)ox >= (unsigned)a.Wo) continue;
        const size_t idx = (((size_t)ri.x * a.Ho + oy) * a.Wo + ox) * a.Cout + ch;
        f32x4 v = *reinterpret_cast<const f32x4*>(stage + rloc * wfl + c4) + bv;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a.act);
        if (a.epi != SNTC_EPI_STORE) {
          const f32x4 rs = *reinterpret_cast<const f32x4*>(a.res + idx);
          switch (a.epi) {
            case SNTC_EPI_ADD: v = v + rs; break;
            case SNTC_EPI_GATE: v = rs + *reinterpret_cast<const f32x4*>(a.aux + idx) * v; break;
            case SNTC_EPI_RES_DIV: v = rs / v; break;
            case SNTC_EPI_RES_MUL: v = rs * v; break;
            case SNTC_EPI_RES_DIV_SQRT:
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = rs[e] / sqrtf(v[e]);
              break;
            case SNTC_EPI_RES_MUL_SQRT:
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = rs[e] * sqrtf(v[e]);
              break;
            case SNTC_EPI_MASK_RELU:
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = rs[e] > 0.0f ? v[e] : 0.0f;
              break;
            case SNTC_EPI_MASK_LEAKY:
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = rs[e] >= 0.0f ? v[e] : 0.2f * v[e];
              break;
            default: break;
          }
        }
        *reinterpret_cast<f32x4*>(a.y + idx) = v;
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + l31;
    if (col >= G.Ncol) continue;
    const unsigned ce = G.cols[col];
    const int ch = ce & 0xffff;
    const int oyo = (int)((ce >> 24) & 0xff) - 128;
    const int oxo = (int)((ce >> 16) & 0xff) - 128;
    const float bv = a.bias ? a.bias[ch] : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int4 ri = rinfo[row];
        if (!ri.w) continue;
        const int oy = ri.y * a.sO + oyo;
        const int ox = ri.z * a.sO + oxo;
        if ((unsigned)oy >= (unsigned)a.Ho || (unsigned)ox >= (unsigned)a.Wo) continue;
        const size_t idx = (((size_t)ri.x * a.Ho + oy) * a.Wo + ox) * a.Cout + ch;
        float v = apply_act(acc[i][j][r] + bv, a.act);
        switch (a.epi) {
          case SNTC_EPI_ADD: v = v + a.res[idx]; break;
          case SNTC_EPI_GATE: v = a.res[idx] + a.aux[idx] * v; break;
          case SNTC_EPI_RES_DIV: v = a.res[idx] / v; break;
          case SNTC_EPI_RES_MUL: v = a.res[idx] * v; break;
          case SNTC_EPI_RES_DIV_SQRT: v = a.res[idx] / sqrtf(v); break;
          case SNTC_EPI_RES_MUL_SQRT: v = a.res[idx] * sqrtf(v); break;
          case SNTC_EPI_MASK_RELU: v = a.res[idx] > 0.0f ? v : 0.0f; break;
          case SNTC_EPI_MASK_LEAKY: v = a.res[idx] >= 0.0f ? v : 0.2f * v; break;
          default: break;
        }
        a.y[idx] = v;
      }
    }
  }
}

// Split-K finish: y = epilogue(act(sum_{s = 0..S-1} slab[g][s][m][col] + bias)), splits added in index
// order (deterministic; the K ranges depend only on the layer and the image shape, never on the batch).
__global__ void __launch_bounds__(256) gg_reduce_kernel(const GGArgs a) {
  const GGGroup G = a.g[blockIdx.y];
  const size_t total = (size_t)a.M * G.Ncol;
  const int per = a.Qh * a.Qw;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / G.Ncol);
    const int col = (int)(i - (size_t)m * G.Ncol);
    float v = 0.0f;
    for (int sp = 0; sp < a.ksplit; ++sp) v += a.slab[G.slab_off + (size_t)sp * total + i];
    const unsigned ce = G.cols[col];
    const int ch = ce & 0xffff;
    const int n = m / per;
    const int rem = m - n * per;
    const int qy = rem / a.Qw + G.q0y, qx = rem - (rem / a.Qw) * a.Qw + G.q0x;
    const int oy = qy * a.sO + (int)((ce >> 24) & 0xff) - 128;
    const int ox = qx * a.sO + (int)((ce >> 16) & 0xff) - 128;
    if ((unsigned)oy >= (unsigned)a.Ho || (unsigned)ox >= (unsigned)a.Wo) continue;
    const size_t idx = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout + ch;
    v = apply_act(v + (a.bias ? a.bias[ch] : 0.0f), a.act);
    switch (a.epi) {
      case SNTC_EPI_ADD: v = v + a.res[idx]; break;
      case SNTC_EPI_GATE: v = a.res[idx] + a.aux[idx] * v; break;
      case SNTC_EPI_RES_DIV: v = a.res[idx] / v; break;
      case SNTC_EPI_RES_MUL: v = a.res[idx] * v; break;
      case SNTC_EPI_RES_DIV_SQRT: v = a.res[idx] / sqrtf(v); break;
      case SNTC_EPI_RES_MUL_SQRT: v = a.res[idx] * sqrtf(v); break;
      case SNTC_EPI_MASK_RELU: v = a.res[idx] > 0.0f ? v : 0.0f; break;
      case SNTC_EPI_MASK_LEAKY: v = a.res[idx] >= 0.0f ? v : 0.2f * v; break;
      default: break;
    }
    a.y[idx] = v;
  }
}

int gg_reduce_launch(const GGArgs& args, hipStream_t stream) {
  size_t most = 0;
  for (int gi = 0; gi < args.ngroups; ++gi) most = std::max(most, (size_t)args.M * args.g[gi].Ncol);
  int blocks = (int)std::min<size_t>((most + 255) / 256, 2048);
  hipLaunchKernelGGL(gg_reduce_kernel, dim3(blocks, args.ngroups), dim3(256), 0, stream, args);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "split-K reduce launch");
  return SNTC_OK;
}

// ---------------------------------------------------------------------------------------------
// variants + launch
// ---------------------------------------------------------------------------------------------
int gg_variant_bm(int v) { return v == 8 ? 64 : 128; }
int gg_variant_bn(int v) { return v == 8 ? 64 : 32 * v; }

static size_t lds_bytes(int v) {
  return (size_t)2 * (gg_variant_bm(v) + gg_variant_bn(v)) * 32 * sizeof(float) + gg_variant_bm(v) * sizeof(int4);
}

template <int TM, int TN, int WM, int WN>
static int launch_t(bool vec, bool pro, const GGArgs& args, int nblocks, size_t lds, hipStream_t stream) {
  if (vec && !pro)
    hipLaunchKernelGGL((gg_kernel<TM, TN, WM, WN, true, false>), dim3(nblocks), dim3(256), lds, stream, args);
  else if (vec)
    hipLaunchKernelGGL((gg_kernel<TM, TN, WM, WN, true, true>), dim3(nblocks), dim3(256), lds, stream, args);
  else
    hipLaunchKernelGGL((gg_kernel<TM, TN, WM, WN, false, true>), dim3(nblocks), dim3(256), lds, stream, args);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "gather-GEMM launch");
  return SNTC_OK;
}

template <int TM, int TN, int WM, int WN>
static hipError_t set_attr(size_t lds) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gg_kernel<TM, TN, WM, WN, true, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gg_kernel<TM, TN, WM, WN, true, true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&gg_kernel<TM, TN, WM, WN, false, true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

int gg_init() {
  static thread_local int done_device = -1;
  int dev = 0;
  SNTC_HIP(hipGetDevice(&dev));
  if (done_device == dev) return SNTC_OK;
  SNTC_HIP((set_attr<1, 1, 4, 1>(lds_bytes(1))));
  SNTC_HIP((set_attr<1, 2, 4, 1>(lds_bytes(2))));
  SNTC_HIP((set_attr<1, 3, 4, 1>(lds_bytes(3))));
  SNTC_HIP((set_attr<1, 4, 4, 1>(lds_bytes(4))));
  SNTC_HIP((set_attr<1, 5, 4, 1>(lds_bytes(5))));
  SNTC_HIP((set_attr<1, 6, 4, 1>(lds_bytes(6))));
  SNTC_HIP((set_attr<1, 7, 4, 1>(lds_bytes(7))));
  SNTC_HIP((set_attr<1, 1, 2, 2>(lds_bytes(8))));
  done_device = dev;
  return SNTC_OK;
}

int gg_launch(int variant, bool vec, const GGArgs& args_in, int nblocks, hipStream_t stream) {
  GGArgs args = args_in;
#ifdef SNTC_DIAG
  if (const char* e = getenv("SNTC_GG_DBG")) args.dbg = atoi(e);
#endif
  const size_t lds = lds_bytes(variant);
  const bool pro = args.pro != SNTC_PRO_NONE;
  switch (variant) {
    case 1: return launch_t<1, 1, 4, 1>(vec, pro, args, nblocks, lds, stream);
    case 2: return launch_t<1, 2, 4, 1>(vec, pro, args, nblocks, lds, stream);
    case 3: return launch_t<1, 3, 4, 1>(vec, pro, args, nblocks, lds, stream);
    case 4: return launch_t<1, 4, 4, 1>(vec, pro, args, nblocks, lds, stream);
    case 5: return launch_t<1, 5, 4, 1>(vec, pro, args, nblocks, lds, stream);
    case 6: return launch_t<1, 6, 4, 1>(vec, pro, args, nblocks, lds, stream);
    case 7: return launch_t<1, 7, 4, 1>(vec, pro, args, nblocks, lds, stream);
    case 8: return launch_t<1, 1, 2, 2>(vec, pro, args, nblocks, lds, stream);
    default: return fail(SNTC_ERR_UNSUPPORTED, "unknown gather-GEMM tile variant");
  }
}

}  // namespace sntc
