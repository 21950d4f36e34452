// What an LDS-fed fp32-MFMA K loop of the gather-GEMM's shape can reach on THIS device with RANDOM operands -- the known-good
// reference the kernel's roofline fraction is read against (cdna_hip_programming.md 5.4 rule 10: no ceiling claims from one's
// own failed attempts; rule 25: zero / constant operands read high because the chip holds a higher clock on them).
//
// One workgroup = 4 waves as 2 x 2, wave tile 64 x 64 (the 128 x 128 instance of csrc/gather_gemm.hip), a ring of three
// 16-deep stages of (128 + 128) rows x 64 B in LDS, fragments by ds_read_b128, 32 v_mfma_f32_32x32x2_f32 per wave and stage.
// Modes (bit mask):
//   1  barrier per stage (the ring hand-off)
//   2  stage traffic: 4 buffer_load_dwordx4 per thread and stage from an L2-resident array + 4 ds_write_b128 into the ring
//   4  constant operands instead of random ones (what tools/microbench/mfma_peak.hip measures)
//   8  no LDS fragment reads (operands stay in registers)
// The loop does exactly the kernel's per-stage instruction mix and nothing else (no tile bookkeeping, no epilogue), so its rate
// is an upper bound for any schedule of that mix; the difference between modes prices barriers, fragment reads and staging.
// Build: hipcc --offload-arch=gfx950 -O3 -o gemm_ceiling gemm_ceiling.hip ; run: ./gemm_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <cmath>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kRows = 256;          // A rows + B rows of a stage
constexpr int kSlot = kRows * 16;   // floats per ring slot

// PAT 0: every wave instruction loads 1 KB contiguous.  PAT 1: the kernel's gather shape -- 4 lanes per row fetch one 64-B
// K slab of a row, 64 rows per instruction, rows `row_stride` bytes apart (an NHWC pixel of Cin channels); the slab advances
// by 64 B per stage and wraps inside the row, so every 128-B line is touched by two different stages.
template <int MODE, int PAT = 0>
__global__ void __launch_bounds__(256, 2) loop_f32(const float* __restrict__ src, float* out, int stages, unsigned src_bytes, unsigned row_stride) {
  extern __shared__ __attribute__((aligned(16))) float ring[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  for (int i = tid; i < 3 * kSlot; i += 256) ring[i] = (MODE & 4) ? 0.5f : src[(blockIdx.x * 977 + i) & ((src_bytes >> 2) - 1)];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)src_bytes, 0x00020000);
  const int swz = (l31 >> 2) & 3;
  const int fa = (wm * 64 + l31) * 16, fb = (128 + wn * 64 + l31) * 16;
  const int o0 = ((0 + h) ^ swz) << 2, o1 = ((2 + h) ^ swz) << 2;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x4 A0[2], B0[2], A1[2], B1[2];
  auto rd = [&](f32x4* A, f32x4* B, int slot, int off) {
    const float* base = ring + slot * kSlot;
    for (int i = 0; i < 2; ++i) A[i] = *reinterpret_cast<const f32x4*>(base + fa + i * 512 + off);
    for (int j = 0; j < 2; ++j) B[j] = *reinterpret_cast<const f32x4*>(base + fb + j * 512 + off);
  };
  auto mm = [&](const f32x4* A, const f32x4* B) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i][e], B[j][e], acc[i][j], 0, 0, 0);
  };
  rd(A0, B0, 0, o0);
  rd(A1, B1, 0, o1);
  f32x4 R[4];
  const int r0 = tid >> 2, c = tid & 3, wsw = (c ^ ((r0 >> 2) & 3)) << 2;
  unsigned goff = ((unsigned)blockIdx.x * 65536u + (unsigned)tid * 16u) & (src_bytes - 1);
  unsigned rowoff[4];
  unsigned slab = 0;
  for (int i = 0; i < 4; ++i) rowoff[i] = (unsigned)(((unsigned long long)(blockIdx.x * 256 + r0 + 64 * i) * row_stride) & (src_bytes - 1)) + c * 16u;
  auto gaddr = [&](int i) { return PAT == 0 ? ((goff + i * 4096u) & (src_bytes - 1)) : ((rowoff[i] + slab) & (src_bytes - 1)); };
  if (MODE & 2)
    for (int i = 0; i < 4; ++i) R[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)gaddr(i), 0, 0));
  int s0 = 0, s1 = 1, s2 = 2;
  for (int st = 0; st < stages; ++st) {
    if (MODE & 2) {
      float* dst = ring + s2 * kSlot;
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(dst + (r0 + 64 * i) * 16 + wsw) = R[i];
      goff = (goff + 16384u) & (src_bytes - 1);
      slab += 64u;
      if (slab >= row_stride) slab = 0;
      for (int i = 0; i < 4; ++i) R[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)gaddr(i), 0, 0));
    }
    if (!(MODE & 8)) rd(A1, B1, s0, o1);
    mm(A0, B0);
    if (!(MODE & 8)) rd(A0, B0, s1, o0);
    mm(A1, B1);
    if (MODE & 2) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 13, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 15, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE & 1) __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    const int t = s0; s0 = s1; s1 = s2; s2 = t;
  }
  float s = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  if (s == 12345.678f) out[0] = s;
}

// The bf16 x 3 counterpart: the same 64 x 64 wave tile, operands as three bf16 planes (32 B per row, plane and stage), six
// v_mfma_f32_32x32x16_bf16 per 32 x 32 tile and 16-deep stage.  MODE 1: barrier per stage; 8: no fragment reads.
template <int MODE>
__global__ void __launch_bounds__(256, 2) loop_bf3(const float* __restrict__ src, float* out, int stages, unsigned src_bytes, unsigned row_stride) {
  extern __shared__ __attribute__((aligned(16))) float ring[];      // 3 slots x 256 rows x 96 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  constexpr int kSlot3 = kRows * 24;
  for (int i = tid; i < 3 * kSlot3; i += 256) {
    const float v = src[(blockIdx.x * 977 + i) & ((src_bytes >> 2) - 1)];
    __bf16 two[2] = {(__bf16)v, (__bf16)(v * 0.37f)};
    ring[i] = __builtin_bit_cast(float, two);
  }
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  bf16x8 A[3][2], B[3][2];
  const int hoff = (h ^ ((l31 >> 3) & 1)) << 4;
  auto rd = [&](int slot) {
    const char* base = reinterpret_cast<const char*>(ring + slot * kSlot3);
    for (int p = 0; p < 3; ++p) {
      for (int i = 0; i < 2; ++i) A[p][i] = *reinterpret_cast<const bf16x8*>(base + p * 128 * 32 + (wm * 64 + i * 32 + l31) * 32 + hoff);
      for (int j = 0; j < 2; ++j) B[p][j] = *reinterpret_cast<const bf16x8*>(base + 3 * 128 * 32 + p * 128 * 32 + (wn * 64 + j * 32 + l31) * 32 + hoff);
    }
  };
  rd(0);
  int s0 = 0;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
  for (int st = 0; st < stages; ++st) {
    if (!(MODE & 8)) rd(s0);
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[PA[t]][i], B[PB[t]][j], acc[i][j], 0, 0, 0);
    if (MODE & 1) __syncthreads();
    s0 = s0 == 2 ? 0 : s0 + 1;
  }
  float s = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  if (s == 12345.678f) out[0] = s;
}

// The bf16 x 3 loop as a kernel would run it on PRE-SPLIT operands (activations stored by their producer as three bf16 planes,
// 96 B per pixel and 16-channel slab, like the packed weights): stages go global -> LDS by direct-to-LDS loads (no staging
// registers, no ds_write, no VALU), row-major [row][96 B] LDS image with the two 16-B halves of a plane swapped on rows
// 8..15 (mod 16) -- applied on the SOURCE side, the DMA destination is linear -- so that fragment ds_read_b128 are
// conflict-free; three ring slots, fragments double-buffered: in step j the loads of stage j+3 go into the slot stage j left,
// stage j+1's fragments are read while stage j multiplies, and the step ends once stage j+2 has landed.
template <int WM, int WN, int TM, int TN, bool DBUF>
__global__ void __launch_bounds__(WM * WN * 64, (WM * WN >= 8) ? 2 : 2) loop_bf3_dma(const float* __restrict__ src, float* out, int stages,
                                                                                  unsigned src_bytes, unsigned row_stride) {
  constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32, ROWS = BM + BN, SLOT = ROWS * 96;
  constexpr int CH = (ROWS * 6 + NT - 1) / NT;               // 16-B chunks per thread and stage
  extern __shared__ __attribute__((aligned(16))) char ringb[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)src_bytes, 0x00020000);
  // row_stride == 0: the blocks form a GEMM grid (5 column tiles wide, XCD-aware: blocks b and b + 8 share an XCD and work on
  // neighbouring tiles), A rows of 2880 B (480 channels, pre-split) shared by a strip's column tiles, B rows of 25920 B (K = 4320)
  // shared by every strip -- the L2 reuse of a real layer.  Otherwise every block streams private rows `row_stride` apart.
  const bool gemm = row_stride == 0;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per_x = (gridDim.x + 7) >> 3;
  const int nt = idx % 5, mt = xcd * ((per_x + 4) / 5) + idx / 5;
  const unsigned strideA = gemm ? 2880u : row_stride, strideB = gemm ? 25920u : row_stride;
  const unsigned baseB = gemm ? (96u << 20) : 0u;
  unsigned off[CH];
  bool isB[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int q = i * NT + tid, row = q / 6, part = q % 6, p = part >> 1, hs = (part & 1) ^ ((row >> 3) & 1);
    isB[i] = row >= BM;
    unsigned long long o;
    if (!gemm) o = (unsigned long long)(blockIdx.x * ROWS + row) * row_stride;
    else o = isB[i] ? baseB + (unsigned long long)(nt * BN + row - BM) * strideB : (unsigned long long)(mt * BM + row) * strideA;
    off[i] = (unsigned)(o & (src_bytes - 1)) + p * 32 + hs * 16;
  }
  unsigned slabA = 0, slabB = 0;
  auto issue = [&](int slot) {
    char* base = ringb + slot * SLOT + wave * 1024;
#pragma unroll
    for (int i = 0; i < CH; ++i)
      if ((ROWS * 6) % NT == 0 || (i * NT + wave * 64) < ROWS * 6)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(base + i * NT * 16), 16, (int)((off[i] + (isB[i] ? slabB : slabA)) & (src_bytes - 1)), 0, 0, 0);
    slabA += 96u;
    if (slabA + 96u > strideA) slabA = 0;
    slabB += 96u;
    if (slabB + 96u > strideB) slabB = 0;
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  struct Fr { bf16x8 a[3][TM], b[3][TN]; };
  Fr F0, F1;
  const int hoff = (h ^ ((l31 >> 3) & 1)) << 4;
  auto rd = [&](Fr& F, int slot) {
    const char* base = ringb + slot * SLOT;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) F.a[p][i] = *reinterpret_cast<const bf16x8*>(base + (wm * TM * 32 + i * 32 + l31) * 96 + p * 32 + hoff);
#pragma unroll
      for (int j = 0; j < TN; ++j) F.b[p][j] = *reinterpret_cast<const bf16x8*>(base + (BM + wn * TN * 32 + j * 32 + l31) * 96 + p * 32 + hoff);
    }
  };
  auto mm = [&](const Fr& F) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[PA[t]][i], F.b[PB[t]][j], acc[i][j], 0, 0, 0);
  };
  // my loads per stage (the last chunk round may not include this wave)
  issue(0); issue(1); issue(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  rd(F0, 0);
  int s0 = 0, s1 = 1;
  auto step = [&](Fr& Fc, Fr& Fn) {
    issue(s0);                        // stage j+3 into the slot whose fragments (stage j) are already in registers
    if (DBUF) {
      rd(Fn, s1);
      mm(Fc);
    } else {
      mm(Fc);
    }
    __builtin_amdgcn_sched_barrier(0);
    // stage j+2 (issued in the previous step) must have landed: leave only this step's loads in flight
    if ((ROWS * 6) % NT == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CH) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CH - 1) : "memory");     // conservative for the waves that issue CH
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (!DBUF) rd(Fn, s1);
    s0 = s1; s1 = s1 == 2 ? 0 : s1 + 1;
  };
  for (int st = 0; st < stages; st += 2) {
    step(F0, DBUF ? F1 : F0);
    step(DBUF ? F1 : F0, F0);
  }
  float s = 0;
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  if (s == 12345.678f) out[0] = s;
}


// A-halo reuse (stride-1 k x k layers): the TAPS taps of one 16-channel slab read the SAME activation rows shifted by whole
// pixels, so the block stages one patch of BM + EXTRA rows per slab (double-buffered) and only the weights per stage.
template <int WM, int WN, int TM, int TN, bool DBUF, int TAPS, int EXTRA>
__global__ void __launch_bounds__(WM * WN * 64, 2) loop_bf3_halo(const float* __restrict__ src, float* out, int stages, unsigned src_bytes, unsigned) {
  constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32, PROWS = BM + EXTRA, PBYTES = PROWS * 96, BBYTES = BN * 96;
  constexpr int CHA = (PROWS * 6 + NT - 1) / NT, CHB = (BN * 6 + NT - 1) / NT;
  constexpr int MINA = (PROWS * 6) / NT, MINB = (BN * 6) / NT;
  extern __shared__ __attribute__((aligned(16))) char ringb[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)src_bytes, 0x00020000);
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per_x = (gridDim.x + 7) >> 3;
  const int nt = idx % 5, mt = xcd * ((per_x + 4) / 5) + idx / 5;
  const unsigned strideA = 2880u, strideB = 25920u, baseB = 96u << 20;
  unsigned offA[CHA], offB[CHB];
#pragma unroll
  for (int i = 0; i < CHA; ++i) {
    const int q = i * NT + tid, row = q / 6, part = q % 6, p = part >> 1, hs = (part & 1) ^ ((row >> 3) & 1);
    offA[i] = (unsigned)(((unsigned long long)(mt * BM + row) * strideA) & (src_bytes - 1)) + p * 32 + hs * 16;
  }
#pragma unroll
  for (int i = 0; i < CHB; ++i) {
    const int q = i * NT + tid, row = q / 6, part = q % 6, p = part >> 1, hs = (part & 1) ^ ((row >> 3) & 1);
    offB[i] = (unsigned)((baseB + (unsigned long long)(nt * BN + row) * strideB) & (src_bytes - 1)) + p * 32 + hs * 16;
  }
  char* const patch = ringb;
  char* const bring = ringb + 2 * PBYTES;
  unsigned slabA = 0, slabB = 0;
  auto issueA = [&](int buf) {
    char* base = patch + buf * PBYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < CHA; ++i)
      if ((PROWS * 6) % NT == 0 || (i * NT + wave * 64) < PROWS * 6)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(base + i * NT * 16), 16, (int)((offA[i] + slabA) & (src_bytes - 1)), 0, 0, 0);
    slabA += 96u;
    if (slabA + 96u > strideA) slabA = 0;
  };
  auto issueB = [&](int slot) {
    char* base = bring + slot * BBYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < CHB; ++i)
      if ((BN * 6) % NT == 0 || (i * NT + wave * 64) < BN * 6)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(base + i * NT * 16), 16, (int)((offB[i] + slabB) & (src_bytes - 1)), 0, 0, 0);
    slabB += 96u;
    if (slabB + 96u > strideB) slabB = 0;
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  struct Fr { bf16x8 a[3][TM], b[3][TN]; };
  Fr F0, F1;
  const int hoffB = (h ^ ((l31 >> 3) & 1)) << 4;
  auto rd = [&](Fr& F, int slot, int buf, int tap) {
    const char* pb = patch + buf * PBYTES;
    const char* bb = bring + slot * BBYTES;
    const int shift = (tap / 3) * 48 + tap % 3;        // a 3 x 3 window on a 48-pixel-wide map
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int prow = wm * TM * 32 + i * 32 + l31 + shift;
        F.a[p][i] = *reinterpret_cast<const bf16x8*>(pb + prow * 96 + p * 32 + ((h ^ ((prow >> 3) & 1)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) F.b[p][j] = *reinterpret_cast<const bf16x8*>(bb + (wn * TN * 32 + j * 32 + l31) * 96 + p * 32 + hoffB);
    }
  };
  auto mm = [&](const Fr& F) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[PA[t]][i], F.b[PB[t]][j], acc[i][j], 0, 0, 0);
  };
  auto waitcnt = [&](int n) {
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    }
  };
  issueA(0); issueB(0); issueB(1); issueB(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  rd(F0, 0, 0, 0);
  int s0 = 0, s1 = 1, buf = 0;
  for (int st = 0; st < stages; st += 2 * TAPS) {
#pragma unroll
    for (int u = 0; u < 2 * TAPS; ++u) {
      const int p = u % TAPS;                 // tap of the stage being multiplied
      Fr& Fc = (DBUF && (u & 1)) ? F1 : F0;
      Fr& Fn = (DBUF && !(u & 1)) ? F1 : F0;
      const int ntap = (p + 1) % TAPS, nbuf = ntap == 0 ? buf ^ 1 : buf;
      issueB(s0);
      if (p == 0) issueA(buf ^ 1);            // the next slab's patch, while this slab's nine taps run
      if (DBUF) { rd(Fn, s1, nbuf, ntap); mm(Fc); } else mm(Fc);
      __builtin_amdgcn_sched_barrier(0);
      waitcnt(MINB + (p == 0 ? MINA : 0) + (p == 1 ? MINA : 0));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (!DBUF) rd(Fn, s1, nbuf, ntap);
      s0 = s1; s1 = s1 == 2 ? 0 : s1 + 1;
      buf = nbuf;
    }
  }
  float s = 0;
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  if (s == 12345.678f) out[0] = s;
}

template <class K>
static void run_t(const char* label, K kern, int threads, int blocks, size_t lds, const float* src, float* out, unsigned src_bytes,
                  double flop_per_stage_block, unsigned row_stride) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int stages = 20000;
  float best = 1e30f, sum = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, src, out, rep == 0 ? 2000 : stages, src_bytes, row_stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep) { best = ms < best ? ms : best; sum += ms; }
  }
  hipError_t err = hipGetLastError();
  const double tf = flop_per_stage_block * blocks * stages / (sum / 3) / 1e9;
  printf("%-66s mean %.2f ms  %.1f TFLOP/s-equivalent = %.3f of 157.3 (best %.1f) %s\n", label, sum / 3, tf, tf / 157.3,
         flop_per_stage_block * blocks * stages / best / 1e9, err == hipSuccess ? "" : hipGetErrorString(err));
}

template <class K>
static void run(const char* label, K kern, size_t lds, const float* src, float* out, unsigned src_bytes, double flop_per_stage_block, unsigned row_stride = 1920) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int blocks = 512, stages = 20000;
  float best = 1e30f, sum = 0;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, src, out, rep == 0 ? 2000 : stages, src_bytes, row_stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep) { best = ms < best ? ms : best; sum += ms; }
  }
  const double tf = flop_per_stage_block * blocks * stages / (sum / 5) / 1e9;
  printf("%-58s mean %.2f ms  %.1f TFLOP/s(-equivalent)  = %.3f of 157.3   (best %.1f)\n", label, sum / 5, tf, tf / 157.3,
         flop_per_stage_block * blocks * stages / best / 1e9);
}

int main() {
  const unsigned src_bytes = 16u << 20;      // L2 / Infinity-Cache resident
  const unsigned big_bytes = 1u << 30;       // streams from HBM
  std::vector<float> h(src_bytes / 4);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  // GEMM_CEILING_S3=1: fill the source with genuine S3 data (hi / mid / lo bfloat16 terms of N(0, 1) values, 96-B blocks) instead of
  // raw float bits -- what the pre-split kernels really multiply; the clock the chip holds depends on the operand values
  if (getenv("GEMM_CEILING_S3")) {
    auto bf = [](float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); };
    auto fb = [](unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; };
    unsigned short* q = reinterpret_cast<unsigned short*>(h.data());
    const size_t blocks = (size_t)src_bytes / 96;
    for (size_t b = 0; b < blocks; ++b)
      for (int e = 0; e < 16; ++e) {
        float u1 = ((float)rand() + 1.f) / ((float)RAND_MAX + 2.f), u2 = (float)rand() / RAND_MAX;
        float x = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
        unsigned short hi = bf(x); float r1 = x - fb(hi);
        unsigned short mid = bf(r1); unsigned short lo = bf(r1 - fb(mid));
        q[b * 48 + e] = hi; q[b * 48 + 16 + e] = mid; q[b * 48 + 32 + e] = lo;
      }
  }
  float *src, *out;
  hipMalloc(&src, big_bytes);
  hipMalloc(&out, 4);
  for (unsigned o = 0; o < big_bytes / src_bytes; ++o) hipMemcpy(reinterpret_cast<char*>(src) + (size_t)o * src_bytes, h.data(), src_bytes, hipMemcpyHostToDevice);
  const double f = 2.0 * 128 * 128 * 16;
  const size_t lds = 3 * kSlot * 4, lds3 = 3 * kRows * 96;
  run("fp32 constant operands, registers only (mode 12)", loop_f32<12>, lds, src, out, src_bytes, f);
  run("fp32 random operands, registers only (mode 8)", loop_f32<8>, lds, src, out, src_bytes, f);
  run("fp32 random, LDS fragment reads (mode 0)", loop_f32<0>, lds, src, out, src_bytes, f);
  run("fp32 random, fragment reads + barrier per stage (mode 1)", loop_f32<1>, lds, src, out, src_bytes, f);
  run("fp32 random, reads + barrier + stage loads / writes (mode 3)", loop_f32<3>, lds, src, out, src_bytes, f);
  run("  the same, 1 GiB source (HBM)", loop_f32<3>, lds, src, out, big_bytes, f);
  run("  the same, gathered 64-B row slabs, 16 MiB, stride 1920", (loop_f32<3, 1>), lds, src, out, src_bytes, f, 1920);
  run("  the same, gathered 64-B row slabs, 1 GiB, stride 1920", (loop_f32<3, 1>), lds, src, out, big_bytes, f, 1920);
  run("  the same, gathered 64-B row slabs, 1 GiB, stride 768", (loop_f32<3, 1>), lds, src, out, big_bytes, f, 768);
  run("  the same, gathered 64-B row slabs, 1 GiB, stride 384", (loop_f32<3, 1>), lds, src, out, big_bytes, f, 384);
  run("  the same, gathered 64-B row slabs, 1 GiB, stride 5120", (loop_f32<3, 1>), lds, src, out, big_bytes, f, 5120);
  run("  the same, gathered 64-B row slabs, 16 MiB, stride 5120", (loop_f32<3, 1>), lds, src, out, src_bytes, f, 5120);
  run("  the same, gathered 64-B row slabs, 1 GiB, stride 17280", (loop_f32<3, 1>), lds, src, out, big_bytes, f, 17280);
  run("  the same, gathered 64-B row slabs, 16 MiB, stride 4096", (loop_f32<3, 1>), lds, src, out, src_bytes, f, 4096);
  run("bf16x3 random, registers only (mode 8)", loop_bf3<8>, lds3, src, out, src_bytes, f);
  run("bf16x3 random, LDS fragment reads (mode 0)", loop_bf3<0>, lds3, src, out, src_bytes, f);
  run("bf16x3 random, fragment reads + barrier per stage (mode 1)", loop_bf3<1>, lds3, src, out, src_bytes, f);
  // pre-split operands + direct-to-LDS staging: activations rows of 480 channels (2880 B), 1 GiB footprint
  run_t("bf16x3 DMA 128x128 / 4 waves (64x64), frag dbuf, 2 WG/CU", (loop_bf3_dma<2, 2, 2, 2, true>), 256, 512, 3 * 256 * 96, src, out, big_bytes, 2.0 * 128 * 128 * 16, 2880);
  run_t("bf16x3 DMA 128x128 / 4 waves (64x64), single frag, 2 WG/CU", (loop_bf3_dma<2, 2, 2, 2, false>), 256, 512, 3 * 256 * 96, src, out, big_bytes, 2.0 * 128 * 128 * 16, 2880);
  run_t("bf16x3 DMA 256x128 / 8 waves (64x64), frag dbuf, 1 WG/CU", (loop_bf3_dma<4, 2, 2, 2, true>), 512, 256, 3 * 384 * 96, src, out, big_bytes, 2.0 * 256 * 128 * 16, 2880);
  run_t("bf16x3 DMA 256x256 / 8 waves (64x128), single frag, 1 WG/CU", (loop_bf3_dma<4, 2, 2, 4, false>), 512, 256, 3 * 512 * 96, src, out, big_bytes, 2.0 * 256 * 256 * 16, 2880);
  run_t("bf16x3 DMA 128x256 / 4 waves (64x128), single frag, 1-2 WG/CU", (loop_bf3_dma<2, 2, 2, 4, false>), 256, 512, 3 * 384 * 96, src, out, big_bytes, 2.0 * 128 * 256 * 16, 2880);
  run_t("bf16x3 DMA 128x128, 16 MiB source", (loop_bf3_dma<2, 2, 2, 2, true>), 256, 512, 3 * 256 * 96, src, out, src_bytes, 2.0 * 128 * 128 * 16, 2880);
  run_t("bf16x3 DMA 128x128 / 4 waves, GEMM-grid reuse", (loop_bf3_dma<2, 2, 2, 2, true>), 256, 512, 3 * 256 * 96, src, out, big_bytes, 2.0 * 128 * 128 * 16, 0);
  run_t("bf16x3 DMA 256x128 / 8 waves, GEMM-grid reuse", (loop_bf3_dma<4, 2, 2, 2, true>), 512, 256, 3 * 384 * 96, src, out, big_bytes, 2.0 * 256 * 128 * 16, 0);
  run_t("bf16x3 DMA 256x256 / 8 waves (64x128), GEMM-grid reuse", (loop_bf3_dma<4, 2, 2, 4, false>), 512, 256, 3 * 512 * 96, src, out, big_bytes, 2.0 * 256 * 256 * 16, 0);
  run_t("bf16x3 DMA 128x256 / 4 waves (64x128), GEMM-grid reuse", (loop_bf3_dma<2, 2, 2, 4, false>), 256, 256, 3 * 384 * 96, src, out, big_bytes, 2.0 * 128 * 256 * 16, 0);
  run_t("bf16x3 HALO 256x128 / 8 waves, 9 taps per patch, GEMM grid", (loop_bf3_halo<4, 2, 2, 2, true, 9, 98>), 512, 256, 2 * (256 + 98) * 96 + 3 * 128 * 96, src, out, big_bytes, 2.0 * 256 * 128 * 16, 0);
  run_t("bf16x3 HALO 256x256 / 8 waves, 9 taps per patch, GEMM grid", (loop_bf3_halo<4, 2, 2, 4, false, 9, 98>), 512, 256, 2 * (256 + 98) * 96 + 3 * 256 * 96, src, out, big_bytes, 2.0 * 256 * 256 * 16, 0);
  run_t("bf16x3 HALO 128x128 / 4 waves, 9 taps per patch, GEMM grid", (loop_bf3_halo<2, 2, 2, 2, true, 9, 98>), 256, 512, 2 * (128 + 98) * 96 + 3 * 128 * 96, src, out, big_bytes, 2.0 * 128 * 128 * 16, 0);
  return 0;
}
