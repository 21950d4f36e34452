// rans.hip -- the bitstream behind the rate estimates: table-driven, 64-way interleaved rANS.
//
// The reference always runs its entropy models with compression=False (mshyper/models.py:246-251) and reports
// the *estimated* rate; this coder is this build's own wire format, not TFC's.
//
// One stream per (image, segment); ONE WAVE codes a stream: the segment's elements (flat [P, C] order) are dealt
// round-robin to L = 8 .. 64 lanes (step j, lane l <-> element L j + l, so table-id / value accesses are one coalesced
// line per step), every lane owns a 32-bit rANS state and all lanes share one sequence of 16-bit words:
//   decode step:  s = x & 0xffff -> (symbol, f, c);  x = f (x >> 16) + s - c;  lanes with x < 2^16 each take ONE
//                 word, in lane order, from the shared read pointer (wave ballot + popcount of lower lanes);
//                 lanes whose symbol was ESCAPE then read the value: v = (x & 0xffff) - 32768, x >>= 16, and
//                 refill the same way (a second, usually empty, sub-step).
//   encode step:  the exact mirror, steps in reverse, writing backward: ESCAPE lanes emit x & 0xffff and set
//                 x = (x & ~0xffff) | (v + 32768); then lanes with x >= f << 16 emit x & 0xffff, x >>= 16;
//                 x = ((x / f) << 16) + x % f + c.  The highest lane gets the highest address.
//   stream     =  [L x (state hi, state lo)] [words in decode order]; every state starts (encoder) and must end
//                 (decoder) at 2^16, and the read pointer must end at the stream's length: a free integrity check.
// Cost of the parallelism: 4 L bytes of flushed state per stream (the host uses fewer lanes on short streams).  Probability precision 16 bits; CDF tables are
// uint16 (cdf[n] = 65536 implicit) and live in LDS next to a packed (offset, n, vmin) descriptor per table.
#include <algorithm>
#include <type_traits>
#include "sntc_internal.h"

namespace sntc {

struct RansTables {
  const unsigned short* cdf;   // concatenated; table t: cdf[off .. off + n), cdf of symbol n (= 65536) implicit
  const uint2* meta;           // per table: x = off, y = (n << 16) | (vmin & 0xffff); symbol n-1 is ESCAPE
  int ntables;
  int total;                   // entries in cdf
};

// The decoder's own view of the same tables (optional: sntc_rans_decode's dec / lut arguments, built by the host from cdf):
//   dec  entry s of table t, at dec[off_t + 3 t + s] = (cdf[s] << 16) | (freq[s] - 1), followed by three 0xffffffff -- ONE read
//        gives a symbol's (start, frequency), and with key = (slot << 16) | 0xfffe, "key >= entry" is "slot >= cdf[s]" for every
//        real entry (freq - 1 <= 0xfffe: a table has >= 2 symbols) and false for the sentinels;
//   lut  per table a START TABLE of 2^bits entries, lut[lut_off + (slot >> (16 - bits))] = the largest symbol whose cdf is <= the
//        first slot of that bucket: the search starts there.  lmeta[t] = (lut_off << 5) | bits.
struct RansDecTables {
  const unsigned* dec;
  const unsigned short* lut;
  const uint2* meta;           // RansTables::meta
  const unsigned* lmeta;
  int ntables;
  int dec_total;               // entries in dec: total + 3 * ntables, padded to a multiple of 4
  int lut_total;               // entries in lut, padded to a multiple of 8
};

// Staging.  A wave that codes one stream has nobody to hide memory latency behind, so nothing in the coding loops
// touches global memory for input: table ids (and values / stream words) are fetched a CHUNK of 16 steps ahead into
// registers and dropped into LDS rings when the chunk ends; the loops read LDS only, one to two steps ahead of use.
constexpr int kChunk = 16;                        // steps per staging chunk (1024 elements)
constexpr int kWordRing = 4096;                   // decoder: stream words resident in LDS (2 x the most a chunk can eat)
constexpr int kWordRegs = 2 * kChunk;             // decoder: words one lane fetches per chunk
constexpr int kStagingBytes = 2 * kChunk * 64 * 2 + 2 * kChunk * 64 * 4;   // encoder: id + value rings; decoder: id + word rings
static_assert(2 * kChunk * 64 * 2 + kWordRing * 2 <= kStagingBytes, "decoder rings must fit the staging area");
constexpr int kRansLdsLimit = 150 * 1024 - kStagingBytes;
constexpr unsigned short kNoTable = 0xffffu;      // ring entry of a lane with no element in that step

template <bool LDS>
__device__ __forceinline__ void rans_stage_tables(const RansTables& T, unsigned char* smem, const uint2*& meta,
                                                  const unsigned short*& cdf) {
  if (LDS) {
    uint2* m = reinterpret_cast<uint2*>(smem);
    unsigned short* c = reinterpret_cast<unsigned short*>(smem + (size_t)T.ntables * sizeof(uint2));
    for (int i = threadIdx.x; i < T.ntables; i += blockDim.x) m[i] = T.meta[i];
    const unsigned* src = reinterpret_cast<const unsigned*>(T.cdf);   // host pads the table to an even count
    unsigned* dst = reinterpret_cast<unsigned*>(c);
    for (int i = threadIdx.x; i < (T.total + 1) / 2; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    meta = m;
    cdf = c;
  } else {
    meta = T.meta;
    cdf = T.cdf;
  }
}

// one wave per stream s = (image b, segment sg): elements [b E + sg Eseg, min((b + 1) E, b E + (sg + 1) Eseg))
template <bool LDS>
__global__ void __launch_bounds__(64) rans_encode_kernel(const int* __restrict__ values, const unsigned short* __restrict__ tid,
                                                         int segs, int L, long long E, long long Eseg, RansTables T, long long cap,
                                                         unsigned short* __restrict__ scratch, int* __restrict__ len_words) {
  extern __shared__ unsigned char smem[];
  unsigned short* tring = reinterpret_cast<unsigned short*>(smem);                   // [2][kChunk * 64]
  int* vring = reinterpret_cast<int*>(smem + 2 * kChunk * 64 * 2);                   // [2][kChunk * 64]
  const uint2* meta;
  const unsigned short* cdf;
  rans_stage_tables<LDS>(T, smem + kStagingBytes, meta, cdf);
  const int s = blockIdx.x, lane = threadIdx.x;
  const int b = s / segs, sg = s - b * segs;
  const long long e0 = (long long)b * E + (long long)sg * Eseg;
  const long long e1 = std::min((long long)(b + 1) * E, e0 + Eseg);
  const long long steps = e1 > e0 ? (e1 - e0 + L - 1) / L : 0;      // L = lanes in use (8 .. 64): short streams flush fewer states
  const long long nchunks = (steps + kChunk - 1) / kChunk;
  unsigned short* out = scratch + (size_t)s * cap;
  long long wp = cap;
  const unsigned long long gt = lane == 63 ? 0ull : (~0ull << (lane + 1));
  unsigned x = 1u << 16;

  unsigned short TR[kChunk];
  int VR[kChunk];
  auto fetch = [&](long long k) {
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
      const long long e = e0 + (k * kChunk + i) * L + lane;
      const bool a = k >= 0 && lane < L && e < e1;
      TR[i] = a ? tid[e] : kNoTable;
      VR[i] = a ? values[e] : 0;
    }
  };
  auto spill = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
      tring[buf * kChunk * 64 + i * 64 + lane] = TR[i];
      vring[buf * kChunk * 64 + i * 64 + lane] = VR[i];
    }
  };
  // everything that depends only on the symbol (descriptor, ESCAPE test, f, c) runs two steps ahead of the state
  // arithmetic; the loop-carried chain is: escape emit, renormalise, divide
  struct Sym { unsigned f, c, raw; bool active, esc; };
  auto lookup = [&](int t, int v, uint2 m) -> Sym {
    const bool active = t != kNoTable;
    const int n = (int)(m.y >> 16), vmin = (int)(short)(m.y & 0xffffu);
    int sym = v - vmin;
    const bool esc = active && (sym < 0 || sym >= n - 1);
    if (esc) sym = n - 1;
    if (!active) sym = 0;
    const unsigned cl = cdf[m.x + sym];
    const unsigned ch = sym + 1 < n ? (unsigned)cdf[m.x + sym + 1] : 65536u;
    return Sym{active ? ch - cl : 1u, cl, (unsigned)(std::min(std::max(v, -32768), 32767) + 32768), active, esc};
  };
  if (nchunks > 0) {
    fetch(nchunks - 1);
    spill((int)((nchunks - 1) & 1));
  }
  for (long long k = nchunks - 1; k >= 0; --k) {
    fetch(k - 1);
    const int buf = (int)(k & 1);
    const unsigned short* tb = tring + buf * kChunk * 64 + lane;
    const int* vb = vring + buf * kChunk * 64 + lane;
    const int cnt = (int)std::min<long long>(kChunk, steps - k * kChunk);
    auto rd = [&](int i, int& t, int& v) {
      const int ii = std::max(i, 0);
      t = tb[ii * 64];
      v = vb[ii * 64];
      if (i < 0) t = kNoTable;
    };
    int t1, v1, t2, v2;
    rd(cnt - 1, t1, v1);
    Sym cur = lookup(t1, v1, meta[t1 == kNoTable ? 0 : t1]);
    rd(cnt - 2, t1, v1);
    uint2 m1 = meta[t1 == kNoTable ? 0 : t1];
    rd(cnt - 3, t2, v2);
    for (int i = cnt - 1; i >= 0; --i) {
      const Sym sy = cur;
      cur = lookup(t1, v1, m1);                             // step i - 1
      t1 = t2;
      v1 = v2;
      m1 = meta[t1 == kNoTable ? 0 : t1];                   // step i - 2
      rd(i - 3, t2, v2);
      const unsigned long long emask = __ballot(sy.esc);
      if (emask) {                                          // value first (reverse order), then the ESCAPE symbol
        if (sy.esc) {
          out[wp - 1 - __popcll(emask & gt)] = (unsigned short)(x & 0xffffu);
          x = (x & 0xffff0000u) | sy.raw;
        }
        wp -= __popcll(emask);
      }
      const bool need = sy.active && (unsigned long long)x >= ((unsigned long long)sy.f << 16);
      const unsigned long long mask = __ballot(need);
      if (need) {
        out[wp - 1 - __popcll(mask & gt)] = (unsigned short)(x & 0xffffu);
        x >>= 16;
      }
      wp -= __popcll(mask);
      if (sy.active) x = ((x / sy.f) << 16) + (x % sy.f) + sy.c;
    }
    spill((int)((k - 1) & 1));
  }
  wp -= 2 * L;
  if (lane < L) {
    out[wp + 2 * lane] = (unsigned short)(x >> 16);
    out[wp + 2 * lane + 1] = (unsigned short)(x & 0xffffu);
  }
  if (lane == 0) len_words[s] = (int)(cap - wp);
}

__global__ void rans_compact_kernel(const unsigned short* __restrict__ src, long long cap, const int* __restrict__ len_words,
                                    const long long* __restrict__ offsets, int nstreams, unsigned short* __restrict__ dst) {
  const int s = blockIdx.x;
  if (s >= nstreams) return;
  const int len = len_words[s];
  const unsigned short* from = src + (size_t)(s + 1) * cap - len;
  unsigned short* to = dst + offsets[s];
  for (int i = threadIdx.x; i < len; i += blockDim.x) to[i] = from[i];
}

template <bool LDS>
__global__ void __launch_bounds__(64) rans_decode_kernel(const unsigned short* __restrict__ payload, const long long* __restrict__ offsets,
                                                         const unsigned short* __restrict__ tid, int segs, int L, long long E, long long Eseg,
                                                         RansTables T, int* __restrict__ values, int* __restrict__ bad) {
  extern __shared__ unsigned char smem[];
  unsigned short* tring = reinterpret_cast<unsigned short*>(smem);                   // [2][kChunk * 64]
  unsigned short* wring = tring + 2 * kChunk * 64;                                   // [kWordRing]
  const uint2* meta;
  const unsigned short* cdf;
  rans_stage_tables<LDS>(T, smem + kStagingBytes, meta, cdf);
  const int s = blockIdx.x, lane = threadIdx.x;
  const int b = s / segs, sg = s - b * segs;
  const long long e0 = (long long)b * E + (long long)sg * Eseg;
  const long long e1 = std::min((long long)(b + 1) * E, e0 + Eseg);
  const long long steps = e1 > e0 ? (e1 - e0 + L - 1) / L : 0;
  const long long nchunks = (steps + kChunk - 1) / kChunk;
  const unsigned short* w = payload + offsets[s];
  const long long len = offsets[s + 1] - offsets[s];
  if (len < 2 * L) {                                         // not even the lane states: malformed
    if (lane == 0) atomicAdd(bad, 1);
    return;
  }
  unsigned x = lane < L ? (((unsigned)w[2 * lane] << 16) | w[2 * lane + 1]) : (1u << 16);
  bool ok = true;
  const unsigned long long lt = (1ull << lane) - 1ull;

  // stream words: the ring holds [ptr, ptr + kWordRing) at the start of a chunk; a chunk eats at most kWordRing / 2
  int ptr = 2 * L, filled = 2 * L, wfrom = 2 * L, wto = 2 * L;  // stream positions fit 32 bits (host: cap_words < 2^30)
  const int len32 = (int)len;
  unsigned short WR[kWordRegs];
  auto word_fetch = [&](int target) {
    wfrom = filled;
    wto = std::min(target, len32);
#pragma unroll
    for (int i = 0; i < kWordRegs; ++i) {
      const int q = wfrom + i * 64 + lane;
      WR[i] = q < wto ? w[q] : (unsigned short)0;
    }
  };
  auto word_spill = [&]() {
#pragma unroll
    for (int i = 0; i < kWordRegs; ++i) {
      const int q = wfrom + i * 64 + lane;
      if (q < wto) wring[q & (kWordRing - 1)] = WR[i];
    }
    filled = std::max(filled, wto);
  };
  unsigned short TR[kChunk];
  auto tid_fetch = [&](long long k) {
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
      const long long e = e0 + (k * kChunk + i) * L + lane;
      TR[i] = (k < nchunks && lane < L && e < e1) ? tid[e] : kNoTable;
    }
  };
  auto tid_spill = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kChunk; ++i) tring[buf * kChunk * 64 + i * 64 + lane] = TR[i];
  };
  word_fetch(ptr + kWordRing / 2);
  word_spill();
  word_fetch(ptr + kWordRing);
  word_spill();
  tid_fetch(0);
  tid_spill(0);

  for (long long k = 0; k < nchunks; ++k) {
    tid_fetch(k + 1);
    word_fetch(ptr + kWordRing);
    const unsigned short* tb = tring + (int)(k & 1) * kChunk * 64 + lane;
    const int cnt = (int)std::min<long long>(kChunk, steps - k * kChunk);
    int* vout = values + e0 + k * kChunk * L + lane;
    int t1 = tb[0];
    uint2 m1 = meta[t1 == kNoTable ? 0 : t1];
    int t2 = cnt > 1 ? (int)tb[64] : (int)kNoTable;
    for (int i = 0; i < cnt; ++i) {
      const bool active = t1 != kNoTable;
      const uint2 m = m1;
      t1 = t2;
      m1 = meta[t1 == kNoTable ? 0 : t1];                    // descriptor of step i + 1
      t2 = tb[std::min(i + 2, kChunk - 1) * 64];             // table id of step i + 2
      if (i + 2 >= cnt) t2 = kNoTable;
      const int n = active ? (int)(m.y >> 16) : 1, vmin = (int)(short)(m.y & 0xffffu);
      const unsigned short* c = cdf + m.x;
      const unsigned slot = x & 0xffffu;
      // binary search for the largest sym with cdf[sym] <= slot; the bracketing cdf values are carried along so that no
      // read follows the search.  (A lone wave is issue-bound: the 4-ary, three-probes-per-level form needs fewer LDS
      // round trips but ~4x the instructions per level and measured slower.)
      int lo = 0, hi = n;
      unsigned clo = 0u, chi = 65536u;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        const unsigned v = c[mid];
        const bool ge = slot >= v;
        lo = ge ? mid : lo;
        clo = ge ? v : clo;
        hi = ge ? hi : mid;
        chi = ge ? chi : v;
      }
      if (active) x = (chi - clo) * (x >> 16) + slot - clo;
      const bool need = active && x < (1u << 16);
      const unsigned long long mask = __ballot(need);
      if (mask) {                                            // one word each, in lane order, from the shared pointer
        if (need) {
          const int q = ptr + __popcll(mask & lt);
          ok &= q < len32;
          x = (x << 16) | wring[q & (kWordRing - 1)];
        }
        ptr += __popcll(mask);
      }
      int v = lo + vmin;
      const bool esc = active && lo == n - 1;
      const unsigned long long emask = __ballot(esc);
      if (emask) {
        if (esc) {
          const int q = ptr + __popcll(emask & lt);
          ok &= q < len32;
          v = (int)(x & 0xffffu) - 32768;
          x = (x & 0xffff0000u) | wring[q & (kWordRing - 1)];
        }
        ptr += __popcll(emask);
      }
      if (active) vout[i * L] = v;
    }
    word_spill();
    tid_spill((int)((k + 1) & 1));
  }
  // a well-formed stream ends with every state back at its initial value and the pointer at the end
  const bool good = ok && x == (1u << 16) && ptr == len32;
  if (__ballot(!good) && lane == 0) atomicAdd(bad, 1);
}

// The same decoder with the RansDecTables in LDS.  A lone wave per stream is bound by its chain of dependent LDS round trips and
// by its own instruction issue (nobody to interleave with), so a step is: start symbol from the start table (1 read), four
// consecutive entries from there (independent reads; the symbol is among the first three in > 98 % of the slots of every table,
// else the round repeats from the fourth), refill word (1 read) -- three round trips instead of the binary search's
// log2(n) + 2 (11 + 2 on the widest scale table, and a wave waits for its slowest lane) -- and selects instead of divergent
// branches.  Decoded values are those of rans_decode_kernel, bit for bit (tests/test_hip_bitstream.py).
__global__ void __launch_bounds__(64) rans_decode_fast_kernel(const unsigned short* __restrict__ payload, const long long* __restrict__ offsets,
                                                              const unsigned short* __restrict__ tid, int segs, int L, long long E,
                                                              long long Eseg, RansDecTables T, int* __restrict__ values,
                                                              int* __restrict__ bad) {
  extern __shared__ unsigned char smem[];
  unsigned short* tring = reinterpret_cast<unsigned short*>(smem);                   // [2][kChunk * 64]
  unsigned short* wring = tring + 2 * kChunk * 64;                                   // [kWordRing]
  uint4* m4 = reinterpret_cast<uint4*>(smem + kStagingBytes);                        // {entry offset, n | vmin, lut offset, 16 - bits}
  unsigned* dec = reinterpret_cast<unsigned*>(m4 + T.ntables);
  unsigned short* lut = reinterpret_cast<unsigned short*>(dec + T.dec_total);
  const int lane = threadIdx.x;
  for (int i = lane; i < T.ntables; i += 64) {
    const uint2 a = T.meta[i];
    const unsigned lm = T.lmeta[i];
    m4[i] = make_uint4(a.x + 3u * (unsigned)i, a.y, lm >> 5, 16u - (lm & 31u));
  }
  {
    const uint4* src = reinterpret_cast<const uint4*>(T.dec);
    uint4* dst = reinterpret_cast<uint4*>(dec);
    for (int i = lane; i < T.dec_total / 4; i += 64) dst[i] = src[i];
    src = reinterpret_cast<const uint4*>(T.lut);
    dst = reinterpret_cast<uint4*>(lut);
    for (int i = lane; i < T.lut_total / 8; i += 64) dst[i] = src[i];
  }
  __syncthreads();
  const int s = blockIdx.x;
  const int b = s / segs, sg = s - b * segs;
  const long long e0 = (long long)b * E + (long long)sg * Eseg;
  const long long e1 = std::min((long long)(b + 1) * E, e0 + Eseg);
  const long long steps = e1 > e0 ? (e1 - e0 + L - 1) / L : 0;
  const long long nchunks = (steps + kChunk - 1) / kChunk;
  const unsigned short* w = payload + offsets[s];
  const long long len = offsets[s + 1] - offsets[s];
  if (len < 2 * L) {                                         // not even the lane states: malformed
    if (lane == 0) atomicAdd(bad, 1);
    return;
  }
  unsigned x = lane < L ? (((unsigned)w[2 * lane] << 16) | w[2 * lane + 1]) : (1u << 16);

  int ptr = 2 * L, filled = 2 * L, wfrom = 2 * L, wto = 2 * L;  // as rans_decode_kernel: the ring holds [ptr, ptr + kWordRing)
  const int len32 = (int)len;
  unsigned short WR[kWordRegs];
  auto word_fetch = [&](int target) {
    wfrom = filled;
    wto = std::min(target, len32);
#pragma unroll
    for (int i = 0; i < kWordRegs; ++i) {
      const int q = wfrom + i * 64 + lane;
      WR[i] = q < wto ? w[q] : (unsigned short)0;
    }
  };
  auto word_spill = [&]() {
#pragma unroll
    for (int i = 0; i < kWordRegs; ++i) {
      const int q = wfrom + i * 64 + lane;
      if (q < wto) wring[q & (kWordRing - 1)] = WR[i];
    }
    filled = std::max(filled, wto);
  };
  unsigned short TR[kChunk];
  auto tid_fetch = [&](long long k) {
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
      const long long e = e0 + (k * kChunk + i) * L + lane;
      TR[i] = (k < nchunks && lane < L && e < e1) ? tid[e] : kNoTable;
    }
  };
  auto tid_spill = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kChunk; ++i) tring[buf * kChunk * 64 + i * 64 + lane] = TR[i];
  };
  word_fetch(ptr + kWordRing / 2);
  word_spill();
  word_fetch(ptr + kWordRing);
  word_spill();
  tid_fetch(0);
  tid_spill(0);

  // One step.  FULL: every lane has an element in every step of the chunk (all chunks but a stream's last, with 64 lanes) -- no
  // per-lane "active" selects, no store predicate.  A word index at or past the stream's end is not checked here: the
  // pointer only grows, so it ends past the length and the stream is counted as bad below; the ring read itself is masked.
  auto step = [&](auto full_tag, const unsigned short* tb, int* vout, int i, int cnt, int& t1, int& t2, uint4& m1) {
    constexpr bool FULL = decltype(full_tag)::value;
    const bool active = FULL || t1 != kNoTable;
    const uint4 m = m1;
    t1 = t2;
    m1 = m4[(!FULL && t1 == kNoTable) ? 0 : t1];             // descriptor of step i + 1
    t2 = tb[std::min(i + 2, kChunk - 1) * 64];               // table id of step i + 2 (FULL: past the chunk a repeat, unused)
    if (!FULL && i + 2 >= cnt) t2 = kNoTable;
    const unsigned slot = x & 0xffffu, key = (x << 16) | 0xfffeu;
    const unsigned* e = dec + m.x;
    unsigned lo = lut[m.z + (slot >> m.w)];
    unsigned esel;
    for (;;) {
      const unsigned c0 = e[lo], c1 = e[lo + 1], c2 = e[lo + 2], c3 = e[lo + 3];
      asm volatile("" ::"v"(c0), "v"(c1), "v"(c2), "v"(c3));   // all four in registers here: issued together, none deferred into a branch
      const bool g1 = key >= c1, g2 = key >= c2, g3 = key >= c3;
      esel = g2 ? c2 : (g1 ? c1 : c0);
      lo += (g1 ? 1u : 0u) + (g2 ? 1u : 0u);
      if (!__ballot(g3)) break;                              // a lane that has its symbol finds it again: c0 = its entry, g1 false
      lo += g3 ? 1u : 0u;
    }
    const unsigned xn = ((esel & 0xffffu) + 1u) * (x >> 16) + slot - (esel >> 16);
    x = active ? xn : x;                                     // a lane without an element keeps its state (always >= 2^16)
    const bool need = x < (1u << 16);
    const unsigned long long mask = __ballot(need);          // one word each, in lane order, from the shared pointer
    {
      const int q = ptr + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
      const unsigned word = wring[q & (kWordRing - 1)];
      x = need ? ((x << 16) | word) : x;
      ptr += __popcll(mask);
    }
    const int n = (int)(m.y >> 16), vmin = (int)(short)(m.y & 0xffffu);
    int v = (int)lo + vmin;
    const bool esc = active && (int)lo == n - 1;
    const unsigned long long emask = __ballot(esc);
    if (emask) {
      const int q = ptr + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(emask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)emask, 0u));
      const unsigned word = wring[q & (kWordRing - 1)];
      v = esc ? (int)(x & 0xffffu) - 32768 : v;
      x = esc ? ((x & 0xffff0000u) | word) : x;
      ptr += __popcll(emask);
    }
    if (active) vout[(long long)i * L] = v;
  };

  for (long long k = 0; k < nchunks; ++k) {
    tid_fetch(k + 1);
    word_fetch(ptr + kWordRing);
    const unsigned short* tb = tring + (int)(k & 1) * kChunk * 64 + lane;
    const int cnt = (int)std::min<long long>(kChunk, steps - k * kChunk);
    int* vout = values + e0 + k * kChunk * L + lane;
    int t1 = tb[0];
    uint4 m1 = m4[t1 == kNoTable ? 0 : t1];
    int t2 = cnt > 1 ? (int)tb[64] : (int)kNoTable;
    if (L == 64 && e0 + (k + 1) * kChunk * 64 <= e1) {
#pragma unroll
      for (int i = 0; i < kChunk; ++i) step(std::true_type{}, tb, vout, i, kChunk, t1, t2, m1);
    } else {
      for (int i = 0; i < cnt; ++i) step(std::false_type{}, tb, vout, i, cnt, t1, t2, m1);
    }
    word_spill();
    tid_spill((int)((k + 1) & 1));
  }
  const bool good = x == (1u << 16) && ptr == len32;
  if (__ballot(!good) && lane == 0) atomicAdd(bad, 1);
}

// indexes = round(clamp(exp(raw), 0, 63)) as the table id of every y element (the integer scale table TFC's
// compress() path uses); raw = hyper[..., C:]
__global__ void scale_index_kernel(const float* __restrict__ hyper, long long npix, int c, unsigned short* __restrict__ tid) {
  const long long total = npix * c;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / c;
    const int ch = (int)(i - p * c);
    const float idx = fminf(fmaxf(expf(hyper[p * 2 * c + c + ch]), 0.0f), 63.0f);
    tid[i] = (unsigned short)rintf(idx);
  }
}

__global__ void channel_index_kernel(long long npix, int c, unsigned short* __restrict__ tid) {
  const long long total = npix * c;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    tid[i] = (unsigned short)(i % c);
}

__global__ void round_to_int_kernel(const float* __restrict__ x, long long total, int* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    out[i] = (int)rintf(x[i]);
}

__global__ void int_to_float_kernel(const int* __restrict__ x, long long total, float* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    out[i] = (float)x[i];
}

}  // namespace sntc

using namespace sntc;

static int blocks_for(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

static long long segment_elems(long long elems, int segments) {
  const long long per = (elems + segments - 1) / segments;
  return (per + 63) / 64 * 64;
}

static int table_bytes(int ntables, int total) { return ntables * (int)sizeof(uint2) + ((total + 1) / 2) * 4; }

extern "C" int64_t sntc_rans_cap_words(int64_t elems_per_image, int segments) {
  if (elems_per_image < 1 || segments < 1) return -1;
  return 2 * segment_elems(elems_per_image, segments) + 128;
}

static bool lanes_ok(int lanes) { return lanes == 8 || lanes == 16 || lanes == 32 || lanes == 64; }

extern "C" int sntc_rans_encode(const int32_t* values, const uint16_t* table_ids, int nimages, int64_t elems_per_image,
                                int segments, int lanes, const uint16_t* cdf, const uint32_t* meta, int ntables, int total_entries,
                                int64_t cap_words, uint16_t* scratch, int32_t* len_words, void* stream) {
  if (!values || !table_ids || !cdf || !meta || !scratch || !len_words)
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_rans_encode: null argument");
  if (nimages < 1 || elems_per_image < 1 || segments < 1 || !lanes_ok(lanes) || ntables < 1 || total_entries < 1 ||
      cap_words < sntc_rans_cap_words(elems_per_image, segments) || cap_words > 0x3fffffff)
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_rans_encode: bad sizes (cap_words must be >= sntc_rans_cap_words())");
  const RansTables T{cdf, reinterpret_cast<const uint2*>(meta), ntables, total_entries};
  const long long eseg = segment_elems(elems_per_image, segments);
  const int ns = nimages * segments, tb = table_bytes(ntables, total_entries), lds = kStagingBytes + tb;
  hipStream_t s = (hipStream_t)stream;
  if (tb <= kRansLdsLimit) {
    SNTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rans_encode_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(rans_encode_kernel<true>, dim3(ns), dim3(64), lds, s, values, table_ids, segments, lanes, (long long)elems_per_image,
                       eseg, T, (long long)cap_words, scratch, len_words);
  } else {
    hipLaunchKernelGGL(rans_encode_kernel<false>, dim3(ns), dim3(64), kStagingBytes, s, values, table_ids, segments, lanes, (long long)elems_per_image,
                       eseg, T, (long long)cap_words, scratch, len_words);
  }
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_rans_compact(const uint16_t* scratch, int64_t cap_words, const int32_t* len_words, const int64_t* offsets,
                                 int nstreams, uint16_t* payload, void* stream) {
  if (!scratch || !len_words || !offsets || !payload || nstreams < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_rans_compact: bad argument");
  hipLaunchKernelGGL(rans_compact_kernel, dim3(nstreams), dim3(256), 0, (hipStream_t)stream, scratch, (long long)cap_words, len_words,
                     reinterpret_cast<const long long*>(offsets), nstreams, payload);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

// LDS the decoder's own tables may take: what is left of a CU's next to the staging rings and the descriptors
extern "C" int64_t sntc_rans_lut_budget(int ntables, int total_entries) {
  if (ntables < 1 || total_entries < 1) return -1;
  const long long dec_bytes = 4LL * (((long long)total_entries + 3LL * ntables + 3) / 4 * 4);
  const long long rest = (long long)kRansLdsLimit - (long long)ntables * (long long)sizeof(uint4) - dec_bytes;
  return rest < 16 ? 0 : rest / 2 / 8 * 8;
}

extern "C" int sntc_rans_decode(const uint16_t* payload, const int64_t* offsets, const uint16_t* table_ids, int nimages,
                                int64_t elems_per_image, int segments, int lanes, const uint16_t* cdf, const uint32_t* meta,
                                int ntables, int total_entries, const uint32_t* dec, const uint16_t* lut, const uint32_t* lut_meta,
                                int lut_entries, int32_t* values, int32_t* bad_streams, void* stream) {
  if (!payload || !offsets || !table_ids || !cdf || !meta || !values || !bad_streams)
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_rans_decode: null argument");
  if (nimages < 1 || elems_per_image < 1 || segments < 1 || !lanes_ok(lanes) || ntables < 1 || total_entries < 1)
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_rans_decode: bad sizes");
  const bool fast = dec != nullptr;
  if (fast != (lut != nullptr) || fast != (lut_meta != nullptr) ||
      (fast && (lut_entries < ntables || (lut_entries & 7) || lut_entries > sntc_rans_lut_budget(ntables, total_entries))))
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_rans_decode: dec / lut / lut_meta come together, lut_entries a multiple of 8 within sntc_rans_lut_budget()");
  const long long eseg = segment_elems(elems_per_image, segments);
  const int ns = nimages * segments;
  hipStream_t s = (hipStream_t)stream;
  if (int zrc = zero_async(bad_streams, sizeof(int32_t), s)) return zrc;
  if (fast) {
    const int dec_total = (total_entries + 3 * ntables + 3) / 4 * 4;
    const RansDecTables D{dec, lut, reinterpret_cast<const uint2*>(meta), lut_meta, ntables, dec_total, lut_entries};
    const int lds = kStagingBytes + ntables * (int)sizeof(uint4) + dec_total * 4 + lut_entries * 2;
    SNTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rans_decode_fast_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(rans_decode_fast_kernel, dim3(ns), dim3(64), lds, s, payload, reinterpret_cast<const long long*>(offsets), table_ids,
                       segments, lanes, (long long)elems_per_image, eseg, D, values, bad_streams);
    SNTC_HIP(hipGetLastError());
    return SNTC_OK;
  }
  const RansTables T{cdf, reinterpret_cast<const uint2*>(meta), ntables, total_entries};
  const int tb = table_bytes(ntables, total_entries), lds = kStagingBytes + tb;
  if (tb <= kRansLdsLimit) {
    SNTC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rans_decode_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(rans_decode_kernel<true>, dim3(ns), dim3(64), lds, s, payload, reinterpret_cast<const long long*>(offsets),
                       table_ids, segments, lanes, (long long)elems_per_image, eseg, T, values, bad_streams);
  } else {
    hipLaunchKernelGGL(rans_decode_kernel<false>, dim3(ns), dim3(64), kStagingBytes, s, payload, reinterpret_cast<const long long*>(offsets),
                       table_ids, segments, lanes, (long long)elems_per_image, eseg, T, values, bad_streams);
  }
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_scale_table_ids(const float* hyper, int64_t npix, int c, uint16_t* table_ids, void* stream) {
  if (!hyper || !table_ids || npix < 1 || c < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_scale_table_ids: bad argument");
  hipLaunchKernelGGL(scale_index_kernel, dim3(blocks_for(npix * c)), dim3(256), 0, (hipStream_t)stream, hyper, (long long)npix, c, table_ids);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_channel_table_ids(int64_t npix, int c, uint16_t* table_ids, void* stream) {
  if (!table_ids || npix < 1 || c < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_channel_table_ids: bad argument");
  hipLaunchKernelGGL(channel_index_kernel, dim3(blocks_for(npix * c)), dim3(256), 0, (hipStream_t)stream, (long long)npix, c, table_ids);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_round_to_int(const float* x, int64_t total, int32_t* out, void* stream) {
  if (!x || !out || total < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_round_to_int: bad argument");
  hipLaunchKernelGGL(round_to_int_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, (long long)total, out);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_int_to_float(const int32_t* x, int64_t total, float* out, void* stream) {
  if (!x || !out || total < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_int_to_float: bad argument");
  hipLaunchKernelGGL(int_to_float_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, (long long)total, out);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}
