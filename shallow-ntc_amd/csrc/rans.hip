// rans.hip -- the bitstream behind the rate estimates: table-driven rANS with 16-bit
// quantised CDFs, one independent stream per (image, group of G channels) so that hundreds of streams code
// in parallel -- one thread per stream, symbols visited position-major / channel-minor.  G trades stream
// overhead (6 bytes each: u16 length + 32-bit final state) against parallelism.
//
// The reference always runs its entropy models with compression=False (mshyper/models.py:246-251) and
// reports the *estimated* rate; this coder is this build's own wire format, not TFC's.
// State x in [2^16, 2^32), 16-bit renormalisation, probability precision 16 bits:
//   encode (symbols in reverse):  if x >= f << 16: emit(x & 0xffff), x >>= 16;  x = ((x / f) << 16) + x % f + c
//   decode (forward):             s = x & 0xffff -> (symbol, f, c);  x = f (x >> 16) + s - c;  refill below 2^16
// The last symbol of every table is ESCAPE: it is followed by the value + 32768 coded as a uniform 16-bit symbol.
#include <algorithm>
#include "sntc_internal.h"

namespace sntc {

struct RansTables {
  const unsigned* cdf;   // concatenated, table t: cdf[off[t] .. off[t] + n[t]]  (n[t] + 1 entries, last = 65536)
  const int* off;
  const int* n;          // symbols incl. ESCAPE
  const int* vmin;       // value of symbol 0
};

__device__ __forceinline__ void rans_put(unsigned& x, unsigned f, unsigned c, unsigned short*& wp) {
  if ((unsigned long long)x >= ((unsigned long long)f << 16)) {
    *--wp = (unsigned short)(x & 0xffffu);
    x >>= 16;
  }
  x = ((x / f) << 16) + (x % f) + c;
}

// one thread per stream (image b, channel group grp of G channels); elements at ((b * P + p) * C + c),
// visited position-major, channel-minor inside the group
__global__ void __launch_bounds__(64) rans_encode_kernel(const int* __restrict__ values, const unsigned short* __restrict__ tid,
                                                         int nstreams, int P, int C, int G, RansTables T, int cap,
                                                         unsigned short* __restrict__ out, int* __restrict__ len_words) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nstreams) return;
  const int sg = (C + G - 1) / G;
  const int b = s / sg, c0 = (s - b * sg) * G, cn = min(G, C - c0);
  unsigned short* end = out + (size_t)(s + 1) * cap;
  unsigned short* wp = end;
  unsigned x = 1u << 16;
  for (int q = P * cn - 1; q >= 0; --q) {
    const int p = q / cn;
    const size_t e = ((size_t)b * P + p) * C + c0 + (q - p * cn);
    const int t = tid[e];
    const int n = T.n[t];
    const unsigned* cdf = T.cdf + T.off[t];
    const int v = values[e];
    int sym = v - T.vmin[t];
    if (sym < 0 || sym >= n - 1) {                       // escape: (reverse order) raw value first, then ESCAPE
      const unsigned raw = (unsigned)(min(max(v, -32768), 32767) + 32768);
      rans_put(x, 1u, raw, wp);
      sym = n - 1;
    }
    rans_put(x, cdf[sym + 1] - cdf[sym], cdf[sym], wp);
  }
  *--wp = (unsigned short)(x & 0xffffu);
  *--wp = (unsigned short)(x >> 16);
  len_words[s] = (int)(end - wp);
}

__global__ void rans_compact_kernel(const unsigned short* __restrict__ src, int cap, const int* __restrict__ len_words,
                                    const long long* __restrict__ offsets, int nstreams, unsigned short* __restrict__ dst) {
  const int s = blockIdx.x;
  if (s >= nstreams) return;
  const int len = len_words[s];
  const unsigned short* from = src + (size_t)(s + 1) * cap - len;
  unsigned short* to = dst + offsets[s];
  for (int i = threadIdx.x; i < len; i += blockDim.x) to[i] = from[i];
}

__global__ void __launch_bounds__(64) rans_decode_kernel(const unsigned short* __restrict__ payload, const long long* __restrict__ offsets,
                                                         const unsigned short* __restrict__ tid, int nstreams, int P, int C,
                                                         int G, RansTables T, int* __restrict__ values, int* __restrict__ bad) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nstreams) return;
  const int sg = (C + G - 1) / G;
  const int b = s / sg, c0 = (s - b * sg) * G, cn = min(G, C - c0);
  const unsigned short* rp = payload + offsets[s];
  const unsigned short* rend = payload + offsets[s + 1];
  unsigned x = ((unsigned)rp[0] << 16) | rp[1];
  rp += 2;
  bool ok = true;
  for (int q = 0; q < P * cn; ++q) {
    const int p = q / cn;
    const size_t e = ((size_t)b * P + p) * C + c0 + (q - p * cn);
    const int t = tid[e];
    const int n = T.n[t];
    const unsigned* cdf = T.cdf + T.off[t];
    const unsigned slot = x & 0xffffu;
    int lo = 0, hi = n;                                   // largest sym with cdf[sym] <= slot
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= slot) lo = mid; else hi = mid;
    }
    const unsigned cl = cdf[lo], f = cdf[lo + 1] - cl;
    x = f * (x >> 16) + slot - cl;
    if (x < (1u << 16)) { ok &= rp < rend; x = (x << 16) | (rp < rend ? *rp++ : 0); }
    int v = lo + T.vmin[t];
    if (lo == n - 1) {                                    // ESCAPE: uniform 16-bit value follows
      v = (int)(x & 0xffffu) - 32768;
      x >>= 16;
      if (x < (1u << 16)) { ok &= rp < rend; x = (x << 16) | (rp < rend ? *rp++ : 0); }
    }
    values[e] = v;
  }
  if (!ok || rp != rend || x != (1u << 16)) atomicAdd(bad, 1);   // a well-formed stream ends exactly at its initial state
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

static RansTables tables(const uint32_t* cdf, const int32_t* off, const int32_t* n, const int32_t* vmin) {
  return RansTables{cdf, off, n, vmin};
}

extern "C" int sntc_rans_encode(const int32_t* values, const uint16_t* table_ids, int nimages, int64_t positions, int channels,
                                int group, const uint32_t* cdf, const int32_t* tab_off, const int32_t* tab_n, const int32_t* tab_min,
                                int cap_words, uint16_t* scratch, int32_t* len_words, void* stream) {
  if (!values || !table_ids || !cdf || !tab_off || !tab_n || !tab_min || !scratch || !len_words)
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_rans_encode: null argument");
  if (nimages < 1 || positions < 1 || channels < 1 || group < 1 || cap_words < 2 * positions * std::min(group, channels) + 4)
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_rans_encode: bad sizes (cap_words must be >= 2 * positions * group + 4)");
  const int ns = nimages * ((channels + group - 1) / group);
  hipLaunchKernelGGL(rans_encode_kernel, dim3((ns + 63) / 64), dim3(64), 0, (hipStream_t)stream, values, table_ids, ns,
                     (int)positions, channels, group, tables(cdf, tab_off, tab_n, tab_min), cap_words, scratch, len_words);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_rans_compact(const uint16_t* scratch, int cap_words, const int32_t* len_words, const int64_t* offsets,
                                 int nstreams, uint16_t* payload, void* stream) {
  if (!scratch || !len_words || !offsets || !payload || nstreams < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_rans_compact: bad argument");
  hipLaunchKernelGGL(rans_compact_kernel, dim3(nstreams), dim3(64), 0, (hipStream_t)stream, scratch, cap_words, len_words,
                     reinterpret_cast<const long long*>(offsets), nstreams, payload);
  SNTC_HIP(hipGetLastError());
  return SNTC_OK;
}

extern "C" int sntc_rans_decode(const uint16_t* payload, const int64_t* offsets, const uint16_t* table_ids, int nimages,
                                int64_t positions, int channels, int group, const uint32_t* cdf, const int32_t* tab_off,
                                const int32_t* tab_n, const int32_t* tab_min, int32_t* values, int32_t* bad_streams,
                                void* stream) {
  if (!payload || !offsets || !table_ids || !cdf || !tab_off || !tab_n || !tab_min || !values || !bad_streams)
    return fail(SNTC_ERR_BAD_SHAPE, "sntc_rans_decode: null argument");
  if (nimages < 1 || positions < 1 || channels < 1 || group < 1) return fail(SNTC_ERR_BAD_SHAPE, "sntc_rans_decode: bad sizes");
  const int ns = nimages * ((channels + group - 1) / group);
  hipStream_t s = (hipStream_t)stream;
  SNTC_HIP(hipMemsetAsync(bad_streams, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(rans_decode_kernel, dim3((ns + 63) / 64), dim3(64), 0, s, payload, reinterpret_cast<const long long*>(offsets),
                     table_ids, ns, (int)positions, channels, group, tables(cdf, tab_off, tab_n, tab_min), values, bad_streams);
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
