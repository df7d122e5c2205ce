/*
 * xmpi_oracle.c -- CPU oracle (plain C).  TEST INFRASTRUCTURE ONLY; see xmpi_oracle.h.
 *
 * Collective semantics restated here are NOT reference code (the reference has none,
 * mpi.go:130); they are the rank-order composition over the reference's lossless
 * Send/Receive (network.go:518-625) spelled out in xmpi_oracle.h.  PARITY UNPINNED for the
 * collectives: no golden vector exists upstream.  The arithmetic helpers (half/bf16
 * conversion) are cross-checked against numpy in tests/test_oracle.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 */
#include "xmpi_oracle.h"

#include <math.h>
#include <string.h>

size_t oracle_dtype_size(int dtype) {
  switch (dtype) {
    case OR_U8: return 1;
    case OR_I32: return 4;
    case OR_I64: return 8;
    case OR_F16: return 2;
    case OR_F32: return 4;
    case OR_F64: return 8;
    case OR_BF16: return 2;
    default: return 0;
  }
}

/* ---- half / bfloat16 ----------------------------------------------------------------------- */

float oracle_half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu;
  uint32_t man = h & 0x3FFu;
  uint32_t out;
  if (exp == 0) {
    if (man == 0) {
      out = sign;
    } else { /* subnormal: value = man * 2^-24 */
      float f = (float)man * 5.9604644775390625e-08f; /* exact */
      uint32_t fb;
      memcpy(&fb, &f, 4);
      out = fb | sign;
    }
  } else if (exp == 31) {
    out = sign | 0x7F800000u | (man << 13);
  } else {
    out = sign | ((exp + 112u) << 23) | (man << 13);
  }
  float r;
  memcpy(&r, &out, 4);
  return r;
}

uint16_t oracle_double_to_half(double d) {
  uint64_t bits;
  memcpy(&bits, &d, 8);
  uint16_t sign = (uint16_t)((bits >> 48) & 0x8000u);
  int bexp = (int)((bits >> 52) & 0x7FF);
  uint64_t man = bits & 0xFFFFFFFFFFFFFULL;
  if (bexp == 0x7FF) return man ? (uint16_t)(sign | 0x7E00u) : (uint16_t)(sign | 0x7C00u);
  if (bexp == 0) return sign; /* double zero / subnormal: far below half's range */
  int e = bexp - 1023;
  if (e > 15) return (uint16_t)(sign | 0x7C00u);
  if (e >= -14) {
    uint64_t m = man >> 42, rem = man & ((1ULL << 42) - 1), halfway = 1ULL << 41;
    uint32_t h = ((uint32_t)(e + 15) << 10) | (uint32_t)m;
    if (rem > halfway || (rem == halfway && (h & 1u))) h++;
    if (h > 0x7C00u) h = 0x7C00u;
    return (uint16_t)(sign | h);
  }
  if (e < -25) return sign;
  {
    uint64_t full = (1ULL << 52) | man; /* value = full * 2^(e-52); unit = 2^-24 */
    int shift = 52 - (e + 24);          /* 43..53 */
    uint64_t q = full >> shift, rem = full & ((1ULL << shift) - 1), halfway = 1ULL << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) q++;
    return (uint16_t)(sign | (uint16_t)q);
  }
}

float oracle_bf16_to_float(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

uint16_t oracle_float_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

/* ---- deterministic inputs ------------------------------------------------------------------ */

uint64_t oracle_hash(uint64_t seed, uint64_t i) {
  uint64_t z = seed * 0xD1342543DE82EF95ULL + i * 0x9E3779B97F4A7C15ULL + 0x2545F4914F6CDD1DULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* the value of element i as a double that is exactly representable in `dtype` */
static double pattern_real(int dtype, int pattern, uint64_t seed, uint64_t i) {
  uint64_t h = oracle_hash(seed, i);
  switch (pattern) {
    case OR_PAT_UNIFORM:
      if (dtype == OR_F64) return (double)(h >> 11) * 0x1p-53;
      if (dtype == OR_F32) return (double)(h >> 40) * 0x1p-24;
      if (dtype == OR_F16) return (double)(h & 63u) * 0x1p-6; /* k/64: 8-way sums exact in fp16 */
      return (double)(h & 15u) * 0x1p-4;                      /* bf16: k/16 */
    case OR_PAT_INDEX: /* every value exactly representable in its dtype */
      if (dtype == OR_F16) return (double)(i & 63u) * 0x1p-6 + (double)(seed & 7u);
      if (dtype == OR_BF16) return (double)(i & 15u) * 0x1p-4 + (double)(seed & 7u);
      return (double)(i % 251u) * 0x1p-8 + (double)(seed & 0xFFu);
    case OR_PAT_CONST:
      return (double)((seed & 0xFFu) + 1u);
    default: { /* OR_PAT_SIGNED: sign and binade vary, 8 significant bits */
      double m = (double)((h >> 40) & 0xFFu) * 0x1p-8 - 0.5;
      int sh = (int)((h >> 8) & 7u) - 4;
      return ldexp(m, sh);
    }
  }
}

int oracle_fill_range(void* buf, size_t start, size_t count, int dtype, int pattern, uint64_t seed) {
  size_t i;
  if (pattern < 0 || pattern > OR_PAT_SIGNED) return -1;
  switch (dtype) {
    case OR_U8: {
      uint8_t* p = (uint8_t*)buf;
      for (i = 0; i < count; i++) {
        if (pattern == OR_PAT_INDEX) p[i] = (uint8_t)(seed * 31u + (start + i));
        else if (pattern == OR_PAT_CONST) p[i] = (uint8_t)(seed + 1u);
        else p[i] = (uint8_t)(oracle_hash(seed, start + i) >> 56);
      }
      return 0;
    }
    case OR_I32: {
      int32_t* p = (int32_t*)buf;
      for (i = 0; i < count; i++) {
        if (pattern == OR_PAT_INDEX) p[i] = (int32_t)(((uint32_t)seed << 24) | ((uint32_t)(start + i) & 0xFFFFFFu));
        else if (pattern == OR_PAT_CONST) p[i] = (int32_t)(seed + 1u);
        else p[i] = (int32_t)(uint32_t)(oracle_hash(seed, start + i) >> 32);
      }
      return 0;
    }
    case OR_I64: {
      int64_t* p = (int64_t*)buf;
      for (i = 0; i < count; i++) {
        if (pattern == OR_PAT_INDEX) p[i] = (int64_t)((seed << 40) | (uint64_t)(start + i)); /* BASELINE cfg 3 */
        else if (pattern == OR_PAT_CONST) p[i] = (int64_t)(seed + 1u);
        else p[i] = (int64_t)oracle_hash(seed, start + i);
      }
      return 0;
    }
    case OR_F16: {
      uint16_t* p = (uint16_t*)buf;
      for (i = 0; i < count; i++) p[i] = oracle_double_to_half(pattern_real(dtype, pattern, seed, start + i));
      return 0;
    }
    case OR_BF16: {
      uint16_t* p = (uint16_t*)buf;
      for (i = 0; i < count; i++) p[i] = oracle_float_to_bf16((float)pattern_real(dtype, pattern, seed, start + i));
      return 0;
    }
    case OR_F32: {
      float* p = (float*)buf;
      for (i = 0; i < count; i++) p[i] = (float)pattern_real(dtype, pattern, seed, start + i);
      return 0;
    }
    case OR_F64: {
      double* p = (double*)buf;
      for (i = 0; i < count; i++) p[i] = pattern_real(dtype, pattern, seed, start + i);
      return 0;
    }
    default:
      return -1;
  }
}

int oracle_fill(void* buf, size_t count, int dtype, int pattern, uint64_t seed) {
  return oracle_fill_range(buf, 0, count, dtype, pattern, seed);
}

/* ---- elementwise combine -------------------------------------------------------------------
 * min: (b < a) ? b : a      max: (a < b) ? b : a      (NaN in either operand keeps a)
 * integer sum/prod wrap (two's complement), as Go's do. */

#define COMBINE_FLOAT(T, a, b, op, r)          \
  do {                                         \
    T _a = (a), _b = (b);                      \
    switch (op) {                              \
      case OR_SUM: (r) = _a + _b; break;       \
      case OR_PROD: (r) = _a * _b; break;      \
      case OR_MIN: (r) = (_b < _a) ? _b : _a; break; \
      default: (r) = (_a < _b) ? _b : _a; break;     \
    }                                          \
  } while (0)

static uint16_t combine_f16(uint16_t a, uint16_t b, int op) {
  double x = (double)oracle_half_to_float(a), y = (double)oracle_half_to_float(b);
  switch (op) {
    case OR_SUM: return oracle_double_to_half(x + y); /* exact in double, one rounding */
    case OR_PROD: return oracle_double_to_half(x * y);
    case OR_MIN: return (y < x) ? b : a;
    default: return (x < y) ? b : a;
  }
}

static uint16_t combine_bf16(uint16_t a, uint16_t b, int op) {
  float x = oracle_bf16_to_float(a), y = oracle_bf16_to_float(b);
  volatile float r;
  switch (op) {
    case OR_SUM: r = x + y; return oracle_float_to_bf16(r);
    case OR_PROD: r = x * y; return oracle_float_to_bf16(r);
    case OR_MIN: return (y < x) ? b : a;
    default: return (x < y) ? b : a;
  }
}

int oracle_reduce2(void* dst, const void* a, const void* b, size_t count, int dtype, int op) {
  size_t i;
  if (op < 0 || op > OR_MAX) return -1;
  switch (dtype) {
    case OR_U8: {
      uint8_t* d = (uint8_t*)dst; const uint8_t *x = (const uint8_t*)a, *y = (const uint8_t*)b;
      for (i = 0; i < count; i++) {
        uint8_t r;
        switch (op) {
          case OR_SUM: r = (uint8_t)(x[i] + y[i]); break;
          case OR_PROD: r = (uint8_t)(x[i] * y[i]); break;
          case OR_MIN: r = (y[i] < x[i]) ? y[i] : x[i]; break;
          default: r = (x[i] < y[i]) ? y[i] : x[i]; break;
        }
        d[i] = r;
      }
      return 0;
    }
    case OR_I32: {
      int32_t* d = (int32_t*)dst; const int32_t *x = (const int32_t*)a, *y = (const int32_t*)b;
      for (i = 0; i < count; i++) {
        int32_t r;
        switch (op) {
          case OR_SUM: r = (int32_t)((uint32_t)x[i] + (uint32_t)y[i]); break;
          case OR_PROD: r = (int32_t)((uint32_t)x[i] * (uint32_t)y[i]); break;
          case OR_MIN: r = (y[i] < x[i]) ? y[i] : x[i]; break;
          default: r = (x[i] < y[i]) ? y[i] : x[i]; break;
        }
        d[i] = r;
      }
      return 0;
    }
    case OR_I64: {
      int64_t* d = (int64_t*)dst; const int64_t *x = (const int64_t*)a, *y = (const int64_t*)b;
      for (i = 0; i < count; i++) {
        int64_t r;
        switch (op) {
          case OR_SUM: r = (int64_t)((uint64_t)x[i] + (uint64_t)y[i]); break;
          case OR_PROD: r = (int64_t)((uint64_t)x[i] * (uint64_t)y[i]); break;
          case OR_MIN: r = (y[i] < x[i]) ? y[i] : x[i]; break;
          default: r = (x[i] < y[i]) ? y[i] : x[i]; break;
        }
        d[i] = r;
      }
      return 0;
    }
    case OR_F16: {
      uint16_t* d = (uint16_t*)dst; const uint16_t *x = (const uint16_t*)a, *y = (const uint16_t*)b;
      for (i = 0; i < count; i++) d[i] = combine_f16(x[i], y[i], op);
      return 0;
    }
    case OR_BF16: {
      uint16_t* d = (uint16_t*)dst; const uint16_t *x = (const uint16_t*)a, *y = (const uint16_t*)b;
      for (i = 0; i < count; i++) d[i] = combine_bf16(x[i], y[i], op);
      return 0;
    }
    case OR_F32: {
      float* d = (float*)dst; const float *x = (const float*)a, *y = (const float*)b;
      for (i = 0; i < count; i++) COMBINE_FLOAT(float, x[i], y[i], op, d[i]);
      return 0;
    }
    case OR_F64: {
      double* d = (double*)dst; const double *x = (const double*)a, *y = (const double*)b;
      for (i = 0; i < count; i++) COMBINE_FLOAT(double, x[i], y[i], op, d[i]);
      return 0;
    }
    default:
      return -1;
  }
}

int oracle_reduce_ranks(void* out, const void* const* in, int nranks, size_t count, int dtype,
                        int op) {
  size_t es = oracle_dtype_size(dtype);
  int r;
  if (!es || nranks < 1) return -1;
  memmove(out, in[0], count * es);
  for (r = 1; r < nranks; r++) {
    int rc = oracle_reduce2(out, out, in[r], count, dtype, op);
    if (rc) return rc;
  }
  return 0;
}

int oracle_allgather(void* out, const void* const* in, int nranks, size_t count, int dtype) {
  size_t es = oracle_dtype_size(dtype);
  int r;
  if (!es || nranks < 1) return -1;
  for (r = 0; r < nranks; r++) memcpy((char*)out + (size_t)r * count * es, in[r], count * es);
  return 0;
}

/* ---- comparison helpers -------------------------------------------------------------------- */

uint64_t oracle_count_mismatch(const void* a, const void* b, size_t bytes) {
  const uint8_t *x = (const uint8_t*)a, *y = (const uint8_t*)b;
  uint64_t n = 0;
  size_t i;
  for (i = 0; i < bytes; i++) n += (x[i] != y[i]);
  return n;
}

uint64_t oracle_checksum(const void* buf, size_t bytes) {
  const uint8_t* p = (const uint8_t*)buf;
  uint64_t s = 0;
  size_t i, w = bytes / 4;
  for (i = 0; i < w; i++) {
    uint32_t v;
    memcpy(&v, p + 4 * i, 4);
    s += v;
  }
  for (i = 4 * w; i < bytes; i++) s += p[i];
  return s;
}

static double elem_as_double(const void* p, size_t i, int dtype) {
  switch (dtype) {
    case OR_F16: return (double)oracle_half_to_float(((const uint16_t*)p)[i]);
    case OR_BF16: return (double)oracle_bf16_to_float(((const uint16_t*)p)[i]);
    case OR_F32: return (double)((const float*)p)[i];
    default: return ((const double*)p)[i];
  }
}

int oracle_diff_stats(const void* a, const void* b, size_t count, int dtype, double stats[3]) {
  size_t i;
  double mx = 0, sb = 0, nn = 0;
  if (dtype != OR_F16 && dtype != OR_BF16 && dtype != OR_F32 && dtype != OR_F64) return -1;
  for (i = 0; i < count; i++) {
    double x = elem_as_double(a, i, dtype), y = elem_as_double(b, i, dtype);
    if (isnan(x) || isnan(y)) {
      if (isnan(x) != isnan(y)) nn += 1;
      continue;
    }
    if (fabs(x - y) > mx) mx = fabs(x - y);
    sb += fabs(y);
  }
  stats[0] = mx;
  stats[1] = sb;
  stats[2] = nn;
  return 0;
}

/* The whole-buffer check of an allreduce at ANY size: elements [start, start + n) of the result `got` against the
 * rank-order fold of the ranks' inputs, which are regenerated here block by block (rank r's input is
 * oracle_fill(pattern, seed0 + r)) -- nothing of the size of the buffers is ever held on the host but `got`.
 * Returns the number of elements whose bits differ; *first_bad (optional) = index of the first. */
uint64_t oracle_check_allreduce(const void* got, size_t start, size_t n, int dtype, int pattern, uint64_t seed0,
                                int nranks, int op, uint64_t* first_bad) {
  enum { BLK = 1 << 16 };
  const size_t es = oracle_dtype_size(dtype);
  uint64_t bad = 0;
  size_t done = 0;
  int r;
  unsigned char *in, *acc;
  if (es == 0 || nranks < 1) return ~0ull;
  in = (unsigned char*)malloc((size_t)BLK * es);
  acc = (unsigned char*)malloc((size_t)BLK * es);
  if (!in || !acc) {
    free(in);
    free(acc);
    return ~0ull;
  }
  while (done < n) {
    const size_t m = n - done < (size_t)BLK ? n - done : (size_t)BLK;
    size_t i;
    oracle_fill_range(acc, start + done, m, dtype, pattern, seed0);
    for (r = 1; r < nranks; r++) {
      oracle_fill_range(in, start + done, m, dtype, pattern, seed0 + (uint64_t)r);
      oracle_reduce2(acc, acc, in, m, dtype, op);
    }
    if (memcmp(acc, (const unsigned char*)got + done * es, m * es) != 0) {
      for (i = 0; i < m; i++)
        if (memcmp(acc + i * es, (const unsigned char*)got + (done + i) * es, es) != 0) {
          if (bad == 0 && first_bad) *first_bad = (uint64_t)(start + done + i);
          bad++;
        }
    }
    done += m;
  }
  free(in);
  free(acc);
  return bad;
}
