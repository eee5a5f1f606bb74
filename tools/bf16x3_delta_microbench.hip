// Microbenchmark (feasibility, not product code): the weight-delta GEMM of one generation,
//   dW[1024 x 1024] = sum over k of X[k][i] * E[k][h],  k = (step, stream) = 5120 rows,
// (recur-nn.c:344-356 summed over streams, recur-nn.c:734-748) computed on the BF16 matrix
// pipe from FP32 operands split on the fly into three bf16 terms each,
//   x = hi + mid + lo   (hi = x truncated to bf16, mid = (x - hi) truncated, lo = the rest: exact),
// with six products per pair (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; the three dropped
// ones are <= 2^-23 of the product) accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
// The product's k_delta_dma does the same GEMM with v_mfma_f32_32x32x2_f32 (exact fp32,
// 157.3 TF peak, which runs on the SIMD's vector ALU) in 96 us = 0.71 of that peak.
// Question asked: what does the 6-product form cost, and how large is its error against
// fp64 compared with an fp32 sum?  (Answer and numbers: DESIGN.md section 8.)
//   hipcc --offload-arch=gfx950 -O3 tools/bf16x3_delta_microbench.hip -o build/bf16x3_delta
//   build/bf16x3_delta            # prints us per launch, TF (fp32-equivalent), error table
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);  \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

constexpr int BM = 128, BN = 128, BK = 32; // block tile; two MFMA K steps per stage
#ifndef INTERLEAVE
#define INTERLEAVE 1
#endif
#ifndef VALU_PER_MFMA
#define VALU_PER_MFMA 5
#endif
#ifndef NOSTAGE
#define NOSTAGE 0 // 1: timing only (wrong results): no fetch / split / LDS write / barrier in the loop
#endif
#ifndef XCD_MAP
#define XCD_MAP 1
#endif
#ifndef KSPLIT_N
#define KSPLIT_N 4
#endif
#ifndef COL0
#define COL0 0 // 1: operand rows start at column 1 (4-byte aligned only), as the product's error planes do
#endif
constexpr int KSPLIT = KSPLIT_N; // 8 x 8 tiles x 4 K slabs = 256 workgroups = one per CU

// 8 consecutive-k fp32 values of one row -> the three bf16x8 fragments of that row
__device__ inline void split8(const float (&x)[8], u32x4 &hi, u32x4 &mid, u32x4 &lo) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    unsigned u = __float_as_uint(x[j]);
    float r1 = x[j] - __uint_as_float(u & 0xffff0000u);
    unsigned v = __float_as_uint(r1);
    float r2 = r1 - __uint_as_float(v & 0xffff0000u);
    h[j] = u;
    m[j] = v;
    l[j] = __float_as_uint(r2);
  }
#pragma unroll
  for (int j = 0; j < 4; j++) { // upper halves of two words -> one word (element 2j low, 2j+1 high)
    hi[j] = __builtin_amdgcn_perm(h[2 * j + 1], h[2 * j], 0x07060302u);
    mid[j] = __builtin_amdgcn_perm(m[2 * j + 1], m[2 * j], 0x07060302u);
    lo[j] = __builtin_amdgcn_perm(l[2 * j + 1], l[2 * j], 0x07060302u);
  }
}

__device__ inline f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// NPROD = 6: fp32-grade; 3: hi*hi + hi*mid + mid*hi (2^-16-grade); 1: plain bf16 (timing floor)
template <int NPROD>
__global__ __launch_bounds__(256) void k_delta_bf16x3(const float *__restrict__ X, int ldx,
                                                      const float *__restrict__ E, int lde,
                                                      float *__restrict__ slab, int M, int N,
                                                      int kper) {
  // [buffer][operand][term][k group of 8][row]: 16 bytes per entry = the 8 bf16 a lane feeds to
  // one MFMA, rows contiguous -> fragment reads are lane-contiguous ds_read_b128
  __shared__ u32x4 sm[2][2][3][4][128];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // XCD-aware placement (workgroups go round-robin over the 8 XCDs): an XCD owns ONE K slab of
  // half the row tiles and all column tiles, so its L2 sees 4 + 8 operand panels, not 32 + 4
  static_assert(KSPLIT == 4, "the placement below is written for 8 x 8 tiles x 4 K slabs");
  const int wg = blockIdx.x + 8 * blockIdx.y + 64 * blockIdx.z, xcd = wg & 7, slot = wg >> 3;
#if XCD_MAP
  const int kz = xcd >> 1, m0 = ((xcd & 1) * 4 + (slot >> 3)) * BM, n0 = (slot & 7) * BN;
#else
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, kz = blockIdx.z;
#endif
  // staging: waves 0-1 fetch and split X, waves 2-3 E; a thread owns 4 adjacent rows of one k group:
  // eight 16-byte loads (k = 0..7 of the group), each lane-contiguous over the rows
  const int op = tid >> 7, kg = (tid >> 5) & 3, rq = tid & 31;
  const int ld = op ? lde : ldx;
  const float *base = (op ? E + n0 : X + m0) + (size_t)(kz * kper) * ld + COL0; // wave-uniform
  const unsigned off = kg * 8 * ld + rq * 4;
  const int wm = w >> 1, wn = w & 1, fr = lane & 31, fk = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  // raw fp32 operand rows in flight: a ring of PF register sets, so that a set is fetched PF
  // stages before it is split (the split is interleaved with the MFMAs of the stage before)
  constexpr int PF = 2;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 raw[PF][8];
  const int nst = kper / BK;
  auto fetch = [&](f32x4 (&p)[8], int s) {
    s = s < nst ? s : nst - 1; // past the end: a harmless repeat of the last stage
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float *q = base + (size_t)(s * BK + j) * ld + off;
      __builtin_memcpy(&p[j], q, 16); // 4-byte aligned when COL0 = 1: still one global_load_dwordx4
    }
  };
  auto stage = [&](const f32x4 (&p)[8], int buf) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; j++) x[j] = p[j][c];
      u32x4 h, m, l;
      split8(x, h, m, l);
      sm[buf][op][0][kg][rq * 4 + c] = h;
      if (NPROD > 1) sm[buf][op][1][kg][rq * 4 + c] = m;
      if (NPROD > 3) sm[buf][op][2][kg][rq * 4 + c] = l;
    }
  };
  constexpr int NT = NPROD > 3 ? 3 : NPROD > 1 ? 2 : 1;
  // one stage: multiply stage s out of LDS buffer s & 1 while splitting the register set that
  // holds stage s + 1 into the other buffer, then refill that set with stage s + 1 + PF
  auto step = [&](f32x4 (&p)[8], int s) {
    const int buf = s & 1;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      u32x4 a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int t = 0; t < NT; t++) {
          a[i][t] = sm[buf][0][t][ks * 2 + fk][wm * 64 + i * 32 + fr];
          b[i][t] = sm[buf][1][t][ks * 2 + fk][wn * 64 + i * 32 + fr];
        }
      // small terms first, the leading product last
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          if (NPROD > 3) {
            acc[i][j] = mfma(a[i][1], b[j][1], acc[i][j]);
            acc[i][j] = mfma(a[i][0], b[j][2], acc[i][j]);
            acc[i][j] = mfma(a[i][2], b[j][0], acc[i][j]);
          }
          if (NPROD > 1) {
            acc[i][j] = mfma(a[i][0], b[j][1], acc[i][j]);
            acc[i][j] = mfma(a[i][1], b[j][0], acc[i][j]);
          }
          acc[i][j] = mfma(a[i][0], b[j][0], acc[i][j]);
        }
    }
#if NOSTAGE == 0
    stage(p, buf ^ 1);
    fetch(p, s + 1 + PF);
#endif
#if INTERLEAVE && NOSTAGE == 0
    // issue order: every MFMA followed by a few of the split's vector instructions (they hide
    // in the 32 cycles the matrix pipe takes per MFMA)
#pragma unroll
    for (int g = 0; g < 8 * NPROD; g++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);
    }
#endif
#if NOSTAGE == 0
    __syncthreads();
#endif
  };

  fetch(raw[0], 0);
  stage(raw[0], 0);
#pragma unroll
  for (int q = 0; q < PF; q++) fetch(raw[q], 1 + q);
  __syncthreads();
  for (int s = 0; s < nst; s += PF) { // nst is a multiple of PF (checked on the host)
    step(raw[0], s);
    step(raw[1], s + 1);
  }
  float *out = slab + (size_t)kz * M * N;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fk;
        int col = n0 + wn * 64 + j * 32 + fr;
        out[(size_t)row * N + col] = acc[i][j][e];
      }
}


// ---- the same product with the split made ONCE by a producer pass ------------------------
// Planes: P[term][k / 8][row][8] bf16 = 16-byte entries, exactly one lane's operand of one MFMA.
__global__ __launch_bounds__(256) void k_split_planes(const float *__restrict__ src, int ld, int rows,
                                                      int K, u32x4 *__restrict__ planes) {
  const int row = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y; // k group
  if (row >= rows) return;
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; j++) x[j] = src[(size_t)(g * 8 + j) * ld + row + COL0];
  u32x4 h, m, l;
  split8(x, h, m, l);
  const size_t plane = (size_t)(K / 8) * rows, at = (size_t)g * rows + row;
  planes[at] = h;
  planes[plane + at] = m;
  planes[2 * plane + at] = l;
}

// the GEMM over the planes: per stage (K = 32) a workgroup copies 2 operands x 3 terms x 4 k groups
// x 128 rows x 16 bytes = 48 KB global -> registers -> LDS (12 entries per thread, no arithmetic)
__global__ __launch_bounds__(256) void k_delta_presplit(const u32x4 *__restrict__ PX,
                                                        const u32x4 *__restrict__ PE,
                                                        float *__restrict__ slab, int M, int N, int K,
                                                        int kper) {
  __shared__ u32x4 sm[2][2 * 3 * 4 * 128];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // XCD-aware placement (workgroups go round-robin over the 8 XCDs): an XCD owns ONE K slab of
  // half the row tiles and all column tiles, so its L2 sees 4 + 8 operand panels, not 32 + 4
  static_assert(KSPLIT == 4, "the placement below is written for 8 x 8 tiles x 4 K slabs");
  const int wg = blockIdx.x + 8 * blockIdx.y + 64 * blockIdx.z, xcd = wg & 7, slot = wg >> 3;
#if XCD_MAP
  const int kz = xcd >> 1, m0 = ((xcd & 1) * 4 + (slot >> 3)) * BM, n0 = (slot & 7) * BN;
#else
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, kz = blockIdx.z;
#endif
  const int wm = w >> 1, wn = w & 1, fr = lane & 31, fk = lane >> 5;
  const size_t planeX = (size_t)(K / 8) * M, planeE = (size_t)(K / 8) * N;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
  const int nst = kper / BK;
  u32x4 raw[12];
  auto fetch = [&](int s) {
    s = s < nst ? s : nst - 1;
#pragma unroll
    for (int e = 0; e < 12; e++) { // entry e * 256 + tid of the stage image [op][term][kg][row]
      const int op = e / 6, t = (e >> 1) % 3, kg = (e & 1) * 2 + (tid >> 7), row = tid & 127;
      const size_t g = (size_t)(kz * kper + s * BK) / 8 + kg;
      raw[e] = op ? PE[t * planeE + g * N + n0 + row] : PX[t * planeX + g * M + m0 + row];
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 12; e++) sm[buf][e * 256 + tid] = raw[e];
  };
  fetch(0);
  stage(0);
  fetch(1);
  __syncthreads();
  for (int s = 0; s < nst; s++) {
    const int buf = s & 1;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      u32x4 a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int t = 0; t < 3; t++) {
          a[i][t] = sm[buf][((0 * 3 + t) * 4 + ks * 2 + fk) * 128 + wm * 64 + i * 32 + fr];
          b[i][t] = sm[buf][((1 * 3 + t) * 4 + ks * 2 + fk) * 128 + wn * 64 + i * 32 + fr];
        }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          acc[i][j] = mfma(a[i][1], b[j][1], acc[i][j]);
          acc[i][j] = mfma(a[i][0], b[j][2], acc[i][j]);
          acc[i][j] = mfma(a[i][2], b[j][0], acc[i][j]);
          acc[i][j] = mfma(a[i][0], b[j][1], acc[i][j]);
          acc[i][j] = mfma(a[i][1], b[j][0], acc[i][j]);
          acc[i][j] = mfma(a[i][0], b[j][0], acc[i][j]);
        }
    }
    stage(buf ^ 1);
    fetch(s + 2);
    __syncthreads();
  }
  float *out = slab + (size_t)kz * M * N;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fk;
        int col = n0 + wn * 64 + j * 32 + fr;
        out[(size_t)row * N + col] = acc[i][j][e];
      }
}

static unsigned long long rs = 88172645463325252ull;
static double urand() {
  rs ^= rs << 13;
  rs ^= rs >> 7;
  rs ^= rs << 17;
  return (double)(rs >> 11) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2.0 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

static void report(const char *name, double us, const float *dS, int M, int N, int K, int ldx, int lde,
                   const std::vector<float> &X, const std::vector<float> &E) {
  const double flop = 2.0 * M * N * K;
  std::vector<float> S((size_t)KSPLIT * M * N);
  CHECK(hipMemcpy(S.data(), dS, S.size() * sizeof(float), hipMemcpyDeviceToHost));
  // error against fp64 on a sample of outputs, in units of eps32 * sum |x||e| (the scale of an
  // fp32 dot product's rounding error), beside a sequential fp32 fma sum of the same entries
  double worst_g = 0, worst_c = 0, rms_g = 0, rms_c = 0;
  int n = 0;
  for (int t = 0; t < 2048; t++) {
    int i = (int)(urand() * M), h = (int)(urand() * N);
    double ref = 0, mag = 0;
    float f = 0;
    for (int k = 0; k < K; k++) {
      double p = (double)X[(size_t)k * ldx + i + COL0] * (double)E[(size_t)k * lde + h + COL0];
      ref += p;
      mag += fabs(p);
      f = fmaf(X[(size_t)k * ldx + i + COL0], E[(size_t)k * lde + h + COL0], f);
    }
    double g = 0;
    for (int z = 0; z < KSPLIT; z++) g += (double)S[((size_t)z * M + i) * N + h];
    const double unit = mag * 5.9604644775390625e-08 + 1e-300;
    double eg = fabs(g - ref) / unit, ec = fabs((double)f - ref) / unit;
    worst_g = fmax(worst_g, eg);
    worst_c = fmax(worst_c, ec);
    rms_g += eg * eg;
    rms_c += ec * ec;
    n++;
  }
  printf("%-28s %8.2f us  %7.1f TF fp32-equivalent (%.2f of the 157.3 TF fp32 matrix peak)   "
         "error / (eps32 * sum|x||e|): max %.3g rms %.3g   [sequential fp32 fma sum: max %.3g rms %.3g]\n",
         name, us, flop / us * 1e-6, flop / us * 1e-6 / 157.3, worst_g, sqrt(rms_g / n), worst_c,
         sqrt(rms_c / n));
}

template <int NPROD>
static void run(const char *name, const float *dX, int ldx, const float *dE, int lde, float *dS, int M,
                int N, int K, const std::vector<float> &X, const std::vector<float> &E, int reps) {
  dim3 grid(N / BN, M / BM, KSPLIT), block(256);
  const int kper = K / KSPLIT;
  if (kper % (2 * BK)) exit(2);
  for (int i = 0; i < 20; i++)
    hipLaunchKernelGGL(k_delta_bf16x3<NPROD>, grid, block, 0, 0, dX, ldx, dE, lde, dS, M, N, kper);
  CHECK(hipGetLastError());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; i++)
    hipLaunchKernelGGL(k_delta_bf16x3<NPROD>, grid, block, 0, 0, dX, ldx, dE, lde, dS, M, N, kper);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  report(name, 1e3 * ms / reps, dS, M, N, K, ldx, lde, X, E);
}

static void run_presplit(const float *dX, int ldx, const float *dE, int lde, float *dS, int M, int N, int K,
                         const std::vector<float> &X, const std::vector<float> &E, int reps) {
  u32x4 *pX, *pE;
  CHECK(hipMalloc(&pX, (size_t)3 * (K / 8) * M * 16));
  CHECK(hipMalloc(&pE, (size_t)3 * (K / 8) * N * 16));
  dim3 grid(N / BN, M / BM, KSPLIT), block(256), sgx(M / 256, K / 8), sge(N / 256, K / 8);
  const int kper = K / KSPLIT;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float ms = 0;
  for (int pass = 0; pass < 2; pass++) { // 0: the split of both operands, 1: the GEMM
    for (int i = 0; i < reps + 20; i++) {
      if (i == 20) CHECK(hipEventRecord(e0, 0));
      if (pass == 0) {
        hipLaunchKernelGGL(k_split_planes, sgx, block, 0, 0, dX, ldx, M, K, pX);
        hipLaunchKernelGGL(k_split_planes, sge, block, 0, 0, dE, lde, N, K, pE);
      } else {
        hipLaunchKernelGGL(k_delta_presplit, grid, block, 0, 0, pX, pE, dS, M, N, K, kper);
      }
    }
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (pass == 0)
      printf("split of X and E into bf16 planes (stand-alone pass; in a product the producers' epilogues): %.2f us "
             "(%.0f MB read, %.0f MB written)\n", 1e3 * ms / reps, (M + N) * (double)K * 4e-6, (M + N) * (double)K * 6e-6);
  }
  report("bf16x3 over pre-split planes", 1e3 * ms / reps, dS, M, N, K, ldx, lde, X, E);
  CHECK(hipFree(pX));
  CHECK(hipFree(pE));
}

int main() {
  // the north-star generation: hidden 1024, 256 streams, depth 20; rows padded as in the product
  const int M = 1024, N = 1024, K = 5120, ldx = 1068, lde = 1068;
  std::vector<float> X((size_t)K * ldx), E((size_t)K * lde);
  for (auto &v : X) { // ReLU hidden values: 55 % zeros, the rest spread over five decades
    v = urand() < 0.55 ? 0.f : (float)(fabs(nrand()) * pow(10.0, -3.0 * urand()));
  }
  for (auto &v : E) v = (float)(nrand() * 1e-3 * pow(10.0, -4.0 * urand()));
  float *dX, *dE, *dS;
  CHECK(hipMalloc(&dX, X.size() * sizeof(float)));
  CHECK(hipMalloc(&dE, E.size() * sizeof(float)));
  CHECK(hipMalloc(&dS, (size_t)KSPLIT * M * N * sizeof(float)));
  CHECK(hipMemcpy(dX, X.data(), X.size() * sizeof(float), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dE, E.data(), E.size() * sizeof(float), hipMemcpyHostToDevice));
  run<6>("bf16x3, 6 products", dX, ldx, dE, lde, dS, M, N, K, X, E, 200);
  run<3>("bf16x2, 3 products", dX, ldx, dE, lde, dS, M, N, K, X, E, 200);
  run<1>("plain bf16, 1 product", dX, ldx, dE, lde, dS, M, N, K, X, E, 200);
  run_presplit(dX, ldx, dE, lde, dS, M, N, K, X, E, 200);
  return 0;
}
