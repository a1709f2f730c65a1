// layout_probe.hip — memory skeleton of the C1 CartPole step under different state layouts.
// Build: hipcc --offload-arch=gfx950 -O3 -o layout_probe layout_probe.hip ; run on the GPU box.
// Every variant moves the same 120 B per env-step (the algorithmic bytes of DESIGN.md §4) with a
// trivial amount of arithmetic, so the numbers bound what a layout can give the real kernel.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e = (x);                                                          \
    if (e != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                     \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Rows {
  double* phys;   // [4][N]
  double* theta;  // [N]
  int32_t* t;
  float* er;
  uint8_t* status;
  const int32_t* action;
  f32x4* obs;
  float* reward;
  uint8_t* term;
  uint8_t* trunc;
  uint8_t* flag;
  float* delta;
  unsigned char* blocked;  // [chunk][12544 B] state rows of one 256-env chunk, contiguous
  unsigned char* blocked_out;  // [chunk][7424 B] outputs of one chunk, contiguous
};

__device__ __forceinline__ void compute(double s[4], double& th, int& t, float& er, unsigned& st, int a, f32x4& o, float& rew,
                                        unsigned& term, unsigned& trunc, unsigned& flag, float& delta) {
  th += 0.1;
  const double f = a ? 10.0 : -10.0;
  s[1] += 0.02 * (f + th * s[2]);
  s[0] += 0.02 * s[1];
  s[3] += 0.02 * (s[2] * 9.8 - f * 0.05);
  s[2] += 0.02 * s[3];
  t += 1;
  rew = 1.f;
  er += rew;
  term = (s[0] > 2.4 || s[0] < -2.4 || s[2] > 0.21 || s[2] < -0.21) ? 1u : 0u;
  trunc = t >= 500 ? 1u : 0u;
  st = (term | trunc) ? 1u : 0u;
  if (st) { s[0] = 0.01; s[1] = -0.02; s[2] = 0.03; s[3] = 0.01; t = 0; er = 0.f; }
  o = f32x4{(float)s[0], (float)s[1], (float)s[2], (float)s[3]};
  flag = 1u;
  delta = 0.1f;
}

// A: plain SoA rows (the library's layout)
__global__ __launch_bounds__(256) void soa_kernel(Rows r, int64_t N) {
  const int64_t chunks = (N + 255) / 256;
  for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
    const int64_t i = c * 256 + threadIdx.x;
    if (i >= N) continue;
    unsigned st = r.status[i];
    int t = r.t[i];
    double s[4];
    for (int k = 0; k < 4; k++) s[k] = r.phys[k * N + i];
    const int a = r.action[i];
    float er = r.er[i];
    double th = r.theta[i];
    f32x4 o;
    float rew, delta;
    unsigned term, trunc, flag;
    compute(s, th, t, er, st, a, o, rew, term, trunc, flag, delta);
    r.theta[i] = th;
    r.flag[i] = flag;
    r.delta[i] = delta;
    for (int k = 0; k < 4; k++) r.phys[k * N + i] = s[k];
    r.obs[i] = o;
    r.t[i] = t;
    r.reward[i] = rew;
    r.term[i] = term;
    r.trunc[i] = trunc;
    r.status[i] = st;
    r.er[i] = er;
  }
}

// B: state rows blocked per chunk (one contiguous 12.25 KB region per workgroup), outputs dense SoA
// C: outputs blocked per chunk as well
template <bool OUT_BLOCKED>
__global__ __launch_bounds__(256) void blocked_kernel(Rows r, int64_t N) {
  const int64_t chunks = (N + 255) / 256;
  const int l = threadIdx.x;
  for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
    const int64_t i = c * 256 + l;
    if (i >= N) continue;
    unsigned char* blk = r.blocked + c * 12544;
    double* bp = (double*)blk;                    // [4][256]
    double* bth = (double*)(blk + 8192);          // [256]
    int32_t* bt = (int32_t*)(blk + 10240);
    float* ber = (float*)(blk + 11264);
    uint8_t* bst = blk + 12288;
    unsigned st = bst[l];
    int t = bt[l];
    double s[4];
    for (int k = 0; k < 4; k++) s[k] = bp[k * 256 + l];
    const int a = r.action[i];
    float er = ber[l];
    double th = bth[l];
    f32x4 o;
    float rew, delta;
    unsigned term, trunc, flag;
    compute(s, th, t, er, st, a, o, rew, term, trunc, flag, delta);
    bth[l] = th;
    for (int k = 0; k < 4; k++) bp[k * 256 + l] = s[k];
    bt[l] = t;
    bst[l] = st;
    ber[l] = er;
    if (OUT_BLOCKED) {
      unsigned char* ob = r.blocked_out + c * 7168;  // obs 4096 | reward 1024 | delta 1024 | term 256 | trunc 256 | flag 256 | pad 256
      ((f32x4*)ob)[l] = o;
      ((float*)(ob + 4096))[l] = rew;
      ((float*)(ob + 5120))[l] = delta;
      (ob + 6144)[l] = term;
      (ob + 6400)[l] = trunc;
      (ob + 6656)[l] = flag;
    } else {
      r.flag[i] = flag;
      r.delta[i] = delta;
      r.obs[i] = o;
      r.reward[i] = rew;
      r.term[i] = term;
      r.trunc[i] = trunc;
    }
  }
}

// D: SoA rows, flag bytes packed into ONE byte row (term | trunc<<1 | flag<<2), status folded into t's sign
__global__ __launch_bounds__(256) void soa_packed_kernel(Rows r, int64_t N) {
  const int64_t chunks = (N + 255) / 256;
  for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
    const int64_t i = c * 256 + threadIdx.x;
    if (i >= N) continue;
    int tt = r.t[i];
    unsigned st = tt < 0;
    int t = tt & 0x7fffffff;
    double s[4];
    for (int k = 0; k < 4; k++) s[k] = r.phys[k * N + i];
    const int a = r.action[i];
    float er = r.er[i];
    double th = r.theta[i];
    f32x4 o;
    float rew, delta;
    unsigned term, trunc, flag;
    compute(s, th, t, er, st, a, o, rew, term, trunc, flag, delta);
    r.theta[i] = th;
    r.delta[i] = delta;
    for (int k = 0; k < 4; k++) r.phys[k * N + i] = s[k];
    r.obs[i] = o;
    r.t[i] = t | (st << 31);
    r.reward[i] = rew;
    r.term[i] = term | (trunc << 1) | (flag << 2);
    r.er[i] = er;
  }
}

// F: plain SoA rows, but the four byte rows leave the workgroup as FULL 128-B lines: every lane parks its four flag
// bytes in LDS as one dword, after a barrier wavefront k writes row k as 64 lanes x 4 B (256 contiguous bytes)
__global__ __launch_bounds__(256) void soa_linebytes_kernel(Rows r, int64_t N) {
  __shared__ uint32_t flags[256];
  const int64_t chunks = (N + 255) / 256;
  for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
    const int64_t i = c * 256 + threadIdx.x;
    unsigned st = 0, term = 0, trunc = 0, flag = 0;
    if (i < N) {
      st = r.status[i];
      int t = r.t[i];
      double s[4];
      for (int k = 0; k < 4; k++) s[k] = r.phys[k * N + i];
      const int a = r.action[i];
      float er = r.er[i];
      double th = r.theta[i];
      f32x4 o;
      float rew, delta;
      compute(s, th, t, er, st, a, o, rew, term, trunc, flag, delta);
      r.theta[i] = th;
      r.delta[i] = delta;
      for (int k = 0; k < 4; k++) r.phys[k * N + i] = s[k];
      r.obs[i] = o;
      r.t[i] = t;
      r.reward[i] = rew;
      r.er[i] = er;
    }
    flags[threadIdx.x] = term | (trunc << 8) | (flag << 16) | (st << 24);
    __syncthreads();
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;   // wavefront w writes row w
    const uint4 q = ((const uint4*)flags)[l];
    const unsigned sh = 8 * w;
    const uint32_t word = ((q.x >> sh) & 0xff) | (((q.y >> sh) & 0xff) << 8) | (((q.z >> sh) & 0xff) << 16) | (((q.w >> sh) & 0xff) << 24);
    uint8_t* row = w == 0 ? r.term : w == 1 ? r.trunc : w == 2 ? r.flag : r.status;
    if (c * 256 + 4 * l + 3 < N) ((uint32_t*)(row + c * 256))[l] = word;
    __syncthreads();
  }
}

// E: pure copy of the same byte count with 16-B accesses (upper bound of the memory system for this footprint)
__global__ __launch_bounds__(256) void copy16_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    f32x4 v = src[i];
    v.x += 1.f;
    dst[i] = v;
  }
}

// ---- FrozenLake-shaped skeleton (C3): per-env PCG64 record read + state half written every step ----
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
struct GridRows {
  u64x2* rng_aos;        // [N][2] x u64x2 = 32-byte records
  unsigned long long* rng_soa;  // [4][N]
  int32_t* cell; int32_t* t; uint8_t* status; const int32_t* action; double* prob3;  // [3][N]
  float* er; float* reward; uint8_t* term; uint8_t* trunc; float* prob; uint8_t* flag; float* delta;
};
__device__ __forceinline__ void pcg_step(unsigned long long& sh, unsigned long long& sl, unsigned long long ih, unsigned long long il) {
  const unsigned long long MH = 2549297995355413924ULL, ML = 4865540595714422341ULL;
  unsigned long long lo = sl * ML;
  unsigned long long hi = __umul64hi(sl, ML) + sh * ML + sl * MH;
  unsigned long long lo2 = lo + il;
  sl = lo2;
  sh = hi + ih + (lo2 < lo ? 1ULL : 0ULL);
}
template <bool SOA>
__global__ __launch_bounds__(256) void grid_kernel(GridRows r, int64_t N) {
  const int64_t chunks = (N + 255) / 256;
  for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
    const int64_t i = c * 256 + threadIdx.x;
    if (i >= N) continue;
    unsigned long long sh, sl, ih, il;
    if (SOA) { sh = r.rng_soa[i]; sl = r.rng_soa[N + i]; ih = r.rng_soa[2 * N + i]; il = r.rng_soa[3 * N + i]; }
    else { u64x2 a = r.rng_aos[2 * i], b = r.rng_aos[2 * i + 1]; sh = a.x; sl = a.y; ih = b.x; il = b.y; }
    unsigned st = r.status[i];
    int t = r.t[i];
    int cell = r.cell[i];
    const int a = r.action[i];
    float er = r.er[i];
    const double p0 = r.prob3[i], p1 = r.prob3[N + i], p2 = r.prob3[2 * N + i];
    pcg_step(sh, sl, ih, il);
    const unsigned long long x = sh ^ sl;
    const unsigned rot = (unsigned)(sh >> 58);
    const double u = (double)(((x >> rot) | (x << ((64u - rot) & 63u))) >> 11) * (1.0 / 9007199254740992.0);
    const int idx = p0 > u ? 0 : p0 + p1 > u ? 1 : 2;
    cell = (cell + a + idx + (int)st) & 63;
    t += 1;
    const unsigned term = cell == 63, trunc = t >= 100;
    st = term | trunc;
    er += (float)term;
    if (SOA) { r.rng_soa[i] = sh; r.rng_soa[N + i] = sl; }
    else r.rng_aos[2 * i] = u64x2{sh, sl};
    r.cell[i] = cell; r.t[i] = st ? 0 : t; r.status[i] = st; r.reward[i] = (float)term; r.term[i] = term; r.trunc[i] = trunc;
    r.prob[i] = (float)(idx == 0 ? p0 : idx == 1 ? p1 : p2); r.flag[i] = 0; r.delta[i] = 0.f; r.er[i] = er;
  }
}

template <typename F> float time_it(F&& launch, int iters) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int k = 0; k < 20; k++) launch();
  CHECK(hipEventRecord(e0));
  for (int k = 0; k < iters; k++) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : (1 << 20);
  const int iters = 300;
  const int64_t chunks = (N + 255) / 256;
  Rows r;
  CHECK(hipMalloc(&r.phys, 4 * N * 8));
  CHECK(hipMalloc(&r.theta, N * 8));
  CHECK(hipMalloc(&r.t, N * 4));
  CHECK(hipMalloc(&r.er, N * 4));
  CHECK(hipMalloc(&r.status, N));
  CHECK(hipMalloc((void**)&r.action, N * 4));
  CHECK(hipMalloc(&r.obs, N * 16));
  CHECK(hipMalloc(&r.reward, N * 4));
  CHECK(hipMalloc(&r.term, N));
  CHECK(hipMalloc(&r.trunc, N));
  CHECK(hipMalloc(&r.flag, N));
  CHECK(hipMalloc(&r.delta, N * 4));
  CHECK(hipMalloc(&r.blocked, chunks * 12544));
  CHECK(hipMalloc(&r.blocked_out, chunks * 7168));
  CHECK(hipMemset(r.phys, 0, 4 * N * 8));
  CHECK(hipMemset(r.theta, 0, N * 8));
  CHECK(hipMemset(r.t, 0, N * 4));
  CHECK(hipMemset(r.er, 0, N * 4));
  CHECK(hipMemset(r.status, 0, N));
  CHECK(hipMemset((void*)r.action, 0, N * 4));
  CHECK(hipMemset(r.blocked, 0, chunks * 12544));
  const int64_t copy_bytes = 60 * N;  // 60 B read + 60 B written per env = 120 B
  f32x4 *src, *dst;
  CHECK(hipMalloc(&src, copy_bytes));
  CHECK(hipMalloc(&dst, copy_bytes));
  CHECK(hipMemset(src, 0, copy_bytes));
  for (int grid : {4096, 2048, 1024}) {
    const int g = (int)(chunks < grid ? chunks : grid);
    float a = time_it([&] { hipLaunchKernelGGL(soa_kernel, dim3(g), dim3(256), 0, 0, r, N); }, iters);
    float b = time_it([&] { hipLaunchKernelGGL(blocked_kernel<false>, dim3(g), dim3(256), 0, 0, r, N); }, iters);
    float c = time_it([&] { hipLaunchKernelGGL(blocked_kernel<true>, dim3(g), dim3(256), 0, 0, r, N); }, iters);
    float d = time_it([&] { hipLaunchKernelGGL(soa_packed_kernel, dim3(g), dim3(256), 0, 0, r, N); }, iters);
    float f = time_it([&] { hipLaunchKernelGGL(soa_linebytes_kernel, dim3(g), dim3(256), 0, 0, r, N); }, iters);
    printf("N=%lld grid=%d  soa with byte rows written as full lines through LDS: %.2f us\n", (long long)N, g, f);
    float e = time_it([&] { hipLaunchKernelGGL(copy16_kernel, dim3(g), dim3(256), 0, 0, src, dst, copy_bytes / 16); }, iters);
    printf("N=%lld grid=%d  soa %.2f us | state-blocked %.2f | all-blocked %.2f | soa-packed-flags %.2f | copy16(120B/env) %.2f   [120 B/env-step: %.0f %.0f %.0f %.0f %.0f GB/s]\n",
           (long long)N, g, a, b, c, d, e, 120.0 * N / a / 1e3, 120.0 * N / b / 1e3, 120.0 * N / c / 1e3, 120.0 * N / d / 1e3,
           120.0 * N / e / 1e3);
  }
  GridRows gr;
  CHECK(hipMalloc(&gr.rng_aos, N * 32)); CHECK(hipMalloc(&gr.rng_soa, N * 32));
  CHECK(hipMemset(gr.rng_aos, 1, N * 32)); CHECK(hipMemset(gr.rng_soa, 1, N * 32));
  CHECK(hipMalloc(&gr.prob3, N * 24)); CHECK(hipMemset(gr.prob3, 0, N * 24));
  CHECK(hipMalloc(&gr.cell, N * 4)); CHECK(hipMemset(gr.cell, 0, N * 4));
  CHECK(hipMalloc(&gr.prob, N * 4));
  gr.t = r.t; gr.status = r.status; gr.action = r.action; gr.er = r.er; gr.reward = r.reward; gr.term = r.term; gr.trunc = r.trunc;
  gr.flag = r.flag; gr.delta = r.delta;
  {
    const int g = (int)(chunks < 4096 ? chunks : 4096);
    float a = time_it([&] { hipLaunchKernelGGL(grid_kernel<false>, dim3(g), dim3(256), 0, 0, gr, N); }, iters);
    float b = time_it([&] { hipLaunchKernelGGL(grid_kernel<true>, dim3(g), dim3(256), 0, 0, gr, N); }, iters);
    printf("FrozenLake-shaped skeleton (117 B/env-step real): rng AoS 32-B records %.2f us | rng SoA rows %.2f us\n", a, b);
  }
  return 0;
}
