// stream_probe.hip — memory skeleton of the C1 CartPole step in the HBM-streaming regime (2^22-2^24 envs per GPU).
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip ; run on the GPU box: ./stream_probe [log2N ...]
//
// Every variant moves the algorithmic bytes of one C1 env-step (48 B read + 71-72 B written) with a
// trivial amount of arithmetic, so the numbers bound what a layout can give the real kernel.  Knobs (template flags):
//   STATE  0 = library layout of round 1: phys chunk-blocked, theta / t / status as flat [N] rows
//          1 = every persistent row of a 256-env chunk in ONE contiguous block (phys | theta | t | status)
//   OUT    0 = outputs as flat [N] rows (obs [N][4] f32, reward, delta, term, trunc, flag)
//          1 = outputs of a chunk in one contiguous block
//   PACK   0 = status byte row + three flag byte rows
//          1 = needs-reset bit in t's sign bit, term / trunc / flag as ONE dword row (4-B stores, full 256-B wave stores)
//   CPW    consecutive chunks a workgroup walks before it strides by the grid (longer contiguous runs per stream)
//   NT     non-temporal stores of the write-once outputs
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

struct Bufs {
  unsigned char* state;  // STATE 0: phys [chunk][4][256] f64 | theta [N] f64 | t [N] i32 | status [N] u8 (sub-pointers below)
  double* phys;
  double* theta;
  int32_t* t;
  uint8_t* status;
  const int32_t* action;
  unsigned char* out;  // OUT 1: [chunk][out_stride]
  f32x4* obs;
  float* reward;
  float* delta;
  uint8_t* term;
  uint8_t* trunc;
  uint8_t* flag;
  uint32_t* flags32;
};

template <bool NT, typename T> __device__ __forceinline__ void st_out(T* p, T v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

__device__ __forceinline__ void compute(double s[4], double& th, int& t, unsigned& st, int a, f32x4& o, float& rew,
                                        unsigned& term, unsigned& trunc, unsigned& flag, float& delta) {
  th += 0.1;
  const double f = a ? 10.0 : -10.0;
  s[1] += 0.02 * (f + th * s[2]);
  s[0] += 0.02 * s[1];
  s[3] += 0.02 * (s[2] * 9.8 - f * 0.05);
  s[2] += 0.02 * s[3];
  t += 1;
  rew = 1.f;
  term = (s[0] > 2.4 || s[0] < -2.4 || s[2] > 0.21 || s[2] < -0.21) ? 1u : 0u;
  trunc = t >= 500 ? 1u : 0u;
  if (st) { s[0] = 0.01; s[1] = -0.02; s[2] = 0.03; s[3] = 0.01; t = 0; th = 0.1; term = trunc = 0; }
  st = (term | trunc) ? 1u : 0u;
  o = f32x4{(float)s[0], (float)s[1], (float)s[2], (float)s[3]};
  flag = 1u;
  delta = 0.1f;
}

constexpr int state_stride(int STATE, int PACK) { return 8192 + 2048 + 1024 + (PACK ? 0 : 256); }
constexpr int out_stride(int PACK) { return 4096 + 1024 + 1024 + (PACK ? 1024 : 768); }

template <int STATE, int OUT, int PACK, int CPW, bool NT>
__global__ __launch_bounds__(256) void skel(Bufs r, int64_t N) {
  const int64_t chunks = (N + 255) / 256;
  const int l = threadIdx.x;
  for (int64_t c0 = (int64_t)blockIdx.x * CPW; c0 < chunks; c0 += (int64_t)gridDim.x * CPW) {
#pragma unroll 1
    for (int cc = 0; cc < CPW; cc++) {
      const int64_t c = c0 + cc;
      if (c >= chunks) break;
      const int64_t i = c * 256 + l;
      if (i >= N) continue;
      double* bp;
      double* bth;
      int32_t* bt;
      uint8_t* bst;
      if (STATE) {
        unsigned char* blk = r.state + c * state_stride(STATE, PACK);
        bp = (double*)blk; bth = (double*)(blk + 8192) + l; bt = (int32_t*)(blk + 10240) + l; bst = blk + 11264 + l;
      } else {
        bp = r.phys + c * 1024; bth = r.theta + i; bt = r.t + i; bst = r.status + i;
      }
      unsigned st;
      int t = *bt;
      if (PACK) { st = (unsigned)t >> 31; t &= 0x7fffffff; }
      else st = *bst;
      double s[4];
#pragma unroll
      for (int k = 0; k < 4; k++) s[k] = bp[k * 256 + l];
      const int a = r.action[i];
      double th = *bth;
      f32x4 o;
      float rew, delta;
      unsigned term, trunc, flag;
      compute(s, th, t, st, a, o, rew, term, trunc, flag, delta);
      *bth = th;
#pragma unroll
      for (int k = 0; k < 4; k++) bp[k * 256 + l] = s[k];
      if (PACK) *bt = t | (int)(st << 31);
      else { *bt = t; *bst = (uint8_t)st; }
      if (OUT) {
        unsigned char* ob = r.out + c * out_stride(PACK);
        st_out<NT>((f32x4*)ob + l, o);
        st_out<NT>((float*)(ob + 4096) + l, rew);
        st_out<NT>((float*)(ob + 5120) + l, delta);
        if (PACK) st_out<NT>((uint32_t*)(ob + 6144) + l, term | (trunc << 8) | (flag << 16));
        else {
          st_out<NT>(ob + 6144 + l, (unsigned char)term); st_out<NT>(ob + 6400 + l, (unsigned char)trunc);
          st_out<NT>(ob + 6656 + l, (unsigned char)flag);
        }
      } else {
        st_out<NT>(r.obs + i, o);
        st_out<NT>(r.reward + i, rew);
        st_out<NT>(r.delta + i, delta);
        if (PACK) st_out<NT>(r.flags32 + i, term | (trunc << 8) | (flag << 16));
        else { st_out<NT>(r.term + i, (uint8_t)term); st_out<NT>(r.trunc + i, (uint8_t)trunc); st_out<NT>(r.flag + i, (uint8_t)flag); }
      }
    }
  }
}


// ---- latency-chain probe: the lib-layout skeleton plus W dependent float64 FMAs per lane (the integrator's place), optionally two
// workgroup barriers (the reset hand-over's), optionally the NEXT chunk's rows requested before this chunk's arithmetic ----
struct RowsIn { unsigned st; int t; double s[4]; int a; double th; };
__device__ __forceinline__ RowsIn load_rows(const Bufs& r, int64_t c, int l) {
  RowsIn x;
  const int64_t i = c * 256 + l;
  const double* bp = r.phys + c * 1024;
  x.st = r.status[i]; x.t = r.t[i];
#pragma unroll
  for (int k = 0; k < 4; k++) x.s[k] = bp[k * 256 + l];
  x.a = r.action[i]; x.th = r.theta[i];
  return x;
}
template <int W, bool BARRIERS, bool PREFETCH>
__global__ __launch_bounds__(256) void chain(Bufs r, int64_t N) {
  __shared__ double hand[256];
  const int64_t chunks = N / 256;
  const int l = threadIdx.x;
  int64_t c = blockIdx.x;
  if (c >= chunks) return;
  RowsIn cur = load_rows(r, c, l);
  for (; c < chunks; c += gridDim.x) {
    const int64_t cn = c + gridDim.x;
    RowsIn nxt = cur;
    if (PREFETCH && cn < chunks) nxt = load_rows(r, cn, l);
    const int64_t i = c * 256 + l;
    double s[4] = {cur.s[0], cur.s[1], cur.s[2], cur.s[3]};
    double th = cur.th; int t = cur.t; unsigned st = cur.st;
    f32x4 o; float rew, delta; unsigned term, trunc, flag;
    compute(s, th, t, st, cur.a, o, rew, term, trunc, flag, delta);
    double acc = s[1];
#pragma unroll 8
    for (int w = 0; w < W; w++) acc = acc * 0.999999 + s[w & 3] * 1e-9;   // dependent chain: W x (mul + add), not contracted away
    s[1] = acc;
    if (BARRIERS) {
      hand[l] = s[0];
      __syncthreads();
      if (l < 13) hand[l * 7] = hand[l * 3] * 0.5 + 1e-3;
      __syncthreads();
      if (st) s[0] = hand[l];
    }
    double* bp = r.phys + c * 1024;
    r.theta[i] = th;
#pragma unroll
    for (int k = 0; k < 4; k++) bp[k * 256 + l] = s[k];
    r.t[i] = t; r.status[i] = (uint8_t)st;
    st_out<true>(r.obs + i, o); st_out<true>(r.reward + i, rew); st_out<true>(r.delta + i, delta);
    st_out<true>(r.term + i, (uint8_t)term); st_out<true>(r.trunc + i, (uint8_t)trunc); st_out<true>(r.flag + i, (uint8_t)flag);
    if (PREFETCH) cur = nxt;
    else if (cn < chunks) cur = load_rows(r, cn, l);
  }
}
template <int W, bool BARRIERS, bool PREFETCH> void run_chain(const Bufs& r, int64_t N, int grid, int iters) {
  const int64_t chunks = N / 256;
  const int g = (int)(chunks < grid ? chunks : grid);
  float us = time_it([&] { hipLaunchKernelGGL((chain<W, BARRIERS, PREFETCH>), dim3(g), dim3(256), 0, 0, r, N); }, iters);
  printf("{\"n_log2\": %d, \"variant\": \"chain\", \"fma_pairs\": %d, \"barriers\": %d, \"prefetch\": %d, \"grid\": %d, \"us\": %.2f, \"algorithmic_GBs\": %.0f}\n",
         (int)__builtin_ctzll((unsigned long long)N), W, (int)BARRIERS, (int)PREFETCH, g, us, 120.0 * N / us / 1e3);
  fflush(stdout);
}


// ---- scattered-record probe: the lib-layout skeleton plus, for ~5 % of the lanes (hash of env index and launch number), a
// 32-byte record read (two 16-B loads) and / or a 16-B write-back into a [N][4] u64 array: the reset path's stream accesses ----
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
template <int MODE>   // 0 none, 1 load+store, 2 load only, 3 store only, 4 load+store of 64-B aligned records ([N][8] u64), 5 = 1 with 4-B scattered stores too
__global__ __launch_bounds__(256) void scat(Bufs r, int64_t N, u64x2_t* rec, float* lastr, unsigned launch) {
  const int64_t chunks = N / 256;
  const int l = threadIdx.x;
  for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
    const int64_t i = c * 256 + l;
    RowsIn cur = load_rows(r, c, l);
    const bool hit = (((uint32_t)i * 2654435761u + launch * 40503u) >> 22) < 52u;   // ~5 %
    u64x2_t a = {0, 0}, b = {0, 0};
    const int64_t ri = MODE == 4 ? 4 * i : 2 * i;
    if (hit && (MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5)) { a = rec[ri]; b = rec[ri + 1]; }
    double s[4] = {cur.s[0], cur.s[1], cur.s[2], cur.s[3]};
    double th = cur.th; int t = cur.t; unsigned st = cur.st;
    f32x4 o; float rew, delta; unsigned term, trunc, flag;
    compute(s, th, t, st, cur.a, o, rew, term, trunc, flag, delta);
    if (hit) {
      a.x = a.x * 6364136223846793005ull + b.x; a.y = a.y * 6364136223846793005ull + b.y + 1;
      s[0] += (double)(a.x >> 11) * 1e-30;
      if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5) rec[ri] = a;
      if (MODE == 5) { lastr[i] = (float)t; lastr[N + i] = (float)t; }
    }
    double* bp = r.phys + c * 1024;
    r.theta[i] = th;
#pragma unroll
    for (int k = 0; k < 4; k++) bp[k * 256 + l] = s[k];
    r.t[i] = t; r.status[i] = (uint8_t)st;
    st_out<true>(r.obs + i, o); st_out<true>(r.reward + i, rew); st_out<true>(r.delta + i, delta);
    st_out<true>(r.term + i, (uint8_t)term); st_out<true>(r.trunc + i, (uint8_t)trunc); st_out<true>(r.flag + i, (uint8_t)flag);
  }
}
template <int MODE> void run_scat(const Bufs& r, int64_t N, int iters, u64x2_t* rec, float* lastr, const char* tag) {
  unsigned launch = 0;
  float us = time_it([&] { hipLaunchKernelGGL((scat<MODE>), dim3(4096), dim3(256), 0, 0, r, N, rec, lastr, launch++); }, iters);
  printf("{\"n_log2\": %d, \"variant\": \"scattered records: %s\", \"us\": %.2f, \"algorithmic_GBs\": %.0f}\n",
         (int)__builtin_ctzll((unsigned long long)N), tag, us, 120.0 * N / us / 1e3);
  fflush(stdout);
}


// ---- in-block records: the chunk's persistent rows AND its 256 stream records in one contiguous block
//   [phys 8192 | theta 2048 | t 1024 | epi 1024 (needs-reset bit + last episode length: replaces status + the two sparse 4-B stores) | records 8192]
// so that the sparse record accesses fall next to the bytes the workgroup streams anyway (same DRAM neighbourhood).
template <int RECS_IN_BLOCK, int DENSE_EPI>
__global__ __launch_bounds__(256) void inblk(Bufs r, int64_t N, unsigned char* blk_base, u64x2_t* rec_far, float* lastr, unsigned launch) {
  const int64_t chunks = N / 256;
  const int l = threadIdx.x;
  constexpr int STRIDE = 8192 + 2048 + 1024 + 1024 + 8192;
  for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
    const int64_t i = c * 256 + l;
    unsigned char* blk = blk_base + c * STRIDE;
    double* bp = (double*)blk;
    double* bth = (double*)(blk + 8192) + l;
    int32_t* bt = (int32_t*)(blk + 10240) + l;
    int32_t* be = (int32_t*)(blk + 11264) + l;
    u64x2_t* rec = RECS_IN_BLOCK ? (u64x2_t*)(blk + 12288) + 2 * l : rec_far + 2 * i;
    int t = *bt;
    int epi = DENSE_EPI ? *be : (int)r.status[i];
    unsigned st = epi & 1;
    double s[4];
#pragma unroll
    for (int k = 0; k < 4; k++) s[k] = bp[k * 256 + l];
    const int a = r.action[i];
    double th = *bth;
    const bool hit = (((uint32_t)i * 2654435761u + launch * 40503u) >> 22) < 52u;   // ~5 %
    u64x2_t ra = {0, 0}, rb = {0, 0};
    if (hit) { ra = rec[0]; rb = rec[1]; }
    f32x4 o; float rew, delta; unsigned term, trunc, flag;
    compute(s, th, t, st, a, o, rew, term, trunc, flag, delta);
    if (hit) {
      ra.x = ra.x * 6364136223846793005ull + rb.x; ra.y = ra.y * 6364136223846793005ull + rb.y + 1;
      s[0] += (double)(ra.x >> 11) * 1e-30;
      rec[0] = ra;
      if (DENSE_EPI) epi = (t << 1);
      else { lastr[i] = (float)t; lastr[N + i] = (float)t; }
    }
    *bth = th;
#pragma unroll
    for (int k = 0; k < 4; k++) bp[k * 256 + l] = s[k];
    *bt = t;
    if (DENSE_EPI) *be = (epi & ~1) | (int)st;
    else r.status[i] = (uint8_t)st;
    st_out<true>(r.obs + i, o); st_out<true>(r.reward + i, rew); st_out<true>(r.delta + i, delta);
    st_out<true>(r.term + i, (uint8_t)term); st_out<true>(r.trunc + i, (uint8_t)trunc); st_out<true>(r.flag + i, (uint8_t)flag);
  }
}
template <int RECS_IN_BLOCK, int DENSE_EPI> void run_inblk(const Bufs& r, int64_t N, int iters, unsigned char* blk, u64x2_t* rec, float* lastr, const char* tag) {
  unsigned launch = 0;
  float us = time_it([&] { hipLaunchKernelGGL((inblk<RECS_IN_BLOCK, DENSE_EPI>), dim3(4096), dim3(256), 0, 0, r, N, blk, rec, lastr, launch++); }, iters);
  printf("{\"n_log2\": %d, \"variant\": \"blocked state, %s\", \"us\": %.2f, \"algorithmic_GBs\": %.0f}\n",
         (int)__builtin_ctzll((unsigned long long)N), tag, us, 120.0 * N / us / 1e3);
  fflush(stdout);
}

// pure copy of the same byte count, 16-B accesses (ceiling of the memory system for this footprint and grid)
__global__ __launch_bounds__(256) void copy16_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    f32x4 v = src[i];
    v.x += 1.f;
    dst[i] = v;
  }
}
// read-modify-write in place (the state rows' pattern: every byte read is written back), 16-B accesses
__global__ __launch_bounds__(256) void rmw16_kernel(f32x4* __restrict__ buf, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    f32x4 v = buf[i];
    v.x += 1.f;
    buf[i] = v;
  }
}

template <typename F> float time_it(F&& launch, int iters) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int k = 0; k < 5; k++) launch();
  CHECK(hipEventRecord(e0));
  for (int k = 0; k < iters; k++) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return ms * 1000.f / iters;
}

template <int STATE, int OUT, int PACK, int CPW, bool NT> void run(const Bufs& r, int64_t N, int grid, int iters, const char* tag) {
  const int64_t chunks = (N + 255) / 256;
  const int g = (int)((chunks + CPW - 1) / CPW < grid ? (chunks + CPW - 1) / CPW : grid);
  float us = time_it([&] { hipLaunchKernelGGL((skel<STATE, OUT, PACK, CPW, NT>), dim3(g), dim3(256), 0, 0, r, N); }, iters);
  printf("{\"n_log2\": %d, \"variant\": \"%s\", \"state_blocked\": %d, \"out_blocked\": %d, \"pack\": %d, \"cpw\": %d, \"nt\": %d, \"grid\": %d, "
         "\"us\": %.2f, \"algorithmic_GBs\": %.0f}\n",
         (int)__builtin_ctzll((unsigned long long)N), tag, STATE, OUT, PACK, CPW, (int)NT, g, us, 120.0 * N / us / 1e3);
  fflush(stdout);
}

int main(int argc, char** argv) {
  std::vector<int> sizes;
  for (int k = 1; k < argc; k++) sizes.push_back(atoi(argv[k]));
  if (sizes.empty()) sizes = {20, 22, 24};
  for (int lg : sizes) {
    const int64_t N = 1LL << lg;
    const int iters = lg >= 24 ? 40 : lg >= 22 ? 100 : 300;
    const int64_t chunks = (N + 255) / 256;
    Bufs r;
    // one allocation per role so that the flat-row and the blocked variants do not alias (both are kept zero-initialised)
    const size_t state_bytes = (size_t)chunks * 11520, out_bytes = (size_t)chunks * 7168;
    unsigned char *state, *out, *rows, *orow;
    CHECK(hipMalloc(&state, state_bytes));
    CHECK(hipMalloc(&out, out_bytes));
    CHECK(hipMalloc(&rows, (size_t)N * (32 + 8 + 4 + 1) + 4096));
    CHECK(hipMalloc(&orow, (size_t)N * (16 + 4 + 4 + 4 + 3) + 4096));
    CHECK(hipMemset(state, 0, state_bytes));
    CHECK(hipMemset(rows, 0, (size_t)N * 45 + 4096));
    CHECK(hipMalloc((void**)&r.action, N * 4));
    CHECK(hipMemset((void*)r.action, 0, N * 4));
    r.state = state;
    r.out = out;
    r.phys = (double*)rows;
    r.theta = (double*)(rows + (size_t)N * 32);
    r.t = (int32_t*)(rows + (size_t)N * 40);
    r.status = rows + (size_t)N * 44;
    r.obs = (f32x4*)orow;
    r.reward = (float*)(orow + (size_t)N * 16);
    r.delta = (float*)(orow + (size_t)N * 20);
    r.flags32 = (uint32_t*)(orow + (size_t)N * 24);
    r.term = orow + (size_t)N * 28;
    r.trunc = orow + (size_t)N * 29;
    r.flag = orow + (size_t)N * 30;
    const int64_t copy_bytes = 60 * N;
    f32x4 *src, *dst;
    CHECK(hipMalloc(&src, copy_bytes));
    CHECK(hipMalloc(&dst, copy_bytes));
    CHECK(hipMemset(src, 0, copy_bytes));
    if (getenv("PROBE_SCAT")) {
      u64x2_t* rec; float* lastr;
      CHECK(hipMalloc(&rec, (size_t)N * 64)); CHECK(hipMemset(rec, 1, (size_t)N * 64));
      CHECK(hipMalloc(&lastr, (size_t)N * 8)); CHECK(hipMemset(lastr, 0, (size_t)N * 8));
      for (int rep = 0; rep < 2; rep++) {
        run_scat<0>(r, N, iters, rec, lastr, "none");
        run_scat<1>(r, N, iters, rec, lastr, "32-B read + 16-B write-back");
        run_scat<2>(r, N, iters, rec, lastr, "32-B read only");
        run_scat<3>(r, N, iters, rec, lastr, "16-B write only");
        run_scat<4>(r, N, iters, rec, lastr, "64-B aligned records, 32-B read + 16-B write-back");
        run_scat<5>(r, N, iters, rec, lastr, "32-B read + 16-B write-back + two 4-B stores");
      }
      unsigned char* blk;
      CHECK(hipMalloc(&blk, (size_t)chunks * 20480)); CHECK(hipMemset(blk, 0, (size_t)chunks * 20480));
      for (int rep = 0; rep < 2; rep++) {
        run_inblk<0, 0>(r, N, iters, blk, rec, lastr, "far records, sparse last-return stores");
        run_inblk<1, 0>(r, N, iters, blk, rec, lastr, "IN-BLOCK records, sparse last-return stores");
        run_inblk<0, 1>(r, N, iters, blk, rec, lastr, "far records, dense epi row");
        run_inblk<1, 1>(r, N, iters, blk, rec, lastr, "IN-BLOCK records, dense epi row");
      }
      CHECK(hipFree(blk));
      CHECK(hipFree(rec)); CHECK(hipFree(lastr));
      continue;
    }
    if (getenv("PROBE_CHAIN")) {
      for (int G : {4096, 2048}) {
        run_chain<0, false, false>(r, N, G, iters); run_chain<0, false, true>(r, N, G, iters);
        run_chain<0, true, false>(r, N, G, iters); run_chain<0, true, true>(r, N, G, iters);
        run_chain<100, false, false>(r, N, G, iters); run_chain<100, false, true>(r, N, G, iters);
        run_chain<100, true, false>(r, N, G, iters); run_chain<100, true, true>(r, N, G, iters);
        run_chain<200, false, false>(r, N, G, iters); run_chain<200, false, true>(r, N, G, iters);
        run_chain<200, true, false>(r, N, G, iters); run_chain<200, true, true>(r, N, G, iters);
        run_chain<400, false, false>(r, N, G, iters); run_chain<400, false, true>(r, N, G, iters);
        run_chain<400, true, false>(r, N, G, iters); run_chain<400, true, true>(r, N, G, iters);
      }
      run_chain<200, true, false>(r, N, 16384, iters); run_chain<200, false, false>(r, N, 16384, iters);
      continue;
    }
    for (int rep = 0; rep < 2; rep++) {
      const int G = 4096;
      run<0, 0, 0, 1, true>(r, N, G, iters, "lib: phys blocked, flat rows, nt outputs");
      run<0, 0, 0, 1, false>(r, N, G, iters, "lib layout, plain stores");
      run<1, 0, 0, 1, true>(r, N, G, iters, "state blocked");
      run<1, 1, 0, 1, true>(r, N, G, iters, "state+out blocked");
      run<0, 0, 1, 1, true>(r, N, G, iters, "lib layout, packed flags");
      run<1, 0, 1, 1, true>(r, N, G, iters, "state blocked, packed flags");
      run<1, 1, 1, 1, true>(r, N, G, iters, "state+out blocked, packed flags");
      run<1, 1, 1, 1, false>(r, N, G, iters, "state+out blocked, packed flags, plain stores");
      run<0, 0, 0, 4, true>(r, N, G, iters, "lib layout, 4 chunks per wg");
      run<1, 0, 1, 4, true>(r, N, G, iters, "state blocked, packed, 4 chunks per wg");
      run<1, 1, 1, 4, true>(r, N, G, iters, "state+out blocked, packed, 4 chunks per wg");
      run<1, 1, 1, 2, true>(r, N, G, iters, "state+out blocked, packed, 2 chunks per wg");
      run<1, 1, 1, 1, true>(r, N, 2048, iters, "state+out blocked, packed flags, grid 2048");
      run<1, 1, 1, 1, true>(r, N, 16384, iters, "state+out blocked, packed flags, grid 16384");
      run<0, 0, 0, 1, true>(r, N, 16384, iters, "lib layout, grid 16384");
      float e = time_it([&] { hipLaunchKernelGGL(copy16_kernel, dim3(G), dim3(256), 0, 0, src, dst, copy_bytes / 16); }, iters);
      printf("{\"n_log2\": %d, \"variant\": \"copy16 60 B in + 60 B out per env\", \"us\": %.2f, \"algorithmic_GBs\": %.0f}\n", lg, e, 120.0 * N / e / 1e3);
      float f = time_it([&] { hipLaunchKernelGGL(rmw16_kernel, dim3(G), dim3(256), 0, 0, src, copy_bytes / 16); }, iters);
      printf("{\"n_log2\": %d, \"variant\": \"rmw16 in place, 60 B per env read and written back\", \"us\": %.2f, \"algorithmic_GBs\": %.0f}\n", lg, f, 120.0 * N / f / 1e3);
      fflush(stdout);
    }
    CHECK(hipFree(state)); CHECK(hipFree(out)); CHECK(hipFree(rows)); CHECK(hipFree(orow)); CHECK(hipFree((void*)r.action));
    CHECK(hipFree(src)); CHECK(hipFree(dst));
  }
  return 0;
}
