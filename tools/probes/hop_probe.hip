// hop_probe.hip — what does ONE all-to-all hand-off between the resident workgroups of a persistent kernel cost, against a kernel boundary?
// This is the number a persistent decode engine (or a fused gate_up + down launch) lives or dies by: a Llama layer has four such hand-offs.
//
// Loop: ITERS x { every workgroup reads the whole vector v_i (VEC floats), reduces it, writes its VEC / G slice of v_{i+1} }
//   mode 0: one launch per iteration inside a hipGraph               (the hand-off is the kernel boundary)
//   mode 1: one persistent launch, per-workgroup epoch flags          (data: agent-scope write-through stores; one release per workgroup;
//                                                                      wave 0 polls the G flags; data read with sc1 loads)
//   mode 2: one persistent launch, {value, tag} granules              (no flags, no fences: readers re-read until every tag is current)
//   mode 3: as mode 1 but the flags are ONE counter (atomic add), polled by lane 0
//   mode 4: as mode 1, then ONE agent-scope acquire per workgroup (buffer_inv sc1) and plain loads, so that the workgroups of an XCD share
//           the vector's lines in their L2 the way the launches of mode 0 do
// Every mode computes the same values (checksum printed).  Build: hipcc -O3 --offload-arch=gfx950 hop_probe.hip -o hop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned long long u64;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false)); }
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1, 0xf>(v); v += dpp_mov<0x4E, 0xf>(v); v += dpp_mov<0x141, 0xf>(v); v += dpp_mov<0x140, 0xf>(v);
  v += dpp_mov<0x142, 0xa>(v); v += dpp_mov<0x143, 0xc>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

struct Args {
  float* v[2];          // modes 0, 1, 3: the vector, double-buffered by iteration parity
  u64* g[2];            // mode 2: granules {float bits | tag << 32}
  unsigned* flags;      // [G] epoch of each workgroup (mode 1) / [0] arrival counter (mode 3)
  unsigned* err;
  int vec, iters, first;
};

__device__ __forceinline__ float block_total(float s, float* red) {
  s = wave_sum(s);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();                      // red reuse
  if (lane == 0) red[wv] = s;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float next_value(float total, int vec, int k) { return total * (0.5f / (float)vec) + (float)k * 1e-4f; }

// mode 0: one iteration per launch
__global__ __launch_bounds__(256) void step_kernel(const float* vin, float* vout, int vec) {
  __shared__ float red[4];
  const f32x4* v4 = reinterpret_cast<const f32x4*>(vin);
  float s = 0.f;
  for (int i = threadIdx.x; i < vec / 4; i += 256) { const f32x4 t = v4[i]; s += (t[0] + t[1]) + (t[2] + t[3]); }
  const float total = block_total(s, red);
  const int sl = vec / gridDim.x;
  if (threadIdx.x < sl) vout[blockIdx.x * sl + threadIdx.x] = next_value(total, vec, blockIdx.x * sl + threadIdx.x);
}

template <int MODE>
__global__ __launch_bounds__(256) void persistent_kernel(const Args a) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int G = gridDim.x, sl = a.vec / G;
  for (int it = 0; it < a.iters; it++) {
    const unsigned epoch = (unsigned)(a.first + it);          // the vector read now was completed by iteration `epoch` (0: the host)
    float s = 0.f;
    if (MODE == 2) {
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(a.g[epoch & 1], 0, 0x7fffffff, 0x00020000);
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
        s = 0.f;
        for (int i = threadIdx.x; i < a.vec / 2; i += 256) {                 // 16 B = two granules
          const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)i * 16u, 0, 16 /*sc1*/);
          ok &= (t[1] == epoch) & (t[3] == epoch);
          s += __uint_as_float(t[0]) + __uint_as_float(t[2]);
        }
        if (__all(ok)) break;
        if (++spins > (1u << 20)) { if (lane == 0) atomicCAS(a.err, 0u, 2u); break; }
      }
    } else {
      if (epoch > 0 || true) {
        if (wv == 0) {
          unsigned spins = 0;
          if (MODE == 1 || MODE == 4) {
            for (;;) {
              bool ok = true;
              for (int i = lane; i < G; i += 64) ok &= __hip_atomic_load(a.flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
              if (__all(ok)) break;
              if (++spins > (1u << 20)) { if (lane == 0) atomicCAS(a.err, 0u, 1u); break; }
            }
          } else {
            while (__hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * (unsigned)G) {
              if (++spins > (1u << 20)) { if (lane == 0) atomicCAS(a.err, 0u, 3u); break; }
            }
          }
        }
        if (MODE == 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
      }
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(a.v[epoch & 1], 0, 0x7fffffff, 0x00020000);
      for (int i = threadIdx.x; i < a.vec / 4; i += 256) {
        const u32x4 t = MODE == 4 ? __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)i * 16u, 0, 0)
                                  : __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)i * 16u, 0, 16 /*sc1*/);
        s += (__uint_as_float(t[0]) + __uint_as_float(t[1])) + (__uint_as_float(t[2]) + __uint_as_float(t[3]));
      }
    }
    const float total = block_total(s, red);
    const int k = blockIdx.x * sl + threadIdx.x;
    if (MODE == 2) {
      // granule sums pair differently from the f32x4 sums of the other modes: checksums are compared per mode family
      if (threadIdx.x < sl)
        __hip_atomic_store(a.g[(epoch + 1) & 1] + k, (u64)__float_as_uint(next_value(total, a.vec, k)) | ((u64)(epoch + 1) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (threadIdx.x < sl)                                                     // sl <= 64: wave 0 alone writes
        __hip_atomic_store(a.v[(epoch + 1) & 1] + k, next_value(total, a.vec, k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (wv == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the write-through stores have left
        if (lane == 0) {
          if (MODE == 1 || MODE == 4) __hip_atomic_store(a.flags + blockIdx.x, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else __hip_atomic_fetch_add(a.flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256;
  const int iters = 400, reps = 10;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (int vec : {2048, 8192}) {
    float* v[2]; u64* g[2]; unsigned *flags, *err;
    for (int i = 0; i < 2; i++) { CK(hipMalloc(&v[i], vec * 4)); CK(hipMalloc(&g[i], vec * 8)); }
    CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&err, 4));
    std::vector<float> h0(vec); std::vector<u64> g0(vec);
    for (int k = 0; k < vec; k++) { h0[k] = (float)(k % 17) * 0.01f; unsigned u; memcpy(&u, &h0[k], 4); g0[k] = u; }
    auto reset = [&]() {
      CK(hipMemcpy(v[0], h0.data(), vec * 4, hipMemcpyHostToDevice)); CK(hipMemset(v[1], 0, vec * 4));
      CK(hipMemcpy(g[0], g0.data(), vec * 8, hipMemcpyHostToDevice)); CK(hipMemset(g[1], 0xff, vec * 8));
      CK(hipMemset(flags, 0, 4096)); CK(hipMemset(err, 0, 4));
    };
    auto checksum = [&](bool gran, int final_buf) {
      double cs = 0;
      if (gran) { std::vector<u64> o(vec); CK(hipMemcpy(o.data(), g[final_buf], vec * 8, hipMemcpyDeviceToHost)); for (auto x : o) { unsigned u = (unsigned)x; float f; memcpy(&f, &u, 4); cs += f; } }
      else { std::vector<float> o(vec); CK(hipMemcpy(o.data(), v[final_buf], vec * 4, hipMemcpyDeviceToHost)); for (auto x : o) cs += x; }
      return cs;
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // mode 0
    {
      reset();
      hipGraph_t gr; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int it = 0; it < iters; it++) step_kernel<<<G, 256, 0, st>>>(v[it & 1], v[(it + 1) & 1], vec);
      CK(hipStreamEndCapture(st, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      const double cs = checksum(false, iters & 1);
      float best = 1e30f;
      for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("vec %5d  G %3d  mode 0 (graph of launches)      %7.3f us / iteration   checksum %.6f\n", vec, G, best * 1e3 / iters, cs);
    }
    for (int mode = 1; mode <= 4; mode++) {
      float best = 1e30f; double cs = 0; unsigned herr = 0;
      for (int r = 0; r < reps + 1; r++) {
        reset();
        Args a{}; a.v[0] = v[0]; a.v[1] = v[1]; a.g[0] = g[0]; a.g[1] = g[1]; a.flags = flags; a.err = err; a.vec = vec; a.iters = iters; a.first = 0;
        CK(hipEventRecord(e0, st));
        if (mode == 1) persistent_kernel<1><<<G, 256, 0, st>>>(a);
        if (mode == 2) persistent_kernel<2><<<G, 256, 0, st>>>(a);
        if (mode == 3) persistent_kernel<3><<<G, 256, 0, st>>>(a);
        if (mode == 4) persistent_kernel<4><<<G, 256, 0, st>>>(a);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r > 0 && ms < best) best = ms;
        if (r == 0) { cs = checksum(mode == 2, iters & 1); CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); }
      }
      const char* names[] = {"", "persistent, epoch flags   ", "persistent, granules      ", "persistent, one counter   ", "flags + acquire, L2 loads "};
      printf("vec %5d  G %3d  mode %d (%s) %7.3f us / iteration   checksum %.6f%s\n", vec, G, mode, names[mode], best * 1e3 / iters, cs, herr ? "  SPIN LIMIT HIT" : "");
    }
  }
  return 0;
}
