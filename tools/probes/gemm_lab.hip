// gemm_lab.hip — the eight-wave prefill GEMMs of kernels/gemm_dma.h as a standalone harness (round 5): the product kernels and candidate variants on the
// three shapes of a Llama-3.2-1B layer at S = 2048, timed with HIP events over distinct weight buffers (a prefill streams every layer's weights cold),
// value-checked against the product kernel, with parts compiled out (DIS bits) to see where a launch's time goes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I../../tinygpt_amd/csrc -I. gemm_lab.hip -o build/gemm_lab
//   build/gemm_lab [reps]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "archive/gemm_dma_r05_lab.h"      // round 5's header with its lab-only template switches (DIS / PP / TEPI / WJ); the product header dropped them in round 6
#include "gemm_lab_variants.h"

using namespace tgx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

struct Shape { const char* name; int M, N, K; int epi; };   // epi: 0 residual (o_proj / down), 1 silu (gate_up: N = 2 * inter)

constexpr int NL = 4;      // distinct weight buffers cycled by the timed loop

struct Bufs {
  bf16_t *Ah, *Al, *Ai, *B[NL], *oh, *ol;      // Ai: the two terms interleaved per k32 block, [M][K/32][hi 32 | lo 32]
  float *C, *Cref;
};

template <typename F>
static double time_us(int reps, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 12; i++) launch(i % NL);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) launch(i % NL);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return 1000.0 * ms / reps;
}

static GemmArgs make_args(const Shape& s, const Bufs& b, int l, float* C) {
  GemmArgs g{};
  g.A_hi = b.Ah; g.A_lo = b.Al; g.A_lo2 = nullptr; g.B = b.B[l]; g.bias = nullptr; g.C = C;
  g.M = s.M; g.N = s.N; g.K = s.K; g.ldc = s.N; g.inter = s.N / 2; g.out_hi = b.oh; g.out_lo = b.ol;
  g.three_from = 1 << 30; g.xcd_tiles = 1;
  return g;
}

template <typename K>
static void set_lds(K kern, size_t bytes) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)); }

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 40;
  const Shape shapes[] = {{"o_proj  2048x2048x2048", 2048, 2048, 2048, 0}, {"down    2048x2048x8192", 2048, 2048, 8192, 0}, {"gate_up 2048x16384x2048", 2048, 16384, 2048, 1}};
  constexpr size_t LDS8 = 3 * 3 * 256 * 32 * 2, LDS8K = 3 * 3 * 128 * 64 * 2;
  // kernels under test
  auto k8_res = gemm_dma8_kernel<DT_BF16, GEMM_RESIDUAL>; auto k8_silu = gemm_dma8_kernel<DT_BF16, GEMM_SILU>;
  auto w2_silu = gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 2>;          // rounds 2-4: wave = 128 x 64
  auto k8_silu_direct = gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 0, false>;    // the direct (2 bytes per lane) siluMul epilogue
  set_lds(k8_silu_direct, LDS8);
  auto k8k_res = gemm_dma8k_kernel<DT_BF16, GEMM_RESIDUAL>;
  set_lds(k8_res, LDS8); set_lds(k8_silu, LDS8); set_lds(w2_silu, LDS8); set_lds(k8k_res, LDS8K);
#define DISK(name, kern, lds) auto name = kern; set_lds(name, lds);
  constexpr size_t LDS8I = 5 * 256 * 128;
  DISK(i_silu, (gemm_dma8i_kernel<DT_BF16, GEMM_SILU>), LDS8I)  DISK(i_silu_d1, (gemm_dma8i_kernel<DT_BF16, GEMM_SILU, 1>), LDS8I)
  DISK(i_silu_d2, (gemm_dma8i_kernel<DT_BF16, GEMM_SILU, 2>), LDS8I)  DISK(i_silu_d4, (gemm_dma8i_kernel<DT_BF16, GEMM_SILU, 4>), LDS8I)
  DISK(i_silu_d8, (gemm_dma8i_kernel<DT_BF16, GEMM_SILU, 8>), LDS8I)
  DISK(k8_silu_d1, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 1>), LDS8)  DISK(k8_silu_d2, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 2>), LDS8)
  DISK(k8_silu_d4, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 4>), LDS8)  DISK(k8_silu_d8, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 8>), LDS8)
  DISK(k8_silu_d6, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 6>), LDS8)  DISK(k8_silu_d7, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 7>), LDS8)
  DISK(k8k_res_d1, (gemm_dma8k_kernel<DT_BF16, GEMM_RESIDUAL, true, 1>), LDS8K)  DISK(k8k_res_d2, (gemm_dma8k_kernel<DT_BF16, GEMM_RESIDUAL, true, 2>), LDS8K)
  DISK(pp1_silu, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 0, true, 1>), LDS8)  DISK(pp3_silu, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 0, true, 3>), LDS8)
  DISK(pp5_silu, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 0, true, 5>), LDS8)  DISK(pp1_silu_d2, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 2, true, 1>), LDS8)
  DISK(pp1_silu_d4, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 4, true, 1>), LDS8)  DISK(pp1_silu_d8, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 8, true, 1>), LDS8)
  DISK(pp1_silu_d1, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 1, true, 1>), LDS8)  DISK(pp1_silu_d6, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 6, true, 1>), LDS8)
  DISK(pp9_silu, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 0, true, 9>), LDS8)
  DISK(ip0_silu, (gemm_dma8ip_kernel<DT_BF16, GEMM_SILU, 0, 0>), LDS8I)  DISK(ip2_silu, (gemm_dma8ip_kernel<DT_BF16, GEMM_SILU, 0, 2>), LDS8I)
  DISK(ip4_silu, (gemm_dma8ip_kernel<DT_BF16, GEMM_SILU, 0, 4>), LDS8I)  DISK(ip6_silu, (gemm_dma8ip_kernel<DT_BF16, GEMM_SILU, 0, 6>), LDS8I)
  DISK(ip20_silu, (gemm_dma8ip_kernel<DT_BF16, GEMM_SILU, 0, 20>), LDS8I)
  DISK(ip4_silu_d1, (gemm_dma8ip_kernel<DT_BF16, GEMM_SILU, 1, 4>), LDS8I)  DISK(ip4_silu_d2, (gemm_dma8ip_kernel<DT_BF16, GEMM_SILU, 2, 4>), LDS8I)
  DISK(ip4_silu_d4, (gemm_dma8ip_kernel<DT_BF16, GEMM_SILU, 4, 4>), LDS8I)  DISK(ip4_silu_d8, (gemm_dma8ip_kernel<DT_BF16, GEMM_SILU, 8, 4>), LDS8I)
  DISK(i_part, (gemm_dma8i_kernel<DT_BF16, GEMM_PARTIAL>), LDS8I)
  DISK(k8_part, (gemm_dma8_kernel<DT_BF16, GEMM_PARTIAL>), LDS8)  DISK(k8k_part, (gemm_dma8k_kernel<DT_BF16, GEMM_PARTIAL>), LDS8K)
  DISK(k8_silu_d18, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 18>), LDS8)
  DISK(k8_silu_d17, (gemm_dma8_kernel<DT_BF16, GEMM_SILU, true, 4, 17>), LDS8)
  DISK(pp1_res, (gemm_dma8_kernel<DT_BF16, GEMM_RESIDUAL, true, 4, 0, true, 1>), LDS8)
  DISK(k8k_res_d4, (gemm_dma8k_kernel<DT_BF16, GEMM_RESIDUAL, true, 4>), LDS8K)  DISK(k8k_res_d8, (gemm_dma8k_kernel<DT_BF16, GEMM_RESIDUAL, true, 8>), LDS8K)
  DISK(k8k_res_d6, (gemm_dma8k_kernel<DT_BF16, GEMM_RESIDUAL, true, 6>), LDS8K)  DISK(k8k_res_d7, (gemm_dma8k_kernel<DT_BF16, GEMM_RESIDUAL, true, 7>), LDS8K)

  for (const Shape& s : shapes) {
    Bufs b{};
    const size_t nA = (size_t)s.M * s.K, nB = (size_t)s.N * s.K, nC = (size_t)s.M * s.N;
    std::vector<uint16_t> hAh(nA), hAl(nA), hB(nB);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) & 0xFFFFFF) / 16777216.0f - 0.5f; };
    for (size_t i = 0; i < nA; i++) { const float x = 2.f * rnd(); const uint16_t h = f2bf(x); hAh[i] = h; hAl[i] = f2bf(x - bf2f(h)); }
    CK(hipMalloc(&b.Ah, nA * 2)); CK(hipMalloc(&b.Al, nA * 2)); CK(hipMalloc(&b.C, nC * 4)); CK(hipMalloc(&b.Cref, nC * 4));
    CK(hipMalloc(&b.oh, nC)); CK(hipMalloc(&b.ol, nC));          // [M][N/2] 16-bit
    CK(hipMemcpy(b.Ah, hAh.data(), nA * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(b.Al, hAl.data(), nA * 2, hipMemcpyHostToDevice));
    {
      std::vector<uint16_t> hAi(2 * nA);
      for (int m = 0; m < s.M; m++)
        for (int k = 0; k < s.K; k++) {
          const size_t o = ((size_t)m * (s.K / 32) + k / 32) * 64 + k % 32;
          hAi[o] = hAh[(size_t)m * s.K + k]; hAi[o + 32] = hAl[(size_t)m * s.K + k];
        }
      CK(hipMalloc(&b.Ai, 2 * nA * 2)); CK(hipMemcpy(b.Ai, hAi.data(), 2 * nA * 2, hipMemcpyHostToDevice));
    }
    for (int l = 0; l < NL; l++) {
      for (size_t i = 0; i < nB; i++) hB[i] = f2bf(0.04f * rnd());
      CK(hipMalloc(&b.B[l], nB * 2)); CK(hipMemcpy(b.B[l], hB.data(), nB * 2, hipMemcpyHostToDevice));
    }
    const double gf = 2.0 * s.M * s.N * (double)s.K * 2.0 * 1e-9;      // executed GFLOP (two terms)
    printf("== %s  (executed %.1f GFLOP; MFMA-ideal at 2.5 PF %.1f us)\n", s.name, gf, gf / 2.5e6 * 1e3);
    auto report = [&](const char* name, double us) { printf("  %-66s %8.1f us  %6.0f TF executed\n", name, us, gf / us * 1e-3); fflush(stdout); };
    const dim3 g256(s.N / 256, s.M / 256), g128(s.N / 128, s.M / 128), blk(512);
    // value check helper: run `launch` once on buffer C (zeroed), compare with Cref
    auto check = [&](const char* name, auto launch, bool silu) {
      std::vector<float> r(1 << 16), t(1 << 16);
      if (silu) {
        CK(hipMemset(b.oh, 0, nC)); CK(hipMemset(b.ol, 0, nC)); launch(0, b.C); CK(hipDeviceSynchronize());
        std::vector<uint16_t> x(1 << 16), y(1 << 16);
        CK(hipMemcpy(x.data(), b.oh, x.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), (uint16_t*)b.Cref, y.size() * 2, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < x.size(); i++) bad += x[i] != y[i];
        std::vector<uint16_t> xl(1 << 16), yl(1 << 16);
        CK(hipMemcpy(xl.data(), (uint16_t*)b.ol + ((size_t)s.M * s.N / 2 - xl.size()), xl.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(yl.data(), (uint16_t*)b.Cref + (size_t)s.M * s.N / 2 + ((size_t)s.M * s.N / 2 - yl.size()), yl.size() * 2, hipMemcpyDeviceToHost));
        size_t badl = 0; for (size_t i = 0; i < xl.size(); i++) badl += xl[i] != yl[i];
        printf("  check %-52s %zu of %zu hi words (first rows) + %zu of %zu lo words (last rows) differ from the product kernel\n", name, bad, x.size(), badl, xl.size());
      } else {
        CK(hipMemset(b.C, 0, nC * 4)); launch(0, b.C); CK(hipDeviceSynchronize());
        CK(hipMemcpy(t.data(), b.C, t.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), b.Cref, r.size() * 4, hipMemcpyDeviceToHost));
        double md = 0, mr = 0; for (size_t i = 0; i < r.size(); i++) { md = fmax(md, fabs((double)t[i] - r[i])); mr = fmax(mr, fabs((double)r[i])); }
        printf("  check %-52s max |d| / max |ref| = %.2e\n", name, md / mr);
      }
    };
    if (s.epi == 0) {
      auto L = [&](auto kern, dim3 grid, size_t lds) { return [&, kern, grid, lds](int l, float* C = nullptr) { GemmArgs g = make_args(s, b, l, C ? C : b.C); hipLaunchKernelGGL(kern, grid, blk, lds, 0, g); }; };
      report("product: gemm_dma8k (128x128, 8 waves = 2 row halves x 4 k quarters)", time_us(reps, [&](int l) { L(k8k_res, g128, LDS8K)(l); }));
      report("  DIS 1  (DMA + barriers only)", time_us(reps, [&](int l) { L(k8k_res_d1, g128, LDS8K)(l); }));
      report("  DIS 2  (no DMA after the prologue: MFMAs + fragment reads)", time_us(reps, [&](int l) { L(k8k_res_d2, g128, LDS8K)(l); }));
      report("  DIS 4  (no fragment reads)", time_us(reps, [&](int l) { L(k8k_res_d4, g128, LDS8K)(l); }));
      report("  DIS 8  (no epilogue)", time_us(reps, [&](int l) { L(k8k_res_d8, g128, LDS8K)(l); }));
      report("  DIS 6  (MFMAs on zero operands only)", time_us(reps, [&](int l) { L(k8k_res_d6, g128, LDS8K)(l); }));
      report("  DIS 7  (skeleton: barriers + LDS reduce + epilogue)", time_us(reps, [&](int l) { L(k8k_res_d7, g128, LDS8K)(l); }));
      report("gemm_dma8 (256x256: 64 tiles on 256 CUs)", time_us(reps, [&](int l) { L(k8_res, g256, LDS8)(l); }));
      if (s.K >= 4096) {
        float* part; CK(hipMalloc(&part, 4 * nC * 4));
        auto LP = [&](auto kern, dim3 grid, size_t lds, int z) { return [&, kern, grid, lds, z](int l) { GemmArgs g = make_args(s, b, l, b.C); g.part = part; g.nsplit = z; g.k_per = s.K / z; hipLaunchKernelGGL(kern, dim3(grid.x, grid.y, z), blk, lds, 0, g); }; };
        for (int rep = 0; rep < 2; rep++) {
          report("gemm_dma8k one slab (the product's launch: 128x128, store, the next norm launch adds)", time_us(reps, LP(k8k_part, g128, LDS8K, 1)));
          report("gemm_dma8 256x256 x 4 K slabs (256 workgroups, fp32 slab stores)", time_us(reps, LP(k8_part, g256, LDS8, 4)));
          report("gemm_dma8i 256x256 x 4 K slabs, full lines (interleaved A)", time_us(reps, [&](int l) { GemmArgs g = make_args(s, b, l, b.C); g.A_hi = b.Ai; g.A_lo = nullptr; g.part = part; g.nsplit = 4; g.k_per = s.K / 4; hipLaunchKernelGGL(i_part, dim3(g256.x, g256.y, 4), blk, LDS8I, 0, g); }));
          report("gemm_dma8 256x256 x 2 K slabs (128 workgroups)", time_us(reps, LP(k8_part, g256, LDS8, 2)));
          report("gemm_dma8k 128x128 x 2 K slabs (512 workgroups)", time_us(reps, LP(k8k_part, g128, LDS8K, 2)));
        }
        // value check: sum of the four slabs against the product kernel's C (C was zero: RESIDUAL adds)
        CK(hipMemset(b.C, 0, nC * 4)); L(k8k_res, g128, LDS8K)(0, b.C); LP(k8_part, g256, LDS8, 4)(0); CK(hipDeviceSynchronize());
        std::vector<float> r(1 << 16), t(4 << 16);
        CK(hipMemcpy(r.data(), b.C, r.size() * 4, hipMemcpyDeviceToHost));
        for (int z = 0; z < 4; z++) CK(hipMemcpy(t.data() + ((size_t)z << 16), part + (size_t)z * nC, r.size() * 4, hipMemcpyDeviceToHost));
        double md = 0, mr = 0; for (size_t i = 0; i < r.size(); i++) { const double v = ((double)t[i] + t[i + (1 << 16)]) + ((double)t[i + (2 << 16)] + t[i + (3 << 16)]); md = fmax(md, fabs(v - r[i])); mr = fmax(mr, fabs((double)r[i])); }
        printf("  check 4 slabs summed vs the 128x128 kernel: max |d| / max |ref| = %.2e\n", md / mr);
        { GemmArgs g = make_args(s, b, 0, b.C); g.A_hi = b.Ai; g.A_lo = nullptr; g.part = part; g.nsplit = 4; g.k_per = s.K / 4; CK(hipMemset(part, 0, 4 * nC * 4)); hipLaunchKernelGGL(i_part, dim3(g256.x, g256.y, 4), blk, LDS8I, 0, g); CK(hipDeviceSynchronize());
          std::vector<float> t2(4 << 16);
          for (int z = 0; z < 4; z++) CK(hipMemcpy(t2.data() + ((size_t)z << 16), part + (size_t)z * nC, r.size() * 4, hipMemcpyDeviceToHost));
          size_t bad = 0; for (size_t i = 0; i < t2.size(); i++) bad += t2[i] != t[i];
          printf("  check full-line 4 slabs vs half-line 4 slabs: %zu of %zu words differ\n", bad, t2.size()); }
        CK(hipFree(part));
      }
      report("gemm_dma8 ping-pong (256x256: 64 tiles on 256 CUs)", time_us(reps, [&](int l) { L(pp1_res, g256, LDS8)(l); }));
    } else {
      auto L = [&](auto kern, dim3 grid, size_t lds) { return [&, kern, grid, lds](int l, float* C = nullptr) { GemmArgs g = make_args(s, b, l, b.C); if (C == b.Cref) { g.out_hi = (bf16_t*)b.Cref; g.out_lo = g.out_hi + (size_t)s.M * s.N / 2; } hipLaunchKernelGGL(kern, grid, blk, lds, 0, g); }; };
      auto LI = [&](auto kern) { return [&, kern](int l, float* C = nullptr) { GemmArgs g = make_args(s, b, l, b.C); g.A_hi = b.Ai; g.A_lo = nullptr; hipLaunchKernelGGL(kern, g256, blk, LDS8I, 0, g); }; };
      auto base = L(k8_silu, g256, LDS8);
      base(0, b.Cref); CK(hipDeviceSynchronize());
      report("(clock warm-up pass, not a figure)", time_us(reps, [&](int l) { base(l); }));
      report("product: gemm_dma8 (256x256, wave = 64 x 128)", time_us(reps, [&](int l) { base(l); }));
      check("ping-pong (must be bit-identical)", L(pp1_silu, g256, LDS8), true);
      report("gemm_dma8 ping-pong (waves w / w + 4 half a stage apart)", time_us(reps, [&](int l) { L(pp1_silu, g256, LDS8)(l); }));
      report("  ping-pong + s_setprio 1 around the matrix phase", time_us(reps, [&](int l) { L(pp3_silu, g256, LDS8)(l); }));
      report("  ping-pong + static s_setprio 1 for waves 4-7", time_us(reps, [&](int l) { L(pp5_silu, g256, LDS8)(l); }));
      report("  ping-pong, DMA issue ahead of the fragment reads", time_us(reps, [&](int l) { L(pp9_silu, g256, LDS8)(l); }));
      report("  ping-pong DIS 1  (DMA + reads + barriers only)", time_us(reps, [&](int l) { L(pp1_silu_d1, g256, LDS8)(l); }));
      report("  ping-pong DIS 2  (no DMA after the prologue)", time_us(reps, [&](int l) { L(pp1_silu_d2, g256, LDS8)(l); }));
      report("  ping-pong DIS 4  (no fragment reads)", time_us(reps, [&](int l) { L(pp1_silu_d4, g256, LDS8)(l); }));
      report("  ping-pong DIS 6  (MFMAs on zero operands only)", time_us(reps, [&](int l) { L(pp1_silu_d6, g256, LDS8)(l); }));
      report("  ping-pong DIS 8  (no epilogue)", time_us(reps, [&](int l) { L(pp1_silu_d8, g256, LDS8)(l); }));
      report("product again", time_us(reps, [&](int l) { base(l); }));
      check("direct siluMul epilogue (must be bit-identical)", L(k8_silu_direct, g256, LDS8), true);
      report("gemm_dma8 with the direct siluMul epilogue (2-byte stores)", time_us(reps, [&](int l) { L(k8_silu_direct, g256, LDS8)(l); }));
      report("product again (LDS-transposed epilogue, 16-byte stores)", time_us(reps, [&](int l) { base(l); }));
      check("wave = 128 x 64 (rounds 2-4; must be bit-identical)", L(w2_silu, g256, LDS8), true);
      report("gemm_dma8 WJ 2 (wave = 128 x 64, rounds 2-4)", time_us(reps, [&](int l) { L(w2_silu, g256, LDS8)(l); }));
      check("gemm_dma8i (full lines: interleaved A, k64 B units)", LI(i_silu), true);
      report("gemm_dma8i (full lines: interleaved A, k64 B units, 5 x 32 KB ring)", time_us(reps, [&](int l) { LI(i_silu)(l); }));
      check("gemm_dma8ip (full lines + ping-pong)", LI(ip0_silu), true);
      for (int rep = 0; rep < 2; rep++) {
        report("gemm_dma8i again", time_us(reps, [&](int l) { LI(i_silu)(l); }));
        report("gemm_dma8ip (full lines + ping-pong)", time_us(reps, [&](int l) { LI(ip0_silu)(l); }));
        report("  + s_setprio 1 around the matrix phase", time_us(reps, [&](int l) { LI(ip2_silu)(l); }));
        report("  + static s_setprio 1 for waves 4-7", time_us(reps, [&](int l) { LI(ip4_silu)(l); }));
        report("  + both", time_us(reps, [&](int l) { LI(ip6_silu)(l); }));
        report("  static prio + one DMA piece per two fragment reads", time_us(reps, [&](int l) { LI(ip20_silu)(l); }));
        report("product (k32 stages, lock step, LDS-transposed epilogue)", time_us(reps, [&](int l) { base(l); }));
      }
#ifdef LAB_TIMING
      { long long* tmg; CK(hipMalloc(&tmg, 64 * 8)); CK(hipMemset(tmg, 0, 64 * 8));
        GemmArgs g = make_args(s, b, 0, b.C); g.A_hi = b.Ai; g.A_lo = nullptr; g.ssq_out = reinterpret_cast<float*>(tmg);
        hipLaunchKernelGGL(getenv("LAB_IP20") ? ip20_silu : ip4_silu, g256, blk, LDS8I, 0, g); CK(hipDeviceSynchronize());
        long long ht[64]; CK(hipMemcpy(ht, tmg, 64 * 8, hipMemcpyDeviceToHost));
        printf("  dma8ip per-phase cycles of one workgroup, per k32 step (%lld k64 blocks): fragment reads | DMA issue | landing wait | barrier | 32 MFMAs | barrier\n", ht[6]);
        for (int w = 0; w < 8; w++) { printf("    group %d wave %d:", w >> 2, w & 3); for (int i = 0; i < 6; i++) printf(" %6lld", ht[w * 8 + i] / (2 * ht[6])); printf("\n"); }
        CK(hipFree(tmg)); }
#endif
      report("  dma8ip DIS 1  (DMA + reads + barriers only)", time_us(reps, [&](int l) { LI(ip4_silu_d1)(l); }));
      report("  dma8ip DIS 2  (no DMA after the prologue)", time_us(reps, [&](int l) { LI(ip4_silu_d2)(l); }));
      report("  dma8ip DIS 4  (no fragment reads)", time_us(reps, [&](int l) { LI(ip4_silu_d4)(l); }));
      report("  dma8ip DIS 8  (no epilogue)", time_us(reps, [&](int l) { LI(ip4_silu_d8)(l); }));
      report("  dma8i DIS 1  (DMA + barriers only)", time_us(reps, [&](int l) { LI(i_silu_d1)(l); }));
      report("  dma8i DIS 2  (no DMA after the prologue)", time_us(reps, [&](int l) { LI(i_silu_d2)(l); }));
      report("  dma8i DIS 4  (no fragment reads)", time_us(reps, [&](int l) { LI(i_silu_d4)(l); }));
      report("  dma8i DIS 8  (no epilogue)", time_us(reps, [&](int l) { LI(i_silu_d8)(l); }));
      report("  dma8 DIS 1  (DMA + barriers only)", time_us(reps, [&](int l) { L(k8_silu_d1, g256, LDS8)(l); }));
      report("  dma8 DIS 17 (the operand stream free-running: no ring discipline, one wait at the end)", time_us(reps, [&](int l) { L(k8_silu_d17, g256, LDS8)(l); }));
      report("  dma8 DIS 18 (the same through buffer_load_dwordx4 ... lds)", time_us(reps, [&](int l) { L(k8_silu_d18, g256, LDS8)(l); }));
      report("  dma8 DIS 17 again", time_us(reps, [&](int l) { L(k8_silu_d17, g256, LDS8)(l); }));
      report("  dma8 DIS 2  (no DMA after the prologue: MFMAs + fragment reads)", time_us(reps, [&](int l) { L(k8_silu_d2, g256, LDS8)(l); }));
      report("  dma8 DIS 4  (no fragment reads)", time_us(reps, [&](int l) { L(k8_silu_d4, g256, LDS8)(l); }));
      report("  dma8 DIS 8  (no epilogue)", time_us(reps, [&](int l) { L(k8_silu_d8, g256, LDS8)(l); }));
      report("  dma8 DIS 6  (MFMAs on zero operands only)", time_us(reps, [&](int l) { L(k8_silu_d6, g256, LDS8)(l); }));
      report("  dma8 DIS 7  (skeleton: barriers + epilogue)", time_us(reps, [&](int l) { L(k8_silu_d7, g256, LDS8)(l); }));
    }
    CK(hipFree(b.Ai)); CK(hipFree(b.Ah)); CK(hipFree(b.Al)); CK(hipFree(b.C)); CK(hipFree(b.Cref)); CK(hipFree(b.oh)); CK(hipFree(b.ol));
    for (int l = 0; l < NL; l++) CK(hipFree(b.B[l]));
  }
  return 0;
}
