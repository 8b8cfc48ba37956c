// Round 6 probe (gfx950): what one LDS-DMA copy (`global_load_lds_dwordx4`) costs the wave that issues it, and what its immediate
// offset addresses.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/glds_probe tools/glds_probe.hip ; run on the GPU box.
//   part 1 (semantics): M0 = 256, `global_load_lds_dwordx4 v_off, s[base:base+1] offset:1024`: where do the 1 KiB land in LDS, and
//           which global bytes are they?  (answers whether ONE M0 write can serve several copies of a stage through immediates)
//   part 2 (issue cost): 1 / 4 / 12 waves per CU (256 workgroups), every wave issues 9 copies per round, 64 rounds, sources L2-resident:
//           form A = the round-1..5 helper (save M0, set, s_nop, copy, restore; per-copy VALU address add),
//           form B = one M0 write per 3 copies + immediates, addresses precomputed, SGPR base advanced per round.
//           s_memtime cycles per copy, with and without a drain (`vmcnt(0)`) per round.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void sem_kernel(const unsigned* src, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned voff = (unsigned)lane * 16u;
  const unsigned m0v = 256u;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(src), "s"(m0v) : "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}

template <int FORM, int DRAIN>
__global__ __launch_bounds__(768) void issue_kernel(const char* src, unsigned long long* cyc, int rounds, unsigned row_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds_base = (unsigned)(size_t)(lptr_t)lds + (unsigned)wave * 9u * 1024u;
  // 16 rows x 64 B per copy; rows row_bytes apart (a [rows][C] bf16 tensor), 9 copies = 144 rows
  unsigned voff[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) voff[i] = (unsigned)((wave * 144 + i * 16 + (lane >> 2)) * row_bytes + (lane & 3) * 16);
  const char* base = src + (size_t)blockIdx.x * 4096;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < rounds; ++r) {
    if constexpr (FORM == 0) {
      const unsigned cofb = (unsigned)(r & 7) * 64u;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff[i] + cofb), "s"(base), "s"(lds_base + (unsigned)i * 1024u)
                     : "memory");
      }
    } else {
      const char* b = base + (size_t)(r & 7) * 64;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        asm volatile("s_add_u32 m0, %4, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024\n\tglobal_load_lds_dwordx4 %2, %3 offset:2048"
                     ::"v"(voff[3 * g]), "v"(voff[3 * g + 1] - 1024u), "v"(voff[3 * g + 2] - 2048u), "s"(b), "s"(lds_base), "i"(g * 3072)
                     : "memory");
      }
    }
    if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  if (lds[lane] == 0x12345u) cyc[0] = 0;   // keep the LDS live
}

template <int FORM, int DRAIN>
static void run_issue(const char* src, unsigned long long* cyc_d, int waves, unsigned row_bytes) {
  const int rounds = 64, grid = 256;
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(issue_kernel<FORM, DRAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int rep = 0; rep < 3; ++rep) {
    CHK(hipMemset(cyc_d, 0, grid * 16 * 8));
    hipLaunchKernelGGL((issue_kernel<FORM, DRAIN>), dim3(grid), dim3(waves * 64), (size_t)waves * 9 * 1024, 0, src, cyc_d, rounds, row_bytes);
    CHK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> h(grid * 16);
  CHK(hipMemcpy(h.data(), cyc_d, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0; unsigned long long mx = 0; int n = 0;
  for (int b = 0; b < grid; ++b) for (int w = 0; w < waves; ++w) { sum += (double)h[b * 16 + w]; mx = mx > h[b * 16 + w] ? mx : h[b * 16 + w]; ++n; }
  const double per_copy = sum / n / (rounds * 9.0);
  printf("  form %c drain %d waves/CU %2d row_bytes %4u: %7.1f cycles per copy per wave (max wave %.1f) -> %.1f B/clk/CU\n", FORM ? 'B' : 'A', DRAIN, waves,
         row_bytes, per_copy, (double)mx / (rounds * 9.0), 1024.0 * waves / per_copy);
}

int main() {
  // part 1
  unsigned *src_d, *out_d;
  std::vector<unsigned> src(4096), out(1024);
  for (int i = 0; i < 4096; ++i) src[i] = (unsigned)i;
  CHK(hipMalloc(&src_d, 4096 * 4)); CHK(hipMalloc(&out_d, 1024 * 4));
  CHK(hipMemcpy(src_d, src.data(), 4096 * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 4096, 0, src_d, out_d);
  CHK(hipDeviceSynchronize());
  CHK(hipMemcpy(out.data(), out_d, 1024 * 4, hipMemcpyDeviceToHost));
  int first = -1, last = -1;
  for (int i = 0; i < 1024; ++i) if (out[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; }
  printf("part 1: M0 = 256, offset:1024, lane offsets 16 B: LDS dwords [%d, %d] written (byte %d ..); first value = source dword %u (byte %u)\n", first, last,
         first * 4, first >= 0 ? out[first] : 0u, first >= 0 ? out[first] * 4 : 0u);
  printf("        => the immediate %s the LDS address and %s the global address\n", first * 4 == 256 + 1024 ? "IS ADDED TO" : (first * 4 == 256 ? "is NOT added to" : "?? "),
         (first >= 0 && out[first] == 256u) ? "IS ADDED TO" : ((first >= 0 && out[first] == 0u) ? "is NOT added to" : "??"));
  // part 2
  const size_t src_bytes = (size_t)64 << 20;
  char* big; unsigned long long* cyc_d;
  CHK(hipMalloc(&big, src_bytes)); CHK(hipMemset(big, 1, src_bytes)); CHK(hipMalloc(&cyc_d, 256 * 16 * 8));
  printf("part 2: 256 workgroups, 9 copies of 1 KiB per wave and round, 64 rounds\n");
  for (unsigned rb : {512u, 2048u}) {
    run_issue<0, 1>(big, cyc_d, 1, rb); run_issue<1, 1>(big, cyc_d, 1, rb);
    run_issue<0, 1>(big, cyc_d, 4, rb); run_issue<1, 1>(big, cyc_d, 4, rb);
    run_issue<0, 1>(big, cyc_d, 12, rb); run_issue<1, 1>(big, cyc_d, 12, rb);
    run_issue<0, 0>(big, cyc_d, 4, rb); run_issue<1, 0>(big, cyc_d, 4, rb);
    run_issue<0, 0>(big, cyc_d, 12, rb); run_issue<1, 0>(big, cyc_d, 12, rb);
  }
  return 0;
}
