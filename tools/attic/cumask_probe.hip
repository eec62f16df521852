// cumask_probe.hip -- how does hipExtStreamCreateWithCUMask map mask bits to
// XCDs / SEs / CUs on MI355X?  (experiment, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_=(x); if(e_!=hipSuccess){fprintf(stderr,"%s: %s\n",#x,hipGetErrorString(e_)); exit(1);} } while(0)

__global__ void where_kernel(uint32_t *out) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // burn a little time so blocks spread over the allowed CUs
  float x = threadIdx.x;
  for (int i = 0; i < 20000; ++i) x = x * 1.000001f + 0.5f;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = (xcc & 0xF) | (x == 0.123f ? 1u << 31 : 0); }
}

int main() {
  const int nb = 4096;
  uint32_t *d; CK(hipMalloc(&d, nb * 8));
  std::vector<uint32_t> h(nb * 2);
  auto run = [&](const char *name, std::vector<uint32_t> mask) {
    hipStream_t st;
    if (mask.empty()) CK(hipStreamCreate(&st));
    else CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    where_kernel<<<nb, 64, 0, st>>>(d);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
    std::map<int, std::set<int>> per_xcc;  // xcc -> set of (se,sh,cu)
    for (int b = 0; b < nb; ++b) {
      uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
      int cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
      per_xcc[xcc].insert(se * 32 + sh * 16 + cu);
    }
    int total = 0;
    printf("%-28s:", name);
    for (auto &kv : per_xcc) { printf(" xcc%d:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
    printf("  total distinct CUs=%d\n", total);
    if (total <= 16) for (auto &kv : per_xcc) { printf("    xcc%d:", kv.first); for (int c : kv.second) printf(" se%d.sh%d.cu%d", c / 32, (c / 16) & 1, c & 15); printf("\n"); }
    CK(hipStreamDestroy(st));
  };
  run("no mask", {});
  run("bits 0-7", {0xFFu, 0, 0, 0, 0, 0, 0, 0});
  run("bits 0-31", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0});
  run("bits 248-255", {0, 0, 0, 0, 0, 0, 0, 0xFF000000u});
  run("bit 0 of each word", {1, 1, 1, 1, 1, 1, 1, 1});
  run("bits 0,8,16,..,56", {0x01010101u, 0x01010101u, 0, 0, 0, 0, 0, 0});
  run("all but bits 0-7", {0xFFFFFF00u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u});
  run("all but bit0 of each word", {~1u, ~1u, ~1u, ~1u, ~1u, ~1u, ~1u, ~1u});
  return 0;
}
