// fp32 atomic-add throughput by where the contributors of an output tile run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 scripts/atomic_probe.hip -o /tmp/ap && /tmp/ap
// Block b runs on XCD b % 8.  Each block adds a 256 x 256 fp32 tile (one 256-float row per step, 256 threads) to output tile
// `tile(b)`.  "same XCD": tile = b % ntiles with ntiles % 8 == 0 (all contributors of a tile share an L2); "all XCDs": ntiles odd.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void add_tiles(float* out, int ntiles, int mode, int rows) {
  const int b = blockIdx.x;
  int t;
  if (mode == 0) t = b % ntiles;                 // contributors of tile t: b = t, t + ntiles, ... (XCD = b % 8)
  else t = (b / 8) % ntiles;                     // mode 1: consecutive blocks (8 XCDs) share a tile
  float* o = out + (size_t)t * rows * 256;
  for (int r = 0; r < rows; r++) __hip_atomic_fetch_add(o + r * 256 + threadIdx.x, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
int main() {
  const int rows = 256;
  float* out; hipMalloc(&out, (size_t)256 * rows * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Case { int blocks, ntiles, mode; const char* what; };
  std::vector<Case> cases = {
    {144, 8, 0, "8 tiles, same XCD per tile (18 contributors)"}, {144, 9, 0, "9 tiles, all XCDs (16 contributors)"},
    {144, 9, 1, "9 tiles, consecutive blocks share (all XCDs)"}, {144, 16, 0, "16 tiles same XCD (9 contributors)"},
    {144, 18, 0, "18 tiles, 4 XCDs each"}, {256, 16, 0, "256 blocks 16 tiles same XCD"}, {256, 17, 0, "256 blocks 17 tiles all XCDs"},
    {256, 256, 0, "256 blocks 256 tiles: no sharing"}, {1024, 64, 0, "1024 blocks 64 tiles same XCD"}, {1024, 63, 0, "1024 blocks 63 tiles all XCDs"},
  };
  for (auto& c : cases) {
    hipMemset(out, 0, (size_t)256 * rows * 256 * 4);
    for (int i = 0; i < 3; i++) add_tiles<<<c.blocks, 256>>>(out, c.ntiles, c.mode, rows);
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; i++) add_tiles<<<c.blocks, 256>>>(out, c.ntiles, c.mode, rows);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)c.blocks * rows * 256;
    printf("%-50s %7.1f us  %6.1f G adds/s\n", c.what, ms / it * 1e3, ops / (ms / it * 1e-3) / 1e9);
  }
  return 0;
}
