// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE for the access shape of the per-ballot comb tables.
// MI355X_MICROARCH.md states the x2 under-report only for 16-B-per-lane COALESCED streaming reads and asks for a calibration
// on a known byte count for any other shape.  The comb-table lookup of k_eq_table is: every lane reads one 160-byte entry
// (10 x uint4) of its own 5-KiB table (32 entries), tables contiguous per lane -> 160 useful bytes that straddle two 128-B lines.
//   kernel calib_stream : 16 B per lane, coalesced, every byte of the buffer once            (known: bytes = buffer)
//   kernel calib_gather : lane i reads entry e_i of table i, 160 B                           (known: useful = 160 N, lines = 256 N)
// Run:   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//        rocprofv3 --pmc FETCH_SIZE -d out -o pmc --output-format csv -- ./fetch_calib
// and compare the counter (KiB) per kernel with the byte counts this program prints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void calib_stream(const uint4* in, size_t n_quads, uint4* sink) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_quads; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = in[i];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if (acc.x == 0x12345678u) sink[0] = acc;     // never true for the fill pattern; keeps the loads alive
}
__global__ void calib_gather(const uint4* tables, size_t n_tables, unsigned seed, uint4* sink) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_tables; t += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)t * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned e = h & 31u;                          // entry of this lane's table
    const uint4* p = tables + t * 320 + e * 10;          // 5 KiB per table, 160 B per entry
#pragma unroll
    for (int q = 0; q < 10; ++q) { const uint4 v = p[q]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  }
  if (acc.x == 0x12345678u) sink[0] = acc;
}

int main() {
  const size_t n_tables = (size_t)4 << 20;               // 4 Mi tables x 5 KiB = 20 GiB: far beyond L2 (32 MiB) and the Infinity Cache
  const size_t bytes = n_tables * 5120;
  uint4 *buf = nullptr, *sink = nullptr;
  CHK(hipMalloc((void**)&buf, bytes));
  CHK(hipMalloc((void**)&sink, 64));
  CHK(hipMemset(buf, 0x5a, bytes));
  CHK(hipDeviceSynchronize());
  const size_t stream_bytes = (size_t)8 << 30;            // 8 GiB of the buffer, streamed once
  hipLaunchKernelGGL(calib_stream, dim3(2048), dim3(256), 0, 0, buf, stream_bytes / 16, sink);
  CHK(hipDeviceSynchronize());
  hipLaunchKernelGGL(calib_gather, dim3(2048), dim3(256), 0, 0, buf, n_tables, 1u, sink);
  CHK(hipDeviceSynchronize());
  printf("calib_stream: bytes read = %zu (%.1f KiB)\n", stream_bytes, stream_bytes / 1024.0);
  printf("calib_gather: useful bytes = %zu (%.1f KiB), 128-B lines touched = %zu bytes (%.1f KiB)\n", n_tables * 160, n_tables * 160 / 1024.0,
         n_tables * 256, n_tables * 256 / 1024.0);
  CHK(hipFree(buf)); CHK(hipFree(sink));
  return 0;
}
