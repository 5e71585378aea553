// calib_fetch.hip -- calibrates rocprofv3's FETCH_SIZE for k_phase's access
// pattern (MI355X_MICROARCH.md: "calibrate on a known byte count in your own
// access pattern before trusting an absolute"): scattered 16-byte and 4-byte
// reads over a buffer far larger than L2 + Infinity Cache, each address read
// exactly once, plus a coalesced 16 B/lane stream for reference.
//   hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o gpurun_out/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// each thread reads REC bytes at a pseudo-random, REC-aligned slot of a 128-byte line of its own
template <int REC>
__global__ void scattered(const unsigned char* __restrict__ buf, uint64_t nlines, unsigned* out, uint64_t nthreads) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nthreads) return;
  // a bijection on line indices: every line is touched by exactly one thread
  uint64_t line = (t * 0x9E3779B97F4A7C15ull) % nlines;  // odd multiplier modulo a power of two
  uint64_t off = line * 128 + (mix(t) % (128 / REC)) * REC;
  unsigned acc = 0;
  if (REC == 16) {
    uint4 v = *reinterpret_cast<const uint4*>(buf + off);
    acc = v.x ^ v.y ^ v.z ^ v.w;
  } else {
    acc = *reinterpret_cast<const unsigned*>(buf + off);
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void stream16(const uint4* __restrict__ buf, uint64_t n, unsigned* out) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  uint4 v = buf[t];
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) out[0] = 1;
}
int main() {
  const uint64_t bytes = 8ull << 30;  // 8 GiB
  unsigned char* buf;
  unsigned* out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  hipMemset(buf, 1, bytes);
  const uint64_t nlines = bytes / 128;  // 2^26, a power of two
  const uint64_t nthreads = nlines;     // one access per line
  scattered<16><<<(unsigned)((nthreads + 255) / 256), 256>>>(buf, nlines, out, nthreads);
  scattered<4><<<(unsigned)((nthreads + 255) / 256), 256>>>(buf, nlines, out, nthreads);
  stream16<<<(unsigned)((bytes / 16 + 255) / 256), 256>>>((const uint4*)buf, bytes / 16, out);
  hipDeviceSynchronize();
  printf("accesses=%llu scattered16_useful=%llu scattered4_useful=%llu lines128=%llu stream_bytes=%llu\n",
         (unsigned long long)nthreads, (unsigned long long)nthreads * 16, (unsigned long long)nthreads * 4,
         (unsigned long long)nthreads * 128, (unsigned long long)bytes);
  return 0;
}
