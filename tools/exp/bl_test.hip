#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const int* p, int nbytes, int* out, int soff, int mode) {
  __shared__ __attribute__((aligned(16))) int smem[2048];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2048; i += 64) smem[i] = -1;
  __syncthreads();
  if (mode == 0) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)p + soff + lane * 16),
                                     (__attribute__((address_space(3))) void*)(smem + 256), 16, 0, 0);
  } else {
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + 256), 16, lane * 16, soff, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out[i] = smem[i];
}
int main() {
  const int N = 4096;
  std::vector<int> h(N);
  for (int i = 0; i < N; ++i) h[i] = i;
  int *d, *o;
  hipMalloc(&d, N * 4); hipMalloc(&o, 2048 * 4);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) for (int soff : {0, 4096}) {
    k<<<1, 64>>>(d, N * 4, o, soff, mode);
    std::vector<int> r(2048);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    printf("mode %d soff %d: first written idx:", mode, soff);
    int first = -1, cnt = 0;
    for (int i = 0; i < 2048; ++i) if (r[i] != -1) { if (first < 0) first = i; ++cnt; }
    printf(" %d count %d; values at first..+8:", first, cnt);
    for (int i = 0; i < 8 && first >= 0; ++i) printf(" %d", r[first + i]);
    printf(" ... lane1 chunk:");
    for (int i = 0; i < 4 && first >= 0; ++i) printf(" %d", r[first + 4 + i]);
    printf("\n");
  }
  return 0;
}
