// wait_value.hip - can a stream wait for a word that a RUNNING kernel of another stream writes (hipStreamWaitValue32 on
// device memory), and how soon after the write does the waiting stream's next kernel start?  (Round 6: the frame loop would
// start frame i + 1's primary pass the moment frame i's AO pass finds its queues dry.)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/wait_value.hip -o /tmp/wait_value && /tmp/wait_value
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ unsigned long long wall() { return __builtin_amdgcn_s_memrealtime(); }

__global__ void long_kernel(uint32_t *flag, uint32_t value, unsigned long long *stamps, int spin_us) {
    const unsigned long long t0 = wall();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t0;
    while (wall() - t0 < (unsigned long long)spin_us * 100ull / 2) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamps[1] = wall();
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    while (wall() - t0 < (unsigned long long)spin_us * 100ull) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2] = wall();
}
__global__ void next_kernel(unsigned long long *stamps) {
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[3] = wall();
}

int main() {
    int can = 0;
    CHECK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    uint32_t *flag;
    unsigned long long *stamps, h[4];
    CHECK(hipMalloc(&flag, 64));
    CHECK(hipMalloc(&stamps, 64));
    hipStream_t a, b;
    CHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipMemset(flag, 0, 64));
        CHECK(hipMemset(stamps, 0, 64));
        CHECK(hipDeviceSynchronize());
        CHECK(hipStreamWaitValue32(b, flag, (uint32_t)(rep + 1), hipStreamWaitValueGte, 0xffffffffu));
        hipLaunchKernelGGL(next_kernel, dim3(1), dim3(64), 0, b, stamps);
        hipLaunchKernelGGL(long_kernel, dim3(1), dim3(64), 0, a, flag, (uint32_t)(rep + 1), stamps, 400);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h, stamps, 32, hipMemcpyDeviceToHost));
        printf("long kernel 0 .. %.1f us, flag written at %.1f us, the waiting stream's kernel started at %.1f us (%.1f us after the write)\n",
               (h[2] - h[0]) / 100.0, (h[1] - h[0]) / 100.0, (h[3] - h[0]) / 100.0, ((double)h[3] - (double)h[1]) / 100.0);
    }
    return 0;
}
