// probe: how much dynamic LDS can a workgroup get on this device/runtime?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out, int n) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = lds[n - 1];
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int a = 0, b = 0, c = 0;
    hipDeviceGetAttribute(&a, hipDeviceAttributeMaxSharedMemoryPerBlock, 0);
    hipDeviceGetAttribute(&b, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0);
    hipDeviceGetAttribute(&c, hipDeviceAttributeSharedMemPerBlockOptin, 0);
    printf("%s arch=%s sharedMemPerBlock=%zu attrPerBlock=%d perCU=%d optin=%d CUs=%d clock=%d\n", p.name, p.gcnArchName,
           p.sharedMemPerBlock, a, b, c, p.multiProcessorCount, p.clockRate);
    float* out; hipMalloc(&out, 4);
    for (int kb : {32, 64, 96, 128, 160}) {
        size_t bytes = (size_t)kb * 1024;
        hipError_t e1 = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        hipGetLastError();
        hipLaunchKernelGGL(k, dim3(1), dim3(256), bytes, 0, out, (int)(bytes / 4));
        hipError_t e2 = hipGetLastError();
        hipError_t e3 = hipDeviceSynchronize();
        float v = -1; hipMemcpy(&v, out, 4, hipMemcpyDeviceToHost);
        printf("  %3d KiB: setattr=%s launch=%s sync=%s value=%.0f\n", kb, hipGetErrorName(e1), hipGetErrorName(e2), hipGetErrorName(e3), v);
    }
    return 0;
}
