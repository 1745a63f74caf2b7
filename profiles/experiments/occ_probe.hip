// How many 256-thread workgroups fit on a CU for a given dynamic-LDS size?  (hipcc --offload-arch=gfx950 occ_probe.hip -o occ_probe.bin)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(float* p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[255 - threadIdx.x]; }
int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int kb : {48, 52, 56, 57, 60, 64, 72, 76, 77, 78, 79, 80, 81, 82, 96, 128, 160}) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, (size_t)kb * 1024);
        printf("%3d KB -> %d blocks/CU (%s)\n", kb, n, hipGetErrorString(e));
    }
    return 0;
}
