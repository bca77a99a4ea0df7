// ds_write_addtid_b32 semantics check: LDS address = M0[15:0] + offset + 4 * lane.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* __restrict__ in, float* __restrict__ out) {
    __shared__ float buf[2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float v0 = in[blockIdx.x * 2048 + wave * 64 + lane];
    float v1 = in[blockIdx.x * 2048 + 1024 + wave * 64 + lane];
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)buf + wave * 256);
    asm volatile("s_mov_b32 m0, %0\n\t"
                 "s_nop 0\n\t"
                 "ds_write_addtid_b32 %1 offset:0\n\t"
                 "ds_write_addtid_b32 %2 offset:4096\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 :: "s"(base), "v"(v0), "v"(v1) : "memory", "m0");
    __syncthreads();
    out[blockIdx.x * 2048 + tid] = buf[tid];
    out[blockIdx.x * 2048 + 1024 + tid] = buf[1024 + tid];
}
int main() {
    const int nb = 64, n = nb * 2048;
    std::vector<float> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *di, *dout;
    hipMalloc(&di, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(di, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nb), dim3(1024), 0, 0, di, dout);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += o[i] != h[i];
    printf("addtid test: %d mismatches of %d\n", bad, n);
    return bad != 0;
}
