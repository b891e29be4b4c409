// Where does ds_write_addtid_b32 put a lane's dword? (gfx950)   hipcc --offload-arch=gfx950 addtid_probe.hip -o addtid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out, int m0_step, int pad) {
    extern __shared__ int s[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = -1;
    __syncthreads();
    const int v = 1000 * (threadIdx.x >> 6) + (threadIdx.x & 63);     // wave * 1000 + lane
    const int m0v = __builtin_amdgcn_readfirstlane(pad + m0_step * (int)(threadIdx.x >> 6));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 4\n\tds_write_addtid_b32 %0 offset:0\n\ts_waitcnt lgkmcnt(0)" ::"v"(v), "s"(m0v) : "m0", "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) out[i] = s[i];
}
int main() {
    int* d;
    hipMalloc(&d, 4096 * 4);
    const int cfg[][2] = {{0, 0}, {256, 0}, {1024, 0}, {1024, 256}, {512, 128}};
    for (auto& c : cfg) {
        k<<<1, 256, 4096 * 4>>>(d, c[0], c[1]);
        int h[4096];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("M0 = %d + %d * wave:", c[1], c[0]);
        int n = 0;
        for (int i = 0; i < 4096; ++i) {
            n += h[i] != -1;
            if (h[i] != -1 && h[i] % 1000 == 0) printf("  wave %d lane 0 at byte %d;", h[i] / 1000, 4 * i);
        }
        printf("  %d dwords written\n", n);
    }
    return 0;
}
