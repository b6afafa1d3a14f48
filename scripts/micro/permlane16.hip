// Which rows does v_permlane16_swap exchange?  (gfx950; used by the GEMM epilogue's register transpose, gemm.hip)
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/permlane16.hip -o scripts/micro/bin/permlane16
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[threadIdx.x + 64] = r[1];
}
int main() {
    unsigned* d; unsigned h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int v = 0; v < 2; ++v) {
        printf("result %d, first lane of each row of 16:", v);
        for (int r = 0; r < 4; ++r) printf("  row%d=%u", r, h[64 * v + 16 * r]);
        printf("\n");
    }
    // expected (swap odd rows of the first operand with even rows of the second):
    //   result 0: row0=100 row1=200 row2=132 row3=232     result 1: row0=116 row1=216 row2=148 row3=248
    return 0;
}
