// Prints the operand / result lane layout of v_mfma_f32_4x4x1_16B_f32 (used by csrc/gru_small.hip): for every lane `la` an A
// operand that is 1 in that lane only, B = 1 + lane; the non-zero accumulator entries tell which (block, row) lane `la` feeds and
// which B lane lands in which column.  Build and run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o probe mfma4x4_probe.hip && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(lane == la ? 1.f : 0.f, (float)(1 + lane), acc, 0, 0, 0);
        for (int v = 0; v < 4; ++v) out[(la * 4 + v) * 64 + lane] = acc[v];
    }
}
int main() {
    float* d;
    hipMalloc(&d, 64 * 4 * 64 * sizeof(float));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    static float h[64 * 4 * 64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; la += 1) {
        if (la % 4 > 1 && la > 8) continue;
        printf("A lane %2d ->", la);
        for (int v = 0; v < 4; ++v)
            for (int l = 0; l < 64; ++l)
                if (h[(la * 4 + v) * 64 + l] != 0.f) printf(" [v%d lane%2d = B lane %2d]", v, l, (int)h[(la * 4 + v) * 64 + l] - 1);
        printf("\n");
    }
    return 0;
}
