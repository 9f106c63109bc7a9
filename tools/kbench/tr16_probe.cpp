// What ds_read_b64_tr_b16 (gfx950) returns: LDS holds element index as a 16-bit value; every lane passes the address of 4 contiguous
// elements; prints, per lane, the four 16-bit values it received.
//   hipcc --offload-arch=gfx950 -O3 tools/kbench/tr16_probe.cpp -o tools/kbench/bin/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short *out, int pitch) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, li = l & 15, lg = l >> 4;
    // lane (li, lg): block row (li >> 2) of k-group lg, 4 columns starting at 4 * (li & 3); row pitch `pitch` elements
    const int idx = (4 * lg + (li >> 2)) * pitch + 4 * (li & 3);
    const v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s *)(lds + idx));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short *d, h[256];
    hipMalloc(&d, 512);
    for (int pitch : {16, 64}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pitch);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("pitch %d: lane -> values (as row*pitch + col => (row, col))\n", pitch);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h[l * 4 + j] / pitch, h[l * 4 + j] % pitch);
            printf("\n");
        }
    }
    return 0;
}
