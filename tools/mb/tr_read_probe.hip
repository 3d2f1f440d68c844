// What ds_read_b64_tr_b16 delivers: LDS tile t[row][col] = row * 256 + col (u16, 72-element rows); lane 4q + p of each 16-lane
// group g supplies the address of row 4g + q, columns 4p .. 4p + 3; prints, per lane, the (row, col) of its four elements.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned long long *out)
{
    __shared__ __attribute__((aligned(16))) unsigned short t[64 * 72];
    for (int i = threadIdx.x; i < 64 * 72; i += 64) t[i] = (unsigned short)((i / 72) * 256 + (i % 72));
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    unsigned addr = (unsigned)(size_t)(&t[(4 * g + q) * 72 + 4 * p]);
    unsigned long long r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[threadIdx.x] = r;
}
int main()
{
    unsigned long long *d, h[64];
    hipMalloc(&d, 64 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; e++) {
            unsigned v = (h[l] >> (16 * e)) & 0xffff;
            printf(" (r%u,c%u)", v >> 8, v & 255);
        }
        printf("\n");
    }
    return 0;
}
